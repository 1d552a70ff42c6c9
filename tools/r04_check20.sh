#!/usr/bin/env bash
# Round-4 call 20: SQ / GRBM counters per kernel on the final sources (vector-busy fraction, instructions per call).
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04t; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$R/$O/pmc_sq" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "sq rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_grbm" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "grbm rc=$?"
python tools/pmc_summary.py $O > $O/pmc_counters.json 2>$O/pmc_summary.err; head -c 300 $O/pmc_summary.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04t/pmc_counters.json"))
for k in ("raster_bwd_kernel", "raster_fwd_kernel", "radix_scatter_kernel", "radix_hist_ranges_kernel", "project_fwd_kernel", "project_bwd_kernel"):
    if k in d: print(k, {c: round(v) for c, v in d[k].items() if c.startswith(("SQ_", "GRBM"))})
PY
rm -rf $O/pmc_sq $O/pmc_grbm
