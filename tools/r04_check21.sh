#!/usr/bin/env bash
# Round-4 call 21: PMC traffic and SQ counters again with tools/pmc_summary.py fixed (since the DET template flag was added the summary
# counted both twins of raster_bwd as calls and halved every per-call figure of that kernel; radix_hist_ranges_kernel was not listed).
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04u; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for w in c2 c3 c5; do
  rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
  WORKLOAD=$w BENCH_ARGS="--workload $w" bash tools/pmc_traffic.sh > $O/pmc_traffic_$w.log 2>&1
  cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json; cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary_$w.json
  rm -rf gpurun_out/pmc_traffic
done
python - $O/pmc_traffic.merged.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for w, e in d.items():
    print(w, e.get("source_sha16"), {k: round(v["hbm_bytes_per_launch"] / 1e6) for k, v in e.items() if isinstance(v, dict)})
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$R/$O/pmc_sq" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "sq rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_grbm" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "grbm rc=$?"
python tools/pmc_summary.py $O > $O/pmc_counters.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04u/pmc_counters.json"))
for k in ("raster_bwd_kernel", "raster_fwd_kernel", "radix_scatter_kernel", "radix_hist_ranges_kernel", "project_fwd_kernel", "project_bwd_kernel"):
    if k in d: print(k, {c: round(v) for c, v in d[k].items() if c.startswith(("SQ_", "GRBM"))})
PY
rm -rf $O/pmc_sq $O/pmc_grbm
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
