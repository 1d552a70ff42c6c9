#!/usr/bin/env bash
# profiles refresh: full GPU suite, bench line, rocprofv3 kernel stats, PMC traffic (bench) and the per-kernel replay traffic
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/profiles_refresh; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
echo "== bench"; python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-400
echo "== kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
echo "== PMC traffic of the bench (FETCH_SIZE / WRITE_SIZE in separate passes)"
mkdir -p gpurun_out/pmc_traffic; rm -rf gpurun_out/pmc_traffic/*
bash tools/pmc_traffic.sh 2>&1 | tail -40
echo "== replay traffic: each kernel launched back to back on its own (no neighbours whose dirty lines could be flushed in its window)"
for e in dnsplat_project_bwd dnsplat_project_fwd dnsplat_raster_bwd; do
  for c in WRITE_SIZE FETCH_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$O/replay_${e}_$c" -o p -- python "$R/tools/ab_kernels.py" --entry $e --libs "$R/dn-splatter_amd/libdnsplat.so" --rounds 6 --iters 4 > /dev/null 2>&1)
    echo "$e $c"; python tools/pmc_by_kernel.py $O/replay_${e}_$c | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if any(s in k for s in ('project_bwd_kernel','project_fwd_kernel','raster_bwd_kernel')): print('   ',k[-60:],v)"
  done
done
echo "== VALU busy counters"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$R/$O/pmc_sq" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1); echo "sq rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_grbm" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1); echo "grbm rc=$?"
python tools/pmc_summary.py $O > $O/pmc_sq_summary.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/profiles_refresh/pmc_sq_summary.json"))
for k in ("raster_bwd_kernel","raster_fwd_kernel"):
    if k in d: print(k, {c:v for c,v in d[k].items() if c.startswith(("SQ_","GRBM"))})
PY
