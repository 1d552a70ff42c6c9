#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r02m; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for w in c3 c5; do echo "== bench $w"; timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; tail -1 $O/bench_$w.json | cut -c1-300; done
echo "== bench c5 with the loss stack (fused / torch)"; timeout 600 python bench.py --workload c5 --losses fused --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5_fused.json 2> $O/bench_c5_fused.err; tail -1 $O/bench_c5_fused.json | cut -c1-200
echo "== single-rank RCCL (DNSPLAT_FORCE_DIST=1): the whole exchange sequence + the multi_gpu report"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2_forcedist.json 2> $O/bench_c2_forcedist.err; tail -1 $O/bench_c2_forcedist.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['multi_gpu']))"; tail -3 $O/bench_c2_forcedist.err
