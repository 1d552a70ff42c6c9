#!/usr/bin/env bash
# Round-4 call 10: full GPU suite on the new defaults (backward trims on, ABI 12: factor slab from project_bwd), one rank through RCCL.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04j; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu"; DNSPLAT_MARGIN_LOG=$R/$O/margins.tsv timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_gpu.log | head -20
echo "== one rank through RCCL"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads > $O/bench_c2_single_rank_rccl.json 2>$O/rccl.err; tail -1 $O/bench_c2_single_rank_rccl.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step_p10_p50_p90'], d.get('multi_gpu'))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['gpu_ms_per_step_p10_p50_p90'], d['launch'][:60])"
