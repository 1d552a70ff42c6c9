#!/usr/bin/env bash
# Round-4 validation call: smoke, GPU tests, default bench line (with the C3 / C5 sections), deterministic-mode sweeps.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r04_check.sh'
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04a; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
if [ -z "${SKIP_TESTS:-}" ]; then
echo "== pytest -m gpu"; DNSPLAT_MARGIN_LOG=$O/margins.tsv timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
fi
if [ -z "${SKIP_BENCH:-}" ]; then
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"; tail -1 $O/bench_c2.json | cut -c1-400; tail -3 $O/bench_c2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04a/bench_c2.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items()})
print("strict", {k: d["strict_index_parity"].get(k) for k in ("value", "ms_per_step", "launch", "error")})
for k, v in (d.get("extra_workloads") or {}).items():
    print(k, {x: v.get(x) for x in ("value", "ms_per_step", "Nv", "n_isects_sorted", "stages_ms", "error")})
print("valu", d["roofline_valu"]["dnsplat_raster_bwd"], d["roofline_valu"]["dnsplat_raster_fwd"])
PY
echo "== single rank through RCCL, graphed compute + eager exchange"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_single_rank_rccl.json 2> $O/bench_rccl.err; echo "rc=$?"; tail -1 $O/bench_c2_single_rank_rccl.json | cut -c1-300; tail -3 $O/bench_rccl.err
fi
if [ -z "${SKIP_SWEEP:-}" ]; then
echo "== deterministic mode: 60 unseen scenes x 5 runs, seed 121 x 20"
DNSPLAT_DETERMINISTIC=1 timeout 1200 python tools/parity_seed_sweep.py 100 30 5 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_deterministic.txt; tail -2 $O/parity_seed_sweep_deterministic.txt; grep -c FAIL $O/parity_seed_sweep_deterministic.txt
DNSPLAT_DETERMINISTIC=1 timeout 600 python tools/parity_seed_sweep.py 121 1 20 2>&1 | grep -v amdgpu > $O/seed121_deterministic_x20.txt; cat $O/seed121_deterministic_x20.txt | tail -4
timeout 600 python tools/parity_seed_sweep.py 121 1 20 2>&1 | grep -v amdgpu > $O/seed121_atomics_x20.txt; tail -3 $O/seed121_atomics_x20.txt
fi
