#!/usr/bin/env bash
# Round-4 call 2: bench under faulthandler, the new tests, library A/B (binning / projection load batching), deterministic sweeps.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== bench (strict section only)"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads > $O/bench_c2_strict.json 2> $O/bench_c2_strict.err; echo "rc=$?"
grep -v "UserWarning\|run_backward\|amdgpu.ids" $O/bench_c2_strict.err | tail -30
echo "== bench (default: strict + extras + cpu baseline)"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"
grep -v "UserWarning\|run_backward\|amdgpu.ids" $O/bench_c2.err | tail -30
python - <<'PY'
import json
for f in ("bench_c2_strict", "bench_c2"):
    try:
        d = json.loads(open(f"gpurun_out/r04b/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    print(f, "value", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items()})
    s = d["strict_index_parity"]; print("  strict", {k: s.get(k) for k in ("value", "ms_per_step", "launch", "error", "stages_ms")})
    for k, v in (d.get("extra_workloads") or {}).items():
        print("  ", k, {x: v.get(x) for x in ("value", "ms_per_step", "Nv", "n_isects_sorted", "stages_ms", "error")})
PY
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_two_ranks.py -q -x 2>&1 | tail -5
echo "== library A/B in the frame, c2"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh base bin proj new 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
echo "== library A/B in the frame, c5"
STEPS=20 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh base new 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
echo "== deterministic mode vs the order-independent oracle: seed 121 x 20, then 60 unseen scenes x 5"
DNSPLAT_DETERMINISTIC=1 timeout 600 python tools/parity_seed_sweep.py 121 1 20 2>&1 | grep -v amdgpu > $O/seed121_deterministic_x20.txt; tail -2 $O/seed121_deterministic_x20.txt
DNSPLAT_DETERMINISTIC=1 timeout 1200 python tools/parity_seed_sweep.py 100 30 5 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_deterministic.txt; tail -1 $O/parity_seed_sweep_deterministic.txt; grep -c FAIL $O/parity_seed_sweep_deterministic.txt
