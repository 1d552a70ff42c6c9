#!/usr/bin/env bash
# Round-4 call 3: failing test detail, default bench line with child-process extras, A/B of the hoisted parameter loads, full suite.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04c; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== determinism / empty-frame tests"; timeout 900 python -m pytest tests/test_gpu_determinism.py -q > $O/pytest_det.log 2>&1; echo "rc=$?"; grep -E "^E |passed|failed|Error" $O/pytest_det.log | head -30
echo "== bench (default)"; timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "rc=$?"
grep -v "UserWarning\|run_backward\|amdgpu.ids" $O/bench_c2.err | tail -20
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04c/bench_c2.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items()})
    s = d["strict_index_parity"]; print("  strict", {k: s.get(k) for k in ("value", "ms_per_step", "launch", "error", "stages_ms")})
    for k, v in (d.get("extra_workloads") or {}).items():
        print("  ", k, {x: v.get(x) for x in ("value", "ms_per_step", "Nv", "n_isects_sorted", "stages_ms", "roofline", "useful_pair_fraction", "error")})
    print("  valu", d["roofline_valu"]["dnsplat_raster_bwd"], d["roofline_valu"]["dnsplat_raster_fwd"])
    print("  cpu", d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e)
PY
echo "== library A/B in the frame, c2: new (batched loads) vs new2 (+ parameter loads hoisted above the staging)"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh new new2 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
STEPS=20 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh new new2 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
if [ -z "${SKIP_TESTS:-}" ]; then
echo "== pytest -m gpu (full)"; DNSPLAT_MARGIN_LOG=$O/margins.tsv timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
fi
