#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02e; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02e/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed" gpurun_out/r02e/pytest_gpu.log | head -20
echo "== bench (full, with cpu baseline)"; python bench.py --steps 30 --warmup 5 > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err; tail -1 gpurun_out/r02e/bench.json
echo "== kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r02e/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/r02e/prof_bench.json" 2> "$R/gpurun_out/r02e/prof.err"); echo "rocprof rc=$?"
f=$(find gpurun_out/r02e/prof -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py $f
f=$(find gpurun_out/r02e/prof -name '*kernel_stats.csv' | head -1); head -30 $f | cut -c1-150
