#!/usr/bin/env bash
# One gpurun call: smoke, GPU parity tests, bench, rocprof kernel trace.  Outputs under gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
if [ "${SKIP_TESTS:-0}" != "1" ]; then
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
fi
echo "== bench"; timeout 900 python bench.py --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err"; echo "rocprof rc=$?"
cd "$R"; find gpurun_out/prof -name '*stats*' | head; 
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f"
fi
