#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02d; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r02d/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|rounding-sensitive" gpurun_out/r02d/pytest_gpu.log | head -40
tail -15 gpurun_out/r02d/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench"; python bench.py --no-cpu-baseline --steps 30 --warmup 5 | tail -1
