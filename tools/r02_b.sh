#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02b; export TMPDIR=/tmp
echo "== pytest -m gpu (default library = symtab)"; timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r02b/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|rounding-sensitive" gpurun_out/r02b/pytest_gpu.log | head -40
tail -12 gpurun_out/r02b/pytest_gpu.log
echo "== A/B"; CHECK="c1_raster or saturated or test_c2_full" bash tools/ab_libs.sh base sym symtab
