#!/usr/bin/env bash
# Round-4 call 16: depth-sort chunks of 4096 keys (16 per thread) instead of 2048 at the larger sizes.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04p; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests on rsn16"; DNSPLAT_LIB=$R/gpurun_ab/lib_rsn16.so timeout 900 python -m pytest tests -m gpu -q -x -k "binning or c1_ or ragged or full_size_projection or small_frame" > $O/pytest_rsn16.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_rsn16.log | head
echo "== A/B c5"; STEPS=15 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh cur rsn16 2>&1 | grep -v amdgpu | tee $O/ab_rs_items_n_c5.txt
echo "== A/B c3"; STEPS=15 BENCH_ARGS="--workload c3 --no-strict" bash tools/ab_libs.sh cur rsn16 2>&1 | grep -v amdgpu | tee $O/ab_rs_items_n_c3.txt
