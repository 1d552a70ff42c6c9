#!/usr/bin/env bash
# Round-4 call 15: tile passes with 1024-thread workgroups (4 keys per thread, 55-57 VGPRs: 32 instead of 24 waves per CU).
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04o; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests on ti1024"; DNSPLAT_LIB=$R/gpurun_ab/lib_ti1024.so timeout 900 python -m pytest tests -m gpu -q -x -k "binning or c1_ or ragged or multi_camera or tight_tile or full_size_projection or c2_full_frame or small_frame" > $O/pytest_ti1024.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_ti1024.log | head
echo "== A/B c2"; STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh cur ti1024 2>&1 | grep -v amdgpu | tee $O/ab_ti_threads_c2.txt
echo "== A/B c5"; STEPS=15 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh cur ti1024 2>&1 | grep -v amdgpu | tee $O/ab_ti_threads_c5.txt
