#!/usr/bin/env bash
# Round-4 call 7: backward trims (|.| sums as v_fma, chained d/d(alpha)), dummy pixel row, batched group switch; first tile pass's
# histogram from digit ranges.  Selected parity tests per variant, then the variants interleaved in whole bench runs.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04g; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
SEL="c1_ or ragged or mirror_matches or saturated or occluded or c2_full_frame_fused or tight_tile or small_frame or multi_camera or full_size_projection or determinis"
for v in trim dummy dsw; do
  echo "== tests on $v"; DNSPLAT_LIB=$R/gpurun_ab/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -x -k "$SEL" > $O/pytest_$v.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_$v.log | head -8
done
echo "== A/B c2"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh head trim dummy dsw 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
echo "== A/B c5"
STEPS=15 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh head trim dsw 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
