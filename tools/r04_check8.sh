#!/usr/bin/env bash
# Round-4 call 8: dummy pixel row with corrected fold windows + scaled window counter + dx carried over the back edge; batched switch.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04h; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
SEL="determinis or c1_ or ragged or mirror_matches or saturated or occluded or c2_full_frame_fused or tight_tile or small_frame or multi_camera or sh_degrees or render_modes or legacy or golden_fixture or graphed_step_replays"
for v in d2 d2sw; do
  echo "== tests on $v"; DNSPLAT_LIB=$R/gpurun_ab/lib_$v.so timeout 900 python -m pytest tests -m gpu -q -k "$SEL" > $O/pytest_$v.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_$v.log | head -12
done
echo "== A/B c2"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh head trim d2 d2sw 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
echo "== A/B c5"
STEPS=15 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh head d2 d2sw 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
