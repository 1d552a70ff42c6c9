#!/usr/bin/env bash
# Round-4 call 6: tile-pass occupancy (LDS tables sized by digits + aliased, 64-VGPR cap) A/B; selected parity tests on the current tree.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04f; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests (current tree)"; timeout 900 python -m pytest tests -m gpu -q -k "c1_raster or binning or ragged or multi_camera or c2_full_frame_rasterization or fused_loss or without_intersections" > $O/pytest_sel.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_sel.log | head
echo "== A/B c2: head | occ1 (LDS shrink only) | occ8 (+ 64-VGPR cap)"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh head occ1 occ8 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
echo "== A/B c5"
STEPS=20 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh head occ1 occ8 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
