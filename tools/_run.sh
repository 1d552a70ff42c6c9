#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "rasterization_matches or ragged or mirror or sh_degrees or saturated or golden or determin or permutation or fused_loss or multi_camera" 2>&1 | tail -5
timeout 600 python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs gpurun_ab/lib_nofold.so,gpurun_ab/lib_fold.so 2>&1 | tail -6
timeout 600 python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs gpurun_ab/lib_nofold.so,gpurun_ab/lib_fold.so --workload c5 2>&1 | tail -4
timeout 300 python tools/_diag.py 2>&1 | grep -v amdgpu | tail -24
