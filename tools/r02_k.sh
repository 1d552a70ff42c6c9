#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); L=$R/gpurun_ab; export TMPDIR=/tmp
echo "== paired A/B raster_bwd: lanes per group switch"
python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs $L/lib_g16.so,$L/lib_g8.so,$L/lib_g32.so 2>&1 | grep -v amdgpu.ids
python tools/ab_kernels.py --entry dnsplat_raster_bwd --workload c3 --libs $L/lib_g16.so,$L/lib_g8.so --rounds 8 2>&1 | grep -v amdgpu.ids
echo "== parity with g8"; DNSPLAT_LIB=$L/lib_g8.so timeout 600 python -m pytest tests -m gpu -q -x -k "c1_raster or saturated or mirror or ragged" 2>&1 | tail -2
