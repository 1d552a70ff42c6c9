#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r02j; mkdir -p $O; export TMPDIR=/tmp
L=$R/gpurun_ab
echo "== paired A/B dnsplat_bin_emit_sort (host library = bin8: the largest workspace)"
DNSPLAT_LIB=$L/lib_bin8.so python tools/ab_kernels.py --entry dnsplat_bin_emit_sort --libs $L/lib_bin16old.so,$L/lib_bin16.so,$L/lib_bin8.so,$L/lib_bin32.so --rounds 10 --iters 4 2>&1 | grep -v amdgpu.ids
echo "== same at c3"
DNSPLAT_LIB=$L/lib_bin8.so python tools/ab_kernels.py --entry dnsplat_bin_emit_sort --workload c3 --libs $L/lib_bin16old.so,$L/lib_bin16.so,$L/lib_bin8.so,$L/lib_bin32.so --rounds 6 --iters 3 2>&1 | grep -v amdgpu.ids
for v in bin8 bin32; do echo "== parity with $v"; DNSPLAT_LIB=$L/lib_$v.so timeout 600 python -m pytest tests -m gpu -q -x -k "c1_raster or ragged or full_size_proj or multi_camera or small_frame" 2>&1 | tail -2; done
