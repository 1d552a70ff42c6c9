#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02c; export TMPDIR=/tmp
echo "== full-size parity tests (default library = lean)"; timeout 900 python -m pytest tests -m gpu -q -s -k "full_frame or c3_centre or saturated or c1_raster or mirror" > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^E  |passed|failed|rounding-sensitive" gpurun_out/r02c/pytest.log | head -20
L=gpurun_ab
echo "== paired A/B raster_bwd"; python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs $L/lib_base.so,$L/lib_sym.so,$L/lib_symsl.so,$L/lib_lean.so,$L/lib_leantab.so 2>&1 | grep -v amdgpu.ids
echo "== paired A/B raster_fwd"; python tools/ab_kernels.py --entry dnsplat_raster_fwd --libs $L/lib_base.so,$L/lib_lean.so 2>&1 | grep -v amdgpu.ids
echo "== bench (lean)"; python bench.py --no-cpu-baseline --steps 30 --warmup 5 | tail -1
