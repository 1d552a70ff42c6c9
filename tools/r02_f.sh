#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02f; export TMPDIR=/tmp
L=gpurun_ab
echo "== paired A/B raster_bwd, forward's keep masks ON"; DNSPLAT_KEEP_MASKS=1 python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs $L/lib_dual.so,$L/lib_clamp.so 2>&1 | grep -v amdgpu.ids
echo "== paired A/B raster_bwd, keep masks OFF"; DNSPLAT_KEEP_MASKS=0 python tools/ab_kernels.py --entry dnsplat_raster_bwd --libs $L/lib_dual.so,$L/lib_clamp.so 2>&1 | grep -v amdgpu.ids
echo "== paired A/B raster_fwd masks ON vs OFF (same lib)"; DNSPLAT_KEEP_MASKS=1 python tools/ab_kernels.py --entry dnsplat_raster_fwd --libs $L/lib_clamp.so 2>&1 | grep -v amdgpu.ids; DNSPLAT_KEEP_MASKS=0 python tools/ab_kernels.py --entry dnsplat_raster_fwd --libs $L/lib_clamp.so 2>&1 | grep -v amdgpu.ids
echo "== parity (default lib: clamp + masks)"; timeout 900 python -m pytest tests -m gpu -q -x -k "c1_raster or saturated or mirror or full_frame or c3_centre or golden or ragged" 2>&1 | tail -3
echo "== bench"; python bench.py --no-cpu-baseline --steps 30 --warmup 5 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['other_ms_torch_postops_autograd_host'], {k:v['ms'] for k,v in d['stages'].items()})"
