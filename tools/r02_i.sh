#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r02i; mkdir -p $O; export TMPDIR=/tmp
L=gpurun_ab
echo "== paired A/B project_bwd / project_fwd: before (lib_clamp) vs scratch-free (libdnsplat)"
python tools/ab_kernels.py --entry dnsplat_project_bwd --libs $L/lib_clamp.so,dn-splatter_amd/libdnsplat.so 2>&1 | grep -v amdgpu.ids
python tools/ab_kernels.py --entry dnsplat_project_fwd --libs $L/lib_clamp.so,dn-splatter_amd/libdnsplat.so 2>&1 | grep -v amdgpu.ids
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
echo "== PMC traffic"; rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic; bash tools/pmc_traffic.sh 2>&1 | grep -A4 "project\|raster_bwd\"" | head -40
echo "== bench"; python bench.py --no-cpu-baseline --steps 30 --warmup 5 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['other_ms_torch_postops_autograd_host'], {k:v['ms'] for k,v in d['stages'].items()})"
