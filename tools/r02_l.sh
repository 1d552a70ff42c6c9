#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); L=$R/gpurun_ab; export TMPDIR=/tmp
echo "== parity / sort order with look-back (default library), bounded by timeouts"
timeout 300 python -m pytest tests -m gpu -q -x -k "c1_raster or ragged or small_frame or capacity or multi_camera or edge_empty" 2>&1 | tail -3
timeout 600 python -m pytest tests -m gpu -q -x -k "full_size or full_frame or centre_crop or c3_centre" 2>&1 | tail -3
echo "== paired A/B dnsplat_bin_emit_sort: table (hist + scan launches) vs look-back"
timeout 300 python tools/ab_kernels.py --entry dnsplat_bin_emit_sort --libs $L/lib_tab.so,$L/lib_lb.so --rounds 10 --iters 4 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/ab_kernels.py --entry dnsplat_bin_emit_sort --workload c5 --libs $L/lib_tab.so,$L/lib_lb.so --rounds 6 --iters 3 2>&1 | grep -v amdgpu.ids
echo "== bench"; timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['other_ms_torch_postops_autograd_host'], {k:v['ms'] for k,v in d['stages'].items()})"
