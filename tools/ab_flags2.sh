#!/usr/bin/env bash
# A/B of a -D flag applied to BOTH compositing kernels (raster_fwd + raster_bwd) inside one gpurun call.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_flags2.sh "-DDNS_TILE_ORDER=0" "-DDNS_TILE_ORDER=2"'
cd "${GRAFT_REPO_ROOT:-.}"
C=dn-splatter_amd/csrc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for flags in "$@" "$1"; do
  ( cd $C && /opt/rocm/bin/hipcc $COMMON $flags -c raster_fwd.hip -o _obj/raster_fwd.o && /opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize $flags -c raster_bwd.hip -o _obj/raster_bwd.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj/project.o _obj/binning.o _obj/raster_fwd.o _obj/raster_bwd.o _obj/c_api.o _obj/postops.o _obj/losses.o -o ../libdnsplat.so ) || exit 1
  if [ -n "${CHECK:-}" ]; then timeout 600 python -m pytest tests -m gpu -x -q -k "$CHECK" 2>&1 | tail -1; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
print('$flags |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()))"
  done
done
