#!/usr/bin/env bash
# A/B timing of raster_bwd.hip build variants inside ONE gpurun call (box-to-box variation is ~2 %).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_bwd.sh "-DDNS_BWD_GROUP=16" "-DDNS_BWD_GROUP=8"'
cd "${GRAFT_REPO_ROOT:-.}"
R=$(pwd)
mkdir -p gpurun_out
C=dn-splatter_amd/csrc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for flags in "$@" "$1"; do
  ( cd $C && /opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize $flags -c raster_bwd.hip -o _obj/raster_bwd.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj/project.o _obj/binning.o _obj/raster_fwd.o _obj/raster_bwd.o _obj/c_api.o _obj/postops.o _obj/losses.o -o ../libdnsplat.so ) || exit 1
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$flags', d['value'], d['ms_per_step'], 'bwd', d['stages']['dnsplat_raster_bwd']['ms'], 'fwd', d['stages']['dnsplat_raster_fwd']['ms'])"
  done
done
