#!/usr/bin/env bash
# Builds raster_bwd.hip with -DDNS_BWD_TIMELINE, runs tools/bwd_timeline.py, restores the normal library.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/bwd_timeline.sh'
cd "${GRAFT_REPO_ROOT:-.}"
C=dn-splatter_amd/csrc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize"
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/_obj/project.o $C/_obj/binning.o $C/_obj/raster_fwd.o $C/_obj/raster_bwd.o $C/_obj/c_api.o $C/_obj/postops.o $C/_obj/losses.o -o dn-splatter_amd/libdnsplat.so; }
( cd $C && /opt/rocm/bin/hipcc $COMMON -DDNS_BWD_TIMELINE -c raster_bwd.hip -o _obj/raster_bwd.o ) && link || exit 1
timeout 200 python tools/bwd_timeline.py 2>&1 | grep -v amdgpu.ids
( cd $C && /opt/rocm/bin/hipcc $COMMON -c raster_bwd.hip -o _obj/raster_bwd.o ) && link
