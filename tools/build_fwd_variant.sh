#!/usr/bin/env bash
# Fast variant of libdnsplat.so in which only raster_fwd.hip is recompiled (extra -D flags); the other objects are the tree's
# csrc/_obj/*.o (run csrc/build.sh first).   tools/build_fwd_variant.sh name "-DDNS_FWD_X=1"  ->  gpurun_ab/lib_<name>.so
set -euo pipefail
NAME=$1; FLAGS=${2:-}
cd "$(dirname "$0")/../dn-splatter_amd/csrc"
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c raster_fwd.hip -o /tmp/raster_fwd_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj/project.o _obj/binning.o /tmp/raster_fwd_$NAME.o _obj/raster_bwd.o _obj/c_api.o _obj/postops.o _obj/losses.o -o ../../gpurun_ab/lib_$NAME.so
echo "built gpurun_ab/lib_$NAME.so ($FLAGS)"
