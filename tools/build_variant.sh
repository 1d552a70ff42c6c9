#!/usr/bin/env bash
# Cross-compiles a VARIANT of libdnsplat.so (extra -D flags) into gpurun_ab/lib_<name>.so, here in the CPU container;
# gpurun_ab/ travels with the snapshot, so a GPU call only has to time the libraries (tools/ab_libs.sh).
#   tools/build_variant.sh base ""            tools/build_variant.sh sym "-DDNS_EXP_SYM=1"
# <FILE>_SRC=path (BINNING_SRC, PROJECT_SRC, RASTER_FWD_SRC, RASTER_BWD_SRC) compiles another version of that one source file
# (e.g. `git show <rev>:dn-splatter_amd/csrc/binning.hip > /tmp/b.hip`) in place of the tree's — for changes that have no switch.
set -euo pipefail
NAME=$1; FLAGS=${2:-}
cd "$(dirname "$0")/../dn-splatter_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$(pwd) $FLAGS"
O=../../gpurun_ab/_obj_$NAME; mkdir -p $O
src() { local v="$1_SRC"; local f="${!v:-}"; if [ -n "$f" ]; then cp "$f" "$O/$2"; echo "$O/$2"; else echo "$2"; fi; }
pids=()
$HIPCC $COMMON -ffp-contract=off -c $(src PROJECT project.hip)       -o $O/project.o & pids+=($!)
$HIPCC $COMMON                   -c $(src BINNING binning.hip)       -o $O/binning.o & pids+=($!)
$HIPCC $COMMON                   -c $(src RASTER_FWD raster_fwd.hip) -o $O/raster_fwd.o & pids+=($!)
$HIPCC $COMMON -fno-slp-vectorize -c $(src RASTER_BWD raster_bwd.hip) -o $O/raster_bwd.o & pids+=($!)
$HIPCC $COMMON                   -c c_api.hip      -o $O/c_api.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c postops.hip    -o $O/postops.o & pids+=($!)
$HIPCC $COMMON                   -c losses.hip     -o $O/losses.o & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $O/*.o -o ../../gpurun_ab/lib_$NAME.so
rm -rf $O
echo "built gpurun_ab/lib_$NAME.so ($FLAGS)"
