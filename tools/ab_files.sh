#!/usr/bin/env bash
# A/B timing of whole-file variants of one kernel source inside a single gpurun call.
#   cp variant sources to gpurun_ab/<name>.hip first (that directory travels with the snapshot), then
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_files.sh raster_bwd base m12'
# Every variant is compiled as dn-splatter_amd/csrc/<kernel>.hip would be; the first one is repeated at the end.
cd "${GRAFT_REPO_ROOT:-.}"
C=dn-splatter_amd/csrc
KERNEL=$1; shift
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I$C"
EXTRA=""
[ "$KERNEL" = raster_bwd ] && EXTRA="-fno-slp-vectorize"
[ "$KERNEL" = project ] && EXTRA="-ffp-contract=off"
cp $C/$KERNEL.hip /tmp/keep.hip
for v in "$@" "$1"; do
  cp gpurun_ab/$v.hip $C/$KERNEL.hip
  ( cd $C && /opt/rocm/bin/hipcc $COMMON $EXTRA -c $KERNEL.hip -o _obj/$KERNEL.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _obj/project.o _obj/binning.o _obj/raster_fwd.o _obj/raster_bwd.o _obj/c_api.o _obj/postops.o _obj/losses.o -o ../libdnsplat.so ) || exit 1
  if [ -n "${CHECK:-}" ]; then timeout 600 python -m pytest tests -m gpu -x -q -k "$CHECK" 2>&1 | tail -1; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
print('$v |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()))"
  done
done
cp /tmp/keep.hip $C/$KERNEL.hip
