#!/usr/bin/env bash
# A/B timing of build variants of ONE kernel file inside a single gpurun call (box-to-box variation is ~2 %).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/ab_kernel.sh raster_fwd "-DDNS_FWD_PX=2" "-DDNS_FWD_PX=4"'
# The first variant is run again at the end; the library is left built with it.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
C=dn-splatter_amd/csrc
KERNEL=$1; shift
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
EXTRA=""
[ "$KERNEL" = raster_bwd ] && EXTRA="-fno-slp-vectorize"
[ "$KERNEL" = project ] && EXTRA="-ffp-contract=off"
[ "$KERNEL" = postops ] && EXTRA="-ffp-contract=off"
for flags in "$@" "$1"; do
  ( cd $C && /opt/rocm/bin/hipcc $COMMON $EXTRA $flags -c $KERNEL.hip -o _obj/$KERNEL.o &&
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
