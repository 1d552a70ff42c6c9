#!/usr/bin/env bash
# Times pre-built library variants (tools/build_variant.sh) inside ONE gpurun call, interleaved, first variant repeated last.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'CHECK="c1_raster or mirror" bash tools/ab_libs.sh base sym symtab'
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd)
for v in "$@" "$1"; do
  export DNSPLAT_LIB=$R/gpurun_ab/lib_$v.so
  if [ -n "${CHECK:-}" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$CHECK" 2>&1 | tail -2; fi
  for rep in 1 2; do
    python bench.py --no-cpu-baseline --steps ${STEPS:-30} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
print('$v |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()))"
  done
done
