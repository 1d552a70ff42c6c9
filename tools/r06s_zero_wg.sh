# Zero gradient rows skipped per workgroup (DNSPLAT_SH_ZERO_STATE=1: sh_zero_state + zero_state_geometry) x row order of the scene.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06s
timeout 600 python -m pytest tests -m gpu -x -q -k "zero_rows or reordered" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-strict --no-extra-workloads --steps 20 --warmup 3 $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
print('$1 |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()))"; }
for w in c2 c5; do
for rep in 1 2; do
for sc in reference_init morton; do
DNSPLAT_SH_ZERO_STATE=0 run "$w $sc zero_state=0" "--workload $w --scene $sc"
DNSPLAT_SH_ZERO_STATE=1 run "$w $sc zero_state=1" "--workload $w --scene $sc"
done; done; done
