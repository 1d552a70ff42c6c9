# Row order of the Gaussians: the reference's random initialisation vs the same set along a Morton curve (bench.py --scene morton),
# paired inside one gpurun call.   gpurun --timeout 1500 -- 'bash tools/r06s_morton.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/r06s
timeout 600 python -m pytest tests -m gpu -x -q -k "reordered or refinement or densif" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-strict --no-extra-workloads --steps 20 --warmup 3 $2 2>/dev/null | tail -1 | tee gpurun_out/r06s/morton_$3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
st=d['stages']
print('$1 |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()))"; }
for w in c2 c3 c5; do
for rep in 1 2; do
run "$w reference_init" "--workload $w" ${w}_ref_$rep
run "$w morton        " "--workload $w --scene morton" ${w}_morton_$rep
done; done
