#!/usr/bin/env bash
# Round-4 call 13: is the one slow step of r04d_c2_bench.json an accident?  The bench five times, per-step series of each.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04m; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['gpu_ms_per_step_series']
print('run $i |', d['value'], 'fps', d['ms_per_step'], 'ms | max step %.3f at index %d | p50 %.4f |' % (max(s), s.index(max(s)), d['gpu_ms_per_step_p10_p50_p90'][1]), [x for x in s if x > 2.3])"
done 2>&1 | tee $O/bench_repeats.txt
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['gpu_ms_per_step_series']
print('40 steps |', d['value'], 'fps', d['ms_per_step'], 'ms | slow steps', [(i, x) for i, x in enumerate(s) if x > 2.3])" | tee -a $O/bench_repeats.txt
