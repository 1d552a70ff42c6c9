cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06s
timeout 600 python -m pytest tests -m gpu -x -q -k "ssim or fused_loss or install" 2>&1 | tail -5 > gpurun_out/r06s/pytest_ssim.log
cat gpurun_out/r06s/pytest_ssim.log
for mode in torch_hip_ssim torch_capturable fused; do
  python bench.py --workload c5 --losses $mode --no-cpu-baseline --no-strict --no-extra-workloads --steps 15 --warmup 3 2>gpurun_out/r06s/bench_c5_$mode.err | tail -1 > gpurun_out/r06s/bench_c5_$mode.json
  python -c "
import json
d=json.load(open('gpurun_out/r06s/bench_c5_$mode.json'))
print('$mode', d['value'], 'fps', d['ms_per_step'], 'ms', d.get('config',{}).get('graph'))"
done
