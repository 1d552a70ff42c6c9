#!/usr/bin/env bash
# Round 6, VERDICT r05 item 2: one rank through RCCL — what the exchange step costs before a byte crosses xGMI — for the forms of the SH
# exchange (bench.py --exchange), at C2 and C5, plus the kernel timeline of one frame (rocprofv3 kernel trace).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r06_exchange.sh'   ->  gpurun_out/r06_exchange/
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r06_exchange; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
export DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
port=29520
for w in c2 c5; do
  for mode in ${MODES:-auto rebuild packed own+packed}; do
    port=$((port+1))
    MASTER_PORT=$port python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $w --exchange $mode > $O/bench_${w}_${mode}.json 2> $O/bench_${w}_${mode}.err
    tail -1 $O/bench_${w}_${mode}.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d['multi_gpu']
print('$w $mode |', d['value'], 'fps | step', m['step_ms'], 'compute_only', m['compute_only_ms'], 'single_gpu', m['single_gpu_graphed_step_ms'], 'exposed_vs_single', m['exchange_exposed_vs_single_gpu_step_ms'], 'exchange_alone', m['exchange_alone_ms'], '| slab bytes', m.get('slab_bytes_per_rank'))"
  done
done | tee $O/summary.txt
for w in c5; do for mode in auto rebuild "rebuild --slices 4"; do
  port=$((port+1)); tag=$(echo $mode | tr -d ' -')
  (cd /tmp && MASTER_PORT=$port timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof_${w}_${tag}" -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --lean --workload $w --exchange $mode > /dev/null 2> "$R/$O/prof_${w}_${tag}.err")
  f=$(find $O/prof_${w}_${tag} -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python tools/frame_timeline.py "$f" > $O/frame_timeline_single_rank_rccl_${w}_${tag}.txt 2>&1
  rm -rf $O/prof_${w}_${tag}
done; done
