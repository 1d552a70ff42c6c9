#!/usr/bin/env bash
# Round-4 call 14: smoke() as the driver runs it; the reference's op sequence with the drop-ins (--two-call) and the fused pass with torch post-ops.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04n; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for m in "--two-call" "--torch-postops"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m |', d['value'], 'fps', d['ms_per_step'], 'ms |', d['launch'][:50])"
done 2>&1 | tee $O/bench_modes.txt
