#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r02h; mkdir -p $O; export TMPDIR=/tmp
for c in WRITE_SIZE; do
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$O/store_$c" -o p -- "$R/tools/ubench/store_patterns" > "$R/$O/store_$c.log" 2>&1); echo "$c rc=$?"; tail -1 $O/store_$c.log
  python tools/pmc_by_kernel.py $O/store_$c
done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/store_time" -o p -- "$R/tools/ubench/store_patterns" > /dev/null 2>&1); f=$(find $O/store_time -name '*kernel_stats.csv' | head -1); cut -d, -f1-4 $f | head -10
echo "== sh rebuild test"; timeout 300 python -m pytest tests -m gpu -q -k "sh_rebuild" 2>&1 | tail -2
