#!/usr/bin/env bash
# Round-4 call 17: N-dependent depth-sort chunks (4096 keys from 4 M entries on): full-size tests, PMC traffic re-stamped, bench line.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04q; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -k "c5 or full_size or binning or c3_full or big_scene" > $O/pytest_sel.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_sel.log | head
SKIP_TESTS=1 SKIP_RCCL=1 TAG=r04q_set bash tools/r04_set.sh 2>&1 | grep -v "at::native\|elementwise" | head -60
