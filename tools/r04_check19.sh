#!/usr/bin/env bash
# Round-4 call 19: the 60 unseen scenes in the DEFAULT (atomics) gradient mode on the final sources.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04s; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 1100 python tools/parity_seed_sweep.py 100 30 1 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_default.txt; tail -2 $O/parity_seed_sweep_default.txt
