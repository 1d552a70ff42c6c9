#!/usr/bin/env bash
# Round-4 call 11: the 60 unseen scenes of the seed sweep in deterministic gradient mode, 3 repeats each (bit-identity run to run),
# seed 121 (the scene whose atomics order moved one entry past its allowance in round 3) 20 times in both modes.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04k; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== deterministic sweep 100..129 x 3"; DNSPLAT_DETERMINISTIC=1 timeout 1100 python tools/parity_seed_sweep.py 100 30 3 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_deterministic.txt; tail -3 $O/parity_seed_sweep_deterministic.txt; grep -c "FAIL\|VARIES" $O/parity_seed_sweep_deterministic.txt
echo "== seed 121 x 20, deterministic"; DNSPLAT_DETERMINISTIC=1 timeout 300 python tools/parity_seed_sweep.py 121 1 20 2>&1 | grep -v amdgpu > $O/seed121_deterministic_x20.txt; tail -4 $O/seed121_deterministic_x20.txt
echo "== seed 121 x 20, default (atomics)"; timeout 300 python tools/parity_seed_sweep.py 121 1 20 2>&1 | grep -v amdgpu > $O/seed121_default_x20.txt; tail -4 $O/seed121_default_x20.txt
