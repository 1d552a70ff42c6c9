#!/usr/bin/env bash
# Round-2 opening GPU call: the whole -m gpu suite (incl. the new full-size parity tests), the bench line, and the
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/r02a; export TMPDIR=/tmp
nproc; free -g | head -2
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -s --durations=15 > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "\[parity\]|passed|failed|Error|error" gpurun_out/r02a/pytest_gpu.log | sort | uniq -c | sort -rn | head -60
tail -30 gpurun_out/r02a/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"; cat gpurun_out/r02a/bench.json
echo "== PMC calibration"
for c in WRITE_SIZE FETCH_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/r02a/cal_$c" -o p -- python "$R/tools/pmc_calibrate.py" > "$R/gpurun_out/r02a/cal_$c.log" 2>&1); echo "$c rc=$?"
done
python tools/pmc_by_kernel.py gpurun_out/r02a/cal_WRITE_SIZE; python tools/pmc_by_kernel.py gpurun_out/r02a/cal_FETCH_SIZE
