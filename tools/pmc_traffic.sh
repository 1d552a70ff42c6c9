#!/usr/bin/env bash
# HBM traffic of every dnsplat kernel from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (3 + 2 TCC slots), kernel-trace only.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/pmc_traffic.sh'   ->  gpurun_out/pmc_traffic/
#   other workloads: WORKLOAD=c3 BENCH_ARGS='--workload c3' bash tools/pmc_traffic.sh
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/pmc_traffic; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/gpurun_out/pmc_traffic/$c" -o p -- \
     python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean ${BENCH_ARGS:-} > "$R/gpurun_out/pmc_traffic/$c.json" 2> "$R/gpurun_out/pmc_traffic/$c.err"); echo "$c rc=$?"
done
python tools/pmc_summary.py gpurun_out/pmc_traffic > gpurun_out/pmc_traffic/summary.json; python tools/pmc_traffic.py gpurun_out/pmc_traffic/summary.json ${WORKLOAD:-c2}
