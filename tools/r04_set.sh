#!/usr/bin/env bash
# Round-4 measurement set in ONE GPU call: full GPU suite, the driver's bench line (with strict / c3 / c5 sections), the
# rocprofv3 kernel stats of the same command, PMC traffic for c2, the one-rank-through-RCCL line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'TAG=r04a bash tools/r04_set.sh'  ->  gpurun_out/$TAG/
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/${TAG:-r04a}; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then echo "== pytest -m gpu"; DNSPLAT_MARGIN_LOG=$R/$O/margins.tsv timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_gpu.log | head -20; fi
echo "== bench (driver's command)"
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-400
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["value"], "fps", d["ms_per_step"], "ms |", {k.replace("dnsplat_", ""): v["ms"] for k, v in d["stages"].items()})
print("roofline", d.get("roofline")); print("strict", {k: d.get("strict_index_parity", {}).get(k) for k in ("value", "launch", "ms_per_step")})
for k, v in (d.get("extra_workloads") or {}).items(): print(k, {a: v.get(a) for a in ("value", "ms_per_step", "error")} if isinstance(v, dict) else v)
print("valu", d.get("roofline_valu")); print("pairs", d.get("pairs"))
PY
echo "== kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --lean > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{r['Name'][:90]:90s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} us {r['Percentage']}")
PY
if [ -z "${SKIP_PMC:-}" ]; then
echo "== PMC traffic c2"
rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
WORKLOAD=c2 BENCH_ARGS="--workload c2" bash tools/pmc_traffic.sh > $O/pmc_traffic_c2.log 2>&1
cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json; cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary_c2.json
rm -rf gpurun_out/pmc_traffic/FETCH_SIZE gpurun_out/pmc_traffic/WRITE_SIZE
tail -3 $O/pmc_traffic_c2.log
fi
if [ -z "${SKIP_RCCL:-}" ]; then
echo "== one rank through RCCL"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads > $O/bench_c2_single_rank_rccl.json 2>$O/rccl.err; tail -1 $O/bench_c2_single_rank_rccl.json | cut -c1-200
fi
