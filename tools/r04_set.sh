#!/usr/bin/env bash
# Round-4 measurement set in ONE GPU call: full GPU suite, the driver's bench line (with strict / c3 / c5 sections), the
# rocprofv3 kernel stats of the same command, PMC traffic for c2, the one-rank-through-RCCL line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'TAG=r04a bash tools/r04_set.sh'  ->  gpurun_out/$TAG/
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/${TAG:-r04a}; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then echo "== pytest -m gpu"; DNSPLAT_MARGIN_LOG=$R/$O/margins.tsv timeout 1200 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_gpu.log | head -20; fi
if [ -z "${SKIP_PMC:-}" ]; then
echo "== PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes; updates profiles/pmc_traffic.json in place so that the bench line below reports it)"
for w in ${PMC_WORKLOADS:-c2 c3 c5}; do
  rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
  WORKLOAD=$w BENCH_ARGS="--workload $w" bash tools/pmc_traffic.sh > $O/pmc_traffic_$w.log 2>&1
  cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json; cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary_$w.json
  rm -rf gpurun_out/pmc_traffic
done
python - $O/pmc_traffic.merged.json <<'PY2'
import json, sys
d = json.load(open(sys.argv[1]))
for w, e in d.items():
    print(w, e.get("source_sha16"), {k: round(v["hbm_bytes_per_launch"] / 1e6) for k, v in e.items() if isinstance(v, dict)})
PY2
fi
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
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --lean > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(f"{r['Name'][:90]:90s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} us {r['Percentage']}")
PY
if [ -z "${SKIP_RCCL:-}" ]; then
echo "== one rank through RCCL"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads > $O/bench_c2_single_rank_rccl.json 2>$O/rccl.err; tail -1 $O/bench_c2_single_rank_rccl.json | cut -c1-200
fi
echo "== frame timeline c2"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof3" -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof3.err"); echo "rc=$?"
f=$(find $O/prof3 -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py "$f" > $O/frame_timeline_c2.txt; rm -rf $O/prof3; tail -3 $O/frame_timeline_c2.txt
