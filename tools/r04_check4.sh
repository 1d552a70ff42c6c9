#!/usr/bin/env bash
# Round-4 call 4: loss-kernel / post-op load batching, empty-frame fix; per-kernel statistics for c2 and c5 + fused losses.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04d; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -k "fused_loss or without_intersections or mirror or reference or depth_normals or c1_raster" > $O/pytest_sel.log 2>&1; echo "rc=$?"; grep -E "^E |passed|failed" $O/pytest_sel.log | head -20
echo "== bench c5 fused loss / c5 / c2"
for cfg in "c5 --losses fused" "c5" "c2"; do
  python bench.py --workload $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 > $O/bench_$(echo $cfg | tr -d ' -').json
  python - "$O/bench_$(echo $cfg | tr -d ' -').json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
gs = sum(v["ms"] for v in d["stages"].values())
print(sys.argv[1], d["value"], "fps", d["ms_per_step"], "ms | stages", {k.replace("dnsplat_", ""): v["ms"] for k, v in d["stages"].items()}, "| other", round(d["ms_per_step"] - gs, 4))
PY
done
echo "== kernel stats c2 / c5 fused loss"
for cfg in "c2" "c5 --losses fused"; do
  tag=$(echo $cfg | tr -d ' -')
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$tag" -o trace -- python "$R/bench.py" --workload $cfg --steps 10 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof_$tag.err"); echo "rocprof $tag rc=$?"
  f=$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_$tag.csv; rm -rf $O/prof_$tag
  python - $O/kernel_stats_$tag.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{r['Name'][:100]:100s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']}")
PY
done
