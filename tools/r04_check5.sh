#!/usr/bin/env bash
# Round-4 call 5: fused depth-sort histograms + gather de-duplication + loss-kernel atomics — full suite, A/B, c5 fused loss.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04e; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== pytest -m gpu (full)"; DNSPLAT_MARGIN_LOG=$O/margins.tsv timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_gpu.log | head -20
echo "== library A/B in the frame: prev (batched loads) | nofuse (+ gather de-dup) | cur (+ fused depth histograms)"
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh prev nofuse cur 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
STEPS=20 BENCH_ARGS="--workload c5 --no-strict" bash tools/ab_libs.sh prev nofuse cur 2>&1 | grep -v amdgpu | tee $O/ab_libs_c5.txt
echo "== c5 + fused losses"
python bench.py --workload c5 --losses fused --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads 2>/dev/null | tail -1 > $O/bench_c5_fused_loss.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e/bench_c5_fused_loss.json").read().strip().splitlines()[-1])
gs = sum(v["ms"] for v in d["stages"].values())
print("c5 fused loss", d["value"], "fps", d["ms_per_step"], "ms | stages", {k.replace("dnsplat_", ""): v["ms"] for k, v in d["stages"].items()}, "| other", round(d["ms_per_step"] - gs, 4))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --workload c5 --losses fused --steps 10 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_c5_fused_loss.csv; rm -rf $O/prof
python - $O/kernel_stats_c5_fused_loss.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    print(f"{r['Name'][:100]:100s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']}")
PY
