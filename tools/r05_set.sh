#!/usr/bin/env bash
# Round-5 measurement set in ONE GPU call (about 12 minutes): full GPU suite with the row-relative bounds asserted (statistics and
# margins logged), the driver's bench line (eager drop-in, strict, C3 / C5 / losses / anisotropic / train-loop sections), the rocprofv3
# kernel stats of the same command, the frame timeline, one rank through RCCL, the seed sweeps (deterministic x3 and default).
# The kernel sources did not change in round 5: profiles/pmc_traffic.json (hash-keyed) and the SQ counters of r04e stay valid;
# PMC=1 re-measures the traffic anyway (tools/pmc_traffic.sh, separate FETCH_SIZE / WRITE_SIZE passes).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'TAG=r05c bash tools/r05_set.sh'  ->  gpurun_out/$TAG/
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/${TAG:-r05c}; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then echo "== pytest -m gpu"; t0=$(date +%s)
DNSPLAT_ROWREL_LOG=$R/$O/rowrel.tsv DNSPLAT_MARGIN_LOG=$R/$O/margins.tsv timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; grep -E "^E  |passed|failed|^FAILED|BOUND EXCEEDED" $O/pytest_gpu.log | head -20; fi
if [ -n "${PMC:-}" ]; then
for w in ${PMC_WORKLOADS:-c2}; do
  rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
  WORKLOAD=$w BENCH_ARGS="--workload $w" bash tools/pmc_traffic.sh > $O/pmc_traffic_$w.log 2>&1
  cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json; rm -rf gpurun_out/pmc_traffic
done; fi
echo "== bench (driver's command)"; t0=$(date +%s)
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["value"], "fps", d["ms_per_step"], "ms |", {k.replace("dnsplat_", ""): v["ms"] for k, v in d["stages"].items()})
print("roofline", d.get("roofline")); print("strict", {k: d.get("strict_index_parity", {}).get(k) for k in ("value", "launch", "ms_per_step")})
print("eager", d.get("eager_drop_in"))
for k, v in (d.get("extra_workloads") or {}).items(): print(k, v.get("value"), v.get("ms_per_step"), str(v.get("launch"))[:70], v.get("error"))
print("valu", {k: v for k, v in (d.get("roofline_valu") or {}).items() if k.startswith("dnsplat")}); print("cpu", d.get("cpu_baseline"))
PY
echo "== the other ways in (INTEGRATION.md A / B): two drop-in calls inside the reference's op sequence; fused pass with torch post-ops"
for m in --two-call --torch-postops; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads $m 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['value'], 'frames/s', d['ms_per_step'], 'ms |', d['launch'][:50])" | tee -a $O/bench_modes.txt
done
echo "== kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --lean > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/prof
python - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} us {r['Percentage']}")
PY
echo "== frame timeline c2"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof3" -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof3.err"); echo "rc=$?"
f=$(find $O/prof3 -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py "$f" > $O/frame_timeline_c2.txt; rm -rf $O/prof3; tail -2 $O/frame_timeline_c2.txt
echo "== one rank through RCCL (one slice: the default)"
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict --no-extra-workloads > $O/bench_c2_single_rank_rccl.json 2>$O/rccl.err; echo "rc=$?"
python - $O/bench_c2_single_rank_rccl.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("rccl", d["value"], d["ms_per_step"], json.dumps(d.get("multi_gpu"))[:700])
PY
echo "== the whole --gpus 2 flow of bench.py with two REAL ranks sharing this GPU (gloo carries the collectives: RCCL refuses two ranks"
echo "   per device) — not a performance figure: the N > 1 code path of the driver's scaling run, end to end, for the first time"
DNSPLAT_DIST_BACKEND=gloo DNSPLAT_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_c2_two_ranks_one_gpu_gloo.json 2> $O/two_ranks.err; echo "rc=$?"
python - $O/bench_c2_two_ranks_one_gpu_gloo.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print("two ranks / one GPU / gloo:", d["value"], "fps (2 cameras)", d["ms_per_step"], "ms |", d["launch"][:80]); print(json.dumps(d.get("multi_gpu"))[:900])
except Exception as e: print("unreadable", e)
PY
grep -v "amdgpu.ids\|socket.cpp\|UserWarning\|run_backward\|^\[W\|OMP_NUM_THREADS\|^\*\*\*" $O/two_ranks.err | tail -5 | cut -c1-300
if [ -z "${SKIP_SWEEP:-}" ]; then
echo "== parity seed sweeps (60 unseen scenes): deterministic mode x3 runs, default mode"
DNSPLAT_DETERMINISTIC=1 timeout 900 python tools/parity_seed_sweep.py 100 30 3 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_deterministic.txt; tail -2 $O/parity_seed_sweep_deterministic.txt
timeout 600 python tools/parity_seed_sweep.py 100 30 1 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_default.txt; tail -2 $O/parity_seed_sweep_default.txt
fi
