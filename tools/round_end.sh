#!/usr/bin/env bash
# (Round 5 used tools/r05_set.sh; round 4 tools/r04_set.sh: same parts, PMC traffic BEFORE the bench line that reads it, a 40-step profiled run, frame timeline.)
# Everything a round's profiles/ entry is made of, in one GPU call (about 12 minutes):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/round_end.sh'   ->  gpurun_out/round_end/
# then copy: bench_*.json -> profiles/rNNx_*_bench.json, kernel_stats.csv, pmc_traffic.merged.json -> profiles/pmc_traffic.json
# (the traffic file carries the hash of the kernel sources: run this AFTER the last source change), pmc_counters.json.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/round_end; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
# SKIP_TESTS=1 / SKIP_SWEEP=1 leave out the two parts that do not depend on timing
if [ -z "${SKIP_TESTS:-}" ]; then echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log; fi
echo "== PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) for c2, c3, c5"
for w in c2 c3 c5; do
  rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
  WORKLOAD=$w BENCH_ARGS="--workload $w" bash tools/pmc_traffic.sh > $O/pmc_traffic_$w.log 2>&1
  cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json
  cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary_$w.json
done
# the same Gaussians along a Morton curve, zero rows skipped per workgroup (bench.py --scene morton; keys c2_morton / c5_morton)
for w in c2 c5; do
  rm -rf gpurun_out/pmc_traffic; mkdir -p gpurun_out/pmc_traffic
  DNSPLAT_SH_ZERO_STATE=1 WORKLOAD=${w}_morton BENCH_ARGS="--workload $w --scene morton" bash tools/pmc_traffic.sh > $O/pmc_traffic_${w}_morton.log 2>&1
  cp gpurun_out/pmc_traffic/pmc_traffic.merged.json $O/pmc_traffic.merged.json
  cp gpurun_out/pmc_traffic/summary.json $O/pmc_traffic_summary_${w}_morton.json
done
python - <<'PY'
import json
d = json.load(open("gpurun_out/round_end/pmc_traffic.merged.json"))
for w, e in d.items():
    print(w, e.get("source_sha16"), {k: round(v["hbm_bytes_per_launch"] / 1e6) for k, v in e.items() if isinstance(v, dict)})
PY
echo "== bench lines (profiles/pmc_traffic.json of this tree now matches the sources)"
python bench.py --steps 30 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-300
python bench.py --steps 20 --warmup 3 --workload c3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null; tail -1 $O/bench_c3.json | cut -c1-200
python bench.py --steps 20 --warmup 3 --workload c5 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; tail -1 $O/bench_c5.json | cut -c1-200
python bench.py --steps 20 --warmup 3 --workload c5 --no-cpu-baseline --losses fused > $O/bench_c5_fused_loss.json 2>/dev/null; tail -1 $O/bench_c5_fused_loss.json | cut -c1-200
python bench.py --steps 20 --warmup 3 --workload c5 --no-cpu-baseline --no-strict --losses torch_hip_ssim > $O/bench_c5_torch_loss_hip_ssim.json 2>/dev/null; tail -1 $O/bench_c5_torch_loss_hip_ssim.json | cut -c1-200
DNSPLAT_SH_ZERO_STATE=1 python bench.py --steps 20 --warmup 3 --workload c2 --scene morton --no-cpu-baseline --no-strict > $O/bench_c2_morton.json 2>/dev/null; tail -1 $O/bench_c2_morton.json | cut -c1-200
DNSPLAT_SH_ZERO_STATE=1 python bench.py --steps 20 --warmup 3 --workload c5 --scene morton --no-cpu-baseline --no-strict > $O/bench_c5_morton.json 2>/dev/null; tail -1 $O/bench_c5_morton.json | cut -c1-200
DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_single_rank_rccl.json 2>/dev/null; tail -1 $O/bench_c2_single_rank_rccl.json | cut -c1-200
DNSPLAT_TIGHT_TILES=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c2_gsplat_tile_boxes.json 2>/dev/null; tail -1 $O/bench_c2_gsplat_tile_boxes.json | cut -c1-200
echo "== kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --lean > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv 2>/dev/null; head -4 $O/kernel_stats.csv | cut -c1-160
echo "== vector-busy counters"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$R/$O/pmc_sq" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "sq rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/pmc_grbm" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "grbm rc=$?"
python tools/pmc_summary.py $O > $O/pmc_counters.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/round_end/pmc_counters.json"))
for k in ("raster_bwd_kernel", "raster_fwd_kernel"):
    if k in d: print(k, {c: round(v) for c, v in d[k].items() if c.startswith(("SQ_", "GRBM"))})
PY
rm -rf $O/prof $O/pmc_sq $O/pmc_grbm
if [ -z "${SKIP_SWEEP:-}" ]; then echo "== parity seed sweep (60 unseen scenes)"; timeout 900 python tools/parity_seed_sweep.py 100 30 2>&1 | grep -v amdgpu > $O/parity_seed_sweep.txt; tail -1 $O/parity_seed_sweep.txt; grep -c FAIL $O/parity_seed_sweep.txt; fi
