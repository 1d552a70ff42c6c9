#!/usr/bin/env bash
# Second session of round 6, final tree: tools/round_end.sh (GPU suite, PMC traffic incl. the Morton scenes, bench lines, kernel stats, counters,
# default-mode seed sweep), then the driver's own command, a 40-step profiled run of it and the deterministic seed sweep.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); export TMPDIR=/tmp
bash tools/round_end.sh
O=gpurun_out/r06s_final; rm -rf $O; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-300
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --no-strict --no-extra-workloads > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err"); echo "rocprof rc=$?"
cp $(ls $O/prof/*/*kernel_stats.csv $O/prof/*kernel_stats.csv 2>/dev/null | head -1) $O/c2_kernel_stats_40steps.csv; rm -rf $O/prof; head -4 $O/c2_kernel_stats_40steps.csv | cut -c1-200
DNSPLAT_DETERMINISTIC=1 timeout 1200 python tools/parity_seed_sweep.py 100 30 3 2>&1 | grep -v amdgpu > $O/parity_seed_sweep_deterministic.txt; tail -3 $O/parity_seed_sweep_deterministic.txt | cut -c1-300
