#!/usr/bin/env bash
# Round-4 call 9: frame timeline (launch gaps), the one-rank-through-RCCL step under the profiler, forward T' = T - alpha T A/B.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04i; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== frame timeline (graph replay, lean)"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof.err"); echo "rc=$?"
f=$(find $O/prof -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py "$f" > $O/frame_timeline_c2.txt; tail -45 $O/frame_timeline_c2.txt; rm -rf $O/prof
echo "== one rank through RCCL under the profiler"
(cd /tmp && DNSPLAT_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof2" -o trace -- python "$R/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof2.err"); echo "rc=$?"
f=$(find $O/prof2 -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_rccl.csv
f=$(find $O/prof2 -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py "$f" > $O/frame_timeline_rccl.txt; tail -60 $O/frame_timeline_rccl.txt; rm -rf $O/prof2
echo "== A/B c2: cur | fts"
DNSPLAT_LIB=$R/gpurun_ab/lib_fts.so timeout 600 python -m pytest tests -m gpu -q -k "c1_ or mirror_matches or golden or tight_tile or saturated" 2>&1 | tail -2
STEPS=30 BENCH_ARGS="--no-extra-workloads --no-strict" bash tools/ab_libs.sh cur fts 2>&1 | grep -v amdgpu | tee $O/ab_libs_c2.txt
