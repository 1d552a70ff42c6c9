#!/usr/bin/env bash
# Round-4 call 12: tile boxes carry the count (width | height << 16): one gather in the binning's scan instead of two.  Binning tests,
# then C2 / C5 with the switch off and on, interleaved; a C5 frame timeline.
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04l; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "binning or c1_ or ragged or multi_camera or mirror_matches or tight_tile or full_size_projection or c2_full_frame or small_frame or without_intersections or graphed" > $O/pytest_sel.log 2>&1; echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $O/pytest_sel.log | head
run() { DNSPLAT_BIN_BOX_GATHER=$1 python bench.py --no-cpu-baseline --steps $2 --warmup 5 --no-strict --no-extra-workloads $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages']
print('gather=$1 |', d['value'], 'fps', d['ms_per_step'], 'ms | ' + ' '.join('%s %.4f' % (k.replace('dnsplat_',''), v['ms']) for k, v in st.items()), '| prepare %.4f emit_sort %.4f' % (st['binning']['dnsplat_bin_prepare_ms'], st['binning']['dnsplat_bin_emit_sort_ms']))"; }
echo "== c2"; for v in 0 1 0 1; do run $v 30 ""; done 2>&1 | tee $O/ab_box_gather_c2.txt
echo "== c5"; for v in 0 1 0 1; do run $v 15 "--workload c5"; done 2>&1 | tee $O/ab_box_gather_c5.txt
echo "== c5 frame timeline"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --lean > /dev/null 2> "$R/$O/prof.err"); echo "rc=$?"
f=$(find $O/prof -name '*kernel_trace.csv' | head -1); python tools/frame_timeline.py "$f" > $O/frame_timeline_c5.txt; tail -36 $O/frame_timeline_c5.txt | cut -c1-150; rm -rf $O/prof
