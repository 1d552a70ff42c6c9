#!/usr/bin/env bash
# per-kernel time of a short bench run (rocprofv3 --kernel-trace --stats): tools/kstats.sh <outdir> [bench args...]
R=$(pwd); O=$1; shift; mkdir -p "$R/$O"; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --lean "$@" > "$R/$O/prof_bench.json" 2> "$R/$O/prof.err")
cp $(ls $R/$O/prof/*/*kernel_stats.csv $R/$O/prof/*kernel_stats.csv 2>/dev/null | head -1) $R/$O/kernel_stats.csv
rm -rf $R/$O/prof
python3 - "$R/$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):6.2f}")
PY
