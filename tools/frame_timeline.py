"""Per-kernel timeline of the last benchmark frame from a rocprofv3 kernel trace csv.

    python tools/frame_timeline.py gpurun_out/prof/trace_kernel_trace.csv
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "project_fwd_kernel" in n]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
    prev_end = e
print(f"frame: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
