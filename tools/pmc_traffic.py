"""summary.json of tools/pmc_summary.py -> profiles/pmc_traffic.json (what bench.py's roofline.traffic reads).

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: rocprofv3 reports both in KiB, and on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced read at 64 B (MI355X_MICROARCH.md "HBM"), hence the
factor 2 on the read side; WRITE_SIZE is taken as reported (uncalibrated per the same guide).
"""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_traffic/summary.json"
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"
stage_of = {"raster_bwd_kernel": "dnsplat_raster_bwd", "raster_fwd_kernel": "dnsplat_raster_fwd",
            "project_fwd_kernel": "dnsplat_project_fwd", "project_bwd_kernel": "dnsplat_project_bwd"}
d = json.load(open(src))
out = {}
for k, stage in stage_of.items():
    if k in d and "FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k]:
        f, w = d[k]["FETCH_SIZE"], d[k]["WRITE_SIZE"]
        out[stage] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                      "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = out
json.dump(allw, open(path, "w"), indent=1)
print(json.dumps(allw, indent=1))
