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
# binning = every kernel of dnsplat_bin_prepare / dnsplat_bin_emit_sort, per FRAME (launches x per-launch average / frames)
bin_kernels = [k for k in d if k.startswith(("radix_", "emit_", "scan_", "depth_keys", "tile_", "set_u32"))]
if bin_kernels and all("FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k] for k in bin_kernels):
    frames = max(1, d.get("raster_fwd_kernel", {}).get("launches_seen", 1))
    f = sum(d[k]["FETCH_SIZE"] * d[k]["launches_seen"] for k in bin_kernels) / frames
    w = sum(d[k]["WRITE_SIZE"] * d[k]["launches_seen"] for k in bin_kernels) / frames
    out["binning"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                      "kernels": sorted(bin_kernels), "note": "sum over the stage's kernels per frame"}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402  (only for the hash of the kernel sources the numbers belong to)
out["source_sha16"] = bench.kernel_source_sha16()
out["calibration"] = ("WRITE_SIZE checked 1:1 on 1 GiB streaming fills / copies and on 12-byte-per-lane row stores; FETCH_SIZE "
                      "reports 1/2 of a streaming read (tools/pmc_calibrate.py, profiles/r02_pmc_calibration.json)")
path = os.path.join(root, "profiles", "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = out
json.dump(allw, open(path, "w"), indent=1)
# gpurun only brings gpurun_out/ back: leave the merged file next to the summary as well (copy it to profiles/ by hand)
json.dump(allw, open(os.path.join(os.path.dirname(os.path.abspath(src)), "pmc_traffic.merged.json"), "w"), indent=1)
print(json.dumps(allw, indent=1))
