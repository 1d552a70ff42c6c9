"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*/p_counter_collection.csv) per kernel, averaged per launch."""
import collections, csv, glob, json, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
names = ["raster_bwd_kernel", "raster_fwd_kernel", "project_bwd_kernel", "project_fwd_kernel", "radix_scatter_kernel",
         "radix_hist_kernel", "radix_hist_ranges_kernel", "emit_prep_kernel", "radix_scan_kernel", "tile_offsets_fill_kernel", "tile_first_init_kernel",
         "scan_sums_kernel", "scan_sums_excl_kernel", "scan_final_kernel", "depth_keys_kernel", "set_u32_kernel",
         "dn_depth_normals_kernel", "sh_factors_kernel", "densify"]
out = {}
for f in sorted(glob.glob(root + "/*/p_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
    meta = {}
    calls = set()
    for row in csv.DictReader(open(f)):
        k = next((n for n in names if n in row["Kernel_Name"]), None)
        if k is None: continue
        # dnsplat_raster_bwd launches its clamping kernel (last template argument true) and, for the fused pass, the clamp-free
        # twin (false); one of the two leaves at once.  Per CALL of the entry point their counters add up, and a call is
        # counted by its one `true` dispatch.
        # template arguments: D, SPLIT, DN, COUNT, MASKS, CLAMP_LOOP[, DET] — the clamp flag is the SIXTH one (since round 4 a DET
        # flag follows it; taking the last argument counted both twins as calls and halved every per-call figure of this kernel)
        m = re.search(r"raster_bwd_kernel<([^>]*)>", row["Kernel_Name"])
        if m:
            targs = [x.strip() for x in m.group(1).split(",")]
            if (targs[5] if len(targs) > 5 else targs[-1]) == "true": calls.add(row["Dispatch_Id"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); ids[k].add(row["Dispatch_Id"])
        meta[k] = dict(VGPR=row["VGPR_Count"], SGPR=row["SGPR_Count"], LDS=row["LDS_Block_Size"], grid=row["Grid_Size"], wg=row["Workgroup_Size"])
    for k in acc:
        n = len(ids[k])
        if k == "raster_bwd_kernel" and calls: n = len(calls)
        out.setdefault(k, {}).update({c: v / n for c, v in acc[k].items()})
        out[k]["launches_seen"] = n; out[k].update(meta[k])
print(json.dumps(out, indent=1))
