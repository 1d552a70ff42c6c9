"""Per-kernel, per-launch averages of every counter in a rocprofv3 --pmc output directory (any kernel names)."""
import collections, csv, glob, json, sys
root = sys.argv[1]
out = {}
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); ids = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); ids[k].add(row["Dispatch_Id"])
    for k in acc:
        n = len(ids[k])
        out.setdefault(k, {}).update({c: round(v / n, 1) for c, v in acc[k].items()})
        out[k]["launches"] = n
print(json.dumps(out, indent=1))
