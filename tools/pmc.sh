#!/usr/bin/env bash
# PMC passes over the bench (separate rocprofv3 runs: SQ has 8 slots, FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/pmc.sh'
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
run() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc/$name" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$R/gpurun_out/pmc/$name.json" 2> "$R/gpurun_out/pmc/$name.err"); echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python - <<'PY'
import csv, glob, collections, json, os
R = os.getcwd()
out = {}
for d in sorted(glob.glob("gpurun_out/pmc/*/")):
    for f in glob.glob(d + "*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-60:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k].add(row["Dispatch_Id"])
        for k in acc:
            if not any(s in k for s in ("raster", "project", "radix", "emit")): continue
            n = len(cnt[k])
            out.setdefault(k, {}).update({c: v / n for c, v in acc[k].items()}); out[k]["launches"] = n
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc/summary.json", "w"), indent=1)
PY
