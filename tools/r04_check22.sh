#!/usr/bin/env bash
# Round-4 call 22: the driver's bench command on the final tree (profiles/pmc_traffic.json corrected).
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r04v; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; tail -1 $O/bench_c2.json | cut -c1-300
python - $O/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["value"], "fps", d["ms_per_step"], "ms |", {k.replace("dnsplat_", ""): (v["ms"], v.get("hbm_traffic")) for k, v in d["stages"].items()})
print("roofline", d.get("roofline")); print("strict", d.get("strict_index_parity", {}).get("value"))
for k, v in (d.get("extra_workloads") or {}).items(): print(k, v.get("value"), v.get("roofline"))
PY
