# VALU counters of the compositing forward for two library builds (is a packed-fp32 instruction one issue slot?)
cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); O=gpurun_out/r06s/pmc_fwd; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
for v in base fwdpk; do
  export DNSPLAT_LIB=$R/gpurun_ab/lib_$v.so
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$R/$O/$v/sq" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "$v sq rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$R/$O/$v/grbm" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1); echo "$v grbm rc=$?"
  python tools/pmc_summary.py $O/$v > $O/$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/$v.json"))
k="raster_fwd_kernel"
print("$v", {c: round(x) for c, x in d[k].items() if c.startswith(("SQ_","GRBM"))})
PY
  rm -rf $O/$v
done
