cd "${GRAFT_REPO_ROOT:-.}"; R=$(pwd); mkdir -p gpurun_out/pmc2; export TMPDIR=/tmp
run() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc2/$name" -o p -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc2/$name.json" 2> "$R/gpurun_out/pmc2/$name.err"); echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
run sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_IFETCH
