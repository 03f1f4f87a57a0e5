V=${1:-dev}
mkdir -p gpurun_out/r3d && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/$V/libtreeqp_amd.so
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/r3d/pmc_${V}_$i -o p -- python3 tools/prof_flat.py C4 12 > gpurun_out/r3d/pmc_${V}_$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_aggregate.py gpurun_out/r3d | grep -v "rocclr\|k_init" > gpurun_out/r3d/pmc_$V.txt
cat gpurun_out/r3d/pmc_$V.txt
