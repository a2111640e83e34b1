# usage: bash scripts/pmc_sq_one.sh <one_conv args...>  -> SQ counters per launch of the conv kernel (one pass per group)
R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for P in ${PMC_GROUPS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC" "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM"}; do
  i=$((i+1)); D=$R/gpurun_out/pmc_sq_$i
  rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_conv.py "$@" > $D.log 2>&1 || echo "pass failed: $P"
  python3 $R/scripts/pmc_counters.py $D conv
done
