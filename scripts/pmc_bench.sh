# usage: bash scripts/pmc_bench.sh <tag> <kernel substring> <conv_bench args...>
# rocprofv3 PMC passes (one counter group per run, as the gfx950 slot table requires) over the standalone
# conv benchmark; per-launch sums -> gpurun_out/pmc_<tag>.txt
TAG=$1; SUB=$2; shift 2
R=$PWD; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_${TAG}.txt; : > $OUT
for P in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU" \
         "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INST_LEVEL_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  D=$R/gpurun_out/pmcb_${TAG}_$(echo $P | cut -d' ' -f1)
  rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- $R/scripts/hip_probe/conv_bench.bin "$@" > $D.log 2>&1 || echo "pass failed: $P" >> $OUT
  python3 $R/scripts/pmc_summary.py $D "$SUB" | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}' >> $OUT
  rm -rf $D
done
cd $R
