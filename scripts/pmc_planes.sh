# usage: bash scripts/pmc_planes.sh <tag> <one_conv args...>
TAG=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
for P in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INST_LEVEL_VMEM"; do
  D=$R/gpurun_out/pmc_${TAG}_$(echo $P | cut -d' ' -f1)
  rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_conv.py "$@" > $D.log 2>&1 || echo "pass failed: $P"
done
cd $R; for d in gpurun_out/pmc_${TAG}_*/; do python3 scripts/pmc_summary.py $d conv; done | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}' > gpurun_out/pmc_${TAG}_summary.txt 2>&1
