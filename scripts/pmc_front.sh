# usage: bash scripts/pmc_front.sh <tag>   (GPU box) -- the same PMC passes as pmc_round.sh for the narrow layers at the front
# of the network (208x208 / 416x416, 32-64 channels), where the step is furthest from its floors
TAG=${1:-r02_front}
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_layers_pmc.jsonl; : > $OUT
GROUPS_=("GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE")
run_one() {   # <label> <kernel substring> <mode> <layer>
  local L=$1 SUB=$2 MODE=$3 LAYER=$4
  for P in "${GROUPS_[@]}"; do
    D=/tmp/pmc_${TAG}_$$; rm -rf $D
    ( cd /tmp && timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- $R/scripts/hip_probe/conv_bench.bin $MODE 0 1 10 1 $LAYER > /dev/null 2>&1 ) || echo "pass failed: $L $P"
    echo "{\"label\": \"$L\", \"mode\": \"$MODE\", \"layer_H_Cin_Cout_k_s_N\": \"$LAYER\", \"pass\": $(python3 $R/scripts/pmc_collect.py $D "$SUB")}" >> $OUT
    rm -rf $D
  done
}
run_one l4_fwd_3x3_208_32to64 gather_conv_planes_kernel fwd 208,32,64,3,1,32
run_one l4_dgrad_3x3_208_32to64 gather_conv_planes_kernel dgrad 208,32,64,3,1,32
run_one l2_fwd_3x3s2_416_32to64 gather_conv_planes_kernel fwd 416,32,64,3,2,32
run_one l2_dgrad_3x3s2_416_32to64 gather_conv_planes_kernel dgrad 416,32,64,3,2,32
run_one l5_fwd_3x3s2_208_64to128 gather_conv_planes_kernel fwd 208,64,128,3,2,32
run_one l7_dgrad_3x3_104_64to128 gather_conv_planes_kernel dgrad 104,64,128,3,1,32
run_one l2_wgrad_3x3s2_416_32to64 wgrad_planes_kernel wgrad 416,32,64,3,2,32
run_one l4_wgrad_3x3_208_32to64 wgrad_planes_kernel wgrad 208,32,64,3,1,32
python3 $R/scripts/pmc_json.py $OUT > $R/gpurun_out/${TAG}_layers_pmc.json
