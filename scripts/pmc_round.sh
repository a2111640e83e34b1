# usage: bash scripts/pmc_round.sh <round tag, e.g. r02>   (GPU box)
# PMC evidence for the SHIPPED conv kernels on real YOLOv3-416 bs-32 layers: one rocprofv3 pass per counter group
# (gfx950 slot limits; --pmc never combined with other trace domains), the standalone benchmark binary directly
# after `--`. Result: gpurun_out/<tag>_conv_pmc.json (copy to profiles/).
TAG=${1:-r02}
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_conv_pmc.jsonl; : > $OUT
GROUPS_=("GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE")
run_one() {   # <label> <kernel substring> <win option> <mode> <layer>
  local L=$1 SUB=$2 WIN=$3 MODE=$4 LAYER=$5
  for P in "${GROUPS_[@]}"; do
    D=/tmp/pmc_${TAG}_$$; rm -rf $D
    ( cd /tmp && timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- $R/scripts/hip_probe/conv_bench.bin $MODE 0 $WIN 10 1 $LAYER > /dev/null 2>&1 ) || echo "pass failed: $L $P"
    echo "{\"label\": \"$L\", \"mode\": \"$MODE\", \"layer_H_Cin_Cout_k_s_N\": \"$LAYER\", \"pass\": $(python3 $R/scripts/pmc_collect.py $D "$SUB")}" >> $OUT
    rm -rf $D
  done
}
run_one win128_52 conv_win_kernel 1 fwd 52,128,256,3,1,32
run_one win128_26 conv_win_kernel 1 fwd 26,256,512,3,1,32
run_one win256_13 conv_win_kernel 1 fwd 13,512,1024,3,1,32
run_one win128_52_dgrad conv_win_kernel 1 dgrad 52,128,256,3,1,32
run_one patch128x128_104 conv_win_kernel 1 fwd 104,64,128,3,1,32
run_one planes128x128_1x1_52 gather_conv_planes_kernel 1 fwd 52,256,128,1,1,32
run_one planes128x64_1x1_104 gather_conv_planes_kernel 1 fwd 104,128,64,1,1,32
run_one planes128x32_1x1_208 gather_conv_planes_kernel 1 fwd 208,64,32,1,1,32
# filter gradients: the x-window kernel (conv_wgrad_win.hip, the default for 3x3 stride-1 layers; slabs + ordered reduce as in
# the training step) and, with YOLO_WGRAD_WIN=0, the per-tap kernel it replaced
export CONV_BENCH_WGRAD_WS=1
run_one wgradwin_52 wgrad_win_kernel 1 wgrad 52,128,256,3,1,32
run_one wgradwin_26 wgrad_win_kernel 1 wgrad 26,256,512,3,1,32
run_one wgradwin_13 wgrad_win_kernel 1 wgrad 13,512,1024,3,1,32
export YOLO_WGRAD_WIN=0
run_one wgrad128x128_52 wgrad_planes_kernel 1 wgrad 52,128,256,3,1,32
run_one wgrad128x128_26 wgrad_planes_kernel 1 wgrad 26,256,512,3,1,32
run_one wgrad128x256_13 wgrad_planes_kernel 1 wgrad 13,512,1024,3,1,32
unset YOLO_WGRAD_WIN
run_one wgrad64x128_208 wgrad_planes_kernel 1 wgrad 208,32,64,3,1,32
python3 $R/scripts/pmc_json.py $OUT > $R/gpurun_out/${TAG}_conv_pmc.json
