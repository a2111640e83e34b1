#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
python scripts/step_ab.py --k 10 --rounds 4 "off:YOLO_BN_FUSED_REDUCE=0" "all:YOLO_BN_FUSED_REDUCE=1" "w:YOLO_BN_FUSED_REDUCE=w" "ws:YOLO_BN_FUSED_REDUCE=ws" "wsh:YOLO_BN_FUSED_REDUCE=wsh" "wsp:YOLO_BN_FUSED_REDUCE=wsp" > $O/ab_c3.log 2>&1; echo "ab rc $?"; tail -8 $O/ab_c3.log
python scripts/step_ab.py --config c4 --k 8 --rounds 3 "off:YOLO_BN_FUSED_REDUCE=0" "all:YOLO_BN_FUSED_REDUCE=1" "ws:YOLO_BN_FUSED_REDUCE=ws" "wsp:YOLO_BN_FUSED_REDUCE=wsp" > $O/ab_c4.log 2>&1; echo "ab c4 rc $?"; tail -6 $O/ab_c4.log
