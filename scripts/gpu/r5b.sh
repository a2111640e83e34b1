#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5b; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "fused_bn_backward" > $O/t1.log 2>&1; echo "bnred kernel tests rc $?"; tail -8 $O/t1.log
cd /tmp
for V in 0 1; do
  export YOLO_BN_FUSED_REDUCE=$V
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$V -- python3 $R/bench.py --plain --steps 4 --warmup 3 > $O/kt$V.log 2>&1 || echo "prof $V failed"
  python3 $R/scripts/step_timeline.py $O/kt$V 1 --json $O/timeline_fused$V.json > $O/timeline_fused$V.txt 2>&1
  cp $O/kt$V/*/*kernel_stats.csv $O/kstats_fused$V.csv 2>/dev/null
  rm -rf $O/kt$V
done
cat $O/timeline_fused0.txt; echo ======; cat $O/timeline_fused1.txt
