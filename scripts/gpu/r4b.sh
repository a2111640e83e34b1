#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4b; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "inference_unit or epilogue or split_planes" > $O/t1.log 2>&1; echo "conv tests rc $?"; tail -5 $O/t1.log
cd /tmp
for V in 0; do
  YOLO_REDUCE_DBG=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$V -- python3 $R/scripts/infer_bs1_graph.py > $O/prof$V.log 2>&1 || echo "prof failed"
  cp $O/ks$V/*/*kernel_stats.csv $O/ks_$V.csv 2>/dev/null; rm -rf $O/ks$V
  echo "dbg=$V"; grep "graph replay" $O/prof$V.log
  python3 $R/scripts/kstats_summary.py $O/ks_$V.csv 3 | grep reduce
done
cd $R
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/t2.log 2>&1; echo "model tests rc $?"; tail -5 $O/t2.log
python scripts/bench_configs.py c5 2>&1 | grep "inference forward"
