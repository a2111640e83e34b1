#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4s; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "stem" > $O/t1.log 2>&1; echo "stem tests rc $?"; tail -6 $O/t1.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/t2.log 2>&1; echo "model tests rc $?"; tail -3 $O/t2.log
python scripts/bench_configs.py c5 2>&1 | grep "inference forward"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/scripts/infer_bs1_graph.py > $O/prof.log 2>&1 || echo "prof failed"
cp $O/ks/*/*kernel_stats.csv $O/bs1_kernel_stats.csv 2>/dev/null; rm -rf $O/ks
python3 $R/scripts/kstats_summary.py $O/bs1_kernel_stats.csv 22 | grep -i "stem\|absmax\|total\|bn_\|gather_conv_kernel"
