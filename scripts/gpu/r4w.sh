#!/bin/bash
R=$PWD; O=$R/gpurun_out/r4w; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_conv.py -x -q -k "stem" > $O/t1.log 2>&1; echo "stem tests rc $?"; tail -4 $O/t1.log
YOLO_STEM_MFMA=8 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -x -q -k "stem or (3- and not 416) or fwd" > $O/t2.log 2>&1; echo "mfma-forward tests rc $?"; tail -3 $O/t2.log
python scripts/bench_configs.py c5 2>&1 | grep "inference forward"
i=0
for V in 0 8 0 8; do
  i=$((i+1))
  YOLO_STEM_MFMA=$V python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/bench_$i.log 2>$O/bench_$i.err || { tail -5 $O/bench_$i.err; exit 1; }
  echo -n "stem_mfma=$V: "; python scripts/bench_line.py $O/bench_$i.log
done
