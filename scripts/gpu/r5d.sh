#!/bin/bash
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_model.py -x -q -k "not 608 and not 416" > $O/t.log 2>&1; echo "tests rc $?"; tail -3 $O/t.log
bash scripts/gpu/r5c.sh | grep finalize
cp tf2_yolo_amd/libyolo_hip.so $O/new.so
for V in old new old new; do
  if [ $V = old ]; then cp tf2_yolo_amd/libyolo_hip_old.so.bin tf2_yolo_amd/libyolo_hip.so; else cp $O/new.so tf2_yolo_amd/libyolo_hip.so; fi
  python bench.py --no-cpu-baseline --no-kernel-timer --steps 20 > $O/c.log 2>$O/c.err; echo -n "$V: "; python scripts/bench_line.py $O/c.log
done
cp $O/new.so tf2_yolo_amd/libyolo_hip.so; rm $O/new.so
