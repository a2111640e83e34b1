#!/bin/bash
# step A/B: per-tap / window 32x32x16 / window 16x16x32 filter gradients; then one-stream kernel stats of the step
O=$PWD/gpurun_out/r6m; mkdir -p $O; R=$PWD
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
for i in 1 2; do for V in 0 1 3; do
  echo -n "WGRAD_WIN=$V run $i: "; YOLO_WGRAD_WIN=$V python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
cd /tmp; export TMPDIR=/tmp
YOLO_BWD_OVERLAP=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks1 -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-extra-blocks > $O/ks1.log 2>&1
cp $O/ks1/*/*kernel_stats.csv $O/serial_kernel_stats.csv; rm -rf $O/ks1
