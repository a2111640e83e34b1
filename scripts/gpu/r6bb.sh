#!/bin/bash
# does the runtime's kernel-argument placement change the per-launch floor? (HIP_FORCE_DEV_KERNARG: kernargs in device memory)
mkdir -p gpurun_out/r6bb
L=gpurun_out/r6bb/ab.log
for v in 0 1 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> $L
  HIP_FORCE_DEV_KERNARG=$v python scripts/infer_bs1_graph.py 2>/dev/null >> $L
  HIP_FORCE_DEV_KERNARG=$v python scripts/bench_configs.py c1 c2 2>/dev/null | cut -c1-140 >> $L
  HIP_FORCE_DEV_KERNARG=$v python bench.py --plain --steps 20 --warmup 5 2>/dev/null | tail -n 1 | cut -c1-80 >> $L
done
cat $L
