#!/bin/bash
# x-window filter gradient: workgroups per launch (one round of the chip = 512; fewer = fewer slab bytes, longer pixel chunks)
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
for i in 1 2; do for T in 0 256 384 768; do
  echo -n "WGRAD_WIN_TARGET=$T run $i: "; YOLO_WGRAD_WIN_TARGET=$T python bench.py $A 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done; done
