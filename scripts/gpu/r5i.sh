#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5i; mkdir -p $O; export TMPDIR=/tmp
python scripts/step_ab.py --k 10 --rounds 4 "shared:YOLO_DYP_PER_LAYER=0" "own:YOLO_DYP_PER_LAYER=1" > $O/ab_dyp_c3.log 2>&1; echo "ab rc $?"; tail -5 $O/ab_dyp_c3.log
python scripts/step_ab.py --config c4 --k 8 --rounds 3 "shared:YOLO_DYP_PER_LAYER=0" "own:YOLO_DYP_PER_LAYER=1" > $O/ab_dyp_c4.log 2>&1; echo "ab c4 rc $?"; tail -4 $O/ab_dyp_c4.log
for i in 1 2; do for NT in 1 0; do echo -n "YOLO_NT_STORE=$NT: "; YOLO_NT_STORE=$NT python bench.py --plain --steps 10 --warmup 3 2>/dev/null | tail -1; done; done | tee $O/ab_nt_store.log
python -m pytest tests/test_gpu_keras_shell.py -x -q -k "captured or fit" > $O/t1.log 2>&1; echo "shell tests rc $?"; tail -3 $O/t1.log
