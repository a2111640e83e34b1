#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
python -m pytest tests/test_gpu_elementwise.py "tests/test_gpu_model.py::test_model_parity" tests/test_gpu_keras_shell.py -x -q -k "not 608 and not 416" > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -6
for V in 1 0 1 0; do YOLO_CONCAT_PLANES=$V python scripts/bench_configs.py c4 2>&1 | grep images_per_s | sed "s/^/concat_planes=$V /"; done
for V in 1 0; do YOLO_CONCAT_PLANES=$V python bench.py --no-cpu-baseline --steps 20 > $O/bench_$V.log 2>$O/bench_$V.err; python scripts/bench_line.py $O/bench_$V.log; done
python scripts/infer_bs1_graph.py 2>&1 | grep -v amdgpu | tail -3
