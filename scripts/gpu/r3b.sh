#!/bin/bash
# round-3 GPU batch b: patch-kernel tests, same-box A/B of the patch form (C3 bench, C4), gradient-excess with forced pools
set -o pipefail
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests/test_gpu_keras_shell.py "tests/test_gpu_conv.py::test_conv_patch_window_kernel_fwd_dgrad" "tests/test_gpu_conv.py::test_conv_window_kernel_fwd_dgrad" "tests/test_gpu_conv.py::test_conv_split_k" -x -q > $O/tests1.log 2>&1; echo "tests1 rc $?"; tail -3 $O/tests1.log
python bench.py --no-cpu-baseline > $O/bench_patch1.log 2>$O/bench_patch1.err || exit 1
YOLO_CONV_PATCH=0 python bench.py --no-cpu-baseline > $O/bench_patch0.log 2>$O/bench_patch0.err || exit 1
python bench.py --no-cpu-baseline > $O/bench_patch1b.log 2>$O/bench_patch1b.err || exit 1
python scripts/bench_configs.py c4 > $O/c4_patch1.log 2>&1 || exit 1
YOLO_CONV_PATCH=0 python scripts/bench_configs.py c4 > $O/c4_patch0.log 2>&1 || exit 1
python scripts/grad_excess.py 4 608 1 $O/ge_v4_608_1_planes.json > $O/ge_planes.log 2>&1 || exit 1
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q --durations=8 > $O/tests2.log 2>&1; echo "tests2 rc $?"; tail -15 $O/tests2.log
grep -h images_per_s $O/c4_*.log
for f in $O/bench_patch*.log; do python scripts/bench_line.py $f; done
