#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_gpu_elementwise.py tests/test_gpu_kats.py tests/test_gpu_loss.py "tests/test_gpu_model.py::test_model_parity" -x -q -k "not 3-True-416 and not 2-True-416" > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -8
python scripts/bench_configs.py c4 c2 c1 2>&1 | grep -v amdgpu.ids | tee $O/configs.log
python scripts/layer_table.py $O/layer_table.json > $O/layer_table.log 2>&1; grep -v amdgpu.ids $O/layer_table.log | head -70 | cut -c1-250
python bench.py --no-cpu-baseline > $O/bench.log 2>$O/bench.err; python scripts/bench_line.py $O/bench.log
