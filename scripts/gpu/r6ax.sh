#!/bin/bash
# conv_small.hip with the all-blocks-at-once epilogue: tests under every tile, then bs-1 predict
mkdir -p gpurun_out/r6ax
bash scripts/gpu/r6aw.sh || exit 1
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "infer or predict or keras_shell or readme" > gpurun_out/r6ax/t.log 2>&1 || { tail -30 gpurun_out/r6ax/t.log; exit 1; }
tail -1 gpurun_out/r6ax/t.log
for i in 1 2 3; do python scripts/infer_bs1_graph.py 2>/dev/null >> gpurun_out/r6ax/ms.log; done
cat gpurun_out/r6ax/ms.log
