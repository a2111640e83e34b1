#!/bin/bash
# full GPU suite + smoke, as the driver runs them
O=gpurun_out/r3k; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/tests.log 2>&1; echo "tests rc $?"; grep -v "frame #" $O/tests.log | tail -12
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2
