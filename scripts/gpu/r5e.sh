#!/bin/bash
python scripts/absmax_probe.py 2>&1 | grep -v amdgpu
