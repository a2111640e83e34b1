#!/bin/bash
python scripts/fit_rate.py 2>&1 | grep -v amdgpu | tail -6
python scripts/host_enqueue.py 10 2>&1 | grep -v amdgpu | tail -3
