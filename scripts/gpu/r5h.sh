#!/bin/bash
python scripts/instep_1x1.py 2>&1 | grep -v amdgpu | head -44
