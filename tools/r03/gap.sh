#!/bin/bash
O=gpurun_out/r03_gap; mkdir -p $O
timeout 600 python tools/gap_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gap.md
