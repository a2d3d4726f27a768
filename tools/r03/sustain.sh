#!/bin/bash
O=gpurun_out/r03_sustain; mkdir -p $O
REPS=60 KEEP=1 timeout 600 python tools/sustain_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/sustain_keep.md
REPS=60 KEEP=0 timeout 600 python tools/sustain_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/sustain_idle.md
