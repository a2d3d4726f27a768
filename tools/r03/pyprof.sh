#!/bin/bash
O=gpurun_out/r03_pyprof; mkdir -p $O
timeout 300 python tools/pyprofile.py > $O/pyprofile.txt 2>&1; grep -v amdgpu $O/pyprofile.txt | head -70
timeout 200 python tools/timeline.py 2>&1 | grep -v amdgpu | tail -8 | tee $O/timeline.txt
