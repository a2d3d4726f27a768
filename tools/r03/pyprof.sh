#!/bin/bash
O=gpurun_out/r03_pyprof; mkdir -p $O
timeout 300 python tools/pyprofile.py > $O/pyprofile.txt 2>&1; grep -v amdgpu $O/pyprofile.txt | sed -n '/callers of the blocking/,$p' | head -80
