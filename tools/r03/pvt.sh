#!/bin/bash
O=gpurun_out/r03_pvt; mkdir -p $O
timeout 600 python tools/probe_vs_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/probe_vs_trace.md
