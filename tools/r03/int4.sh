#!/bin/bash
O=gpurun_out/r03_int4; mkdir -p $O
timeout 200 python tools/timeline_int.py 2>&1 | grep -v amdgpu | tee $O/timeline.txt
timeout 200 python tools/int_trace.py 2>&1 | grep -v amdgpu | tee $O/trace.txt
timeout 200 python tools/pyprofile_int.py enc 2>&1 | grep -v amdgpu | head -45 | tee $O/pyprof_enc.txt
timeout 200 python tools/int_decode_timeline.py 2>&1 | grep -v amdgpu | tail -15 | tee $O/dec_timeline.txt
