#!/bin/bash
O=gpurun_out/r03_int1; mkdir -p $O
TOP=40 timeout 300 python tools/int_launches.py > $O/launches.txt 2>&1; cat $O/launches.txt | grep -v amdgpu.ids
timeout 300 python tools/int_decode_timeline.py > $O/dec_timeline.txt 2>&1; tail -40 $O/dec_timeline.txt
