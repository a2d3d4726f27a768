#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python3 tools/r05/int_host_profile.py tottime > $O/g24_host_profile.txt 2>&1
timeout 600 python3 tools/r05/int_host_profile.py cumulative > $O/g24_host_profile_cum.txt 2>&1
tail -3 $O/g24_host_profile.txt
