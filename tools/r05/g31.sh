#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ls /sys/devices/system/node/ | grep node; cat /sys/devices/system/node/node*/cpulist
for n in -1 0 1 -1 0 1; do timeout 200 python3 tools/r05/numa_probe.py $n 2>&1 | grep -v amdgpu; done
