#!/bin/bash
# GPU session 42: the default bench line once more (roofline.traffic now read from profiles/r04/final_pmc_traffic.json)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04F; mkdir -p $O
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; head -c 400 $O/bench_default.json; echo
