#!/bin/bash
# the FETCH_SIZE / WRITE_SIZE passes of the bench command once more, for the by-size table of the helper kernels under their round-6 names
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06P; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 3 --warmup 1 --latency-frames 2 --cpu-baseline 0 --secondary 0"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $CMD > $O/write.log 2>&1
FD=$(dirname $(find $O/fetch -name p_counter_collection.csv | head -1)); WD=$(dirname $(find $O/write -name p_counter_collection.csv | head -1))
python3 profiles/hbm_bandwidth_by_size.py $FD $WD > $O/hbm_bandwidth_by_size.md 2>&1
cat $O/hbm_bandwidth_by_size.md
rm -rf $O/fetch $O/write
