#!/bin/bash
# GPU session 22: hunt the intermittent encoder failure: many short depth-1 bench runs (the two sightings were such runs); the exception now says which job
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
fails=0
for i in $(seq 1 36); do
  timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/h$i.json 2> $O/h$i.err
  rc=$?
  if [ $rc != 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; tail -5 $O/h$i.err; else rm -f $O/h$i.err; fi
done
echo "36 runs, $fails failures"
