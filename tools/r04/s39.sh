#!/bin/bash
# GPU session 39: by-shape table of the colour codec's convolution launches (cfg#4)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
timeout 400 python3 tools/color_by_level.py $O/color_launches.txt > $O/color_by_shape.md 2> $O/color_by_shape.err; cat $O/color_by_shape.md; tail -3 $O/color_by_shape.err
