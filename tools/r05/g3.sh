#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
TOP=45 python tools/int_launches.py > gpurun_out/r05/g3_int_launches.log 2>&1
cat gpurun_out/r05/g3_int_launches.log
