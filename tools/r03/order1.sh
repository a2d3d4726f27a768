#!/bin/bash
O=gpurun_out/r03_order1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "grouped or split or bit_exact" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
FPCC_EXPERIMENT=1 timeout 900 python tools/order_sweep.py 8 > $O/sweep.txt 2>&1; cat $O/sweep.txt
