#!/bin/bash
# same-box A/B: process pinned to the GPU's NUMA node (default) against unpinned
O=gpurun_out/r03_numa; mkdir -p $O; rm -f $O/ab.txt
python -c "
import torch
p = torch.cuda.get_device_properties(0)
a = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
print('GPU', a, 'numa', open(f'/sys/bus/pci/devices/{a}/numa_node').read().strip(), 'cpus', open(f'/sys/bus/pci/devices/{a}/local_cpulist').read().strip())
" 2>&1 | grep GPU | tee -a $O/ab.txt
for rep in 1 2 3; do
  for d in 1 0; do
    FPCC_NUMA_BIND=$d timeout 300 python bench.py --secondary 0 --cpu-baseline 0 --steps 20 --warmup 5 2>&1 | tail -1 > $O/b.json
    echo "bind=$d $(grep -o '"ms_per_step": [0-9.]*\|"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*\|"kernel_ms_per_step": [0-9.]*\|"host_binding": [^}]*}' $O/b.json | tr '\n' ' ')" | tee -a $O/ab.txt
  done
done
