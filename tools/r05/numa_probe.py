#!/usr/bin/env python3
"""Does the host side of the integer codec (pinned staging buffers, the serial rANS decoder) care which NUMA node it runs on?
usage: numa_probe.py <node | -1>"""
import os, sys, glob, time
node = int(sys.argv[1])


def cpus_of(node):
    out = []
    for part in open(f'/sys/devices/system/node/node{node}/cpulist').read().strip().split(','):
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


if node >= 0:
    os.sched_setaffinity(0, cpus_of(node))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
gpu_nodes = {p: open(p).read().strip() for p in glob.glob('/sys/class/drm/card*/device/numa_node')}
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(lidar_cloud(3))).cuda()
te, td = [], []
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= 3:
        te.append(t1 - t0); td.append(t2 - t1)
import statistics
print(f'node {node}: enc {1e3*statistics.median(te):.2f} ms dec {1e3*statistics.median(td):.2f} ms  gpu numa {gpu_nodes}', flush=True)
