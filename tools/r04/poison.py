#!/usr/bin/env python3
"""Does any kernel read memory nobody wrote?  Every block the caching allocator hands out is first filled with a poison pattern
(NaN / Inf bit patterns, -1 as an index): compress and decompress must give the same bytes / the same cloud as on a clean heap.
usage: poison.py [resolution=1024] [rounds=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def poison(pattern: int, gib: int = 24):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for _ in range(gib):
        blocks.append(torch.full((1 << 28,), pattern, dtype=torch.int32, device='cuda'))
    for n in (128, 1024, 16384, 131072, 1 << 19, 1 << 21):              # the small pool and mid-sized blocks
        for _ in range(96):
            blocks.append(torch.full((n,), pattern, dtype=torch.int32, device='cuda'))
    torch.cuda.synchronize()
    del blocks                                                            # stays cached: the next torch.empty() gets it


torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
ref = model.compress(frame)
ME.clear_global_coordinate_manager()
ref_rec = model.decompress(ref).clone()
torch.cuda.synchronize()
ME.clear_global_coordinate_manager()
bad = 0
for r in range(rounds):
    for name, pat in (('nan', -1), ('inf', 0x7F800000), ('-inf', -8388608), ('huge index', 0x7FFFFFF0), ('qnan', 0x7FC00000)):
        poison(pat)
        try:
            data = model.compress(frame)
        except Exception as e:
            print(f'round {r} {name}: compress raised {e!r}', flush=True)
            bad += 1
            ME.clear_global_coordinate_manager()
            continue
        ME.clear_global_coordinate_manager()
        if data != ref:
            bad += 1
            first = next(i for i in range(min(len(data), len(ref))) if data[i] != ref[i]) if len(data) == len(ref) else -1
            print(f'round {r} {name}: bytes differ ({len(data)} vs {len(ref)}, first at {first})', flush=True)
        poison(pat)
        try:
            rec = model.decompress(ref)
            torch.cuda.synchronize()
            if rec.shape != ref_rec.shape or not torch.equal(rec, ref_rec):
                bad += 1
                print(f'round {r} {name}: decoded cloud differs ({tuple(rec.shape)} vs {tuple(ref_rec.shape)})', flush=True)
        except Exception as e:
            print(f'round {r} {name}: decompress raised {e!r}', flush=True)
            bad += 1
        ME.clear_global_coordinate_manager()
print(f'{rounds} rounds x 5 patterns, {bad} deviations')
