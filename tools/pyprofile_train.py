#!/usr/bin/env python3
"""cProfile of the host side of the training step (where the Python time per launch goes)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
torch.manual_seed(0)
cfg = TrainConfig()
tr = Trainer(Model(baseline_r1()), cfg, torch.device('cuda', 0))
data = synthetic_batches(0, 1, cfg, torch.device('cuda', 0))
for _ in range(3):
    tr.step(next(data))
batches = [next(data) for _ in range(4)]
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for b in batches:
    tr.step(b)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
