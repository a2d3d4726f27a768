#!/usr/bin/env python3
"""Training step (cfg#5): which source line issues which aten operator (a TorchDispatchMode around one optimisation step, backward
included: the autograd engine's calls into fastpcc_amd/autograd.py are Python frames too)."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
torch.manual_seed(0)
cfg = TrainConfig()
tr = Trainer(Model(baseline_r1()), cfg, torch.device('cuda', 0))
data = synthetic_batches(0, 1, cfg, torch.device('cuda', 0))
for _ in range(3):
    tr.step(next(data))
batch = next(data)
NO_LAUNCH = ('empty', 'as_strided', 'view', 'reshape', 'select', 'slice', 'narrow', 'expand', 't.', 'transpose', 'permute', 'unsqueeze',
             'squeeze', 'detach', 'alias', 'resize_', '_unsafe_view', 'set_', 'lift_fresh', 'unbind', 'split', 'chunk', 'flatten',
             '_reshape_alias', 'unfold', 'is_pinned', 'movedim', 'moveaxis', 'resolve_conj', 'resolve_neg', 'sym_', 'stride', 'size',
             'is_', 'dim', 'numel', 'storage_offset', 'new_empty', '_local_scalar_dense', 'item', 'result_type')


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter()
        self.syncs = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        line = '?'
        for fr in reversed(traceback.extract_stack()[:-1]):
            if 'fastpcc_amd/' in fr.filename and not fr.filename.endswith('hipops.py'):
                line = f"{fr.filename.split('fastpcc_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
        if name.startswith(('_local_scalar_dense', 'item')):
            self.syncs[line] += 1
        elif not name.startswith(NO_LAUNCH):
            self.by[(name, line)] += 1
        return func(*args, **(kwargs or {}))


with Tally() as t:
    tr.step(batch)
    torch.cuda.synchronize()
print(f'## one optimisation step: {sum(t.by.values())} aten operators that launch or copy, {sum(t.syncs.values())} read-backs')
lines = collections.Counter()
for (op, line), n in t.by.items():
    lines[line] += n
for line, n in lines.most_common(60):
    ops = ', '.join(f'{o} x{c}' for (o, l), c in sorted(t.by.items(), key=lambda kv: -kv[1]) if l == line)
    print(f'{n:5d}  {line:60s} {ops[:150]}')
for line, n in t.syncs.most_common():
    print(f'{n:5d}  read-back   {line}')
