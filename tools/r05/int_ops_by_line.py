#!/usr/bin/env python3
"""Integer codec (cfg#3): which source line issues which device operation.  One compress + decompress of the LiDAR-like frame under a
TorchDispatchMode: every aten operator that launches a kernel or a copy, by the innermost frame inside fastpcc_amd/."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
from fastpcc_amd import hipops, _native

model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(lidar_cloud(3))).cuda()
for _ in range(3):
    data = model.compress(frame); rec = model.decompress(data)
torch.cuda.synchronize()

NO_LAUNCH = ('empty', 'as_strided', 'view', 'reshape', 'select', 'slice', 'narrow', 'expand', 't.', 'transpose', 'permute', 'unsqueeze',
             'squeeze', 'detach', 'alias', 'resize_', '_unsafe_view', 'set_', 'lift_fresh', 'unbind', 'split', 'chunk', 'flatten',
             '_reshape_alias', 'unfold', 'is_pinned', 'movedim', 'moveaxis', 'resolve_conj', 'resolve_neg', 'sym_', 'stride', 'size',
             'is_', 'dim', 'numel', 'storage_offset', 'new_empty', '_local_scalar_dense', 'item')
from torch.utils._python_dispatch import TorchDispatchMode


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
        elif name.startswith('_to_copy') or name.startswith('copy_') or name.startswith('_pin_memory'):
            dev = [a.device.type for a in args if isinstance(a, torch.Tensor)]
            self.by[(name + ' ' + '>'.join(dev) + ('>' + str((kwargs or {}).get('device', '')) if 'device' in (kwargs or {}) else ''), line)] += 1
        elif not name.startswith(NO_LAUNCH):
            self.by[(name, line)] += 1
        return func(*args, **(kwargs or {}))


for name, fn in (('compress', lambda: model.compress(frame)), ('decompress', lambda: model.decompress(data))):
    with Tally() as t:
        fn()
        torch.cuda.synchronize()
    print(f'## {name}: {sum(t.by.values())} aten operators that launch or copy, {sum(t.syncs.values())} read-backs (.item / .tolist)')
    for (op, line), n in sorted(t.by.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f'{n:5d}  {op:34s} {line}')
    for line, n in t.syncs.most_common():
        print(f'{n:5d}  read-back                          {line}')
    print()
