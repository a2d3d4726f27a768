import os, sys, gc, collections
sys.path.insert(0, os.getcwd())
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
dev = torch.device('cuda:0')
torch.manual_seed(0); model = Model(baseline_r1()); enliven(model, 0); model = model.to(dev).eval()
frame = torch.from_numpy(batched(body_cloud(512, SCALE.get(512, 1.0), seed=2))).to(dev)
for _ in range(2):
    d = model.compress(frame); ME.clear_global_coordinate_manager(); model.decompress(d); ME.clear_global_coordinate_manager()
gc.collect(); gc.disable(); gc.set_debug(gc.DEBUG_SAVEALL)
d = model.compress(frame); ME.clear_global_coordinate_manager()
n1 = gc.collect(); g1 = list(gc.garbage); gc.garbage.clear()
rec = model.decompress(d); ME.clear_global_coordinate_manager()
n2 = gc.collect(); g2 = list(gc.garbage); gc.garbage.clear()
for name, g in (('compress', g1), ('decompress', g2)):
    c = collections.Counter(type(o).__name__ for o in g)
    tens = sum(o.numel() * o.element_size() for o in g if isinstance(o, torch.Tensor) and o.is_cuda)
    print(name, len(g), 'objects in cycles;', c.most_common(8), 'cuda tensor bytes', tens)
    for o in g:
        if type(o).__name__ in ('function', 'cell') and getattr(o, '__qualname__', ''):
            print('   ', o.__qualname__)
