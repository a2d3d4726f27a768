"""offsets a 32-row block of the production row order executes, from the actual device tables (cfg#2 frame, one level)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
for _ in range(level):
    m = cm._ensure_parent(m)
nbr = cm._nbr27(m)
order = cm._row_order(m)
n = m.n
present = (nbr >= 0)
masks = torch.zeros(n, dtype=torch.int64, device='cuda')
for k in range(27):
    masks |= present[k].to(torch.int64) << k
def stat(o, blk):
    mm = masks[o.long()] if o is not None else masks
    pad = (-n) % blk
    mm = torch.nn.functional.pad(mm, (0, pad)).view(-1, blk)
    u = mm[:, 0].clone()
    for j in range(1, blk):
        u |= mm[:, j]
    pc = torch.zeros_like(u)
    for k in range(27):
        pc += (u >> k) & 1
    # per offset group of the grouped kernel
    return pc.float().mean().item()
print('rows', n, 'pairs/row', present.sum().item() / n)
print('natural: offsets per 32-row block', stat(None, 32))
print('production row order: per 32-row block', stat(order, 32), ' per 64', stat(order, 64))
print('is a permutation', bool((torch.sort(order).values == torch.arange(n, device='cuda')).all()))
