import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0,'tests')
import torch, numpy as np
import test_gpu_train_v3 as t
from fastpcc_amd.synthetic import surface_cloud
cfg, model = t._model(1, num_latents=(0,0,2,2,0), lossl_geo_upsample=(0,1,1,1,1))
xyz, pn = t._sorted_batch([surface_cloud(11,64,5000), surface_cloud(12,64,3500)])
torch.manual_seed(5)
out = model.train_forward(xyz, pn, 10); loss = out['loss']; print('loss', loss.item(), {k: round(v,4) for k,v in out.items() if k!='loss'})
loss.backward()
params = [p for p in model.parameters() if p.grad is not None]
g = torch.Generator(device='cuda').manual_seed(3)
direction = [torch.randn(p.shape, device='cuda', generator=g) * (p.detach().abs().mean() + 1e-3) for p in params]
analytic = sum((p.grad * d).sum() for p, d in zip(params, direction)).item()
def value(eps):
    with torch.no_grad():
        for p, d in zip(params, direction): p.add_(eps * d)
        torch.manual_seed(5)
        o = model.train_forward(xyz, pn, 10)
        for p, d in zip(params, direction): p.sub_(eps * d)
    return o['loss'].item()
print('analytic', analytic)
for eps in (1e-3, 3e-3, 1e-2, 3e-2):
    print(eps, (value(eps)-value(-eps))/(2*eps), value(0.0))
