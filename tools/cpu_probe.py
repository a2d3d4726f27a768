import os, sys, time
sys.path.insert(0, os.getcwd())
thr, res, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.setdefault('OMP_NUM_THREADS', str(thr))
import numpy as np, torch
torch.set_num_threads(thr)
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
from fastpcc_amd.engine import summation_order
from oracle.codec_v2 import OracleV2
import oracle; oracle.build()
cfg = baseline_r1(); torch.manual_seed(0); m = Model(cfg); enliven(m, 0)
w = {k: v.clone() for k, v in m.state_dict().items() if isinstance(v, torch.Tensor)}
xyz = body_cloud(res, SCALE[1024], seed=2); coords = batched(xyz).astype(np.int64)
o = OracleV2(w, cfg, conv=mode, order_fn=summation_order); o.skip_unused_tail = True
t0 = time.perf_counter(); d = o.compress(coords); t1 = time.perf_counter(); o.decompress(d); t2 = time.perf_counter()
print(res, len(xyz), mode, 'threads', thr, f'enc {t1-t0:.2f} dec {t2-t1:.2f}', flush=True)
