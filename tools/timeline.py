#!/usr/bin/env python3
"""Host-side timeline of one encode + decode of the headline workload (where the wall clock goes outside the kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd.synthetic import enliven
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
em = model.em_lossless_based
for it in range(6):
    em.timing = {} if it >= 3 else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    tm_enc = dict(em.timing) if em.timing is not None else None
    ME.clear_global_coordinate_manager()
    t2 = time.perf_counter()
    rec = model.decompress(data); torch.cuda.synchronize(); t3 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    if tm_enc:
        tm = em.timing
        e = tm_enc
        print(f"enc {1e3*(t1-t0):.2f} ms: pre-EM {1e3*(e['enc_t0']-t0):.2f} | EM enqueue {1e3*(e['enc_enqueued']-e['enc_t0']):.2f} | "
              f"wait GPU {1e3*(e['enc_synced']-e['enc_enqueued']):.2f} | occupancy rANS {1e3*(e['enc_occupancy_coded']-e['enc_synced']):.2f} | "
              f"residual+pack {1e3*(e['enc_done']-e['enc_occupancy_coded']):.2f} | tail {1e3*(t1-e['enc_done']):.2f}")
        print(f"dec {1e3*(t3-t2):.2f} ms: residual rANS {1e3*(tm['dec_residual_decoded']-tm['dec_t0']):.2f} | "
              f"level waits (GPU+D2H) {1e3*tm['dec_wait']:.2f} | occupancy rANS+H2D {1e3*tm['dec_host']:.2f}")
