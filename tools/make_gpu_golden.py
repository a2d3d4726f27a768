#!/usr/bin/env python3
"""Writes tests/golden/v2_stream.json: a lossy_coord_v2/baseline_r1 bitstream produced by THIS build on the GPU, with the
checksum of the cloud it decodes to.  The fp32 summation orders of the convolutions are part of the stream format (the
decoder must reproduce the encoder's activations bit for bit, include/fpcc_hip.h "Numerics version"); the committed stream
makes a silent change of an order-selecting constant (offset-split threshold, padded-shape threshold, two-phase conv3->1)
fail a test instead of orphaning streams.  Run on a GPU box:  python tools/make_gpu_golden.py gpurun_out/v2_stream.json"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import hipops
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import batched, enliven, surface_cloud


def cloud_digest(xyz: np.ndarray) -> str:
    x = xyz.astype(np.int64)
    keys = np.sort((x[:, 0] << 42) | (x[:, 1] << 21) | x[:, 2])
    return hashlib.sha256(keys.tobytes()).hexdigest()


def main(path):
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    model = model.cuda().eval()
    xyz = surface_cloud(11, 128, 90000)              # levels from ~23 K rows down to a few dozen: grouped (order 3), padded and two-phase shapes
    data = model.compress(torch.from_numpy(batched(xyz)).to(torch.int32).cuda())
    rec = model.decompress(data).cpu().numpy()
    out = {'numerics_version': hipops.numerics_version(), 'cloud': {'generator': 'surface_cloud(11, 128, 90000)', 'voxels': int(len(xyz))},
           'weights': 'Model(baseline_r1()) after torch.manual_seed(0); enliven(model, 0)',
           'stream_hex': data.hex(), 'decoded_voxels': int(len(rec)), 'decoded_sha256': cloud_digest(rec)}
    with open(path, 'w') as f:
        json.dump(out, f)
    print('wrote', path, len(data), 'bytes,', len(rec), 'decoded voxels')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'v2_stream.json'))
