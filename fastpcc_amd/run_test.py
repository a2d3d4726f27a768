"""Evaluation driver: the role of /root/reference/test.py:76-140 for this path -- build the model named by a YAML file of
the reference's format, run every sample through `model(pc_data)` (compress + decompress + evaluator log), print the
per-file and mean metrics.  Samples are PLY files (`--ply`) or seeded synthetic frames (`--synthetic body:1024`);
clouds above the config's `kd_tree_partition_max_points_num` are partitioned like the reference's collate function.

    python -m fastpcc_amd.run_test --config /path/to/baseline_r1.yaml --ply a.ply b.ply --results-dir out/
    python -m fastpcc_amd.run_test --synthetic body:1024 lidar
"""
import argparse
import importlib
import json
import os
import sys
from typing import List, Optional

import numpy as np
import torch

from .data import PCData, kitti_odometry_sample, pc_data_collate_fn, read_ply_file

# module path of the reference -> the module of this package that replaces it
MODEL_MODULES = {
    'models.convolutional.lossy_coord_v2': 'fastpcc_amd.codecs.lossy_coord_v2',
    'models.convolutional.lossy_coord_lossy_color': 'fastpcc_amd.codecs.lossy_coord_lossy_color',
    'models.convolutional.lossl_coord_int': 'fastpcc_amd.codecs.lossl_coord_int',
    'models.convolutional.lossl_coord': 'fastpcc_amd.codecs.lossl_coord',
}


def _yaml_sections(path: str) -> dict:
    from .codecs.lossy_coord_v2.model_config import _load_with_includes
    merged = {}
    for doc in _load_with_includes(path):
        for k, v in doc.items():
            if isinstance(v, dict) and isinstance(merged.get(k), dict):
                merged[k].update(v)
            else:
                merged[k] = v
    return merged


def build_model(config_path: Optional[str], weights: Optional[str], device: torch.device):
    if config_path is None:
        from .codecs.lossy_coord_v2 import Model
        from .codecs.lossy_coord_v2.model_config import baseline_r1
        cfg = baseline_r1()
        cfg.numerics_version_in_header = True
        model, sections = Model(cfg), {}
    else:
        sections = _yaml_sections(config_path)
        ref_path = sections.get('model_module_path', 'models.convolutional.lossy_coord_v2')
        module = importlib.import_module(MODEL_MODULES.get(ref_path, ref_path))
        cfg_cls = getattr(module, 'Config', None) or getattr(module, 'ModelConfig')
        cfg = cfg_cls(**(sections.get('model') or {}))
        # Streams of the float codecs are only decodable under the summation-order rules they were written with (include/fpcc_hip.h,
        # 'Numerics version'): evaluation runs carry the version byte unless the YAML says otherwise, so that a build with other
        # rules refuses the stream instead of decoding garbage.  (The library default stays the reference's byte layout.)
        if hasattr(cfg, 'numerics_version_in_header') and 'numerics_version_in_header' not in (sections.get('model') or {}):
            cfg.numerics_version_in_header = True
        try:
            model = module.Model(cfg, device)
        except TypeError:
            model = module.Model(cfg)
    if weights:
        ckpt = torch.load(weights, map_location='cpu', weights_only=False)
        state = ckpt.get('ema_state_dict', ckpt.get('state_dict', ckpt))
        missing, unexpected = model.load_state_dict(state, strict=False)
        print(f'loaded {weights}: {len(missing)} missing, {len(unexpected)} unexpected keys', file=sys.stderr)
    else:
        print('no --weights: seeded random initialisation (bpp / PSNR are plumbing checks, not RD results)', file=sys.stderr)
    return model.to(device).eval(), sections


def _samples(args, sections: dict) -> List[PCData]:
    out = []
    for path in args.ply or []:
        xyz, rgb = read_ply_file(path)
        xyz = np.rint(xyz).astype(np.int32)
        res = args.resolution or 1 << int(np.ceil(np.log2(max(int(xyz.max()) + 1, 2))))
        out.append(PCData(xyz=torch.from_numpy(xyz), color=None if rgb is None else torch.from_numpy(rgb.astype(np.float32)),
                          resolution=[res], file_path=[os.path.basename(path)], org_points_num=[len(xyz)]))
    for path in args.kitti or []:
        grid = ((sections.get('test') or {}).get('dataset') or {}).get('resolution', 4096)
        out.append(kitti_odometry_sample(path, args.resolution or grid))
    for spec in args.synthetic or []:
        from .synthetic import SCALE, body_cloud, lidar_cloud
        kind, _, arg = spec.partition(':')
        if kind == 'body':
            res = int(arg or 1024)
            xyz = body_cloud(res, SCALE.get(res, 1.0), seed=2)
        elif kind == 'lidar':
            res, xyz = 65536, lidar_cloud(3)
        else:
            raise ValueError(f'unknown synthetic frame {spec!r}')
        out.append(PCData(xyz=torch.from_numpy(xyz.astype(np.int32)), resolution=[res], file_path=[f'{spec}.ply'],
                          org_points_num=[len(xyz)]))
    if not out:
        raise SystemExit('give --ply files, --kitti sweeps or --synthetic frames')
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--config', help="a YAML file of the reference's format (model_module_path / model / test sections)")
    ap.add_argument('--weights', help='checkpoint with state_dict / ema_state_dict')
    ap.add_argument('--ply', nargs='*')
    ap.add_argument('--kitti', nargs='*', help='KITTI Odometry velodyne sweeps (.bin), voxelised as the reference dataset does')
    ap.add_argument('--synthetic', nargs='*')
    ap.add_argument('--resolution', type=int, default=0)
    ap.add_argument('--results-dir')
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('fastpcc_amd runs on a GPU (no CPU path)')
    device = torch.device('cuda', 0)
    model, sections = build_model(args.config, args.weights, device)
    limit = ((sections.get('test') or {}).get('dataset') or {}).get('kd_tree_partition_max_points_num', 0)
    limit = limit[0] if isinstance(limit, (list, tuple)) else limit
    if args.results_dir:
        os.makedirs(os.path.join(args.results_dir, 'bin'), exist_ok=True)
    per_file = {}
    if hasattr(model, 'pre_test_hook'):             # test.py:115-116 (the float LiDAR codec inserts its observers here)
        model.pre_test_hook()
    for sample in _samples(args, sections):
        sample.results_dir = os.path.join(args.results_dir, 'bin') if args.results_dir else None
        batch = pc_data_collate_fn([sample], int(limit or 0)).to(device)
        with torch.no_grad():
            ret = model(batch)
        per_file[sample.file_path[0]] = {k: v for k, v in ret.items() if isinstance(v, (int, float))}
    if hasattr(model, 'post_test_hook'):            # test.py:147-148 (... and writes the integer parameters here)
        model.post_test_hook()
    evaluator = getattr(model, 'evaluator', None)
    mean = evaluator.show(os.path.join(args.results_dir, 'bin') if args.results_dir else None) if evaluator is not None else {}
    print(json.dumps({'files': per_file, 'mean': mean}, indent=1, default=float))
    return 0


if __name__ == '__main__':
    sys.exit(main())
