#!/usr/bin/env python3
"""Headline benchmark: encode+decode throughput of lossy_coord_v2/baseline_r1 on a 1M-voxel synthetic frame.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one batch of B frames resident in HBM through compress_many + decompress_many (one network traversal over the batch, B
independent streams); `value` is the throughput of a stream of such steps.  `value_one_frame` is the metric exactly as the reference
times it: compress(frame) + decompress(bytes) of ONE frame, each closed by a device synchronise ('encode time' / 'decode time',
models/convolutional/lossy_coord_v2/model.py:196-205).  Frames are independent, so N GPUs run N replicas on different frames (weak
scaling, no data-path collective).  Weights: seeded random initialisation of the architecture (no checkpoints exist here); data: seeded
synthetic voxel surface (fastpcc_amd/synthetic.py).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 157.3       # fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--resolution', type=int, default=1024, help='1024 -> ~1M voxels (cfg#2)')
    ap.add_argument('--cpu-baseline', type=int, default=1, help='0 disables the CPU oracle timing on rank 0')
    ap.add_argument('--cpu-resolution', type=int, default=1024, help='resolution of the CPU sample (same generator; 1024 = the benchmarked frame itself)')
    ap.add_argument('--cpu-baseline-worker', default='', help=argparse.SUPPRESS)
    ap.add_argument('--secondary', type=int, default=1, help='0 skips the figures reported next to the headline: cfg#3 / #4 / #5 at N = 1, '
                                                             'the cfg#5 DDP training step over all N ranks at N > 1')
    ap.add_argument('--batch', type=int, default=16, help='frames per step: B independent ~1M-voxel frames coded in ONE network traversal '
                    '(compress_many / decompress_many; every stream byte-identical to the frame coded alone).  Measured on one box '
                    '(tools/r05/g34.sh): 8 -> 79.9, 12 -> 83.5, 16 -> 85.0, 24 -> 84.4, 32 -> 86.4 Mpoints/s')
    ap.add_argument('--frames-in-flight', type=int, default=2, help='batches a rank keeps in flight on its GPU (fastpcc_amd/serving.py): '
                    '1 = one batch at a time')
    ap.add_argument('--latency-frames', type=int, default=12, help='frames of the one-frame-at-a-time figure `value_one_frame` (median; >= 10)')
    ap.add_argument('--stages', type=int, default=-1, help='named stages of serving.py (one batch encodes while the other decodes): 1 on, 0 off '
                    '(batches in flight run free), -1 = on below 4 frames per batch.  Single frames in flight fall into lock-step without '
                    'them (profiles/r04/frames_in_flight.md); batches of 4-8 frames are GPU-bound either way and run 2-3 %% faster free '
                    '(profiles/r05/batched_traversal.md)')
    ap.add_argument('--own-streams', type=int, default=0, help='1: every frame in flight on a HIP stream of its own (kernels of different frames overlap)')
    ap.add_argument('--ddp-steps', type=int, default=10, help='optimisation steps of the cfg#5 DDP figure (N > 1)')
    ap.add_argument('--ddp-deadline', type=float, default=420.0, help='seconds after which rank 0 prints the headline without the DDP figure')
    ap.add_argument('--dump-trace', default='', help='write the per-launch conv table of the last step to this file')
    return ap.parse_args()


def conv_flops(info, counts_cache):
    """algorithmic flop of one conv launch: 2 * L * C_in * C_out with L = rulebook pairs actually present"""
    import torch
    if 'flops' in info:                     # a fused launch that carries its own count (fpcc_mlp_chain_f32: sum over its layers)
        return info['flops']
    if info['nbr'] is None:
        pairs = info['n_out'] * info['groups']
    else:
        key = (info['nbr'].data_ptr(), info['n_offsets'], info['n_out'])
        if key not in counts_cache:
            t = info['nbr']
            counts_cache[key] = int((t >= 0).sum().item()) if t.numel() else 0
        pairs = counts_cache[key]
    return 2.0 * pairs * info['c_in'] * info['c_out']


def pmc_traffic(batch, launches_per_step):
    """HBM bytes per MFMA convolution launch (k_conv_wave / k_conv_mfma / ...) from the newest committed rocprofv3 PMC passes
    (profiles/r*/..._pmc_traffic.json, made by profiles/pmc_summary.py: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled
    as the gfx950 guide prescribes).  PMC counters cannot be collected from inside this process, so the committed measurement of the
    same command is reported; None when there is none.  The profiled command codes single frames (reference streams, the
    one-frame-at-a-time leg) AND batches, so the file carries how many frame-equivalents it covers (`_meta`); the figure returned is
    family bytes per frame x the frames of a launch of THIS run: bytes / frame_equivalents x batch / launches_per_step -- per launch,
    like `achieved` and `algorithmic_bytes_per_launch`."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*', '*pmc_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        data = json.load(f)
    mfma = [v for k, v in data.items() if k.startswith(('k_conv_mfma', 'k_conv_wave', 'k_pointwise_wave', 'k_mlp_chain'))]
    launches = sum(v['launches'] for v in mfma)
    total = sum(v['fetch_bytes'] + v['write_bytes'] for v in mfma)
    if not launches:
        return None, None
    meta = data.get('_meta') or {}
    src = os.path.relpath(files[-1], ROOT)
    if meta.get('frame_equivalents') and launches_per_step:
        return total / meta['frame_equivalents'] * batch / launches_per_step, f"{src} ({meta.get('command', '')}; {meta['frame_equivalents']} frame-equivalents)"
    return total / launches, src


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


CPU_THREADS_CAP = 32


def cpu_baseline_worker(resolution, budget_s, out_path):
    """Runs in a child process started BEFORE this process touches the GPU (bench.py --cpu-baseline-worker): the protocol of
    BASELINE.md section 3 / SURVEY.md section 8d -- the CPU oracle configured like MinkowskiEngine's CPU backend (per kernel
    offset index_select -> torch.mm -> index_add_, fp32), same seeded weights, one warm-up and the median of up to five
    encode+decode runs on a bounded sample of the workload (the same generator at `resolution`); like the GPU path it stops
    after the last level that feeds the bitstream (identical bytes).  Threads: min(host cores, 32) -- on the 256-core GPU box
    torch's CPU operators get SLOWER beyond 32 threads (15 K voxels: 0.5 s at 16 threads, 2.9 s at 64, no end within 50 s at
    256; measured with tools/cpu_probe.py), so "all threads" would not be a baseline but a pathology.  The OpenMP FMA-chain
    evaluation (oracle/sparse_conv.c, the bit-exact checker) is timed once beside it as a secondary figure."""
    import statistics
    threads = min(os.cpu_count() or 1, CPU_THREADS_CAP)
    os.environ['OMP_NUM_THREADS'] = str(threads)
    # the parent starts this process before it initialises the GPU and releases it AFTER its GPU legs are done: 32 busy host threads
    # right before the timed region cost the frame pipeline 15-20 % (two host threads pace two frames; profiles/r04/README.md)
    if not sys.stdin.readline():
        return
    import numpy as np
    import torch
    import oracle
    from oracle.codec_v2 import OracleV2
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.engine import summation_order as ME_order
    from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
    oracle.build()
    torch.set_num_threads(threads)
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)                                                   # the weights of the GPU run, re-created by seed
    weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    xyz = body_cloud(resolution, SCALE.get(resolution, SCALE[1024]), seed=2)       # at 1024: rank 0's benchmarked frame
    coords = batched(xyz).astype(np.int64)
    small = batched(xyz[: min(len(xyz), 2000)]).astype(np.int64)

    def run(o):
        t0 = time.perf_counter()
        data = o.compress(coords)
        t1 = time.perf_counter()
        rec = o.decompress(data)
        t2 = time.perf_counter()
        assert len(xyz) - max(16, len(xyz) // 1000) <= rec.shape[0] <= len(xyz)      # ties at the pruning threshold, as on the GPU
        return t1 - t0, t2 - t1

    mm = OracleV2(weights, cfg, conv='mm')
    mm.skip_unused_tail = True
    mm.compress(small)                                                  # warm-up (thread pool, allocator)
    times = [run(mm)]
    reps = min(3, max(1, int(budget_s / max(sum(times[0]), 1e-3))))      # ~17 s per run on the 1 M-voxel frame: median of 3
    while len(times) < reps:
        times.append(run(mm))
    enc, dec = statistics.median(t[0] for t in times), statistics.median(t[1] for t in times)
    which = 'the benchmarked frame itself (rank 0, seed 2)' if resolution == 1024 else 'a sample of the same generator'
    out = {'value': round(len(xyz) / (enc + dec) / 1e6, 5), 'unit': 'Mpoints/s', 'cores': threads, 'kind': 'port',
           'cpu': cpu_model(), 'host_cores': os.cpu_count(),
           'sample': f'{which}: {len(xyz)} voxels at {resolution}^3; gather/GEMM/scatter-add oracle (torch.mm, {threads} threads: '
                     f'more are slower on this host), median of {len(times)} encode+decode runs after a warm-up: enc {enc:.2f}s dec {dec:.2f}s; '
                     f'unused encoder tail skipped as on the GPU'}

    def dump():
        with open(out_path, 'w') as f:                                  # the primary figure is safe even if a secondary run stalls
            json.dump(out, f)
    dump()
    out['secondary'] = {}
    if resolution != 512:                                               # the 250 K-voxel sample earlier rounds quoted, one run
        xyz_full, coords_full = xyz, coords
        xyz = body_cloud(512, SCALE.get(512, SCALE[1024]), seed=2)
        coords = batched(xyz).astype(np.int64)
        s_enc, s_dec = run(mm)
        out['secondary']['sample_512'] = {'what': f'same oracle on the {len(xyz)}-voxel 512^3 sample of rounds 1-5, one run',
                                          'value': round(len(xyz) / (s_enc + s_dec) / 1e6, 5), 'unit': 'Mpoints/s'}
        xyz, coords = xyz_full, coords_full
        dump()
    chain = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    chain.skip_unused_tail = True
    chain.compress(small)
    c_enc, c_dec = run(chain)
    out['secondary'].update({'what': f'oracle/sparse_conv.c FMA-chain evaluation (OpenMP, {threads} threads), one run, same frame',
                             'value': round(len(xyz) / (c_enc + c_dec) / 1e6, 5), 'unit': 'Mpoints/s', 'enc_s': round(c_enc, 2), 'dec_s': round(c_dec, 2)})
    dump()


class CpuBaseline:
    """the worker above as a child process with a hard time limit; created before the GPU is initialised (no fork of a process that
    holds a GPU context), idle until start(), which the bench calls after its GPU legs; collected by result()"""

    def __init__(self, resolution, limit_s=240.0):
        import subprocess
        import tempfile
        self.limit_s = limit_s
        self.path = os.path.join(tempfile.gettempdir(), f'fpcc_cpu_baseline_{os.getpid()}.json')
        self.t0 = None
        self.proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', self.path,
                                      '--cpu-resolution', str(resolution)], stdin=subprocess.PIPE, stdout=subprocess.DEVNULL,
                                     stderr=subprocess.DEVNULL)

    def start(self):
        if self.t0 is None:
            self.t0 = time.perf_counter()
            try:
                self.proc.stdin.write(b'go\n')
                self.proc.stdin.flush()
                self.proc.stdin.close()
            except OSError:
                pass

    def wait(self):
        import subprocess
        self.start()
        try:
            self.proc.wait(timeout=max(1.0, self.limit_s - (time.perf_counter() - self.t0)))
        except subprocess.TimeoutExpired:
            self.proc.kill()                      # the exact PID this object started
            self.proc.wait()

    def result(self):
        self.wait()
        try:
            with open(self.path) as f:
                out = json.load(f)
            os.unlink(self.path)
        except (OSError, ValueError):
            return {'value': None, 'unit': 'Mpoints/s', 'cores': min(os.cpu_count() or 1, CPU_THREADS_CAP), 'kind': 'port',
                    'sample': f'the CPU oracle did not finish within {self.limit_s:.0f} s on this host'}
        if 'value' not in (out.get('secondary') or {}):
            out.setdefault('secondary', {}).update({'what': 'FMA-chain evaluation', 'value': None, 'note': f'not finished within {self.limit_s:.0f} s'})
        return out


def secondary(device):
    """Driver-visible numbers of the other BASELINE.json configurations, outside the timed region, a few seconds in all:
    cfg#3 lossl_coord_int on a 64 x 2048 LiDAR-like sweep, cfg#4 lossy_coord_lossy_color on a 2 M-voxel coloured frame, cfg#5
    one-GPU training step of lossy_coord_v2 (global batch 8).  Median of 5 after 2 warm-ups each; same seeded synthetic
    inputs and weights as the GPU tests."""
    import statistics
    import numpy as np
    import torch
    from fastpcc_amd import engine as ME
    from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven, lidar_cloud
    out = {}

    def timed(enc, dec, reps=5, warm=2):
        te, td, data = [], [], None
        for it in range(warm + reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            data = enc(); torch.cuda.synchronize(); t1 = time.perf_counter()
            ME.clear_global_coordinate_manager()
            rec = dec(data); torch.cuda.synchronize(); t2 = time.perf_counter()
            ME.clear_global_coordinate_manager()
            if it >= warm:
                te.append(t1 - t0); td.append(t2 - t1)
        return statistics.median(te) * 1e3, statistics.median(td) * 1e3, data, rec

    try:
        from fastpcc_amd.codecs.lossl_coord_int import Config, Model
        from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
        torch.cuda.empty_cache()        # every secondary workload starts from its own allocations, not the previous one's cached blocks
        model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.to(device).eval()
        xyz = lidar_cloud(3)
        frame = torch.from_numpy(batched(xyz)).to(device)
        enc, dec, data, rec = timed(lambda: model.compress(frame), lambda d: model.decompress(d))
        out['cfg3_lossl_coord_int'] = {'workload': f'{len(xyz)}-voxel 64x2048 LiDAR-like sweep, 16-bit, channels 256', 'encode_ms': round(enc, 3),
                                       'decode_ms': round(dec, 3), 'Mpoints_per_s': round(len(xyz) / (enc + dec) / 1e3, 3), 'bytes': len(data),
                                       'bpp': round(8 * len(data) / len(xyz), 4), 'lossless': bool(rec.shape[0] == len(xyz))}
        # the same codec over eight different sweeps in ONE traversal (compress_many / decompress_many: the clouds are the samples of a
        # batch, one stream per cloud, each byte-identical to the one it gets alone -- checked here for sweep 0)
        sweeps = [frame] + [torch.from_numpy(batched(lidar_cloud(3 + i))).to(device) for i in range(1, 8)]
        enc8, dec8, data8, rec8 = timed(lambda: model.compress_many(sweeps), lambda d: model.decompress_many(d), reps=3, warm=1)
        n8 = sum(f.shape[0] for f in sweeps)
        out['cfg3_lossl_coord_int']['batch_of_8_sweeps'] = {
            'voxels': n8, 'encode_ms': round(enc8, 3), 'decode_ms': round(dec8, 3), 'Mpoints_per_s': round(n8 / (enc8 + dec8) / 1e3, 3),
            'stream_0_identical': bool(data8[0] == data), 'lossless': bool([r.shape[0] for r in rec8] == [f.shape[0] for f in sweeps])}
        del model, frame, sweeps
    except Exception as e:                                               # the headline number must survive a secondary failure
        out['cfg3_lossl_coord_int'] = {'error': repr(e)[:200]}
    try:
        from fastpcc_amd.codecs.lossy_coord_lossy_color import Model as ColorModel
        from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1 as color_cfg
        torch.manual_seed(0)
        torch.cuda.empty_cache()
        model = ColorModel(color_cfg()); enliven(model, 3, gain=2.3); model = model.to(device).eval()
        xyz = body_cloud(2048, SCALE[2048], seed=4)
        rng = np.random.default_rng(1)
        base = 127 + 90 * np.stack((np.sin(xyz[:, 0] / 90.0), np.cos(xyz[:, 1] / 70.0), np.sin((xyz[:, 2] + xyz[:, 0]) / 110.0)), 1)
        rgb = torch.from_numpy(np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8)).to(device)
        frame = torch.from_numpy(batched(xyz)).to(device)
        enc, dec, data, rec = timed(lambda: model.compress(frame, rgb), lambda d: model.decompress(d), reps=3, warm=1)
        out['cfg4_lossy_coord_lossy_color'] = {'workload': f'{len(xyz)}-voxel 2048^3 coloured body-surface frame', 'encode_ms': round(enc, 3),
                                               'decode_ms': round(dec, 3), 'Mpoints_per_s': round(len(xyz) / (enc + dec) / 1e3, 3),
                                               'bytes': len(data), 'bpp': round(8 * len(data) / len(xyz), 4), 'decoded_points': int(rec[0].shape[0])}
        # four such frames in ONE traversal (compress_many / decompress_many; frame 0's stream is checked against the one it gets alone)
        frames4, rgbs4 = [frame], [rgb]
        for i in range(1, 4):
            x4 = body_cloud(2048, SCALE[2048], seed=4 + i)
            b4 = 127 + 90 * np.stack((np.sin(x4[:, 0] / 90.0), np.cos(x4[:, 1] / 70.0), np.sin((x4[:, 2] + x4[:, 0]) / 110.0)), 1)
            frames4.append(torch.from_numpy(batched(x4)).to(device))
            rgbs4.append(torch.from_numpy(np.clip(b4 + rng.normal(0, 8, b4.shape), 0, 255).astype(np.uint8)).to(device))
        enc4, dec4, data4, rec4 = timed(lambda: model.compress_many(frames4, rgbs4), lambda d: model.decompress_many(d), reps=3, warm=1)
        n4 = sum(f.shape[0] for f in frames4)
        out['cfg4_lossy_coord_lossy_color']['batch_of_4_frames'] = {
            'voxels': n4, 'encode_ms': round(enc4, 3), 'decode_ms': round(dec4, 3), 'Mpoints_per_s': round(n4 / (enc4 + dec4) / 1e3, 3),
            'stream_0_identical': bool(data4[0] == data), 'decoded_points': [int(r[0].shape[0]) for r in rec4]}
        del model, frame, rgb, frames4, rgbs4
    except Exception as e:
        out['cfg4_lossy_coord_lossy_color'] = {'error': repr(e)[:200]}
    try:
        from fastpcc_amd.codecs.lossy_coord_v2 import Model as V2
        from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
        from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
        tcfg = TrainConfig()
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        trainer = Trainer(V2(baseline_r1()), tcfg, device)
        data = synthetic_batches(0, 1, tcfg, device, 128)
        ts, voxels = [], 0
        for it in range(2 + 5):
            batch = next(data)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            trainer.step(batch)
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(time.perf_counter() - t0); voxels += batch.xyz.shape[0]
        ms = statistics.median(ts) * 1e3
        out['cfg5_training_1gpu'] = {'workload': f'lossy_coord_v2/baseline_r1 optimisation step, global batch {tcfg.batch_size} ShapeNet-like clouds at 128^3 '
                                                 f'({voxels // 5} voxels per step), forward + backward + AdamW', 'ms_per_step': round(ms, 2),
                                     'clouds_per_s': round(tcfg.batch_size / ms * 1e3, 2)}
    except Exception as e:
        out['cfg5_training_1gpu'] = {'error': repr(e)[:200]}
    ME.clear_global_coordinate_manager()
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.cpu_resolution, 90.0, args.cpu_baseline_worker)
    import statistics
    from fastpcc_amd import replicas
    rank, world, local = replicas.env_rank()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # the CPU baseline runs in a child process that is CREATED here, before this process initialises the GPU, and stays idle until
    # the GPU legs are done (cpu_job.result() at the end releases it): its 32 threads must not run beside or right before the timed
    # region, whose two frames in flight are paced by two host threads
    cpu_job = CpuBaseline(args.cpu_resolution) if (args.cpu_baseline and world == 1 and rank == 0) else None
    import numpy as np
    import torch
    import torch.distributed as dist
    # Dry run of the N > 1 path on a box with ONE GPU (tools/r03/two_ranks.sh): FPCC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0,
    # FPCC_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).  Not a measurement; the driver's runs set neither.
    if os.environ.get('FPCC_BENCH_ONE_DEVICE') == '1':
        local = 0
    torch.cuda.set_device(local)
    # before anything allocates pinned memory: every thread of the process (the HIP runtime's included) onto the CPUs of the GPU's
    # own NUMA node (two-socket hosts)
    numa = replicas.bind_to_device_numa_node(local)
    # RCCL; only the barrier and two scalar reductions of the headline use it.  Collectives are bounded: a rank that died leaves the
    # others with an error after 5 minutes instead of a job that never ends
    replicas.init(os.environ.get('FPCC_BENCH_BACKEND', 'nccl'), timeout_s=300)
    device = torch.device('cuda', local)

    from fastpcc_amd.synthetic import enliven
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.synthetic import SCALE, batched, body_cloud
    from fastpcc_amd import engine as ME

    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)
    weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    model = model.cuda().eval()

    # every rank codes its own frames (different seeds), all ~1M voxels: B = --batch of them per step
    B = max(1, args.batch)
    clouds = [body_cloud(args.resolution, SCALE.get(args.resolution, 1.0), seed=2 + rank * B + i) for i in range(B)]
    frames = [torch.from_numpy(batched(c)).cuda() for c in clouds]
    frame, xyz = frames[0], clouds[0]
    n_points = frame.shape[0]
    points_per_step = sum(f.shape[0] for f in frames)

    # One step = one batch of B frames through compress_many + decompress_many: ONE traversal of the networks over the B clouds (the
    # clouds are the samples of one engine batch; every layer is launched once; B byte-identical independent streams, each coded by its
    # own jobs on the coder pool -- fastpcc_amd/codecs/lossy_coord_v2/model.py).  With --frames-in-flight D > 1 the rank keeps D such
    # batches in flight on its GPU (fastpcc_amd/serving.py: D codec contexts over one set of weights, all enqueueing on ONE stream, so
    # kernels never overlap): while one batch waits for the host -- the serial rANS tail of compress, the probability -> mask round
    # trip of every occupancy level of decompress -- the other's launches run.  A context waits for its own work only (events).
    from fastpcc_amd.serving import FramePipeline, wait_for_my_work
    from fastpcc_amd.codecs.geo_lossl_em import GeoLosslessEntropyModel
    depth = max(1, args.frames_in_flight)
    pipeline = FramePipeline(model, depth, device, own_streams=bool(args.own_streams))

    def one_frame(f):                                    # one frame at a time, each half closed by a device synchronise
        torch.cuda.synchronize()
        a = time.perf_counter()
        data = model.compress(f)
        torch.cuda.synchronize()
        b = time.perf_counter()
        ME.clear_global_coordinate_manager()
        rec = model.decompress(data)
        torch.cuda.synchronize()
        c = time.perf_counter()
        ME.clear_global_coordinate_manager()
        return data, rec, b - a, c - b

    # reference results of every frame, coded alone: what every step of the timed region must reproduce byte for byte
    want_bytes, want_points = [], []
    for f in frames:
        data, rec, _, _ = one_frame(f)
        # the decoder keeps the candidates above the (8M - N)-th smallest logit; logits that tie with that threshold are
        # dropped, exactly as in the reference (lossy_coord_v2/layers.py:164-180), so the count can fall short by the ties
        assert f.shape[0] - max(16, f.shape[0] // 1000) <= rec.shape[0] <= f.shape[0], (rec.shape[0], f.shape[0])
        want_bytes.append(data)
        want_points.append(rec.shape[0])

    # The metric by its BASELINE definition (the reference's Timer blocks, lossy_coord_v2/model.py:196-205): N0 / (t_enc + t_dec),
    # one frame at a time, each half closed by a device synchronise -- median over `--latency-frames` frames (>= 10) after 2 warm-ups.
    for _ in range(2):
        one_frame(frame)
    lat = [one_frame(frame)[2:] for _ in range(max(1, args.latency_frames))]
    enc_ms = statistics.median(t[0] for t in lat) * 1e3
    dec_ms = statistics.median(t[1] for t in lat) * 1e3
    one_frame_ms = statistics.median(t[0] + t[1] for t in lat) * 1e3
    value_one_frame = n_points / one_frame_ms / 1e3

    import contextlib
    use_stages = args.stages == 1 or (args.stages < 0 and B < 4)
    stage = pipeline.stage if use_stages else (lambda name: contextlib.nullcontext())

    def step_of(ctx_model, _):
        with stage('compress'):                          # one batch encodes while the other decodes (serving.py: FramePipeline.stage)
            data = ctx_model.compress_many(frames)       # returns when the bytes of every cloud are written
            ME.clear_global_coordinate_manager()
        with stage('decompress'):
            rec = ctx_model.decompress_many(data)
            wait_for_my_work(device)                     # this batch's last kernel, not the other batch's queue
            ME.clear_global_coordinate_manager()
        # every step, inside the timed region: the streams are the single-frame streams, the decoders returned the coded points
        if data != want_bytes or [r.shape[0] for r in rec] != want_points:
            raise RuntimeError('a step of the timed region did not reproduce the frames\' own streams / point counts')
        return data, rec

    for data, rec in pipeline.map(step_of, range(max(args.warmup, depth))):      # every context warm
        pass

    def barrier():
        replicas.barrier(device)

    # Per-launch HIP events around every convolution cost ~6 us apiece (~1 ms a step now), so they are recorded on the last
    # THREE steps of the timed region (one step when fewer than 10 are timed); `roofline` is the mean over their launches.
    trace_steps = 3 if args.steps >= 10 else 1
    # a serving process does this once its models are loaded: everything allocated so far leaves the cyclic collector's
    # generations, so the collections that do run between frames only scan what the frames themselves left behind
    import gc
    gc.collect()
    gc.freeze()
    hipops.reserve_trace_events(600 * trace_steps)         # before the timed region: the traced steps only record
    # The shader clock the step actually gets: beside every large 3x3x3 launch of ONE EXTRA step after the timed region a one-wave kernel on a second
    # stream counts shader cycles against the constant 100 MHz counter for 100 us (fpcc_clock_probe).  MI355X's power management
    # starts a burst of matrix work near 2.0 GHz and takes ~30 ms of uninterrupted load to reach the 2.4 GHz the peak is quoted at
    # (profiles/r03/clock_ramp.md); a step with host-paced gaps never gets there.
    clock_stream = torch.cuda.Stream(device=device)
    clock_buf = torch.zeros((64, 2), dtype=torch.int64, device=device)
    clock_idx = []

    def clock_hook(ev0, n_out, n_offsets):
        if n_offsets == 27 and n_out >= 50000 and len(clock_idx) < clock_buf.shape[0]:
            clock_stream.wait_event(ev0)
            with torch.cuda.stream(clock_stream):
                hipops.clock_probe(clock_buf[len(clock_idx)], 100)
            clock_idx.append(len(hipops.CONV_TRACE))
    # The timed region: EXACTLY args.steps steps, all through the pipeline (depth 1: in this thread); the last trace_steps
    # record HIP events around every convolution launch of the thread that runs them (hipops.set_thread_trace; start event, launch
    # and end event of a traced launch are enqueued under one lock, so no other batch's launch falls between them).
    step_traces = []

    step_done = [0.0] * args.steps                       # completion time of every timed step (FPCC_BENCH_STEP_TIMES=1 prints them)

    def timed_step(ctx_model, i):
        if i < args.steps - trace_steps:
            try:
                return step_of(ctx_model, i)
            finally:
                step_done[i] = time.perf_counter()
        mine = []
        hipops.set_thread_trace(mine)
        try:
            return step_of(ctx_model, i)
        finally:
            hipops.set_thread_trace(None)
            step_traces.append((i, mine))

    barrier()
    t0 = time.perf_counter()
    results = pipeline.map(timed_step, range(args.steps))
    barrier()
    elapsed = time.perf_counter() - t0
    hbm_peak_gib = torch.cuda.max_memory_allocated(device) / 2 ** 30          # reference frames, warm-up and the timed region
    data, rec = results[-1][0][0], results[-1][1][0]
    del results
    if os.environ.get('FPCC_BENCH_STEP_TIMES') == '1':
        done = sorted(t for t in step_done if t > 0)
        print('step completions, ms after the start of the timed region:', ' '.join(f'{(t - t0) * 1e3:.1f}' for t in done), file=sys.stderr)
    hipops.CONV_TRACE = [e for _, tr in sorted(step_traces, key=lambda x: x[0]) for e in tr]
    pipeline.close()
    del pipeline
    trace, hipops.CONV_TRACE = hipops.CONV_TRACE, None
    # one more frame, outside the timed region and outside `roofline`'s events, for the clock: the probe kernel beside a launch
    # disturbs that launch's own timing (+15 % on the traced kernel time when both were taken in the same steps)
    hipops.reserve_trace_events(600)
    hipops.CONV_TRACE, hipops.CLOCK_HOOK = [], clock_hook
    if B > 1:                                            # the clock of the workload the timed region ran: one batch, alone
        extra = model.compress_many(frames)
        torch.cuda.synchronize()
        ME.clear_global_coordinate_manager()
        model.decompress_many(extra)
        torch.cuda.synchronize()
        ME.clear_global_coordinate_manager()
        del extra
    else:
        one_frame(frame)
    clock_trace, hipops.CONV_TRACE, hipops.CLOCK_HOOK = hipops.CONV_TRACE, None, None

    elapsed_max, total_points = replicas.aggregate(elapsed, float(points_per_step) * args.steps, device)
    # self-validation of an N-rank run: ranks counted by an all-reduce of ones on the device, every rank's own throughput min / max
    spread = replicas.rank_spread(points_per_step * args.steps / elapsed / 1e6, device)
    spread_one = replicas.rank_spread(value_one_frame, device)

    # rate / distortion of the frame just coded (outside the timed region): bpp and D1-PSNR as the reference's evaluator
    # reports them, distortion computed on the device (fastpcc_amd/evaluators.py)
    # (point-to-point and point-to-plane lines of pc_error; normals estimated on the device as the reference does with Open3D when
    # the PLY carries none -- 14 ms for the 1 M-voxel pair, profiles/r06/README.md)
    from fastpcc_amd.evaluators import pc_error_metrics
    quality = pc_error_metrics(frame[:, 1:], rec, args.resolution)

    n_bytes = len(data)
    out = None
    if rank == 0:
        # dominant kernel: the MFMA sparse convolution.  algorithmic flop / measured duration of its launches
        cache = {}
        flops = ms = bytes_fused = 0.0
        n_launch = 0
        ms_valu = 0.0
        for ev0, ev1, info in trace:
            dt = ev0.elapsed_time(ev1)
            if info['mfma']:
                fl = conv_flops(info, cache)
                flops += fl
                # B_fused of SURVEY.md 8(d): inputs and outputs once, 8 B per rulebook pair, weights once
                if 'bytes' in info:
                    bytes_fused += info['bytes']
                else:
                    pairs = fl / (2.0 * info['c_in'] * info['c_out'])
                    rows_in = info['n_out'] if info['nbr'] is None or info['n_offsets'] == 27 else pairs
                    bytes_fused += 4.0 * (rows_in * info['c_in'] + info['n_out'] * info['groups'] * info['c_out']) + \
                        8.0 * pairs + 4.0 * info['groups'] * info['n_offsets'] * info['c_in'] * info['c_out']
                ms += dt
                n_launch += 1
            else:
                ms_valu += dt
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # duration-weighted mean of the shader clock measured beside the large launches
        clk_num = clk_den = 0.0
        for (cyc, ticks), ti in zip(clock_buf[:len(clock_idx)].tolist(), clock_idx):
            if ticks > 0 and ti < len(clock_trace):
                dt = clock_trace[ti][0].elapsed_time(clock_trace[ti][1])
                clk_num += cyc / ticks * 100.0 * dt
                clk_den += dt
        shader_mhz = clk_num / clk_den if clk_den > 0 else None
        if args.dump_trace:
            per_step = len(trace) // trace_steps
            with open(args.dump_trace, 'w') as f:
                f.write('kind c_in c_out n_out n_off groups ms algo_gflop algo_tflops dense_tflops\n')
                for ev0, ev1, info in trace[-per_step:]:
                    dt = ev0.elapsed_time(ev1)
                    fl = conv_flops(info, cache)
                    dense = 2.0 * info['n_out'] * info['groups'] * info['n_offsets'] * info['c_in'] * info['c_out']
                    f.write(f"{'mfma' if info['mfma'] else 'valu'} {info['c_in']} {info['c_out']} {info['n_out']} "
                            f"{info['n_offsets']} {info['groups']} {dt:.4f} {fl / 1e9:.3f} {fl / dt / 1e9:.2f} "
                            f"{dense / dt / 1e9:.2f}\n")
        traffic, traffic_src = pmc_traffic(B, n_launch // trace_steps)
        out = {
            # the protocol is part of the name so that this figure is never compared with a one-frame-at-a-time figure (earlier rounds'
            # `value`, the reference's Timer placement): that one is `value_one_frame`
            'metric': f'encode+decode Mpoints/sec, lossy_coord_v2 baseline_r1 (stream of {B}-frame batches, {depth} in flight; '
                      'one frame at a time = value_one_frame)',
            'value': round(total_points / elapsed_max / 1e6, 4),
            'unit': 'Mpoints/s',
            # the same metric by the BASELINE definition: N0 / (t_enc + t_dec), ONE frame at a time, each half closed by a device
            # synchronise (the reference's Timer blocks) -- the figure to compare with earlier rounds and with the reference's test loop
            'value_one_frame': round(value_one_frame, 4),
            'value_batched': round(total_points / elapsed_max / 1e6, 4),
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'rccl_ranks': spread['ranks'], 'collective_backend': spread['backend'],
            'per_rank_value': {'min': round(spread['min'], 4), 'max': round(spread['max'], 4)},
            'per_rank_value_one_frame': {'min': round(spread_one['min'], 4), 'max': round(spread_one['max'], 4)},
            'ms_per_step': round(elapsed_max / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'lossy_coord_v2/baseline_r1 inference, {n_points}-voxel {args.resolution}^3 '
                                   f'body-surface frames (cfg#2), seeded random-init weights; one step = a batch of {B} such frames per GPU '
                                   f'({points_per_step} voxels) through compress_many + decompress_many = ONE network traversal over the batch, '
                                   f'{B} independent streams byte-identical to the frames coded alone (checked in every timed step); '
                                   f'{depth} batch(es) in flight per GPU',
                       'parallelism': (f'replicas x{world} (independent frames)' if world > 1 else 'single GPU') +
                                      (f', {depth} batches in flight per GPU on one stream (fastpcc_amd/serving.py)' if depth > 1 else ''),
                       'batch_clouds': B, 'frames_in_flight': depth, 'named_stages': bool(use_stages), 'voxels_per_step': points_per_step,
                       'value_note': 'value = steps x voxels_per_step / elapsed of the whole timed region (throughput of a stream of frames); '
                                     f'value_one_frame = N0 / median(t_enc + t_dec) over {len(lat)} frames coded one at a time, each half closed '
                                     'by a device synchronise (the reference\'s timer placement), measured before the timed region; '
                                     'encode_ms / decode_ms = the medians of those frames',
                       'encode_ms': round(enc_ms, 3), 'decode_ms': round(dec_ms, 3), 'one_frame_ms': round(one_frame_ms, 3),
                       'bytes': n_bytes, 'bpp': round(8 * n_bytes / n_points, 4),
                       'd1_psnr_db': round(quality['mseF,PSNR (p2point)'], 3),
                       'd2_psnr_db': round(quality['mseF,PSNR (p2plane)'], 3),
                       'quality_note': 'random-init weights: bpp / PSNR are parity checks, not RD results',
                       'coder_handover_retries': GeoLosslessEntropyModel.handover_retries,
                       'hbm_peak_gib': round(hbm_peak_gib, 1),
                       'host_binding': numa if numa is not None else 'none'},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / MFMA_PEAK_TFLOPS, 4),
                         'traffic': None if traffic is None else round(traffic),
                         'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, 2*FETCH_SIZE + WRITE_SIZE)',
                         'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': round(bytes_fused / max(n_launch, 1)),
                         'kernel': 'k_conv_wave / k_pointwise_wave / k_mlp_chain / k_conv_mfma (fp32 gather->MFMA sparse convolution: wave-autonomous '
                                   'kernel -- grouped over four offset groups for multi-offset layers --, persistent per-point kernel, fused per-point '
                                   'chains, workgroup-tiled kernel for the narrow shapes)',
                         'launches_per_step': n_launch // trace_steps,
                         'kernel_ms_per_step': round(ms / trace_steps, 3),
                         'algorithmic_gflop_per_step': round(flops / trace_steps / 1e9, 2),
                         'other_conv_ms_per_step': round(ms_valu / trace_steps, 3),
                         'event_traced_steps': f'{trace_steps} of {args.steps} (the last of the timed region, traced inside the pipeline: events '
                                               f'around every convolution launch of the threads that run them; a launch covers the {B} frames of its batch)',
                         'shader_clock_mhz': None if shader_mhz is None else round(shader_mhz),
                         'frac_at_shader_clock': None if shader_mhz is None else
                         round(achieved / (MFMA_PEAK_TFLOPS * shader_mhz / 2400.0), 4),
                         'shader_clock_note': 'mean clock measured beside the 3x3x3 launches on maps >= 50 K rows of one extra step (one batch) after the timed '
                                              'region (fpcc_clock_probe; profiles/r03/clock_ramp.md); `peak` and `frac` are quoted at the '
                                              'nominal 2400 MHz'},
        }

    # N > 1: the training configuration (cfg#5) shards over the same ranks -- global batch 8 split 8 / N, gradients
    # all-reduced over RCCL by DDP -- so the driver's own `bench.py --gpus N` runs produce the DDP curve next to the replica
    # curve.  Collective over all ranks, after (outside) the timed region of the headline metric.  The headline record above is
    # complete before this leg starts: should a rank fail inside it and leave the others blocked in a collective, rank 0's watchdog
    # prints the record and ends the job with a non-zero code instead of losing the measurement (the process group's own
    # timeout, 5 minutes, would otherwise be the only way out).
    failed = False
    if args.secondary and world > 1:
        watchdog = None
        if rank == 0:
            import threading

            def give_up():
                out['config']['secondary'] = {'cfg5_training_ddp': {'error': f'no result within {args.ddp_deadline} s: a rank failed or a '
                                                                             'collective hung; headline emitted by the watchdog'}}
                print(json.dumps(out), flush=True)
                os._exit(3)
            watchdog = threading.Timer(args.ddp_deadline, give_up)
            watchdog.daemon = True
            watchdog.start()
        del model, rec
        ME.clear_global_coordinate_manager()
        torch.cuda.empty_cache()
        from fastpcc_amd.train import ddp_training_record
        try:
            ddp_record = ddp_training_record(args.ddp_steps, 3, device)
        except Exception as e:                                           # the headline number must survive a secondary failure
            ddp_record = {'error': repr(e)[:200]}
            failed = True
        if watchdog is not None:
            watchdog.cancel()
        if rank == 0:
            out['config']['secondary'] = {'cfg5_training_ddp': ddp_record}
            failed = failed or (ddp_record is not None and 'error' in ddp_record)
    if rank == 0:
        if args.secondary and world == 1:
            del model, frame, frames
            out['config']['secondary'] = secondary(device)
        if cpu_job is not None:
            out['cpu_baseline'] = cpu_job.result()
        print(json.dumps(out), flush=True)
    if failed:
        # no further collective after a failed one (the other ranks may never arrive): leave without tearing the group down
        sys.stdout.flush()
        os._exit(4)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
