"""lossy_coord_v2/baseline_r1 on the GPU against the CPU oracle, same seeded weights, same seeded cloud.

What is asserted, in decreasing strictness:
  * every layer's activations: BIT-EXACT (device kernels and oracle evaluate the same documented FMA chains);
  * residual symbols, occupancy symbols, point counts: identical;
  * 16-bit occupancy probabilities: identical (numerics version 3 specifies the logistic function), hence the BITSTREAMS are
    byte-identical and cross-decodable -- asserted without tolerance;
  * GPU decode(GPU encode(x)) returns exactly the coded number of points and is deterministic;
  * reference-shaped evaluation (gather/GEMM/scatter-add oracle): bitstream length within 2 %.
"""
import numpy as np
import pytest
import torch

from fastpcc_amd.engine import summation_order as ME_order
from oracle.codec_v2 import OracleV2
from util import batched, enliven, surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup():
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)
    weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    model = model.cuda().eval()
    model.em_lossless_based.keep_symbols = True
    return cfg, model, weights, hipops


def _cloud(seed, res, n, shift=(0, 0, 0)):
    xyz = surface_cloud(seed, res, n)
    return xyz, batched(xyz) + np.array([0, *shift])


def test_encode_decode_against_oracle(setup):
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(1, 64, 20000, (3, 0, 7))
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    # shuffle the input rows: the codec must not depend on the input order
    perm = torch.randperm(dev.shape[0], generator=torch.Generator().manual_seed(1)).cuda()
    data = model.compress(dev[perm])
    sym = model.em_lossless_based.last_symbols
    rec = model.decompress(data).cpu().numpy()
    assert rec.shape == (len(xyz), 3)
    assert len(np.unique(rec, axis=0)) == len(rec)
    assert model.compress(dev) == data                                  # deterministic, order independent

    o = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    want = o.compress(coords)
    assert (sym['residual'].reshape(-1) == o.symbols['residual'].reshape(-1)).all()
    assert (sym['occupancy'].astype(bool) == np.concatenate(o.symbols['occupancy'])).all()
    assert sym['sizes'] == [len(m) for m in o.symbols['occupancy']]
    p_gpu, p_cpu = sym['prob'].astype(np.int64), np.concatenate(o.symbols['prob']).astype(np.int64)
    # numerics version 3 specifies the logistic function: probabilities, bytes and reconstructions are equal, no tolerance
    assert (p_gpu == p_cpu).all()
    assert data == want
    assert (o.decompress(data) == rec).all()
    assert (model.decompress(want).cpu().numpy() == rec).all()

    # reference-shaped evaluation of the same network: only fp32 re-association apart
    o2 = OracleV2(weights, cfg, conv='mm')
    ref = o2.compress(coords)
    assert abs(len(ref) - len(data)) <= 0.02 * len(ref)
    assert np.mean(o2.symbols['residual'].reshape(-1) != sym['residual'].reshape(-1)) < 0.02


def test_layer_activations_bit_exact(setup):
    """run the GPU network layer by layer on the ORACLE's inputs is implied by the symbol equality above; here the
    final decoder features and logits are compared directly on a second cloud"""
    cfg, model, weights, ops = setup
    from fastpcc_amd import engine as ME
    xyz, coords = _cloud(2, 64, 12000)
    coords[:, 1:] -= coords[:, 1:].min(0)                # compress() codes coordinates relative to their minimum
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    o = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    o.keep_trace = True
    o.compress(coords)
    with torch.no_grad():
        x = model.get_sparse_pc(dev)
        y, counts = model.encoder(x)
        assert counts == [[len(xyz)]]
        got = y.F.cpu().numpy()
        assert (got.view(np.uint32) == o.trace['encoder.blocks.1.1'].view(np.uint32)).all()
        feas = model.em_lossless_based.encoder(y, 1)
        for i, f in enumerate(feas[1:]):
            name = f'em_lossless_based.encoder.blocks_out.{i}' if i >= cfg.skip_encoding_fea else \
                f'em_lossless_based.encoder.blocks.{i}.1'
            assert (f.F.cpu().numpy().view(np.uint32) == o.trace[name].view(np.uint32)).all(), name
    ME.clear_global_coordinate_manager()


@pytest.mark.parametrize('seed,res,n', [(3, 32, 3000), (4, 128, 90000)])
def test_roundtrip_sizes(setup, seed, res, n):
    cfg, model, _, _ = setup
    xyz, coords = _cloud(seed, res, n)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    out = model.test_forward(__import__('fastpcc_amd.data', fromlist=['PCData']).PCData(xyz=dev))
    assert out['pred'].shape == (len(xyz), 3)
    assert out['bpp'] > 0 and out['encode time'] > 0 and out['decode time'] > 0
    lo, hi = coords[:, 1:].min(0), coords[:, 1:].max(0)
    rec = out['pred'].cpu().numpy()
    assert (rec >= lo - 1).all() and (rec <= hi + 2).all()


def test_tiny_cloud(setup):
    """a handful of voxels: every pyramid level has one row; exercises the tail handling of every kernel"""
    cfg, model, weights, ops = setup
    coords = np.array([[0, 10, 10, 10], [0, 11, 10, 10], [0, 10, 11, 11], [0, 40, 41, 42]], dtype=np.int64)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    data = model.compress(dev)
    rec = model.decompress(data).cpu().numpy()
    assert rec.shape == (4, 3)
    o = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    want = o.compress(coords)
    assert len(want) == len(data)


def test_partitioned_stream(setup):
    cfg, model, _, _ = setup
    xyz, coords = _cloud(5, 64, 16000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    half = dev[dev[:, 1] < 32].contiguous(), dev[dev[:, 1] >= 32].contiguous()
    blob = model.compress_partitions([dev, *half])
    rec = model.decompress_partitions(blob)
    assert rec.shape[0] == dev.shape[0]
    n0 = int.from_bytes(blob[:3], 'little')
    assert blob[3:3 + n0] == model.compress(half[0])


def test_rate_objective_tracks_the_coded_size(setup):
    """GeoLosslessEntropyModel.forward (training-mode rate terms): the occupancy cross-entropy must price the
    occupancy streams the binary coder actually writes (same logits, same masks) to within 1 % + 64 bits a stream,
    and every level has its term.  State-dict keys of the noisy deep-factorised prior are the reference's."""
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(5, 64, 20000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    model.compress(dev)
    sizes = model.em_lossless_based.last_symbols['sizes']
    sym = model.em_lossless_based.last_symbols
    # ideal code length of the coded masks under the coded 16-bit probabilities
    p = sym['prob'].astype(np.float64) / 65536
    bits_ideal = -(np.log2(np.where(sym['occupancy'].astype(bool), p, 1 - p))).sum()
    em = model.em_lossless_based
    em.train()
    try:
        with torch.no_grad():
            sparse_pc = model.get_sparse_pc(dev)
            feature, _ = model.encoder(sparse_pc)
            torch.manual_seed(3)
            top, loss = em(feature, 1)
    finally:
        em.eval()
    coord_keys = sorted(k for k in loss if k.startswith('coord_'))
    assert len(coord_keys) == len(sizes) == 6
    # the training forward feeds NOISY residuals down the pyramid, so its logits differ slightly from the coded ones
    total = sum(float(loss[k]) for k in coord_keys)
    assert total == pytest.approx(bits_ideal, rel=0.25)
    fea_keys = [k for k in loss if k.startswith('fea_')]
    assert 'fea_bottom_bits_loss' in fea_keys and len(fea_keys) == 1 + 10
    assert all(np.isfinite(float(v)) and float(v) > 0 for v in loss.values())
    assert top.F.shape == feature.F.shape
    keys = [k for k in model.state_dict() if 'bottom_fea_entropy_model' in k]
    assert 'em_lossless_based.bottom_fea_entropy_model.prior._extra_state' in keys
    assert sum('prior_weights' in k for k in keys) == 5 and sum('prior_factors' in k for k in keys) == 4


def test_unused_encoder_tail_does_not_change_the_bytes(setup):
    """compress() stops after the last level that feeds the bitstream; evaluating the remaining feature predictors like
    the reference does (their result is discarded there) must give the same bytes."""
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(6, 128, 60000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    em = model.em_lossless_based
    short = model.compress(dev)
    em.evaluate_unused_tail = True
    try:
        full = model.compress(dev)
    finally:
        em.evaluate_unused_tail = False
    assert short == full
    assert model.decompress(short).shape[0] == len(xyz)


def test_deferred_occupancy_predictors_do_not_change_the_bytes(setup):
    """compress() runs the feature chain first and the occupancy predictors afterwards, finest level first (so that the longest host
    coding pass starts while GPU work is left); in chain order -- the reference's -- the bytes and the kept symbols are the same."""
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(7, 128, 60000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    em = model.em_lossless_based
    assert em.defer_occupancy
    deferred = model.compress(dev)
    sym_deferred = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in em.last_symbols.items()}
    em.defer_occupancy = False
    try:
        in_order = model.compress(dev)
        sym_in_order = em.last_symbols
    finally:
        em.defer_occupancy = True
    assert deferred == in_order
    # the same predictors enqueued on a second stream the moment their input exists (experiment of round 4, off by default: no faster)
    em.occupancy_stream = True
    try:
        for _ in range(3):                           # more than once: a race between the two streams would not show every time
            assert model.compress(dev) == deferred
    finally:
        em.occupancy_stream = False
    assert sym_deferred['sizes'] == sym_in_order['sizes'] and len(sym_deferred['sizes']) >= 2
    assert (sym_deferred['occupancy'] == sym_in_order['occupancy']).all() and (sym_deferred['prob'] == sym_in_order['prob']).all()
    assert model.decompress(deferred).shape[0] == len(xyz)


def test_test_forward_reports_rate_and_d1(setup):
    """PCC.forward in eval mode = the reference's test_forward: bytes, bpp, timings, and the evaluator's D1 numbers
    (computed on the device) equal the CPU oracle's on the same reconstruction"""
    from fastpcc_amd.data import PCData
    from oracle import metrics as om
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(7, 64, 15000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    model.evaluator.reset()
    ret = model(PCData(xyz=dev, resolution=[64], file_path=['cloud7.ply'], org_points_num=[len(xyz)]))
    assert ret['pred'].shape[1] == 3 and ret['bpp'] == 8 * len(ret['compressed_bytes']) / len(xyz)
    info = model.evaluator.file_path_to_info['cloud7.ply']
    want = om.d1(xyz, ret['pred'].cpu().numpy(), 64)
    for k, v in want.items():
        assert info[k] == pytest.approx(v, rel=1e-12), k
    assert info['compressed_bytes'] == len(ret['compressed_bytes']) and info['output_points_num'] == ret['pred'].shape[0]
    mean = model.evaluator.show(None)
    assert mean['samples_num'] == 1 and 'mseF,PSNR (p2point)(mean)' in mean


def test_entropy_model_takes_sparse_tensors(setup):
    """the reference's models hand ME.SparseTensors to the entropy bottleneck through minkowski_tensor_wrapped_fn
    (continuous_batched.py:52,77,116): features go in as [1, N, C], results come back on the same coordinate map"""
    from fastpcc_amd import engine as ME
    cfg, model, weights, ops = setup
    xyz, coords = _cloud(8, 32, 3000)
    cm = ME.CoordinateManager(D=3)
    st = ME.SparseTensor(torch.randn((len(xyz), 1), device='cuda') * 3, coordinates=torch.from_numpy(coords).to(torch.int32).cuda(),
                         coordinate_manager=cm)
    em = model.em_lossless_based.bottom_fea_entropy_model
    em.train()
    try:
        y, loss = em(st)
    finally:
        em.eval()
    assert isinstance(y, ME.SparseTensor) and y.coordinate_map_key == st.coordinate_map_key and y.F.shape == st.F.shape
    assert float(loss['bits_loss'].detach()) > 0
    strings, shape, deq = em.compress(st)
    assert isinstance(deq, ME.SparseTensor) and len(strings) == 1
    back = em.decompress(strings, shape, st.F.device, sparse_tensor_coords_tuple=(st.coordinate_map_key, cm))
    assert isinstance(back, ME.SparseTensor) and torch.equal(back.F, deq.F)
    from fastpcc_amd.sparse_conv_layers import minkowski_tensor_split, minkowski_tensor_wrapped_op
    wide = ME.SparseTensor(torch.randn((len(xyz), 6), device='cuda'), coordinate_map_key=st.coordinate_map_key, coordinate_manager=cm)
    a, b = minkowski_tensor_split(wide, [2, 4])
    assert a.F.shape[1] == 2 and b.F.shape[1] == 4 and torch.equal(torch.cat((a.F, b.F), 1), wide.F)
    doubled = minkowski_tensor_wrapped_op(wide, lambda t: t * 2)
    assert isinstance(doubled, ME.SparseTensor) and torch.equal(doubled.F, wide.F * 2)


def test_edge_inputs(setup):
    """one voxel; duplicated input points (the engine de-duplicates like ME.SparseTensor with UNWEIGHTED_AVERAGE, so the
    stream equals that of the unique cloud); the largest codable offset; an offset that does not fit the 16-bit header"""
    cfg, model, weights, ops = setup
    one = torch.tensor([[0, 5, 6, 7]], dtype=torch.int32).cuda()
    rec = model.decompress(model.compress(one))
    assert rec.shape == (1, 3)
    xyz, coords = _cloud(9, 64, 5000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    dup = torch.cat((dev, dev[::3], dev[:100]))
    assert model.compress(dup[torch.randperm(dup.shape[0], device='cuda')]) == model.compress(dev)
    far = dev.clone()
    far[:, 1:] += torch.tensor([65535 - int(dev[:, 1].min()), 100, 7], dtype=torch.int32, device='cuda')
    data = model.compress(far)
    assert int.from_bytes(data[0:2], 'little') == 65535
    back = model.decompress(data)
    assert int(back[:, 0].min()) >= 65535 and back.shape[0] == len(xyz)
    far[:, 1] += 1
    with pytest.raises(OverflowError):
        model.compress(far)
    with pytest.raises(RuntimeError):
        model.compress(dev.cpu())                                    # no CPU path


def test_fixed_threshold_pruning(setup):
    """adaptive_pruning = False (layers.py:176-180): no point counts in the header, a candidate is kept when its logit is
    positive or the largest of its 2x2x2 cell -- every parent cell keeps at least one child"""
    import dataclasses
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    cfg, model, weights, _ = setup
    fixed = Model(dataclasses.replace(cfg, adaptive_pruning=False))
    fixed.load_state_dict(model.state_dict(), strict=False)
    fixed = fixed.cuda().eval()
    xyz, coords = _cloud(61, 128, 20000)
    dev = torch.from_numpy(coords).to(torch.int32).cuda()
    a, b = model.compress(dev), fixed.compress(dev)
    assert len(a) - len(b) == 3 * (len(model.encoder.blocks) - 1) and a[6 + 3 * (len(model.encoder.blocks) - 1):] == b[6:]
    rec = fixed.decompress(b).cpu().numpy()
    assert rec.shape[1] == 3 and len(rec) > 0
    parents_in = {tuple(p) for p in (xyz >> 1).tolist()}
    parents_out = {tuple(p) for p in (rec >> 1).tolist()}
    assert parents_out == parents_in                      # the lossless part is exact; each of its cells keeps its maximum
    assert len(np.unique(rec, axis=0)) == len(rec)


@pytest.mark.parametrize('enc,dec', [((16, 32, 32), (32, 16)), ((16, 32, 32, 32), (32, 32, 16))])
def test_deeper_lossy_part_keeps_local_maxima_of_the_decoder_input_cells(enc, dec):
    """baseline_r3 / baseline_r5 shapes (config/convolutional/lossy_coord_v2/baseline_r3.yaml:5-6, baseline_r5.yaml:5-6): with
    two and three upsampling stages the local maxima of Decoder.get_keep are taken inside the voxels of the decoder's INPUT
    level (layers.py:153-161, max_stride_lossy_recon = 2^stages), i.e. among 64 / 512 candidates, not among the 8 siblings"""
    import dataclasses
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    cfg = dataclasses.replace(baseline_r1(), encoder_channels=enc, decoder_channels=dec,
                              geo_lossl_if_sample=(0, 1, 0, 1, 0, 1), geo_lossl_channels=(enc[-1], 32, 32, 32, 32, 32, 1),
                              compressed_channels=(1,) * 7)
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 3)
    weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    model = model.cuda().eval()
    xyz, coords = _cloud(5, 128, 30000, (1, 4, 0))
    data = model.compress(torch.from_numpy(coords).to(torch.int32).cuda())
    rec = model.decompress(data).cpu().numpy()
    assert rec.shape == (len(xyz), 3) and len(np.unique(rec, axis=0)) == len(rec)
    o = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    want = o.compress(coords)
    assert data[:6 + 3 * (len(enc) - 1)] == want[:6 + 3 * (len(enc) - 1)]
    rec_o = o.decompress(want)
    assert rec_o.shape == rec.shape
    same = len({tuple(r) for r in rec.tolist()} & {tuple(r) for r in rec_o.tolist()})
    assert same >= 0.995 * len(rec)            # identical up to threshold ties that fp32 exp rounding of the logits may flip
    pts = coords[:, 1:]                          # the cloud as coded (shifted); cells are aligned to its minimum corner
    lo, stages = pts.min(0), len(dec)
    assert {tuple(p) for p in ((rec - lo) >> stages).tolist()} == {tuple(p) for p in ((pts - lo) >> stages).tolist()}


def _golden_runs():
    """the reference runs of codec_v2.json: five stand-in widths + the run at baseline_r1.yaml's real widths (16/64 encoder,
    64 + 11 x 128 lossless levels), whose config the reference's own loader read from its own YAML"""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v2.json')) as f:
        g = json.load(f)
    keys = {f.name for f in __import__('dataclasses').fields(__import__('fastpcc_amd.codecs.lossy_coord_v2.model_config', fromlist=['ModelConfig']).ModelConfig)}
    real = dict(g['baseline_r1_yaml'])
    real['config'] = {k: v for k, v in real['config'].items() if k in keys}
    return g['runs'] + [real]


@pytest.mark.parametrize('run', _golden_runs(), ids=[r['label'] for r in _golden_runs()])
def test_against_the_reference_run(run):
    """tests/golden/codec_v2.json: the reference's own model code and coders executed by make_golden.py (CPU, torch GEMM
    summation order).  The HIP path must write the same header (offsets, per-level point counts, bottom level) and a stream
    of the same length up to fp32-rounding effects, and reconstruct (nearly) the same cloud."""
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    model = model.cuda().eval()
    xyz = np.array(run['xyz'], dtype=np.int32)
    data = model.compress(torch.from_numpy(batched(xyz)).to(torch.int32).cuda())
    want = bytes.fromhex(run['stream_hex'])
    head = 6 + (3 * (len(cfg.encoder_channels) - 1) if cfg.adaptive_pruning else 0)
    assert data[:head + 4] == want[:head + 4]
    assert abs(len(data) - len(want)) <= 0.02 * len(want) + 4
    rec = model.decompress(data).cpu().numpy()
    ref = np.array(run['recon'])
    if cfg.adaptive_pruning:
        assert len(rec) == len(ref)
    same = len({tuple(r) for r in rec.tolist()} & {tuple(r) for r in ref.tolist()})
    assert same >= 0.97 * len(ref)


def _chain_runs():
    """the chain-order reference runs of codec_v2_chain.json (round 4): the reference's model code over the stand-in engine with every
    layer summed in the order the HIP kernels document -- two stand-in widths, and the real widths of baseline_r1.yaml on two clouds"""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v2_chain.json')) as f:
        g = json.load(f)
    keys = {f.name for f in __import__('dataclasses').fields(__import__('fastpcc_amd.codecs.lossy_coord_v2.model_config', fromlist=['ModelConfig']).ModelConfig)}
    return g['numerics_version'], [dict(r, config={k: v for k, v in r['config'].items() if k in keys}) for r in g['runs']]


@pytest.mark.parametrize('run', _chain_runs()[1], ids=[r['label'] for r in _chain_runs()[1]])
def test_bytes_equal_the_reference_run_in_chain_order(run):
    """STRICT link between the HIP path and executed reference code: the reference's model (layers, traversal, coding order, framing,
    pruning rule, rANS coders -- its own source) run with the documented summation orders writes a stream; the HIP path must write
    the same BYTES and decode them to the same POINTS.  Since numerics version 3 the logistic function in front of the 16-bit
    probabilities is specified operation by operation as well (include/fpcc_hip.h), so no arithmetic is left that the two sides do
    not share instruction for instruction."""
    import hashlib
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    assert _chain_runs()[0] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_v2_chain.json'
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    model = model.cuda().eval()
    xyz = np.array(run['xyz'], dtype=np.int32)
    want = bytes.fromhex(run['stream_hex'])
    data = model.compress(torch.from_numpy(batched(xyz)).to(torch.int32).cuda())
    assert data == want
    for stream in (want, data):
        rec = model.decompress(stream).cpu().numpy().astype(np.int64)
        keys = np.sort((rec[:, 0] << 42) | (rec[:, 1] << 21) | rec[:, 2])
        assert len(rec) == run['recon_points'] and hashlib.sha256(keys.tobytes()).hexdigest() == run['recon_sha256']


def test_numerics_version_byte_is_opt_in_and_checked():
    """numerics_version_in_header: one leading byte; default off = the reference's layout; a stream of another version is refused"""
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    frame = torch.from_numpy(batched(surface_cloud(9, 64, 6000))).to(torch.int32).cuda()
    streams = []
    for flag in (False, True):
        cfg = baseline_r1()
        cfg.numerics_version_in_header = flag
        torch.manual_seed(0)
        model = Model(cfg)
        enliven(model, 0)
        model = model.cuda().eval()
        data = model.compress(frame)
        assert model.decompress(data).shape[0] == frame.shape[0]
        streams.append(data)
    assert streams[1] == bytes([hipops.numerics_version()]) + streams[0]
    with pytest.raises(ValueError, match='numerics version'):
        model.decompress(bytes([hipops.numerics_version() + 1]) + streams[0])
