"""fpcc_mlp_chain_f32: a stack of per-point layers (MinkowskiLinear + bias + PReLU / clamp, one channel concatenation) as ONE
launch -- against the layer-by-layer launches of fpcc_conv_f32 (bit for bit: same FMA chains), against the CPU oracle's chain
evaluation (bit for bit) and through the codec's decoder blocks (fused and unfused give the same bytes)."""
import numpy as np
import pytest
import torch

from oracle import sparse_conv as sc

pytestmark = pytest.mark.gpu


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(t):
    return np.ascontiguousarray(t.cpu().numpy()).view(np.uint32)


def _layers(rng, cx, widths, cat_layer, cy, last_clip):
    out, c_in = [], cx
    for l, c_out in enumerate(widths):
        if l == cat_layer:
            c_in += cy
        w = (rng.normal(size=(c_in, c_out)) / np.sqrt(c_in)).astype(np.float32)
        b = rng.normal(size=c_out).astype(np.float32) if l != 1 else None          # one layer without a bias
        act = (1, 2, 1, 0)[l % 4]                                                  # PReLU, ReLU, PReLU, none
        out.append((w, b, act, 0.1 + 0.07 * l, last_clip if l == len(widths) - 1 else 0.0))
        c_in = c_out
    return out


CHAINS = [  # cx, widths, cat_layer, cy
    (1, [64, 128, 128, 128], 2, 128),          # SubDecoderGeoLossl of baseline_r1
    (128, [128, 128], -1, 0),                  # SubDecoderGeoLossl2
    (128, [64, 64], -1, 0),
    (64, [64, 64], -1, 0),                     # SubDecoderGeoLossl2 of the 64-channel top level
    (64, [32], -1, 0),
    (256, [128, 64, 32], -1, 0),
    (1, [32, 64], 1, 32),
    (32, [32, 128, 64, 128], 3, 96),
    (96, [128, 32], 1, 64),
]


@pytest.mark.parametrize('n', [1, 31, 32, 33, 63, 64, 65, 1000, 70001])
@pytest.mark.parametrize('cx,widths,cat_layer,cy', CHAINS)
def test_fused_chain_equals_the_separate_launches(n, cx, widths, cat_layer, cy):
    from fastpcc_amd import hipops as ops
    assert ops.mlp_chain_ok(cx, widths, cat_layer, cy)
    rng = np.random.default_rng(n + 7 * cx + sum(widths) + 3 * cy)
    x = _cuda(rng.normal(size=(n, cx)).astype(np.float32))
    y = _cuda(rng.normal(size=(n, cy)).astype(np.float32)) if cy else None
    spec = _layers(rng, cx, widths, cat_layer, cy, 1.7)
    dev = [(_cuda(w), None if b is None else _cuda(b), act, torch.tensor([s], device='cuda') if act == 1 else None, clip)
           for w, b, act, s, clip in spec]
    h = x
    for l, (w, b, act, slope, clip) in enumerate(dev):
        h = ops.conv_f32(h, w, w.shape[1], n, x2=y if l == cat_layer else None, bias=b, act=act, slope=slope, clip=clip, pack=True)
    before = ops.mlp_chain_set_form(1)
    try:
        for form in (1, 0):                       # workgroup form (weights in registers; the codecs' shapes) and wave form
            ops.mlp_chain_set_form(form)
            got = ops.mlp_chain(x, dev, y=y, cat_layer=cat_layer)
            assert got.shape == h.shape and (_bits(got) == _bits(h)).all(), form
    finally:
        ops.mlp_chain_set_form(before)


@pytest.mark.parametrize('cx,widths,cat_layer,cy', CHAINS[:3] + CHAINS[5:7])
def test_fused_chain_equals_the_oracle_chain(cx, widths, cat_layer, cy):
    """the CPU oracle (oracle/sparse_conv.c: one FMA chain per output element in the documented order) layer by layer"""
    from fastpcc_amd import hipops as ops
    n = 777
    rng = np.random.default_rng(cx + sum(widths))
    x = rng.normal(size=(n, cx)).astype(np.float32)
    y = rng.normal(size=(n, cy)).astype(np.float32) if cy else None
    spec = _layers(rng, cx, widths, cat_layer, cy, 0.9)
    dev = [(_cuda(w), None if b is None else _cuda(b), act, torch.tensor([s], device='cuda') if act == 1 else None, clip)
           for w, b, act, s, clip in spec]
    got = ops.mlp_chain(_cuda(x), dev, y=None if y is None else _cuda(y), cat_layer=cat_layer)
    h = x
    for l, (w, b, act, s, clip) in enumerate(spec):
        c1 = h.shape[1]
        x2 = y if l == cat_layer else None
        order = ops.conv_order(c1, 0 if x2 is None else x2.shape[1], w.shape[1], 1, 1, n)
        h = sc.conv_chain(h, None, w[None], b, n, x2=x2, act=act, slope=s, clip=clip, order=order)
    assert (_bits(got) == np.ascontiguousarray(h).view(np.uint32)).all()


def test_argument_checks():
    from fastpcc_amd import hipops as ops
    x = torch.zeros((10, 48), device='cuda')
    w = torch.zeros((48, 64), device='cuda')
    with pytest.raises(Exception):
        ops.mlp_chain(x, [(w, None, 0, None, 0.0)])                      # 48 input channels: not 1 and not a multiple of 32
    assert not ops.mlp_chain_ok(48, [64]) and not ops.mlp_chain_ok(64, [48]) and not ops.mlp_chain_ok(64, [32] * 5)
    assert not ops.mlp_chain_ok(64, [32, 32], 0, 32) and ops.mlp_chain_ok(64, [32, 32], 1, 32)


def test_decoder_blocks_fused_and_unfused_write_the_same_stream():
    """lossy_coord_v2: compress + decompress with the per-point chains fused (default) and layer by layer: same bytes, same
    reconstruction"""
    from fastpcc_amd import sparse_conv_layers as scl
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    model = model.cuda().eval()
    xyz = surface_cloud(5, 128, 40000)
    frame = torch.from_numpy(batched(xyz)).to(torch.int32).cuda()
    assert scl.FUSE_MLP_CHAINS
    calls = scl.CHAIN_CALLS
    a = model.compress(frame)
    ra = model.decompress(a)
    assert scl.CHAIN_CALLS > calls                                      # the fused path is the one that ran
    scl.FUSE_MLP_CHAINS = False
    try:
        b = model.compress(frame)
        rb = model.decompress(b)
    finally:
        scl.FUSE_MLP_CHAINS = True
    assert a == b and torch.equal(ra, rb)


@pytest.mark.parametrize('n', [1, 255, 5000, 8192, 70001])
@pytest.mark.parametrize('c0,c1', [(16, 8), (8, 4)])
def test_narrow_head_equals_the_engine_blocks(n, c0, c1):
    """fpcc_pointwise_head_f32 (the decoder's classify block as one launch) against the two ConvBlocks on fastpcc_amd.engine, below
    and above the row count from which the hidden layer is evaluated zero-padded on the MFMA kernel (another summation order)"""
    from fastpcc_amd import engine as ME
    from fastpcc_amd.codecs.lossy_coord_v2.layers import classify_head
    from fastpcc_amd.sparse_conv_layers import ConvBlock
    import torch.nn as nn
    torch.manual_seed(n + c0)
    seq = nn.Sequential(ConvBlock(c0, c1, 1, 1, act='prelu'), ConvBlock(c1, 1, 1, 1, act=None)).cuda().eval()
    with torch.no_grad():
        for p in seq.parameters():
            p.normal_(0, 0.4)
        seq[0].act_module.module.weight.fill_(0.17)
        coords = torch.zeros((n, 4), dtype=torch.int32, device='cuda')
        coords[:, 1] = torch.arange(n, device='cuda') % 1024
        coords[:, 2] = torch.arange(n, device='cuda') // 1024
        cm = ME.CoordinateManager(D=3)
        x = ME.SparseTensor(torch.randn((n, c0), device='cuda'), coordinates=coords, coordinate_manager=cm)
        want = seq(x).F
        got = classify_head(seq, x).F
    assert got.shape == want.shape == (n, 1)
    assert (_bits(got) == _bits(want)).all()
