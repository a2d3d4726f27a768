"""Generates the golden fixtures under tests/golden/ from the REFERENCE itself.  Runs only in the build container
(needs /root/reference and oracle/_ref built by `make -C oracle ref`); the fixtures it writes are data -- inputs and the
reference's outputs -- and are what travels to the GPU box.

    python tests/golden/make_golden.py

Sources of truth used:
  rans.json      reference C++ coders compiled from /root/reference (oracle/_ref): IndexedRansCoder (with/without
                 overflow coding, with/without indexes), BinaryRansCoder, simple RansEncoder; plus
                 batched_pmf_to_quantized_cdf incl. the known-answer assertion of
                 /root/reference/lib/entropy_models/rans_coder/__init__.py:72-77
  morton.json    /root/reference/lib/space_filling_curves/__init__.py:65-88 (CPU path, imported with the CUDA loader stubbed)
  byteslist.json /root/reference/lib/entropy_models/hyperprior/noisy_deep_factorized/utils.py:8-77
  entropy_model.json  NoisyDeepFactorizedEntropyModel of /root/reference/lib/entropy_models/continuous_batched.py
                 (imported with lib.entropy_models.rans_coder -> oracle/_ref coder and the MinkowskiEngine wrapper
                 stubbed): seeded parameters, log_prob / prob, quantised CDF tables, compress strings
  kdtree.json    partition sizes / checksums of lib/data_utils.py:168-234 kd_tree_partition on seeded coordinates
  me_semantics.json  the reference's own statements about MinkowskiEngine / torchsparse conventions: child tables, identity
                 kernels of the fold convolutions, state_dict key / shape lists of its models built on a parameter-only stub engine
  hilbert.json   keys of the reference's Hilbert state machine (table read from hilbert3d.cu, loop evaluated on the host)
  codec_v2.json  the reference's lossy_coord_v2 codec (layers / model / geo_lossl_em / ME wrapper layers + its rANS coders) executed on
                 the CPU over a functional MinkowskiEngine stand-in built on oracle/coords.py + conv_mm: streams, reconstructions
  get_keep.json  the reference's Decoder.get_keep (lossy_coord_v2/layers.py:151-180) executed over the same stand-in on seeded
                 candidate sets: logits, requested point counts, the keep masks it returned
  codec_color.json / codec_lossl.json  likewise the reference's lossy_coord_lossy_color codec (over the MinkowskiEngine stand-in) and its
                 float LiDAR model lossl_coord incl. train_forward (over the torchsparse stand-in)
  codec_v3.json  the reference's lossy_coord_v3 model executed on the CPU over a functional torchsparse stand-in (kernel-offset
                 enumeration restated, everything else the reference's code and coder): streams, side information, reconstructions
  codec_int.json the reference's integer LiDAR codec (cuda_ops.py + lossl_coord_int/model.py) executed on the CPU over a stand-in for
                 its CUDA extension (oracle/int_ops.c scalar functions, torch integer GEMMs): streams of seeded runs
  explut.json    sha256 + samples of the 6145-entry table in /root/reference/lib/int_sparse_conv/src/softmax.cu:18-20
"""
import hashlib
import importlib.util
import json
import math
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_rans():
    import rans_ext_cpp as R
    import simple_rans_ext_cpp as S
    rng = np.random.default_rng(20261002)
    out = {'cdf': [], 'indexed': [], 'binary': [], 'simple': []}

    # --- PMF -> CDF --------------------------------------------------------------------------------------------------
    known = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [2 ** -17, 1, 0, 0]], dtype=np.float64)
    cases = [(known, True), (np.array([[1, 2, 3, 1]], np.float64), False), (np.array([[.1, .2, .4, .2, .1]]), True),
             (np.array([[0, 0, 1], [1, 1, 2]], np.float64), False)]
    for _ in range(12):
        n = int(rng.integers(2, 70))
        p = rng.random((2, n)) ** 4
        p[rng.random((2, n)) < 0.3] = 0
        p[:, int(rng.integers(0, n))] += 0.5
        cases.append((p / p.sum(1, keepdims=True) * rng.choice([1.0, 0.7]), bool(rng.integers(0, 2))))
    for pmf, ov in cases:
        off = np.full(pmf.shape[0], -3, dtype=np.int32)
        cdfs = R.batched_pmf_to_quantized_cdf(pmf.copy(), off, ov)
        out['cdf'].append({'pmf': pmf.tolist(), 'overflow': ov, 'offset_in': -3, 'offset_out': off.tolist(),
                           'cdf': [list(map(int, c)) for c in cdfs]})

    # --- indexed coder -----------------------------------------------------------------------------------------------
    for ov in (False, True):
        for with_idx in (False, True):
            for _ in range(3):
                nt, ns = int(rng.integers(1, 4)), int(rng.integers(2, 40))
                pmf = rng.random((nt, ns)) ** 2
                pmf[rng.random((nt, ns)) < 0.2] = 0
                pmf[:, 0] += 1e-2
                pmf /= pmf.sum(1, keepdims=True)
                off = rng.integers(-6, 3, nt).astype(np.int32)
                coder = R.IndexedRansCoder(ov, 1)
                coder.init_with_pmfs(pmf.copy(), off)
                cdfs = coder.get_cdfs()
                n = int(rng.integers(1, 400))
                idx = rng.integers(0, nt, (1, n)).astype(np.int32) if with_idx else (np.arange(n) % nt).astype(np.int32)[None]
                lens = np.array([len(c) - 1 for c in cdfs])
                if ov:
                    sym = rng.integers(-70, 70, (1, n)).astype(np.int32)
                    sym[0, :3] = [2049, -2049, 0][:min(3, n)][:n] if n >= 3 else sym[0, :3]
                else:
                    sym = (rng.integers(0, 1 << 30, (1, n)) % lens[idx] + off[idx]).astype(np.int32)
                enc = coder.encode_with_indexes(sym, idx)[0] if with_idx else coder.encode(sym)[0]
                dec = np.empty_like(sym)
                if with_idx:
                    coder.decode_with_indexes([enc], idx, dec)
                else:
                    coder.decode([enc], dec)
                assert (dec == sym).all()
                out['indexed'].append({'overflow': ov, 'with_indexes': with_idx, 'cdfs': [list(map(int, c)) for c in cdfs],
                                       'offsets': off.tolist(), 'symbols': sym[0].tolist(),
                                       'indexes': idx[0].tolist(), 'stream': enc.hex()})
    # survey samples
    coder = R.IndexedRansCoder(False, 1)
    coder.init_with_pmfs(np.array([[1, 2, 3, 1]], np.float64), np.array([0], np.int32))
    enc = coder.encode(np.array([[0, 1, 1, 2, 2, 2, 3]], np.int32))[0]
    out['indexed'].append({'overflow': False, 'with_indexes': False, 'cdfs': coder.get_cdfs(), 'offsets': [0],
                           'symbols': [0, 1, 1, 2, 2, 2, 3], 'indexes': [0] * 7, 'stream': enc.hex()})

    # --- binary coder ------------------------------------------------------------------------------------------------
    for n in (1, 7, 100, 5000):
        p = np.clip(np.round(rng.beta(.3, .3, (1, n)) * 65536), 1, 65535).astype(np.uint32)
        bits = rng.random((1, n)) < p / 65536
        enc = R.BinaryRansCoder(1).encode(bits, p)[0]
        out['binary'].append({'prob': p[0].tolist(), 'bits': bits[0].astype(int).tolist(), 'stream': enc.hex()})

    # --- simple persistent coder -------------------------------------------------------------------------------------
    quan_cdf = np.array([[1, 2, 3, 4, 65535], [1, 2, 3, 5, 65535], [2, 3, 4, 6, 65535], [2, 3, 4, 7, 65535],
                         [1, 2, 3, 8, 65535], [1, 2, 3, 9, 65535]], dtype=np.uint16)
    quan_cdf2 = np.array([[1, 2, 4000, 5000, 65535], [2, 3, 3000, 6000, 65535], [3, 4, 3000, 7000, 65535],
                          [4, 5, 1000, 8000, 65535], [5, 6, 5000, 9000, 65535], [6, 7, 6000, 10000, 65535]], dtype=np.uint16)
    org = np.array([2, 4, 1, 1, 2, 3, 0, 2, 4, 2, 1, 1], dtype=np.uint16)
    e = S.RansEncoder(1 << 20)
    e.encode(quan_cdf2, org[6:12])
    e.encode(quan_cdf, org[:6])
    out['simple'].append({'blocks': [{'rows': quan_cdf2.tolist(), 'symbols': org[6:12].tolist()},
                                     {'rows': quan_cdf.tolist(), 'symbols': org[:6].tolist()}], 'stream': e.flush().hex()})
    logits = rng.normal(0, 3, (64, 255))
    pm = np.exp(logits)
    pm /= pm.sum(1, keepdims=True)
    f = np.floor(pm * (65536 - 255)).astype(np.int64) + 1
    rows = np.cumsum(f, 1)
    rows[:, -1] = 65535
    rows = rows.astype(np.uint16)
    sym = rng.integers(0, 255, 64).astype(np.uint16)
    e.encode(rows, sym)
    e.encode(rows[:1], sym[:50])
    out['simple'].append({'blocks': [{'rows': rows.tolist(), 'symbols': sym.tolist()},
                                     {'rows': rows[:1].tolist(), 'symbols': sym[:50].tolist()}], 'stream': e.flush().hex()})
    edge = rng.integers(1, 65535, 200).astype(np.uint16)
    bits = rng.random(200) < 0.5
    e.encode_bin(edge, bits)
    out['simple'].append({'bin': {'edge': edge.tolist(), 'bits': bits.astype(int).tolist()}, 'stream': e.flush().hex()})
    return out


def make_morton():
    # import the reference module with its CUDA extension loader stubbed (only the NumPy path is used)
    import torch.utils.cpp_extension as ce
    real = ce.load
    ce.load = lambda *a, **k: types.SimpleNamespace()
    try:
        sfc = _load(os.path.join(REF, 'lib/space_filling_curves/__init__.py'), 'ref_sfc')
    finally:
        ce.load = real
    rng = np.random.default_rng(7)
    xyz = np.concatenate((np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 7], [0x1fffff] * 3, [1023, 0, 512]]),
                          rng.integers(0, 1 << 21, (40, 3)), rng.integers(0, 1024, (40, 3)))).astype(np.int64)
    out = {'xyz': xyz.tolist(), 'keys': {}}
    for order in ('xyz', 'zyx', 'yxz'):
        for inv in (False, True):
            out['keys'][f'{order}|{int(inv)}'] = [int(v) for v in sfc.morton_encode_magicbits(xyz.copy(), order, inv)]
    return out


def make_byteslist():
    ref = _load(os.path.join(REF, 'lib/entropy_models/hyperprior/noisy_deep_factorized/utils.py'), 'ref_bl')
    rng = np.random.default_rng(11)
    out = []
    for lens in ([0, 1], [255, 256], [65535, 65536, 3], [1] * 8, [1] * 9, [300] * 7, [5, 70000, 2, 0, 9, 1000], [3] * 15,
                 [3] * 16, [3] * 17, [400] * 4, [400] * 5):
        strings = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in lens]
        blob = ref.BytesListUtils.concat_bytes_list(strings)
        out.append({'lengths': lens, 'seed_strings_sha': hashlib.sha256(b''.join(strings)).hexdigest(),
                    'strings': [s.hex() if len(s) < 64 else None for s in strings],
                    'head': blob[: len(blob) - sum(lens)].hex()})
    return out


def make_explut():
    text = open(os.path.join(REF, 'lib/int_sparse_conv/src/softmax.cu')).read()
    body = text[text.index('at::tensor('):]
    body = body[body.index('{') + 1: body.index('}')]
    vals = [int(v) for v in re.findall(r'-?\d+', body)]
    assert len(vals) == 12 * 512 + 1, len(vals)
    return {'n': len(vals), 'sha256': hashlib.sha256(np.array(vals, dtype='<i4').tobytes()).hexdigest(),
            'head': vals[:8], 'tail': vals[-8:], 'at_512': vals[512], 'at_3000': vals[3000]}


def make_entropy_model():
    import torch
    import rans_ext_cpp as R
    stub_rc = types.ModuleType('lib.entropy_models.rans_coder')
    stub_rc.IndexedRansCoder = R.IndexedRansCoder
    stub_wr = types.ModuleType('lib.minkowski_sparse_conv_layers')
    stub_wr.minkowski_tensor_wrapped_fn = lambda *a, **k: (lambda f: f)
    stub_wr.minkowski_tensor_wrapped_op = lambda *a, **k: (lambda f: f)
    sys.modules['lib.entropy_models.rans_coder'] = stub_rc
    sys.modules['lib.minkowski_sparse_conv_layers'] = stub_wr
    sys.path.insert(0, REF)
    from lib.entropy_models.continuous_batched import NoisyDeepFactorizedEntropyModel as RefEM
    from lib.entropy_models.utils import lower_bound, upper_bound, grad_scaler

    def tl(t):
        return t.detach().double().reshape(-1).tolist()

    out = {'cases': [], 'bounds': []}
    for ci, (ch, scaler, lo, hi, init_scale, n, spread) in enumerate((
            (1, 1, -64, 64, 10, 100, 3.0), (8, 1, -64, 64, 10, 60, 4.0), (8, 1, -10, 10, 10, 60, 9.0),
            (4, 2, -16, 16, 3, 50, 2.0), (32, 1, -64, 64, 10, 17, 30.0))):
        torch.manual_seed(ci)
        em = RefEM(batch_shape=torch.Size([ch]), coding_ndim=2, bottleneck_process='noise', bottleneck_scaler=scaler,
                   init_scale=init_scale, lower_bound=lo, upper_bound=hi, broadcast_shape_bytes=(3,))
        with torch.no_grad():        # leave the initial point so that factors / unequal weights are exercised
            for p in em.parameters():
                p.add_(torch.randn_like(p) * 0.3)
        x = torch.randn(1, n, ch) * spread
        x[0, 0, 0] = 500.25          # far outside the table: overflow coding
        probe = torch.cat((torch.linspace(-40, 40, 33)[:, None].expand(33, ch), x[0, :8]))
        rec = {'ch': ch, 'scaler': scaler, 'lower': lo, 'upper': hi, 'init_scale': init_scale,
               'weights': [tl(p) for p in em.prior_weights], 'biases': [tl(p) for p in em.prior_biases],
               'factors': [tl(p) for p in em.prior_factors], 'x': tl(x), 'x_shape': list(x.shape),
               'probe': tl(probe), 'probe_shape': list(probe.shape)}
        with torch.no_grad():
            rec['log_prob'] = tl(em.prior.log_prob(probe))
            rec['prob'] = tl(em.prior.prob(probe))
            rec['logits_cdf'] = tl(em.prior.base.base.logits_cdf(probe))
        em.train()
        torch.manual_seed(100 + ci)
        y, loss = em(x)
        rec['train_noise_seed'] = 100 + ci
        rec['train_y'] = tl(y)
        rec['bits_loss'] = float(loss['bits_loss'])
        loss['bits_loss'].backward()
        rec['grad_w0'] = tl(em.prior_weights[0].grad)
        rec['grad_f0'] = tl(em.prior_factors[0].grad)
        rec['grad_b_last'] = tl(em.prior_biases[-1].grad)
        em.eval()
        rec['cdfs'] = [list(map(int, c)) for c in em.prior.cdf_list]
        rec['cdf_offsets'] = [int(v) for v in em.prior.cdf_offset_list]
        strings, bshape, deq, bits = em.compress(x.clone(), estimate_bits=True)
        rec['strings'] = [t.hex() for t in strings]
        rec['est_bits'] = float(bits)
        rec['dequantized'] = tl(deq)
        back = em.decompress(strings, bshape, torch.device('cpu'))
        assert torch.equal(back, deq), ci
        sd = em.state_dict()
        rec['state_keys'] = sorted(sd.keys())
        out['cases'].append(rec)
    # a freshly initialised model (seed 0) and 100 values ~ N(0, 3^2)
    torch.manual_seed(0)
    em = RefEM(batch_shape=torch.Size([1]), coding_ndim=2, init_scale=10, broadcast_shape_bytes=(3,))
    out['fresh'] = {'biases': [tl(p) for p in em.prior_biases], 'weights0': tl(em.prior_weights[0])[:1],
                    'weights': [tl(p) for p in em.prior_weights]}
    x = torch.randn(1, 100, 1) * 3
    em.eval()
    strings, _, _, bits = em.compress(x.clone(), estimate_bits=True)
    out['fresh'].update({'est_bits': float(bits), 'cdf_len': len(em.prior.cdf_list[0]), 'cdf_offset': int(em.prior.cdf_offset_list[0]),
                         'string': strings[0].hex()})
    # gradient-shaping helpers
    for name, fn in (('lower', lower_bound), ('upper', upper_bound)):
        for mode in ('identity_if_towards', 'disconnected'):
            x = torch.tensor([-2.0, -0.5, 0.0, 0.5, 2.0, 1.0, -1.0], requires_grad=True)
            g = torch.tensor([1.0, -1.0, 1.0, -1.0, 1.0, -1.0, 1.0])
            y = fn(x, 0.25, mode) if mode != 'identity_if_towards' else fn(x, 0.25)
            y.backward(g)
            out['bounds'].append({'fn': name, 'mode': mode, 'x': tl(x), 'g': tl(g), 'y': tl(y), 'dx': tl(x.grad)})
    x = torch.tensor([1.0, -3.0], requires_grad=True)
    y = grad_scaler(x, 0.125)
    y.sum().backward()
    out['bounds'].append({'fn': 'grad_scaler', 'mode': '0.125', 'x': tl(x), 'g': [1.0, 1.0], 'y': tl(y), 'dx': tl(x.grad)})
    return out


def make_entropy_model_indexed():
    """lib/entropy_models/continuous_indexed.py + distributions/{uniform_noise,special_math}.py, imported with the same two
    stubs as make_entropy_model: log_ndtr, the noisy normal, index bounding / flattening, the CDF grid of the scale-indexed
    model, strings of a seeded input and the training rate with its gradients (no noise, so that it is reproducible)."""
    import torch
    import rans_ext_cpp as R
    stub_rc = types.ModuleType('lib.entropy_models.rans_coder')
    stub_rc.IndexedRansCoder = R.IndexedRansCoder
    stub_wr = types.ModuleType('lib.minkowski_sparse_conv_layers')
    stub_wr.minkowski_tensor_wrapped_fn = lambda *a, **k: (lambda f: f)
    stub_wr.minkowski_tensor_wrapped_op = lambda *a, **k: (lambda f: f)
    sys.modules['lib.entropy_models.rans_coder'] = stub_rc
    sys.modules['lib.minkowski_sparse_conv_layers'] = stub_wr
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from lib.entropy_models.continuous_indexed import ContinuousIndexedEntropyModel as RefEM, \
        noisy_scale_normal_indexed_entropy_model_init as ref_init
    from lib.entropy_models.distributions.uniform_noise import NoisyNormal as RefNN
    from lib.entropy_models.distributions import special_math as sm
    tl = lambda t: t.detach().to(torch.float64).flatten().tolist()
    out = {}
    xs = torch.tensor([-40.0, -30.0, -12.0, -10.0001, -10.0, -9.99, -7.5, -5.0, -1.0, -0.3, 0.0, 0.3, 1.0, 4.99, 5.0, 5.01, 8.0, 20.0])
    out['log_ndtr'] = {'x': tl(xs), 'f32': tl(sm.log_ndtr(xs)), 'f64': tl(sm.log_ndtr(xs.double()))}
    ys = torch.tensor([-70.0, -20.0, -6.0, -2.0, -1.0, -0.4, 0.0, 0.25, 1.0, 3.0, 9.0, 33.0, 64.0])[:, None]
    scales = torch.tensor([0.11, 0.5, 1.0, 3.7, 40.0, 256.0])[None]
    nn_ = RefNN(0, scales)
    out['noisy_normal'] = {'y': tl(ys), 'scale': tl(scales), 'log_prob': tl(nn_.log_prob(ys)), 'prob': tl(nn_.prob(ys))}

    cases = []
    for seed, kw in ((1, {}), (2, {'indexes_scaler': 0.5, 'indexes_offset': 2.0}), (3, {'quantize_bottleneck_in_eval': True, 'lower_bound': -20, 'upper_bound': 30})):
        torch.manual_seed(seed)
        em = RefEM(RefNN, (64,), ref_init(0.11, 256, 64), coding_ndim=2, bottleneck_process='', **kw)
        n, c = 120, 5
        idx = (torch.rand(1, n, c) * 75 - 6) / kw.get('indexes_scaler', 1) - kw.get('indexes_offset', 0)     # some out of range
        x = torch.randn(1, n, c) * torch.exp(math.log(0.11) + (math.log(256 / 0.11) / 63) * idx.clamp(0, 63) * 0.6)
        case = {'seed': seed, 'kw': kw, 'n': n, 'c': c, 'x': tl(x), 'indexes': tl(idx)}
        case['bounded'] = tl(em.bound_indexes(idx))
        case['flat'] = em.flatten_indexes(em.bound_indexes(idx)).flatten().tolist()
        em.train()
        xg, ig = x.clone().requires_grad_(), idx.clone().requires_grad_()
        y, loss = em(xg, ig)
        loss['bits_loss'].backward()
        case['train'] = {'bits': loss['bits_loss'].item(), 'dx': tl(xg.grad), 'di': tl(ig.grad), 'y_equals_x': bool(torch.equal(y.detach(), x))}
        em.eval()
        table = em.prior.cdf_list
        case['table_sha'] = hashlib.sha256(json.dumps([list(map(int, r)) for r in table]).encode()).hexdigest()[:16]
        case['table_rows'] = {str(i): list(map(int, table[i])) for i in (0, 17, 63)}
        case['offsets'] = list(map(int, em.prior.cdf_offset_list))
        strings, deq, est = em.compress(x.clone(), idx, estimate_bits=True)
        case['strings'] = [b.hex() for b in strings]
        case['estimated_bits'] = est.item()
        rec = em.decompress(strings, idx, torch.device('cpu'))
        case['roundtrip'] = bool(torch.equal(rec, deq))
        case['decoded'] = tl(rec)
        cases.append(case)
    out['models'] = cases
    return out


def make_entropy_model_hyperprior():
    """lib/entropy_models/hyperprior/noisy_deep_factorized/basic.py on plain tensors with linear hyper networks: the reference
    model's state dict (tensors) is stored, so that the same parameters can be loaded; strings, decoded values, estimated
    bits and the training losses (no noise) are the expected outputs."""
    import torch
    import torch.nn as nn
    import rans_ext_cpp as R
    stub_rc = types.ModuleType('lib.entropy_models.rans_coder'); stub_rc.IndexedRansCoder = R.IndexedRansCoder
    stub_wr = types.ModuleType('lib.minkowski_sparse_conv_layers')
    stub_wr.minkowski_tensor_wrapped_fn = lambda *a, **k: (lambda f: f)
    stub_wr.minkowski_tensor_wrapped_op = lambda x, op, **k: op(x)
    stub_wr.get_minkowski_tensor_coords_tuple = lambda x: None
    stub_tu = types.ModuleType('lib.torch_utils')
    def concat_loss_dicts(a, b, f=lambda x: x, t=lambda x: x):
        for k in b:
            a[f(k)] = a[f(k)] + t(b[k]) if f(k) in a else t(b[k])
        return a
    stub_tu.concat_loss_dicts = concat_loss_dicts
    sys.modules['lib.entropy_models.rans_coder'] = stub_rc
    sys.modules['lib.minkowski_sparse_conv_layers'] = stub_wr
    sys.modules['lib.torch_utils'] = stub_tu
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # the reference's decompress passes sparse_tensor_coords_tuple to functions whose decorator (stubbed away here) removes it
    import lib.entropy_models.continuous_batched as cb, lib.entropy_models.continuous_indexed as ci
    for cls in (cb.NoisyDeepFactorizedEntropyModel, ci.ContinuousIndexedEntropyModel):
        plain = cls.decompress
        cls.decompress = (lambda f: lambda self, *a, sparse_tensor_coords_tuple=None, **k: f(self, *a, **k))(plain)
    from lib.entropy_models.hyperprior.noisy_deep_factorized import basic as B
    tl = lambda t: t.detach().to(torch.float64).flatten().tolist()
    out = []
    for name in ('scale_normal', 'deep_factorized_transform'):
        torch.manual_seed(5)
        c, ch, n = 6, 3, 90
        if name == 'scale_normal':
            em = B.ScaleNoisyNormalEntropyModel(nn.Linear(c, ch), nn.Linear(ch, c), torch.Size([ch]), 2, num_scales=32,
                                                scale_min=0.2, scale_max=40, bottleneck_process='')
        else:
            em = B.NoisyDeepFactorizedEntropyModel(nn.Linear(c, ch), nn.Linear(ch, c * 3), torch.Size([ch]), 2,
                                                   index_ranges=(4, 4, 4), parameter_fns_type='transform',
                                                   parameter_fns_factory=lambda i, o: nn.Linear(i, o), num_filters=(1, 2, 1),
                                                   bottleneck_process='')
        with torch.no_grad():
            em.hyper_decoder.weight.mul_(3.0)
        y = torch.randn(1, n, c) * 4
        state = {k: (tl(v), list(v.shape)) for k, v in em.state_dict().items() if isinstance(v, torch.Tensor)}
        em.train()
        yg = y.clone().requires_grad_()
        y_tilde, loss = em(yg)
        (loss['bits_loss'] + loss['hyper_bits_loss']).backward()
        case = {'name': name, 'c': c, 'ch': ch, 'n': n, 'y': tl(y), 'state': state,
                'train': {'bits_loss': loss['bits_loss'].item(), 'hyper_bits_loss': loss['hyper_bits_loss'].item(), 'dy': tl(yg.grad)}}
        em.eval()
        strings, shape, deq, bits = em.compress(y.clone(), estimate_bits=True)
        case['strings'] = [b.hex() for b in strings]
        case['coding_batch_shape'] = list(shape)
        case['estimated_bits'] = bits.item()
        rec = em.decompress(strings, shape, torch.device('cpu'))
        case['decoded'] = tl(rec)
        out.append(case)
    return out


def make_kdtree():
    # lib/data_utils.py imports plyfile / open3d at module level; neither is used by kd_tree_partition
    sys.modules.setdefault('plyfile', types.SimpleNamespace(PlyData=None, PlyElement=None))
    du = _load(os.path.join(REF, 'lib/data_utils.py'), 'ref_data_utils')
    out = []
    for seed, n, rng_max, max_num in ((1, 5000, 1000, 700), (2, 4097, 64, 1000), (3, 300, 1 << 16, 1000), (4, 20000, 1024, 2600),
                                      (5, 9, 4, 2)):
        rng = np.random.default_rng(seed)
        coord = rng.integers(0, rng_max, (n, 3)).astype(np.int32)
        parts = du.kd_tree_partition(coord, max_num)[0]
        out.append({'seed': seed, 'n': n, 'range': rng_max, 'max_num': max_num, 'sizes': [len(p) for p in parts],
                    'sha': [hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest()[:16] for p in parts]})
    return out


def make_ptq_import():
    """float -> fixed-point parameter conversion of the integer operators (lib/int_sparse_conv/cuda_ops.py:65-77 residual
    block, :223-301 sparse conv, :473-509 requantiser, :542-607 linear).  The module imports torchsparse (absent here) only
    for type names; the conversions are plain tensor arithmetic and run on the CPU."""
    import torch
    import torch.nn as nn
    ts = types.ModuleType('torchsparse'); ts_nn = types.ModuleType('torchsparse.nn')
    ts.SparseTensor = type('SparseTensor', (), {}); ts_nn.Conv3d = type('Conv3d', (), {}); ts.nn = ts_nn
    sys.modules.setdefault('torchsparse', ts); sys.modules.setdefault('torchsparse.nn', ts_nn)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    co = _load(os.path.join(REF, 'lib/int_sparse_conv/cuda_ops.py'), 'ref_int_cuda_ops')
    tl = lambda t: t.detach().to(torch.float64 if t.dtype.is_floating_point else torch.int64).flatten().tolist()
    dump = lambda m: {k: tl(v) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    rnd = lambda *shape, s=1.0: (torch.rand(shape, generator=g) * 2 - 1) * s
    f32 = lambda v: torch.tensor([v], dtype=torch.float32)
    i64 = lambda v: torch.tensor([v], dtype=torch.int64)
    out = {'conv': [], 'linear': [], 'requant': [], 'prelu': [], 'resblock': []}
    for (cin, cout, ks, prelu, out8, s_in, zp_in, s_out, zp_out) in (
            (5, 7, (3, 3, 3), True, True, 0.031, 0, 0.017, 0), (5, 7, (3, 3, 3), False, False, 0.2, 0, None, None),
            (8, 3, (2, 2, 2), True, False, 0.004, 0, None, None), (1, 6, (3, 3, 3), False, False, 1.0, 0, None, None),
            (4, 4, (3, 3, 3), False, True, 0.011, 9, 0.05, -13), (3, 9, (2, 2, 2), True, True, 0.07, -3, 0.002, 5)):
        conv = types.SimpleNamespace(kernel=rnd(ks[0] * ks[1] * ks[2], cin, cout, s=0.4), bias=rnd(cout, s=0.3), kernel_size=ks,
                                     stride=(1, 1, 1) if ks[0] == 3 else (2, 2, 2))
        act = types.SimpleNamespace(weight=torch.tensor([0.21])) if prelu else None
        m = co.SparseConvIn8Out8(cin, cout, ks, conv.stride, prelu, out8)
        m.import_parameters(f32(s_in), i64(zp_in), f32(s_out) if out8 else None, i64(zp_out) if out8 else None, conv, act)
        out['conv'].append({'cin': cin, 'cout': cout, 'ks': list(ks), 'stride': list(conv.stride), 'prelu': prelu, 'out8': out8,
                            's_in': s_in, 'zp_in': zp_in, 's_out': s_out, 'zp_out': zp_out, 'kernel': tl(conv.kernel), 'bias': tl(conv.bias),
                            'slope': 0.21 if prelu else None, 'state': dump(m)})
    for (cin, cout, prelu, out8, s_in, zp_in, s_out, zp_out) in (
            (6, 9, True, True, 0.02, 17, 0.013, 0), (12, 5, False, False, 0.3, -40, None, None), (3, 255, False, False, 0.009, 0, None, None),
            (9, 8, True, False, 0.05, 3, None, None), (4, 4, False, True, 0.6, -1, 0.4, 11)):
        lin = nn.Linear(cin, cout)
        with torch.no_grad():
            lin.weight.copy_(rnd(cout, cin, s=0.7)); lin.bias.copy_(rnd(cout, s=0.5))
        act = types.SimpleNamespace(weight=torch.tensor([0.4])) if prelu else None
        m = co.LinearIn8W8(cin, cout, prelu, out8)
        m.import_parameters(f32(s_in), i64(zp_in), f32(s_out) if out8 else None, i64(zp_out) if out8 else None, lin, act)
        out['linear'].append({'cin': cin, 'cout': cout, 'prelu': prelu, 'out8': out8, 's_in': s_in, 'zp_in': zp_in, 's_out': s_out,
                              'zp_out': zp_out, 'weight': tl(lin.weight), 'bias': tl(lin.bias), 'slope': 0.4 if prelu else None,
                              'state': dump(m)})
    for s_out, zp_out in ((0.031, 0), (0.5, 0), (0.0007, -12), (3.0, 100), (1e-9, 0)):
        m = co.RequantFxpToScaledInt8()
        m.import_parameters(f32(s_out), i64(zp_out))
        out['requant'].append({'s_out': s_out, 'zp_out': zp_out, 'state': dump(m)})
    for slope in (0.25, -0.1, 1.0, 3.3333, 1e-4):
        m = co.PReLUIn32Out32()
        m.import_parameters(types.SimpleNamespace(weight=torch.tensor([slope])))
        out['prelu'].append({'slope': slope, 'state': dump(m)})
    ch = 6
    blk = types.SimpleNamespace(
        obs=types.SimpleNamespace(calculate_qparams=lambda: (f32(0.023), i64(0))),
        obs2=types.SimpleNamespace(calculate_qparams=lambda: (f32(0.041), i64(0))),
        conv=types.SimpleNamespace(kernel=rnd(27, ch, ch, s=0.3), bias=rnd(ch, s=0.2), kernel_size=(3, 3, 3), stride=(1, 1, 1)),
        act=types.SimpleNamespace(weight=torch.tensor([0.15])),
        conv2=types.SimpleNamespace(kernel=rnd(27, ch, ch, s=0.3), bias=rnd(ch, s=0.2), kernel_size=(3, 3, 3), stride=(1, 1, 1)),
        act2=types.SimpleNamespace(weight=torch.tensor([0.3])))
    m = co.SparseResBlockIn32W8Out32(ch)
    m.import_parameters(blk)
    out['resblock'].append({'ch': ch, 'scale': 0.023, 'scale2': 0.041, 'kernel': tl(blk.conv.kernel), 'bias': tl(blk.conv.bias),
                            'kernel2': tl(blk.conv2.kernel), 'bias2': tl(blk.conv2.bias), 'slope': 0.15, 'slope2': 0.3, 'state': dump(m)})
    return out


def _stub_engines():
    """Stand-ins for the two sparse-tensor engines, good for CONSTRUCTING the reference's modules only: every class creates
    the parameters / sub-modules that MinkowskiEngine 0.5.4 documents (kernel [volume, C_in, C_out] -- 2-D for volume 1 --,
    bias [1, C_out], MinkowskiLinear.linear = nn.Linear, MinkowskiPReLU.module = nn.PReLU, MinkowskiBatchNorm.bn =
    nn.BatchNorm1d) and nothing can be evaluated.  What the fixtures take from them is the reference's own module tree:
    names, order and shapes of its state_dict."""
    import enum
    import torch
    import torch.nn as nn
    ME = types.ModuleType('MinkowskiEngine')

    class KernelGenerator:
        def __init__(self, kernel_size=-1, stride=1, dilation=1, region_type=None, dimension=3, **_):
            as3 = lambda v: [v] * dimension if isinstance(v, int) else list(v)
            self.kernel_size, self.kernel_stride, self.kernel_dilation = as3(kernel_size), as3(stride), as3(dilation)
            self.kernel_volume = int(np.prod(self.kernel_size))
            self.region_type, self.dimension = region_type, dimension

    class _Conv(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                     kernel_generator=None, expand_coordinates=False, dimension=3, **_):
            super().__init__()
            kg = kernel_generator or KernelGenerator(kernel_size, stride, dilation, dimension=dimension)
            self.kernel_generator = kg
            shape = (in_channels, out_channels) if kg.kernel_volume == 1 else (kg.kernel_volume, in_channels, out_channels)
            self.kernel = nn.Parameter(torch.zeros(shape))
            self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

    def wrap(name, attr, cls):
        def init(self, *a, **k):
            nn.Module.__init__(self)
            setattr(self, attr, cls(*a, **k))
        return type(name, (nn.Module,), {'__init__': init})

    for name in ('MinkowskiConvolution', 'MinkowskiConvolutionTranspose', 'MinkowskiGenerativeConvolutionTranspose'):
        setattr(ME, name, type(name, (_Conv,), {}))
    ME.MinkowskiLinear = wrap('MinkowskiLinear', 'linear', nn.Linear)
    ME.MinkowskiBatchNorm = wrap('MinkowskiBatchNorm', 'bn', nn.BatchNorm1d)
    for name, cls in (('MinkowskiReLU', nn.ReLU), ('MinkowskiPReLU', nn.PReLU), ('MinkowskiLeakyReLU', nn.LeakyReLU),
                      ('MinkowskiSigmoid', nn.Sigmoid)):
        setattr(ME, name, wrap(name, 'module', cls))
    for name in ('MinkowskiPruning', 'MinkowskiMaxPooling', 'MinkowskiPoolingTranspose', 'MinkowskiGlobalPooling',
                 'MinkowskiAvgPooling', 'MinkowskiSumPooling', 'MinkowskiGlobalMaxPooling', 'MinkowskiGlobalAvgPooling'):
        setattr(ME, name, type(name, (nn.Module,), {'__init__': lambda self, *a, **k: nn.Module.__init__(self)}))
    ME.KernelGenerator = KernelGenerator
    ME.RegionType = enum.Enum('RegionType', 'HYPER_CUBE HYPER_CROSS CUSTOM')
    ME.MinkowskiAlgorithm = enum.Enum('MinkowskiAlgorithm', 'DEFAULT MEMORY_EFFICIENT SPEED_OPTIMIZED')
    ME.CoordinateMapType = enum.Enum('CoordinateMapType', 'CPU CUDA')
    ME.SparseTensorOperationMode = enum.Enum('SparseTensorOperationMode', 'SEPARATE_COORDINATE_MANAGER SHARE_COORDINATE_MANAGER')
    ME.SparseTensorQuantizationMode = enum.Enum('SparseTensorQuantizationMode',
                                                'RANDOM_SUBSAMPLE UNWEIGHTED_AVERAGE UNWEIGHTED_SUM NO_QUANTIZATION')
    ME.set_sparse_tensor_operation_mode = lambda mode: None
    for name in ('SparseTensor', 'CoordinateManager', 'CoordinateMapKey', 'TensorField'):
        setattr(ME, name, type(name, (), {}))
    mst = types.ModuleType('MinkowskiEngine.MinkowskiSparseTensor')
    mst.SparseTensorQuantizationMode = ME.SparseTensorQuantizationMode
    mst.SparseTensor = ME.SparseTensor
    ME.MinkowskiSparseTensor = mst
    sys.modules['MinkowskiEngine'] = ME
    sys.modules['MinkowskiEngine.MinkowskiSparseTensor'] = mst
    ts = types.ModuleType('torchsparse'); ts_nn = types.ModuleType('torchsparse.nn')
    ts.SparseTensor = type('SparseTensor', (), {}); ts_nn.Conv3d = type('Conv3d', (), {}); ts.nn = ts_nn
    ts_nn.__path__ = []; ts.__path__ = []                                   # importable as packages
    ts_f = types.ModuleType('torchsparse.nn.functional'); ts_nn.functional = ts_f
    ts_u = types.ModuleType('torchsparse.utils'); ts_u.__path__ = []; ts.utils = ts_u
    ts_c = types.ModuleType('torchsparse.utils.tensor_cache'); ts_c.TensorCache = type('TensorCache', (), {}); ts_u.tensor_cache = ts_c
    for name, mod in (('torchsparse', ts), ('torchsparse.nn', ts_nn), ('torchsparse.nn.functional', ts_f),
                      ('torchsparse.utils', ts_u), ('torchsparse.utils.tensor_cache', ts_c)):
        sys.modules[name] = mod
    sys.modules.setdefault('plyfile', types.SimpleNamespace(PlyData=None, PlyElement=None))
    sys.modules.setdefault('open3d', types.ModuleType('open3d'))        # only used by the evaluators' file IO
    return ME


def make_me_semantics():
    """What the reference itself states about the two engines' conventions, read from the reference (not typed in):
      * child table of `minkowski_expand_coord_2x` (lib/minkowski_sparse_conv_layers.py:403-408) -- offsets of the 8 children of
        a voxel in kernel-index order of the MinkowskiEngine path (x fastest);
      * `unfold_kernel` and the `fold2bin` identity kernel of the two lossless codecs (lossl_coord_me/model.py:328-337: ME
        layout [K, C_in, C_out], x fastest; lossl_coord_int/model.py:240-246: [K, C_out, C_in], z fastest);
      * names, order and shapes of the state_dict of lossy_coord_v2's PCC at baseline_r1 and of lossy_coord_lossy_color's PCC
        at its baseline_r1, built on the stub engine above (the module tree is the reference's).
    lib.* / models.* are imported from /root/reference with their JIT-built extensions stubbed."""
    import torch
    import torch.utils.cpp_extension as ce
    import yaml
    _stub_engines()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]                      # earlier generators install partial stand-ins under these names
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}      # oracle/_ref: the reference's own coders
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from lib.minkowski_sparse_conv_layers import minkowski_expand_coord_2x
        from models.convolutional.lossy_coord_v2.model import PCC as PCCv2
        from models.convolutional.lossy_coord_v2.model_config import ModelConfig as CfgV2
        from models.convolutional.lossy_coord_lossy_color.model import PCC as PCCcolor
        from models.convolutional.lossy_coord_lossy_color.model_config import ModelConfig as CfgColor
        from models.convolutional.lossl_coord_me.model import Model as ModelME
        from models.convolutional.lossl_coord_me.model_config import Config as CfgME
        from models.convolutional.lossl_coord_int.model import Model as ModelInt
        from models.convolutional.lossl_coord_int.model_config import Config as CfgInt
    finally:
        ce.load = real
    out = {}
    zero = torch.zeros((1, 4), dtype=torch.int32)
    out['expand_coord_2x'] = {str(s): minkowski_expand_coord_2x(zero, s)[0].tolist() for s in (2, 4, 16)}

    def with_yaml(cfg, path):
        with open(os.path.join(REF, path)) as f:
            for k, v in yaml.safe_load(f)['model'].items():
                assert hasattr(cfg, k), k
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        cfg.check()                             # the reference's own post-merge hook (lib/simple_config.py:17-37): types, then
        return cfg                              # check_local_value, which broadcasts compressed_channels over the levels

    def keys(model):
        return [[k, list(v.shape) if isinstance(v, torch.Tensor) else None] for k, v in model.state_dict().items()]

    me = ModelME(CfgME())
    out['lossl_coord_me'] = {'unfold_kernel': me.unfold_kernel[0].tolist(),
                             'fold2bin_kernel_shape': list(me.fold2bin_conv.kernel.shape),
                             'fold2bin_kernel': me.fold2bin_conv.kernel.detach().reshape(8, 8).tolist(),
                             'bin2oct_kernel': me.bin2oct_kernel.tolist()}
    mi = ModelInt(CfgInt(), torch.device('cpu'))
    out['lossl_coord_int'] = {'unfold_kernel': mi.unfold_kernel[0].tolist(),
                              'fold2bin_kernel_shape': list(mi.fold2bin_kernel.shape),
                              'fold2bin_kernel': mi.fold2bin_kernel.reshape(8, 8).tolist(),
                              'bin2oct_kernel': mi.bin2oct_kernel.tolist(),
                              'state_dict': keys(mi)}
    out['lossy_coord_v2/baseline_r1'] = keys(PCCv2(with_yaml(CfgV2(), 'config/convolutional/lossy_coord_v2/baseline_r1.yaml')))
    out['lossy_coord_lossy_color/baseline_r1'] = keys(PCCcolor(with_yaml(
        CfgColor(), 'config/convolutional/lossy_coord_lossy_color/baseline_r1.yaml')))
    return out


def make_hilbert():
    """keys of hilbert3d_encode_lut (lib/space_filling_curves/src/hilbert3d.cu:28-60; CUDA only, so the 96-entry state table is
    read out of the source text and the kernel's loop evaluated here): every point of a 4x4x4 cube at 2 bits, seeded points at
    1, 5, 10 and 21 bits, two axis orders"""
    src = open(os.path.join(REF, 'lib/space_filling_curves/src/hilbert3d.cu')).read()
    tab = [int(v) for v in re.search(r'kMortonToHilbertTable\[96\] = \{([^}]*)\}', src).group(1).replace('\n', ' ').split(',') if v.strip()]
    assert len(tab) == 96

    def key(x, y, z, bits):
        t = k = 0
        for b in range(bits - 1, -1, -1):
            v = tab[t | ((x >> b) & 1) | (((y >> b) & 1) << 1) | (((z >> b) & 1) << 2)]
            k, t = (k << 3) | (v & 7), v & ~7
        return k
    rng = np.random.default_rng(11)
    out = []
    cube = [[x, y, z] for x in range(4) for y in range(4) for z in range(4)]
    cases = [(2, cube)] + [(bits, rng.integers(0, 1 << bits, (60, 3)).tolist()) for bits in (1, 5, 10, 21)]
    for bits, pts in cases:
        for order, cols in (('xyz', (0, 1, 2)), ('zxy', (2, 0, 1))):
            out.append({'bits': bits, 'axis_order': order, 'cols': list(cols), 'xyz': pts,
                        'keys': [key(p[cols[0]], p[cols[1]], p[cols[2]], bits) for p in pts]})
    return out


def _functional_torchsparse():
    """A torchsparse stand-in that EVALUATES on the CPU, good enough to run the reference's lossy_coord_v3 codec end to end:
    SparseTensor / TensorCache containers, spnn.Conv3d parameters ([K, C_in, C_out] kernel -- 2-D for K = 1 --, [C_out]
    bias) and SF.conv3d as gather -> torch.mm -> index_add_ per kernel offset.  What is RESTATED here (not taken from
    torchsparse, which cannot be installed) is the enumeration of kernel offsets: odd kernels centred with x fastest, even
    kernels anchored at 0 with z fastest -- the enumeration of the reference's own integer engine
    (lib/int_sparse_conv/src/hashmap/hashmap_cuda.cuh:239-258), whose float->int weight import is a plain permute
    (cuda_ops.py:257-260), and the one the reference's fold / unfold kernels imply (me_semantics.json).  Strided outputs come
    from SF.spdownsample, which the reference replaces with its own function.  Everything the model does WITH these
    operators -- what is coded, in which order, with which side information -- is the reference's code."""
    import enum
    import torch
    import torch.nn as nn
    _stub_engines()
    ts, ts_nn, SF = sys.modules['torchsparse'], sys.modules['torchsparse.nn'], sys.modules['torchsparse.nn.functional']

    class TensorCache:
        def __init__(self):
            self.cmaps, self.kmaps, self.hashmaps = {}, {}, {}

    class SparseTensor:
        def __init__(self, feats, coords, stride=1, spatial_range=None):
            self.F, self.C = feats, coords
            self.stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
            self.spatial_range = spatial_range
            self._caches = TensorCache()
        feats = property(lambda self: self.F)
        coords = property(lambda self: self.C)

    as3 = lambda v: (v,) * 3 if isinstance(v, int) else tuple(int(i) for i in v)

    def key(c):
        c = c.long()
        return (c[:, 0] << 60) | (c[:, 1] << 40) | (c[:, 2] << 20) | c[:, 3]

    def kernel_table(in_c, out_c, ks, st):
        keys = key(in_c)
        order = torch.argsort(keys)
        keys = keys[order]
        volume = ks[0] * ks[1] * ks[2]
        table = torch.full((volume, out_c.shape[0]), -1, dtype=torch.long)
        axes = (0, 1, 2) if volume % 2 else (2, 1, 0)
        for k in range(volume):
            rem, q = k, out_c.long().clone()
            for a in axes:
                q[:, 1 + a] = q[:, 1 + a] * st[a] + rem % ks[a] - (ks[a] - 1) // 2
                rem //= ks[a]
            ok = (q[:, 1:] >= 0).all(1)
            kq = key(torch.where(ok[:, None], q, torch.zeros_like(q)))
            pos = torch.searchsorted(keys, kq).clamp(max=len(keys) - 1)
            hit = ok & (keys[pos] == kq)
            table[k, hit] = order[pos[hit]]
        return table

    def conv3d(input, weight, kernel_size, bias=None, stride=1, padding=0, dilation=1, transposed=False, generative=False,
               config=None, training=False):
        assert not transposed and not generative and as3(dilation) == (1, 1, 1)
        ks, st = as3(kernel_size), as3(stride)
        caches = input._caches
        if st == (1, 1, 1):
            out_c, out_stride = input.C, input.stride
        else:
            out_stride = tuple(a * b for a, b in zip(input.stride, st))
            if out_stride in caches.cmaps:
                out_c = caches.cmaps[out_stride][0]
            else:
                out_c = SF.spdownsample(input.C, st, ks, torch.zeros(3, dtype=torch.int32), input.spatial_range)
        tag = (input.stride, ks, st)
        if tag not in caches.kmaps:
            caches.kmaps[tag] = kernel_table(input.C, out_c, ks, st)
        table = caches.kmaps[tag]
        w = weight.reshape(table.shape[0], -1, weight.shape[-1])
        out = torch.zeros((out_c.shape[0], w.shape[-1]), dtype=input.F.dtype)
        for k in range(table.shape[0]):
            rows = torch.nonzero(table[k] >= 0)[:, 0]
            if len(rows):
                out.index_add_(0, rows, torch.mm(input.F.index_select(0, table[k, rows]), w[k]))
        if bias is not None:
            out += bias
        caches.cmaps.setdefault(input.stride, (input.C, input.spatial_range))
        caches.cmaps.setdefault(out_stride, (out_c, None))
        ret = SparseTensor(out, out_c, out_stride, None)
        ret._caches = caches
        return ret

    class Conv3d(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, bias=False,
                     transposed=False, generative=False, config=None):
            super().__init__()
            self.in_channels, self.out_channels = in_channels, out_channels
            self.kernel_size, self.stride, self.padding, self.dilation = as3(kernel_size), as3(stride), as3(padding), dilation
            volume = int(np.prod(self.kernel_size))
            self.kernel = nn.Parameter(torch.zeros((volume, in_channels, out_channels) if volume > 1 else (in_channels, out_channels)))
            self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

        def forward(self, input):
            return conv3d(input, self.kernel, self.kernel_size, self.bias, self.stride, self.padding, self.dilation)

    class _Cfg(dict):
        __getattr__ = dict.get
        __setattr__ = dict.__setitem__

    cc = types.SimpleNamespace(Dataflow=enum.Enum('Dataflow', 'ImplicitGEMM GatherScatter FetchOnDemand CodedCSR'),
                               get_default_conv_config=lambda conv_mode=None: _Cfg(), set_global_conv_config=lambda c: None)
    SF.conv_config, SF.get_conv_mode, SF.conv3d = cc, (lambda: None), conv3d
    SF.spdownsample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('the model installs its own spdownsample'))
    ts.SparseTensor, ts_nn.Conv3d = SparseTensor, Conv3d
    sys.modules['torchsparse.utils.tensor_cache'].TensorCache = TensorCache
    return ts


def make_codec_v3():
    """The reference's lossy_coord_v3 model EXECUTED on the CPU over the functional torchsparse stand-in above, with the
    reference's own rANS coder (oracle/_ref): state_dict layout, side-information tables, batch_quantize_pmf_torch vectors,
    and whole compress / decompress runs on small seeded clouds with seeded weights (streams, headers, rounded latents,
    symbols, reconstructions)."""
    import torch
    import torch.utils.cpp_extension as ce
    import yaml
    _functional_torchsparse()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_v3 import model as ref_model
        from models.convolutional.lossy_coord_v3.model_config import Config
    finally:
        ce.load = real
    from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
    from fastpcc_amd.synthetic import batched, surface_cloud
    torch.cuda.synchronize = lambda *a, **k: None                    # the reference synchronises unconditionally; CPU run

    def cfg_of(path=None, **kw):
        cfg = Config()
        if path:
            with open(os.path.join(REF, path)) as f:
                text = f.read()
            inc = re.search(r'#\s*include\s+"([^"]+)"', text)
            if inc:
                with open(os.path.join(REF, inc.group(1))) as f:
                    for k, v in yaml.safe_load(f)['model'].items():
                        setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
            for k, v in yaml.safe_load(text)['model'].items():
                assert hasattr(cfg, k), k
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        for k, v in kw.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        return cfg

    out = {'state_dict': {}}
    for name in ('dense_r1', 'dense_r4', 'dense_r7'):
        m = ref_model.Model(cfg_of(f'config/convolutional/lossy_coord_v3/{name}.yaml'))
        out['state_dict'][name] = {'num_latents': list(m.cfg.num_latents), 'lossl_geo_upsample': list(m.cfg.lossl_geo_upsample),
                                   'max_stride': m.cfg.max_stride, 'channels': m.cfg.channels,
                                   'keys': [[k, list(v.shape)] for k, v in m.state_dict().items()]}
    m.eval()
    out['side_info'] = {'cdf1_head': m.fea_side_info_cdf1[0, :4].tolist(), 'cdf1_tail': m.fea_side_info_cdf1[0, -3:].tolist(),
                        'cdf1_len': int(m.fea_side_info_cdf1.shape[1]), 'cdf2': m.fea_side_info_cdf2[0].tolist(),
                        'bin2oct_kernel': m.bin2oct_kernel.tolist(), 'unfold_kernel': m.unfold_kernel[0].tolist()}
    g = torch.Generator().manual_seed(11)
    hist = torch.tensor([5, 0, 0, 17, 1, 1, 250, 3], dtype=torch.float32)
    logits = torch.randn((3, 255), generator=g) * 3
    out['quantize_pmf'] = {'hist': hist.tolist(), 'hist_cdf': ref_model.Model.batch_quantize_pmf_torch((hist / hist.sum())[None], False)[0].tolist(),
                           'logits': logits.tolist(), 'logits_cdf': ref_model.Model.batch_quantize_pmf_torch(logits.clone()).tolist()}
    x = torch.tensor([-25.0, -20.0, -3.5, 0.0, 19.9, 20.0, 31.0], requires_grad=True)
    y = ref_model.BoundFunction.apply(x, torch.tensor(20.0))
    y.backward(torch.tensor([0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5]))
    out['bound'] = {'x': x.detach().tolist(), 'y': y.detach().tolist(), 'grad': x.grad.tolist()}

    probe_a = torch.randn((257, 40), generator=g)
    probe_b = torch.randn((40, 24), generator=g)
    out['float_probe'] = {'seed': 11, 'mm_sha256': hashlib.sha256(torch.mm(probe_a, probe_b).numpy().tobytes()).hexdigest(),
                          'softmax_sha256': hashlib.sha256(torch.softmax(logits, -1).numpy().tobytes()).hexdigest()}

    runs = []
    for label, kw, seed, res, pts in (
            ('r1_like', dict(channels=8, max_stride=32, num_latents=(0, 0, 2, 1), lossl_geo_upsample=(0, 1, 1, 1)), 1, 32, 700),
            ('r4_like', dict(channels=8, max_stride=64, num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 0, 1, 1, 1)), 2, 64, 1500),
            ('lossless', dict(channels=16, max_stride=32, num_latents=(0, 1, 2), lossl_geo_upsample=(1, 1, 1)), 3, 32, 600)):
        cfg = cfg_of(**kw)
        model = ref_model.Model(cfg)
        randomize_(model, seed)
        model.eval()
        xyz = surface_cloud(seed + 20, res, pts) + np.array([3, 0, 6], dtype=np.int32)
        perm = np.random.default_rng(seed).permutation(len(xyz))
        grabbed = {}
        enc_oct, enc_fea = model.rans_encode_oct, model.rans_encode_fea
        model.rans_encode_oct = lambda cdfs, vals: (grabbed.setdefault('symbols', []).append(vals.numpy().astype(int).tolist()),
                                                    grabbed.setdefault('cdf_sha', []).append(hashlib.sha256(cdfs.numpy().tobytes()).hexdigest()),
                                                    enc_oct(cdfs, vals))[-1]
        model.rans_encode_fea = lambda cdf, vals, lo=None: (grabbed.setdefault('fea', []).append(
            {'cdf': cdf.numpy().astype(int).tolist(), 'values': vals.numpy().astype(int).tolist(),
             'lo': None if lo is None else int(lo.item())}), enc_fea(cdf, vals, lo))[-1]
        with torch.no_grad():
            data = model.compress(torch.from_numpy(batched(xyz)[perm]))
            rec = model.decompress(data)
        runs.append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                     'seed': seed, 'xyz': xyz[perm].tolist(),
                     'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
                     'stream_hex': data.hex(), 'oct_symbols_in_coding_order': grabbed.get('symbols', []),
                     'oct_cdf_sha256_in_coding_order': grabbed.get('cdf_sha', []),
                     'fea_in_coding_order': grabbed['fea'], 'recon': rec.tolist()})
        print('codec_v3', label, len(xyz), 'points ->', len(data), 'bytes,', len(rec), 'decoded')
    out['runs'] = runs

    # training objective (train_forward, :411-455) with the uniform noise of the latents replaced by zeros on both sides
    # (the reference draws it from the CPU generator, the product from the device's): loss terms per level
    real_uniform = torch.Tensor.uniform_
    torch.Tensor.uniform_ = lambda self, *a, **k: self.zero_()
    train = []
    try:
        for label, kw, seed, step in (
                ('r1_like', dict(channels=8, max_stride=32, num_latents=(0, 0, 2, 1), lossl_geo_upsample=(0, 1, 1, 1),
                                 coord_recon_loss_factor=2.0, warmup_steps=0), 1, 10),
                ('r4_like_warmup', dict(channels=8, max_stride=64, num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 0, 1, 1, 1),
                                        coord_recon_loss_factor=0.4, warmup_steps=100), 2, 10),
                ('r7_like', dict(channels=8, max_stride=64, num_latents=(0, 0, 0, 2, 2), lossl_geo_upsample=(0, 0, 0, 1, 1),
                                 coord_recon_loss_factor=0.1, warmup_steps=0), 3, 5)):
            cfg = cfg_of(**kw)
            model = ref_model.Model(cfg)
            randomize_(model, seed)
            model.train()
            clouds = [surface_cloud(seed + 60, 32 if cfg.max_stride == 32 else 64, 900), surface_cloud(seed + 70, 32 if cfg.max_stride == 32 else 64, 600)]
            parts = []
            for b, xyz in enumerate(clouds):
                xyz = xyz - xyz.min(0)
                key = sum(((xyz[:, a].astype(np.int64) >> i) & 1) << (3 * i + (2 - a)) for i in range(8) for a in range(3))
                xyz = xyz[np.argsort(key, kind='stable')]                 # Morton order, z on the lowest bit (x most significant)
                parts.append(np.concatenate((np.full((len(xyz), 1), b), xyz), 1))
            batch = torch.from_numpy(np.concatenate(parts, 0).astype(np.int32))
            res = model.train_forward(batch, [len(p) for p in parts], step)
            train.append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                          'seed': seed, 'training_step': step, 'xyz': batch.tolist(), 'points_num': [len(p) for p in parts],
                          'loss': float(res['loss']), 'terms': {k: float(v) for k, v in res.items() if k != 'loss'}})
            print('codec_v3 train', label, float(res['loss']))
    finally:
        torch.Tensor.uniform_ = real_uniform
    out['train'] = train
    return out


def _functional_int_ext():
    """Stand-in for the reference's CUDA extension `int_sparse_conv_ext` (CUTLASS int8 GEMMs, hash table, fixed-point
    element-wise kernels, LUT softmax), evaluated on the CPU: integer GEMMs are torch integer matmuls, the hash lookup is a
    sorted search with the extension's own offset enumeration (hashmap_cuda.cuh:239-258) and the element-wise operators are
    oracle/int_ops.c -- the restatement of the scalar device functions of src/element_wise/*.cu and softmax.cu that the
    oracle itself uses.  So a run over this stand-in does not check that arithmetic again; what it contributes is everything
    ABOVE it, executed from the reference: lib/int_sparse_conv/cuda_ops.py (which operator gets which multiplier, shift,
    zero point and bias, the kernel-map construction, the residual block) and models/convolutional/lossl_coord_int/model.py
    (traversal, multi-step prediction, caches, CDF construction, coding order, header)."""
    import torch
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import codec_int as oi
    ext = types.ModuleType('int_sparse_conv_ext')
    tables = {}

    def ep(x, bias, slope, mul, zp, shift, bits):
        out = oi.epilogue(x.numpy(), None if bias is None else bias.numpy(), None if slope is None else slope.numpy(),
                          oi._np(mul), int(zp.item()), int(shift), bits)
        return torch.from_numpy(out).to({8: torch.int8, 16: torch.int16, 32: torch.int32}[bits])

    for bits in (8, 16, 32):
        setattr(ext, f'requant_to_int{bits}', lambda x, mul, zp, sh, b=bits: ep(x, None, None, mul, zp, sh, b))
        setattr(ext, f'bias_requant_to_int{bits}', lambda x, bias, mul, zp, sh, b=bits: ep(x, bias, None, mul, zp, sh, b))
        setattr(ext, f'prelu_requant_to_int{bits}', lambda x, slope, mul, zp, sh, b=bits: ep(x, None, slope, mul, zp, sh, b))
        setattr(ext, f'bias_prelu_requant_to_int{bits}', lambda x, bias, slope, mul, zp, sh, b=bits: ep(x, bias, slope, mul, zp, sh, b))
    ext.prelu = lambda x, slope: torch.from_numpy(oi.prelu_i32(x.numpy(), int(slope.item())))
    ext.softmax_int32 = lambda x: torch.from_numpy(oi.softmax_i32(x.numpy()))

    def gemm(a, b, c, d):
        d.copy_(a.to(torch.int32) @ b.to(torch.int32).t() + c)

    def gather_gemm_scatter(a, b, c, d, in_map, out_map):
        rows = out_map.long()
        d[rows] = a[in_map.long()].to(torch.int32) @ b.to(torch.int32).t() + c[rows]

    ext.cutlass_gemm_int8, ext.cutlass_gather_gemm_scatter_int8 = gemm, gather_gemm_scatter

    class GPUHashTable:
        def __init__(self, keys, vals):
            self.tag = keys.data_ptr()

        def insert_coords(self, xyzb):
            tables[self.tag] = xyzb[:, [3, 0, 1, 2]].numpy().astype(np.int64)

        def lookup_coords(self, xyzb, ks, st, volume):
            t = oi.kernel_table(tables[self.tag], xyzb[:, [3, 0, 1, 2]].numpy().astype(np.int64), tuple(ks.tolist()), tuple(st.tolist()))
            assert t.shape[0] == volume
            return torch.from_numpy((t + 1).T.astype(np.int32).copy())

    ext.GPUHashTable = GPUHashTable
    return ext


def make_codec_int():
    """The reference's integer LiDAR codec (models/convolutional/lossl_coord_int + lib/int_sparse_conv/cuda_ops.py) EXECUTED on
    the CPU over the stand-in extension above and the reference's own rANS coder: whole compress / decompress runs with
    seeded parameters on small sweeps.  Integer arithmetic throughout, so the streams are reproducible on any machine."""
    import torch
    import torch.utils.cpp_extension as ce
    _functional_torchsparse()
    ext = _functional_int_ext()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from lib.int_sparse_conv import cuda_ops
        from models.convolutional.lossl_coord_int import model as ref_model
        from models.convolutional.lossl_coord_int.model_config import Config
    finally:
        ce.load = real
    cuda_ops.int_sparse_conv_ext = ext
    torch.cuda.synchronize = lambda *a, **k: None
    from fastpcc_amd.codecs.lossl_coord_int import Config as MyConfig, Model as MyModel
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    from fastpcc_amd.synthetic import batched, lidar_cloud
    runs = []
    for label, kw, seed, beams, az in (('c16', dict(channels=16), 1, 8, 128),
                                       ('c16_skip1', dict(channels=16, skip_top_scales_num=1), 2, 6, 160),
                                       ('c16_more_ch', dict(channels=16, use_more_ch_for_multi_step_pred=True), 3, 8, 96),
                                       ('c32_fea8', dict(channels=32, fea_stride=8, max_stride_wo_recurrent=512, max_stride=4096), 4, 8, 128)):
        mine = MyModel(MyConfig(**kw), 'cpu')
        randomize_(mine, seed)
        cfg = Config()
        for k, v in kw.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        model = ref_model.Model(cfg, torch.device('cpu'))
        missing = model.load_state_dict(mine.state_dict(), strict=True)
        model.eval()
        xyz = lidar_cloud(seed + 30, beams=beams, azimuths=az) + np.array([5, 0, 9], dtype=np.int32)
        perm = np.random.default_rng(seed).permutation(len(xyz))
        with torch.no_grad():
            data = model.compress(torch.from_numpy(batched(xyz)[perm]))
            rec = model.decompress(data)
        assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist())), 'the reference run is not lossless'
        runs.append({'label': label, 'config': kw, 'seed': seed, 'xyz': xyz[perm].tolist(), 'stream_hex': data.hex(),
                     'recon_sha256': hashlib.sha256(np.ascontiguousarray(rec.numpy().astype(np.int32)).tobytes()).hexdigest(),
                     'state_dict_keys': [[k, list(v.shape)] for k, v in model.state_dict().items()]})
        print('codec_int', label, len(xyz), 'points ->', len(data), 'bytes')
    return {'runs': runs}


def _functional_minkowski(order_fn=None):
    """order_fn (round 4): None = every sum by torch (conv_mm / nn.Linear); else oracle.orders.summation_order -- the stand-in
    then evaluates every convolution and linear layer as the fixed-order FMA chain the HIP kernels document
    (oracle/sparse_conv.py:conv_chain), so that a run of the REFERENCE's model code over it writes the bytes the GPU path must
    write exactly (tests/test_gpu_codec_v2.py::test_bytes_equal_the_reference_run_in_chain_order).

    A MinkowskiEngine stand-in that EVALUATES on the CPU: the API surface the reference's lossy_coord_v2 test path touches
    (SparseTensor, CoordinateManager / CoordinateMapKey, the three convolutions, linear, activations, pruning, max pooling and
    its transpose, cat) on top of oracle/coords.py (coordinate maps in Morton order, kernel offsets x fastest, strided /
    generated / transposed maps) and oracle/sparse_conv.py:conv_mm (gather -> torch.mm -> index_add_ per kernel offset).
    Those two files ARE the restatement of MinkowskiEngine's conventions and stay unpinned -- the
    engine cannot be installed.  What a run over this stand-in pins is everything above the engine, executed from the
    reference: lib/minkowski_sparse_conv_layers.py, lossy_coord_v2/{layers,model}.py, lossy_coord_lossy_color/geo_lossl_em.py
    and the rANS coders (which level is predicted from which, what is coded in which order with which side information,
    the framing, the adaptive pruning rule).  New coordinate maps get the empty string id (a '#n' suffix when taken),
    pruned maps 'pruned': the reference's own lookups (layers.py:155-157, geo_lossl_em.py:272) rely on exactly that."""
    import enum
    import torch
    import torch.nn as nn
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import coords as oc
    from oracle import sparse_conv as sc
    ME = types.ModuleType('MinkowskiEngine')
    state = {'cm': None}
    as3 = lambda v: (int(v),) * 3 if isinstance(v, int) else tuple(int(i) for i in v)

    class CoordinateMapKey:
        def __init__(self, tensor_stride, string_id=''):
            self._stride, self._id = as3(tensor_stride), string_id

        def get_tensor_stride(self):
            return list(self._stride)

        def get_key(self):
            return list(self._stride), self._id

        def __eq__(self, other):
            return isinstance(other, CoordinateMapKey) and (self._stride, self._id) == (other._stride, other._id)

        def __hash__(self):
            return hash((self._stride, self._id))

        def __repr__(self):
            return f'CoordinateMapKey({list(self._stride)}, {self._id!r})'

    class CoordinateManager:
        def __init__(self, D=3, coordinate_map_type=None, minkowski_algorithm=None, **_):
            self.levels, self._manager, self._cache = {}, self, {}

        def register(self, level, string_id=''):
            key, n = CoordinateMapKey((level.stride,) * 3, string_id), 0
            while key in self.levels and self.levels[key] is not level:
                n += 1
                key = CoordinateMapKey((level.stride,) * 3, f'{string_id}#{n}')
            self.levels[key] = level
            return key

        def derived(self, key, what):
            """the strided / generated map of `key`, made once"""
            if (key, what) not in self._cache:
                fn = {'strided': oc.strided, 'generated': oc.generated}[what]
                self._cache[(key, what)] = self.register(fn(self.levels[key]), '')
            return self._cache[(key, what)]

        def kmap(self, kind, src_key, dst_key):
            tag = (kind, src_key, dst_key)
            if tag not in self._cache:
                src, dst = self.levels[src_key], self.levels[dst_key]
                self._cache[tag] = oc.transposed_map(src, dst) if kind == 'T' else oc.kernel_map(src, dst, kind)
            return self._cache[tag]

        def get_coordinate_map_keys(self, tensor_stride):
            return [k for k in self.levels if k._stride == as3(tensor_stride)]

        def insert_and_map(self, coordinates, tensor_stride=1, string_id=''):
            level = oc.Level(coordinates.cpu().numpy(), as3(tensor_stride)[0])
            return self.register(level, string_id), (torch.from_numpy(level.order.copy()), None)

        def kernel_map(self, in_key, out_key, stride=1, kernel_size=1, **_):
            assert kernel_size == 1
            src, dst = self.levels[in_key], self.levels[out_key]
            rows = dst.rows_of(src.coords)
            hit = np.nonzero(rows >= 0)[0]
            return {0: torch.from_numpy(np.stack((hit, rows[hit]))).to(torch.int32)} if len(hit) else {}

        def stride(self, key, stride):
            for _ in range(as3(stride)[0].bit_length() - 1):
                key = self.derived(key, 'strided')
            return key

    class SparseTensor:
        def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_map_key=None, coordinate_manager=None,
                     quantization_mode=None, **_):
            cm = coordinate_manager if coordinate_manager is not None else state['cm']
            if coordinates is not None:
                coordinate_map_key, (rows, _) = cm.insert_and_map(coordinates, tensor_stride, '')
                features = features[rows]
            self.F, self.coordinate_map_key, self.coordinate_manager = features, coordinate_map_key, cm

        level = property(lambda self: self.coordinate_manager.levels[self.coordinate_map_key])
        C = property(lambda self: torch.from_numpy(self.level.coords).to(torch.int32))
        tensor_stride = property(lambda self: self.coordinate_map_key.get_tensor_stride())
        shape = property(lambda self: self.F.shape)
        device = property(lambda self: self.F.device)
        dtype = property(lambda self: self.F.dtype)

        @property
        def decomposition_permutations(self):
            b = self.level.coords[:, 0]
            return [torch.from_numpy(np.nonzero(b == i)[0]) for i in range(int(b.max()) + 1 if len(b) else 0)]

        @property
        def decomposed_coordinates(self):
            return [self.C[p][:, 1:] for p in self.decomposition_permutations]

    def like(x, f, key=None):
        return SparseTensor(f, coordinate_map_key=key if key is not None else x.coordinate_map_key,
                            coordinate_manager=x.coordinate_manager)

    class KernelGenerator:
        def __init__(self, kernel_size=-1, stride=1, dilation=1, region_type=None, dimension=3, **_):
            self.kernel_size, self.kernel_stride, self.kernel_dilation = list(as3(kernel_size)), list(as3(stride)), list(as3(dilation))
            self.kernel_volume, self.region_type, self.dimension = int(np.prod(self.kernel_size)), region_type, dimension

    class _Conv(nn.Module):
        MODE = 'conv'

        def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                     kernel_generator=None, expand_coordinates=False, dimension=3, **_):
            super().__init__()
            kg = kernel_generator or KernelGenerator(kernel_size, stride, dilation, dimension=dimension)
            assert len(set(kg.kernel_size)) == 1 and len(set(kg.kernel_stride)) == 1 and set(kg.kernel_dilation) == {1}
            self.kernel_generator, self.in_channels, self.out_channels = kg, in_channels, out_channels
            shape = (in_channels, out_channels) if kg.kernel_volume == 1 else (kg.kernel_volume, in_channels, out_channels)
            self.kernel = nn.Parameter(torch.zeros(shape))
            self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

        def forward(self, x, coordinates=None):
            cm, key = x.coordinate_manager, x.coordinate_map_key
            ks, st = self.kernel_generator.kernel_size[0], self.kernel_generator.kernel_stride[0]
            n = x.F.shape[0]
            if self.MODE == 'conv' and (ks, st) == (1, 1):
                dst_key, kmap = key, [(np.arange(n), np.arange(n))]
            elif self.MODE == 'conv' and st == 1:
                dst_key, kmap = key, cm.kmap(ks, key, key)
            elif self.MODE == 'conv' and (ks, st) == (2, 2):
                dst_key = cm.derived(key, 'strided')
                kmap = cm.kmap(2, key, dst_key)
            elif self.MODE == 'transpose' and (ks, st) == (2, 2):
                dst_key = coordinates
                kmap = cm.kmap('T', key, dst_key)
            elif self.MODE == 'generative' and (ks, st) == (2, 2):
                dst_key = cm.derived(key, 'generated')
                kmap = cm.kmap('T', key, dst_key)
            else:
                raise NotImplementedError((self.MODE, ks, st))
            w = self.kernel.reshape(len(kmap), self.in_channels, self.out_channels)
            n_out = cm.levels[dst_key].n
            if order_fn is None:
                out = sc.conv_mm(x.F, kmap, w, None if self.bias is None else self.bias.reshape(-1), n_out)
            else:
                kind = {('conv', 1): 'k1', ('conv', 3): 'k3', ('conv', 2): 'k2s2', ('transpose', 2): 'k2s2T', ('generative', 2): 'gen'}[(self.MODE, ks)]
                out = _chain(x, kind, oc.dense_table(kmap, n_out), w.detach(), self.bias, n_out)
            return like(x, out, dst_key)

    ME.MinkowskiConvolution = type('MinkowskiConvolution', (_Conv,), {})
    ME.MinkowskiConvolutionTranspose = type('MinkowskiConvolutionTranspose', (_Conv,), {'MODE': 'transpose'})
    ME.MinkowskiGenerativeConvolutionTranspose = type('MinkowskiGenerativeConvolutionTranspose', (_Conv,), {'MODE': 'generative'})

    def wrap(name, attr, cls):
        def init(self, *a, **k):
            nn.Module.__init__(self)
            k.pop('inplace', None) if cls in (nn.PReLU, nn.Sigmoid) else None
            setattr(self, attr, cls(*a, **k))
        return type(name, (nn.Module,), {'__init__': init, 'forward': lambda self, x: like(x, getattr(self, attr)(x.F))})

    def _chain(x, kind, table, w, bias, n_out):
        """the layer as the documented FMA chain; a tensor that is the concatenation of two (ME.cat) enters as two sources"""
        f = x.F.detach()
        split = getattr(x, '_split', None)
        c1, c2 = (split if split is not None and len(split) == 2 else (f.shape[1], 0))
        x1, x2 = f[:, :c1].contiguous().numpy(), (f[:, c1:].contiguous().numpy() if c2 else None)
        out = sc.conv_chain(x1, table, w.numpy(), None if bias is None else bias.detach().reshape(-1).numpy(), n_out, x2=x2,
                            order=order_fn(kind, c1, c2, w.shape[-1], n_out))
        return torch.from_numpy(out)

    class _ChainLinear(nn.Module):
        def __init__(self, in_features, out_features, bias=True):
            super().__init__()
            self.linear = nn.Linear(in_features, out_features, bias=bias)

        def forward(self, x):
            n = x.F.shape[0]
            w = self.linear.weight.detach().t().contiguous()
            table = np.arange(n, dtype=np.int32)[None]
            return like(x, _chain(x, 'mlp', table, w.reshape(1, *w.shape), self.linear.bias, n))

    ME.MinkowskiLinear = wrap('MinkowskiLinear', 'linear', nn.Linear) if order_fn is None else _ChainLinear
    ME.MinkowskiBatchNorm = wrap('MinkowskiBatchNorm', 'bn', nn.BatchNorm1d)
    for name, cls in (('MinkowskiReLU', nn.ReLU), ('MinkowskiPReLU', nn.PReLU), ('MinkowskiLeakyReLU', nn.LeakyReLU),
                      ('MinkowskiSigmoid', nn.Sigmoid)):
        setattr(ME, name, wrap(name, 'module', cls))

    class MinkowskiPruning(nn.Module):
        def forward(self, x, mask):
            src = x.level
            keep = mask.cpu().numpy().astype(bool)
            key = x.coordinate_manager.register(oc.Level(src.coords[keep], src.stride), 'pruned')
            return like(x, x.F[mask], key)

    class _Pool(nn.Module):
        def __init__(self, kernel_size, stride=1, dilation=1, kernel_generator=None, dimension=3, **_):
            super().__init__()
            assert as3(kernel_size) == as3(stride)

    class MinkowskiMaxPooling(_Pool):
        def forward(self, x, coordinates=None):
            src, dst = x.level, x.coordinate_manager.levels[coordinates]
            q = src.coords.copy()
            q[:, 1:] = q[:, 1:] // dst.stride * dst.stride
            rows = torch.from_numpy(dst.rows_of(q))
            out = torch.full((dst.n, x.F.shape[1]), float('-inf'), dtype=x.F.dtype)
            out.scatter_reduce_(0, rows[:, None].expand(-1, x.F.shape[1]), x.F, reduce='amax', include_self=True)
            return like(x, out, coordinates)

    class MinkowskiPoolingTranspose(_Pool):
        def forward(self, x, coordinates):
            src, dst = x.level, x.coordinate_manager.levels[coordinates]
            q = dst.coords.copy()
            q[:, 1:] = q[:, 1:] // src.stride * src.stride
            return like(x, x.F[torch.from_numpy(src.rows_of(q))], coordinates)

    def cat(*tensors):
        if len(tensors) == 1 and isinstance(tensors[0], (tuple, list)):
            tensors = tuple(tensors[0])
        assert all(t.coordinate_map_key == tensors[0].coordinate_map_key for t in tensors)
        out = like(tensors[0], torch.cat([t.F for t in tensors], 1))
        out._split = [t.F.shape[1] for t in tensors]       # the layer that consumes it reads two sources (chain mode)
        return out

    ME.MinkowskiPruning, ME.MinkowskiMaxPooling, ME.MinkowskiPoolingTranspose, ME.cat = MinkowskiPruning, MinkowskiMaxPooling, MinkowskiPoolingTranspose, cat
    ME.SparseTensor, ME.CoordinateManager, ME.CoordinateMapKey, ME.KernelGenerator = SparseTensor, CoordinateManager, CoordinateMapKey, KernelGenerator
    ME.RegionType = enum.Enum('RegionType', 'HYPER_CUBE HYPER_CROSS CUSTOM')
    ME.MinkowskiAlgorithm = enum.Enum('MinkowskiAlgorithm', 'DEFAULT MEMORY_EFFICIENT SPEED_OPTIMIZED')
    ME.CoordinateMapType = enum.Enum('CoordinateMapType', 'CPU CUDA')
    ME.SparseTensorOperationMode = enum.Enum('SparseTensorOperationMode', 'SEPARATE_COORDINATE_MANAGER SHARE_COORDINATE_MANAGER')
    ME.SparseTensorQuantizationMode = enum.Enum('SparseTensorQuantizationMode',
                                                'RANDOM_SUBSAMPLE UNWEIGHTED_AVERAGE UNWEIGHTED_SUM NO_QUANTIZATION')
    ME.set_sparse_tensor_operation_mode = lambda mode: None
    ME.set_global_coordinate_manager = lambda cm: state.__setitem__('cm', cm)
    ME.clear_global_coordinate_manager = lambda: state.__setitem__('cm', None)
    mst = types.ModuleType('MinkowskiEngine.MinkowskiSparseTensor')
    mst.SparseTensorQuantizationMode, mst.SparseTensor = ME.SparseTensorQuantizationMode, SparseTensor
    ME.MinkowskiSparseTensor = mst
    sys.modules['MinkowskiEngine'], sys.modules['MinkowskiEngine.MinkowskiSparseTensor'] = ME, mst
    return ME


def make_codec_v2():
    """The reference's lossy_coord_v2 codec -- the headline path -- EXECUTED on the CPU over the MinkowskiEngine stand-in
    above with the reference's own rANS coders (oracle/_ref): whole compress / decompress runs of seeded models on small
    seeded clouds: streams, reconstructions, per-level point counts."""
    import torch
    import torch.utils.cpp_extension as ce
    _stub_engines()                                  # torchsparse / plyfile / open3d names for the imports below
    _functional_minkowski()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_v2.model import PCC
        from models.convolutional.lossy_coord_v2.model_config import ModelConfig
    finally:
        ce.load = real
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    g = torch.Generator().manual_seed(13)
    probe = torch.randn((301, 48), generator=g), torch.randn((48, 32), generator=g), torch.randn((32,), generator=g)
    out = {'float_probe': {'seed': 13, 'mm_sha256': hashlib.sha256(torch.mm(probe[0], probe[1]).numpy().tobytes()).hexdigest(),
                           'linear_sha256': hashlib.sha256(torch.nn.functional.linear(probe[0], probe[1].t().contiguous(), probe[2]).numpy().tobytes()).hexdigest(),
                           'sigmoid_sha256': hashlib.sha256(torch.sigmoid(probe[0]).numpy().tobytes()).hexdigest()}}
    runs = []
    base = dict(activation='prelu', compressed_channels=(1,), skip_encoding_fea=1, adaptive_pruning=True)
    for label, kw, seed, res, pts in (
            ('r1_like', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1, 0, 1),
                             geo_lossl_channels=(16, 32, 32, 32, 32, 32, 1)), 1, 64, 2500),
            ('r3_like_two_stages', dict(encoder_channels=(8, 16, 16), decoder_channels=(16, 8), geo_lossl_if_sample=(0, 1, 0, 1),
                                        geo_lossl_channels=(16, 32, 32, 32, 1)), 2, 64, 3000),
            ('r5_like_three_stages', dict(encoder_channels=(8, 16, 16, 16), decoder_channels=(16, 16, 8), geo_lossl_if_sample=(0, 1, 0, 1),
                                          geo_lossl_channels=(16, 16, 16, 16, 1)), 3, 128, 6000),
            ('fixed_threshold', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1),
                                     geo_lossl_channels=(16, 16, 16, 16, 1), adaptive_pruning=False), 4, 32, 900),
            ('all_levels_coded', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(1, 1, 1),
                                      geo_lossl_channels=(16, 32, 32, 1), skip_encoding_fea=-1), 5, 32, 900)):
        cfg = ModelConfig()
        for k, v in {**base, **kw}.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        cfg.check()
        torch.manual_seed(0)
        model = PCC(cfg)
        enliven(model, seed)
        model.eval()
        xyz = surface_cloud(seed + 40, res, pts) + np.array([2, 0, 5], dtype=np.int32)
        perm = np.random.default_rng(seed).permutation(len(xyz))
        with torch.no_grad():
            data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32))
            rec = model.decompress(data)
        runs.append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in {**base, **kw}.items()},
                     'seed': seed, 'xyz': xyz[perm].tolist(),
                     'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
                     'stream_hex': data.hex(), 'recon': rec.tolist()})
        print('codec_v2', label, len(xyz), 'points ->', len(data), 'bytes,', len(rec), 'decoded')
    out['runs'] = runs

    # the headline configuration at its REAL widths (16/64 encoder, 64 + 11 x 128 lossless levels), read from the reference's
    # own YAML through the reference's own config loader (lib/config.py:104-117 Config.merge_with_yaml ->
    # config/convolutional/lossy_coord_v2/baseline_r1.yaml); the pyramid of 1 + 6 stride halvings needs a 256^3 cloud to have more than one bottom voxel
    from lib.config import Config as RefConfig
    ref_cfg = RefConfig()
    ref_cfg.merge_with_yaml(os.path.join(REF, 'config/convolutional/lossy_coord_v2/baseline_r1.yaml'))
    ref_cfg.check()
    mc = ref_cfg.model
    torch.manual_seed(0)
    model = PCC(mc)
    enliven(model, 6)
    model.eval()
    xyz = surface_cloud(46, 256, 4000) + np.array([1, 4, 2], dtype=np.int32)
    perm = np.random.default_rng(6).permutation(len(xyz))
    with torch.no_grad():
        data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32))
        rec = model.decompress(data)
    out['baseline_r1_yaml'] = {
        'label': 'baseline_r1_yaml', 'yaml': 'config/convolutional/lossy_coord_v2/baseline_r1.yaml', 'seed': 6,
        'config': {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in vars(mc).items() if not k.startswith('_')},
        'xyz': xyz[perm].tolist(),
        'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
        'state_dict_shapes': [[k, list(v.shape)] for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)],
        'stream_hex': data.hex(), 'recon': rec.tolist()}
    print('codec_v2 baseline_r1.yaml', len(xyz), 'points ->', len(data), 'bytes,', len(rec), 'decoded')

    # training objective (PCC.train_forward, model.py:144-183: rate of the lossless levels under the noisy deep-factorised
    # bottleneck, occupancy cross-entropies, reconstruction losses of the lossy part, warm-up factors) on batches of clouds,
    # with the bottleneck's uniform noise replaced by zeros on both sides (CPU generator there, device generator here)
    real_uniform = torch.Tensor.uniform_
    torch.Tensor.uniform_ = lambda self, *a, **k: self.zero_()
    train = []
    try:
        for label, kw, seed, step in (
                ('r1_like', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1, 0, 1),
                                 geo_lossl_channels=(16, 32, 32, 32, 32, 32, 1), warmup_fea_loss_steps=5000, warmup_fea_loss_factor=0.01,
                                 bits_loss_factor=0.4), 1, 100),
                ('r1_like_after_warmup', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1),
                                              geo_lossl_channels=(16, 32, 32, 32, 1), warmup_fea_loss_steps=50, warmup_fea_loss_factor=0.01,
                                              bits_loss_factor=0.4), 2, 100),
                ('linear_warmup_no_adaptive_pruning', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1),
                                                           geo_lossl_channels=(16, 32, 32, 32, 1), warmup_fea_loss_steps=200,
                                                           warmup_fea_loss_factor=0.05, linear_warmup=True, adaptive_pruning=False,
                                                           coord_recon_loss_factor=0.7), 3, 60)):
            cfg = ModelConfig()
            for k, v in {**base, **kw}.items():
                assert hasattr(cfg, k), k
                setattr(cfg, k, v)
            cfg.check()
            torch.manual_seed(0)
            model = PCC(cfg)
            enliven(model, seed)
            model.train()
            parts = []
            for b, (s2, n) in enumerate(((seed + 100, 1500), (seed + 110, 1100), (seed + 120, 800))):
                c = surface_cloud(s2, 64, n)
                parts.append(np.concatenate((np.full((len(c), 1), b), c), 1))
            batch = torch.from_numpy(np.concatenate(parts, 0).astype(np.int32))
            res = model.train_forward(batch, step, len(parts))
            train.append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in {**base, **kw}.items()},
                          'seed': seed, 'training_step': step, 'batch_size': len(parts), 'xyz': batch.tolist(),
                          'loss': float(res['loss']), 'terms': {k: float(v) for k, v in res.items() if k != 'loss'}})
            print('codec_v2 train', label, float(res['loss']))
    finally:
        torch.Tensor.uniform_ = real_uniform
    out['train'] = train
    return out


def make_get_keep():
    """The reference's adaptive pruning rule `Decoder.get_keep` (models/convolutional/lossy_coord_v2/layers.py:151-180) EXECUTED
    over the MinkowskiEngine stand-in on the situation the decoder is in when it calls it: logits on the generated children of
    a PRUNED stride-2 map whose stride-4 parents exist too.  The fixture is data: coordinates, logits, requested point counts
    and the keep masks the reference returned (candidates listed with their coordinates, so a consumer matches rows by
    coordinate and does not depend on any row order)."""
    import torch
    import torch.utils.cpp_extension as ce
    _stub_engines()
    ME = _functional_minkowski()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_v2.layers import Decoder
    finally:
        ce.load = real
    from fastpcc_amd.synthetic import surface_cloud
    dec = Decoder(8, (8,), 'HYPER_CUBE', 'prelu').eval()
    cases = []
    for label, seed, batch in (('one_cloud_a', 1, 1), ('one_cloud_b', 2, 1), ('two_clouds', 7, 2)):
        g = torch.Generator().manual_seed(seed)
        clouds = [np.concatenate((np.full((len(x), 1), b), x), 1) for b, x in
                  enumerate(np.unique(surface_cloud(seed + b, 32, 900) // 4 * 4, axis=0) for b in range(batch))]
        top_c = torch.from_numpy(np.concatenate(clouds)).to(torch.int32)
        cm = ME.CoordinateManager(D=3)
        top = ME.SparseTensor(torch.ones((len(top_c), 1)), coordinates=top_c, tensor_stride=4, coordinate_manager=cm)
        up = ME.MinkowskiGenerativeConvolutionTranspose(1, 1, 2, 2, bias=False, dimension=3)
        with torch.no_grad():
            mid = up(top)                                                         # all 8 children at stride 2: key (2, '')
            mask = torch.rand(mid.F.shape[0], generator=g) < 0.4
            mask[::8] = True                                                      # every parent keeps a child
            mid = ME.MinkowskiPruning()(mid, mask)                                # key (2, 'pruned')
            cand = up(mid)                                                        # candidates at stride 1
        logits = torch.randn((cand.F.shape[0], 1), generator=g)
        pred = ME.SparseTensor(logits, coordinate_map_key=cand.coordinate_map_key, coordinate_manager=cm)
        assert sorted(k.get_key()[1] for k in cm.get_coordinate_map_keys([2, 2, 2])) == ['', 'pruned']
        n = logits.shape[0]
        if batch == 1:
            targets = [[mid.F.shape[0]], [n // 3], [n - 9], None]
        else:
            sizes = [p.numel() for p in pred.decomposition_permutations]
            targets = [[sizes[0] // 4, sizes[1] // 2], None]
        keeps = []
        for t in targets:
            keep = dec.get_keep(pred, None if t is None else [list(t)], [2, 2, 2])
            keeps.append({'points_num': t, 'keep': np.packbits(keep.numpy().astype(np.uint8)).tobytes().hex(), 'kept': int(keep.sum())})
        i16 = lambda t: np.ascontiguousarray(t.numpy().astype('<i2')).tobytes().hex()        # [n, 4] (batch, x, y, z), little-endian int16
        cases.append({'label': label, 'seed': seed, 'batch': batch, 'top_coords_i16': i16(top_c), 'mid_coords_i16': i16(mid.C),
                      'cand_coords_i16': i16(pred.C), 'logits_f32': np.ascontiguousarray(logits.view(-1).numpy().astype('<f4')).tobytes().hex(),
                      'queries': keeps})
        print('get_keep', label, n, 'candidates', [q['kept'] for q in keeps])
    return {'cases': cases}


def make_codec_color():
    """The reference's joint geometry + colour codec (models/convolutional/lossy_coord_lossy_color) EXECUTED on the CPU over the
    MinkowskiEngine stand-in of make_codec_v2 with the reference's rANS coders: streams and reconstructions of seeded runs."""
    import torch
    import torch.utils.cpp_extension as ce
    _stub_engines()
    _functional_minkowski()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_lossy_color.model import PCC
        from models.convolutional.lossy_coord_lossy_color.model_config import ModelConfig
    finally:
        ce.load = real
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    runs = []
    for label, kw, seed, res, pts in (
            ('two_stages', dict(encoder_channels=(8, 16, 16), decoder_channels=(16, 8), geo_lossl_if_sample=(0, 1, 0, 1),
                                geo_lossl_channels=(16, 32, 32, 32, 1)), 1, 64, 3000),
            ('one_stage', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1, 0, 1),
                               geo_lossl_channels=(16, 16, 16, 16, 16, 16, 1)), 2, 64, 2000)):
        cfg = ModelConfig()
        import yaml
        with open(os.path.join(REF, 'config/convolutional/lossy_coord_lossy_color/baseline_r1.yaml')) as f:
            for k, v in yaml.safe_load(f)['model'].items():
                assert hasattr(cfg, k), k
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        for k, v in kw.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        cfg.compressed_channels = (cfg.compressed_channels[0],) if isinstance(cfg.compressed_channels, tuple) else (cfg.compressed_channels,)
        cfg.check()
        torch.manual_seed(0)
        model = PCC(cfg)
        enliven(model, seed)
        model.eval()
        xyz = surface_cloud(seed + 50, res, pts) + np.array([1, 3, 0], dtype=np.int32)
        rng = np.random.default_rng(seed)
        perm = rng.permutation(len(xyz))
        color = np.clip(128 + 60 * np.sin(xyz / 7.0) + rng.normal(0, 12, xyz.shape), 0, 255).round().astype(np.float32)
        with torch.no_grad():
            data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32), torch.from_numpy(color[perm]))
            rec_xyz, rec_rgb = model.decompress(data)
        runs.append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()
                                                if not k.startswith('_')},
                     'seed': seed, 'xyz': xyz[perm].tolist(), 'color': color[perm].astype(int).tolist(),
                     'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
                     'stream_hex': data.hex(), 'recon_xyz': rec_xyz.tolist(), 'recon_rgb': rec_rgb.to(torch.int32).tolist()})
        print('codec_color', label, len(xyz), 'points ->', len(data), 'bytes,', len(rec_xyz), 'decoded')
    return {'runs': runs}


def make_codec_lossl():
    """The reference's FLOAT LiDAR model (models/convolutional/lossl_coord: the model that is trained, calibrated and converted
    into lossl_coord_int) EXECUTED on the CPU over the functional torchsparse stand-in: compress / decompress runs and the
    loss terms of train_forward on a batch of two sweeps."""
    import torch
    import torch.utils.cpp_extension as ce
    _functional_torchsparse()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossl_coord import model as ref_model
        from models.convolutional.lossl_coord.model_config import Config
    finally:
        ce.load = real
    from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
    from fastpcc_amd.synthetic import batched, lidar_cloud
    torch.cuda.synchronize = lambda *a, **k: None
    out = {'runs': [], 'train': []}
    for label, kw, seed in (('c8', dict(channels=8), 1), ('c8_more_ch', dict(channels=8, use_more_ch_for_multi_step_pred=True), 2),
                            ('c16_fea8', dict(channels=16, fea_stride=8, max_stride_wo_recurrent=512, max_stride=4096), 3)):
        cfg = Config()
        for k, v in kw.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        model = ref_model.Model(cfg)
        randomize_(model, seed)
        keys = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        model.eval()
        xyz = lidar_cloud(seed + 80, beams=6, azimuths=128) + np.array([4, 0, 7], dtype=np.int32)
        perm = np.random.default_rng(seed).permutation(len(xyz))
        with torch.no_grad():
            data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32))
            rec = model.decompress(data)
        assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist())), 'the reference run is not lossless'
        out['runs'].append({'label': label, 'config': kw, 'seed': seed, 'xyz': xyz[perm].tolist(), 'stream_hex': data.hex(),
                            'state_dict_keys': keys,
                            'param_abs_sum': float(sum(p.detach().double().abs().sum() for p in model.parameters()))})
        print('codec_lossl', label, len(xyz), 'points ->', len(data), 'bytes')
        # training objective on a batch of two sweeps (per sample Morton-sorted, z on the lowest bit)
        model.train()
        parts = []
        for b, s2 in enumerate((seed + 90, seed + 95)):
            c = lidar_cloud(s2, beams=5, azimuths=96)
            c = c - c.min(0)
            key = sum(((c[:, a].astype(np.int64) >> i) & 1) << (3 * i + (2 - a)) for i in range(17) for a in range(3))
            c = c[np.argsort(key, kind='stable')]
            parts.append(np.concatenate((np.full((len(c), 1), b), c), 1))
        batch = torch.from_numpy(np.concatenate(parts, 0).astype(np.int32))
        res = model.train_forward(batch, [len(p) for p in parts], 0)
        out['train'].append({'label': label, 'config': kw, 'seed': seed, 'xyz': batch.tolist(), 'points_num': [len(p) for p in parts],
                             'loss': float(res['loss']), 'terms': {k: float(v) for k, v in res.items() if k != 'loss'}})
        print('codec_lossl train', label, float(res['loss']))
    return out


def _chain_env_v2():
    """imports the reference's lossy_coord_v2 model code over the stand-in engine in CHAIN mode (see make_codec_v2_chain): -> (PCC,
    ModelConfig, hipops)"""

    import torch
    import torch.utils.cpp_extension as ce
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.orders import summation_order       # the oracle's restatement of the specification, NOT the product's rule
    from fastpcc_amd import hipops                   # (tests/test_orders.py holds the product's rule against it)
    _stub_engines()
    _functional_minkowski(order_fn=summation_order)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_v2.model import PCC
        from models.convolutional.lossy_coord_v2.model_config import ModelConfig
    finally:
        ce.load = real
    return PCC, ModelConfig, hipops


def make_codec_v2_chain():
    """lossy_coord_v2 runs of the REFERENCE's model code (as make_codec_v2) over the stand-in engine in CHAIN mode: every convolution
    and linear layer summed in the order the HIP kernels document (oracle/orders.py, the restatement of the numerics version of
    include/fpcc_hip.h).  These streams are what the GPU path has to write byte for byte, and their reconstructions point for point --
    the mm-mode runs of codec_v2.json sum in torch's CPU order and can only be matched within tolerances."""
    import torch
    PCC, ModelConfig, hipops = _chain_env_v2()
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    out = {'numerics_version': hipops.numerics_version(), 'runs': []}
    base = dict(activation='prelu', compressed_channels=(1,), skip_encoding_fea=1, adaptive_pruning=True)

    from oracle import sparse_conv as sc_
    real_sigmoid = torch.Tensor.sigmoid

    def run(label, cfg, cfg_dict, seed, xyz, extra=None):
        torch.manual_seed(0)
        model = PCC(cfg)
        enliven(model, seed)
        model.eval()
        perm = np.random.default_rng(seed).permutation(len(xyz))
        # the logistic function in front of the 16-bit probabilities (GeoLosslessEntropyModel.init_prob calls dist.sigmoid()) is the
        # one the HIP path specifies from numerics version 3 on -- like the convolution sums, an arithmetic primitive with a documented
        # rounding, not model logic
        torch.Tensor.sigmoid = lambda self: torch.from_numpy(sc_.sigmoid_spec(self)).to(self.device)
        try:
            with torch.no_grad():
                data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32))
                rec = model.decompress(data)
        finally:
            torch.Tensor.sigmoid = real_sigmoid
        rec_np = rec.cpu().numpy().astype(np.int64)
        keys = np.sort((rec_np[:, 0] << 42) | (rec_np[:, 1] << 21) | rec_np[:, 2])
        entry = {'label': label, 'config': cfg_dict, 'seed': seed, 'xyz': xyz[perm].tolist(),
                 'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
                 'stream_hex': data.hex(), 'recon_points': len(rec), 'recon_sha256': hashlib.sha256(keys.tobytes()).hexdigest()}
        entry.update(extra or {})
        out['runs'].append(entry)
        print('codec_v2_chain', label, len(xyz), 'points ->', len(data), 'bytes,', len(rec), 'decoded')

    for label, kw, seed, res, pts in (
            ('r1_like', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1, 0, 1),
                             geo_lossl_channels=(16, 32, 32, 32, 32, 32, 1)), 1, 64, 2500),
            ('r3_like_two_stages', dict(encoder_channels=(8, 16, 16), decoder_channels=(16, 8), geo_lossl_if_sample=(0, 1, 0, 1),
                                        geo_lossl_channels=(16, 32, 32, 32, 1)), 2, 64, 3000)):
        cfg = ModelConfig()
        for k, v in {**base, **kw}.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        cfg.check()
        run(label, cfg, {k: (list(v) if isinstance(v, tuple) else v) for k, v in {**base, **kw}.items()}, seed,
            surface_cloud(seed + 40, res, pts) + np.array([2, 0, 5], dtype=np.int32))
    # the headline configuration at its real widths, from the reference's own YAML through its own config loader
    from lib.config import Config as RefConfig
    ref_cfg = RefConfig()
    ref_cfg.merge_with_yaml(os.path.join(REF, 'config/convolutional/lossy_coord_v2/baseline_r1.yaml'))
    ref_cfg.check()
    mc = ref_cfg.model
    run('baseline_r1_yaml', mc, {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in vars(mc).items() if not k.startswith('_')}, 6,
        surface_cloud(46, 256, 4000) + np.array([1, 4, 2], dtype=np.int32), {'yaml': 'config/convolutional/lossy_coord_v2/baseline_r1.yaml'})
    run('baseline_r1_yaml_12k', mc, {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in vars(mc).items() if not k.startswith('_')}, 7,
        surface_cloud(47, 256, 16000) + np.array([3, 1, 0], dtype=np.int32), {'yaml': 'config/convolutional/lossy_coord_v2/baseline_r1.yaml'})
    return out


def make_codec_v2_partitions_chain():
    """The reference's LIST path -- PCC.compress_partitions / decompress_partitions (lossy_coord_v2/model.py:247-256,277-288), which code
    the clouds of a list one after the other -- executed in chain mode (as make_codec_v2_chain): the bytes and the decoded points that
    the product's batched traversal (compress_many: ONE network pass over all clouds of the list) must reproduce.  The clouds of a run
    differ in size by more than an order of magnitude, so that the row-count rule of the narrow layers (PAD_MIN_ROWS) differs between
    the clouds of one list.  (All at 256^3: the reference's coder asserts on a cloud whose bottom level is a single voxel.)"""
    import torch
    PCC, ModelConfig, hipops = _chain_env_v2()
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    from oracle import sparse_conv as sc_
    from lib.config import Config as RefConfig
    real_sigmoid = torch.Tensor.sigmoid
    out = {'numerics_version': hipops.numerics_version(), 'runs': []}
    ref_cfg = RefConfig()
    ref_cfg.merge_with_yaml(os.path.join(REF, 'config/convolutional/lossy_coord_v2/baseline_r1.yaml'))
    ref_cfg.check()
    mc = ref_cfg.model
    cfg_dict = {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in vars(mc).items() if not k.startswith('_')}
    for label, seed, clouds in (
            ('baseline_r1_yaml_three_clouds', 8, [surface_cloud(60, 256, 9000) + np.array([1, 4, 2], dtype=np.int32),
                                                  surface_cloud(61, 256, 500) + np.array([100, 3, 50], dtype=np.int32),
                                                  surface_cloud(62, 256, 3000)]),
            ('baseline_r1_yaml_two_clouds', 9, [surface_cloud(63, 256, 800), surface_cloud(64, 256, 14000) + np.array([0, 9, 1], dtype=np.int32)])):
        torch.manual_seed(0)
        model = PCC(mc)
        enliven(model, seed)
        model.eval()
        rng = np.random.default_rng(seed)
        parts = [torch.from_numpy(batched(c)[rng.permutation(len(c))]).to(torch.int32) for c in clouds]
        torch.Tensor.sigmoid = lambda self: torch.from_numpy(sc_.sigmoid_spec(self)).to(self.device)
        try:
            with torch.no_grad():
                blob = model.compress_partitions([torch.cat(parts), *parts])        # element 0: the unpartitioned cloud (model.py:249)
                rec = model.decompress_partitions(blob)
        finally:
            torch.Tensor.sigmoid = real_sigmoid
        rec_np = rec.cpu().numpy().astype(np.int64)
        # the clouds' shares of the reconstruction: decompress_partitions concatenates them in list order
        counts, hashes, at, pos = [], [], 0, 0
        while pos != len(blob):
            length = int.from_bytes(blob[pos:pos + 3], 'little')
            torch.Tensor.sigmoid = lambda self: torch.from_numpy(sc_.sigmoid_spec(self)).to(self.device)
            try:
                with torch.no_grad():
                    n_i = len(model.decompress(blob[pos + 3: pos + 3 + length]))
            finally:
                torch.Tensor.sigmoid = real_sigmoid
            part = rec_np[at: at + n_i]
            keys = np.sort((part[:, 0] << 42) | (part[:, 1] << 21) | part[:, 2])
            counts.append(n_i)
            hashes.append(hashlib.sha256(keys.tobytes()).hexdigest())
            at, pos = at + n_i, pos + 3 + length
        assert at == len(rec_np)
        out['runs'].append({'label': label, 'config': cfg_dict, 'seed': seed, 'parts': [p[:, 1:].tolist() for p in parts],
                            'yaml': 'config/convolutional/lossy_coord_v2/baseline_r1.yaml', 'blob_hex': blob.hex(),
                            'recon_points': counts, 'recon_sha256': hashes})
        print('codec_v2_partitions_chain', label, [len(c) for c in clouds], 'points ->', len(blob), 'bytes,', len(rec_np), 'decoded')
    return out


def _chain_env_color():
    """imports the reference's lossy_coord_lossy_color model code over the stand-in engine in CHAIN mode: -> (PCC, ModelConfig, hipops)"""

    import torch
    import torch.utils.cpp_extension as ce
    import yaml
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle.orders import summation_order       # the oracle's restatement of the specification, NOT the product's rule
    from fastpcc_amd import hipops                   # (tests/test_orders.py holds the product's rule against it)
    _stub_engines()
    _functional_minkowski(order_fn=summation_order)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == 'lib' or k.startswith('lib.') or k == 'models' or k.startswith('models.')]:
        del sys.modules[k]
    real = ce.load
    import rans_ext_cpp
    import simple_rans_ext_cpp
    built = {'rans_ext_cpp': rans_ext_cpp, 'simple_rans_ext_cpp': simple_rans_ext_cpp}
    ce.load = lambda *a, **k: built.get(k.get('name', a[0] if a else ''), types.SimpleNamespace())
    try:
        from models.convolutional.lossy_coord_lossy_color.model import PCC
        from models.convolutional.lossy_coord_lossy_color.model_config import ModelConfig
    finally:
        ce.load = real
    return PCC, ModelConfig, hipops


def make_codec_color_partitions_chain():
    """the reference's list path of the colour codec (lossy_coord_lossy_color/model.py:262-275,299-314) in chain mode, on a list of
    coloured clouds of different sizes: see make_codec_v2_partitions_chain"""
    import torch
    import yaml
    PCC, ModelConfig, hipops = _chain_env_color()
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    from oracle import sparse_conv as sc_
    real_sigmoid = torch.Tensor.sigmoid
    out = {'numerics_version': hipops.numerics_version(), 'runs': []}
    cfg = ModelConfig()
    with open(os.path.join(REF, 'config/convolutional/lossy_coord_lossy_color/baseline_r1.yaml')) as f:
        for k, v in yaml.safe_load(f)['model'].items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
    cfg.compressed_channels = (cfg.compressed_channels[0],) if isinstance(cfg.compressed_channels, tuple) else (cfg.compressed_channels,)
    cfg.check()
    seed, gain = 4, 2.3
    torch.manual_seed(0)
    model = PCC(cfg)
    enliven(model, seed, gain=gain)
    model.eval()
    rng = np.random.default_rng(seed)
    clouds = [surface_cloud(70, 256, 7000) + np.array([1, 3, 0], dtype=np.int32), surface_cloud(71, 256, 700), surface_cloud(72, 256, 2500)]
    xyzs, colors = [], []
    for c in clouds:
        perm = rng.permutation(len(c))
        xyzs.append(torch.from_numpy(batched(c)[perm]).to(torch.int32))
        colors.append(torch.from_numpy(np.clip(128 + 60 * np.sin(c[perm] / 7.0) + rng.normal(0, 12, c.shape), 0, 255).round().astype(np.float32)))
    torch.Tensor.sigmoid = lambda self: torch.from_numpy(sc_.sigmoid_spec(self)).to(self.device)
    try:
        with torch.no_grad():
            blob = model.compress_partitions([torch.cat(xyzs), *xyzs], [torch.cat(colors), *colors])
            parts, pos = [], 0
            while pos != len(blob):
                length = int.from_bytes(blob[pos:pos + 3], 'little')
                rec_xyz, rec_rgb = model.decompress(blob[pos + 3: pos + 3 + length])
                parts.append({'recon_xyz': rec_xyz.tolist(), 'recon_rgb': rec_rgb.to(torch.int32).tolist()})
                pos += 3 + length
    finally:
        torch.Tensor.sigmoid = real_sigmoid
    out['runs'].append({'label': 'baseline_r1_yaml_three_clouds', 'seed': seed, 'gain': gain,
                        'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items() if not k.startswith('_')},
                        'xyz': [x[:, 1:].tolist() for x in xyzs], 'color': [c.to(torch.int32).tolist() for c in colors],
                        'blob_hex': blob.hex(), 'parts': parts})
    print('codec_color_partitions_chain', [len(c) for c in clouds], 'points ->', len(blob), 'bytes,', [len(p['recon_xyz']) for p in parts], 'decoded')
    return out


def make_codec_color_chain():
    """lossy_coord_lossy_color runs of the REFERENCE's model code (as make_codec_color) over the stand-in engine in CHAIN mode (see
    make_codec_v2_chain): the streams the GPU path has to write byte for byte and the coloured clouds it has to decode point for point."""
    import torch
    import yaml
    PCC, ModelConfig, hipops = _chain_env_color()
    from fastpcc_amd.synthetic import batched, enliven, surface_cloud
    from oracle import sparse_conv as sc_
    real_sigmoid = torch.Tensor.sigmoid
    out = {'numerics_version': hipops.numerics_version(), 'runs': []}
    for label, kw, seed, res, pts, gain in (
            ('two_stages', dict(encoder_channels=(8, 16, 16), decoder_channels=(16, 8), geo_lossl_if_sample=(0, 1, 0, 1),
                                geo_lossl_channels=(16, 32, 32, 32, 1)), 1, 64, 3000, None),
            ('one_stage', dict(encoder_channels=(8, 16), decoder_channels=(8,), geo_lossl_if_sample=(0, 1, 0, 1, 0, 1),
                               geo_lossl_channels=(16, 16, 16, 16, 16, 16, 1)), 2, 64, 2000, None),
            # the reference's own YAML at its real widths (a larger gain: at the default one every residual of this depth quantises to 0)
            ('baseline_r1_yaml', {}, 3, 256, 6000, 2.3)):
        cfg = ModelConfig()
        with open(os.path.join(REF, 'config/convolutional/lossy_coord_lossy_color/baseline_r1.yaml')) as f:
            for k, v in yaml.safe_load(f)['model'].items():
                assert hasattr(cfg, k), k
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        for k, v in kw.items():
            assert hasattr(cfg, k), k
            setattr(cfg, k, v)
        cfg.compressed_channels = (cfg.compressed_channels[0],) if isinstance(cfg.compressed_channels, tuple) else (cfg.compressed_channels,)
        cfg.check()
        torch.manual_seed(0)
        model = PCC(cfg)
        enliven(model, seed, **({} if gain is None else {'gain': gain}))
        model.eval()
        xyz = surface_cloud(seed + 50, res, pts) + np.array([1, 3, 0], dtype=np.int32)
        rng = np.random.default_rng(seed)
        perm = rng.permutation(len(xyz))
        color = np.clip(128 + 60 * np.sin(xyz / 7.0) + rng.normal(0, 12, xyz.shape), 0, 255).round().astype(np.float32)
        torch.Tensor.sigmoid = lambda self: torch.from_numpy(sc_.sigmoid_spec(self)).to(self.device)      # see make_codec_v2_chain
        try:
            with torch.no_grad():
                data = model.compress(torch.from_numpy(batched(xyz)[perm]).to(torch.int32), torch.from_numpy(color[perm]))
                rec_xyz, rec_rgb = model.decompress(data)
        finally:
            torch.Tensor.sigmoid = real_sigmoid
        out['runs'].append({'label': label, 'config': {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()
                                                       if not k.startswith('_')},
                            'seed': seed, 'gain': gain, 'xyz': xyz[perm].tolist(), 'color': color[perm].astype(int).tolist(),
                            'param_abs_sum': float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)),
                            'stream_hex': data.hex(), 'recon_xyz': rec_xyz.tolist(), 'recon_rgb': rec_rgb.to(torch.int32).tolist()})
        print('codec_color_chain', label, len(xyz), 'points ->', len(data), 'bytes,', len(rec_xyz), 'decoded')
    return out


def main():
    for name, fn in (('get_keep', make_get_keep), ('codec_lossl', make_codec_lossl), ('codec_color', make_codec_color), ('codec_v2', make_codec_v2), ('codec_v2_chain', make_codec_v2_chain), ('codec_v2_partitions_chain', make_codec_v2_partitions_chain), ('codec_color_chain', make_codec_color_chain), ('codec_color_partitions_chain', make_codec_color_partitions_chain), ('codec_int', make_codec_int), ('codec_v3', make_codec_v3), ('hilbert', make_hilbert), ('me_semantics', make_me_semantics), ('entropy_model_hyperprior', make_entropy_model_hyperprior), ('entropy_model_indexed', make_entropy_model_indexed), ('ptq_import', make_ptq_import), ('kdtree', make_kdtree), ('entropy_model', make_entropy_model), ('rans', make_rans), ('morton', make_morton), ('byteslist', make_byteslist), ('explut', make_explut)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        data = fn()
        path = os.path.join(HERE, name + '.json')
        with open(path, 'w') as f:
            json.dump(data, f, separators=(',', ':'))
        print(name, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
