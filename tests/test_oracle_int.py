"""Integer pipeline oracle: hand-derived known answers for the fixed-point epilogues (the reference has no runnable
integer path) and a lossless round trip of the oracle codec."""
import numpy as np

from fastpcc_amd.codecs.lossl_coord_int import Config, Model
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import batched, lidar_cloud
from oracle import codec_int as oi


def test_round_half_away_and_saturation():
    # (in * mul + zp) >> shift with round-half-away-from-zero of the magnitude (requant.cu:14-25)
    x = np.array([[3, -3, 1, -1, 5, -5, 0, 100000]], dtype=np.int32)
    out = oi.epilogue(x, None, None, [1], 0, 1, 32)
    assert out.tolist() == [[2, -2, 1, -1, 3, -3, 0, 50000]]              # 1.5 -> 2, -1.5 -> -2, 0.5 -> 1, -0.5 -> -1
    assert oi.epilogue(x, None, None, [1], 0, 0, 32).tolist() == x.tolist()   # shift 0: identity
    assert oi.epilogue(x, None, None, [4], 0, 1, 8).tolist() == [[6, -6, 2, -2, 10, -10, 0, 127]]   # int8 saturation
    assert oi.epilogue(np.array([[-100000]], np.int32), None, None, [4], 0, 1, 8).tolist() == [[-128]]
    # zero point is added before the shift; multiplier is unsigned 32 bit
    assert oi.epilogue(np.array([[1]], np.int32), None, None, [0xffffffff], 1 << 31, 32, 32).tolist() == [[1]]
    assert oi.epilogue(np.array([[2]], np.int32), None, None, [0x80000000], 0, 32, 32).tolist() == [[1]]


def test_bias_prelu_requant_order():
    # v = in + bias; negative v -> rha(v * slope, 25); then requant (bias_prelu_requant.cu:15-36)
    slope = [1 << 24]                                   # 0.5 in Q6.25
    x = np.array([[10, -10, -3, -1]], dtype=np.int32)
    b = np.array([-4, 4, 0, 0], dtype=np.int32)
    out = oi.epilogue(x, b, slope, [1, 1, 1, 1], 0, 0, 32)
    assert out.tolist() == [[6, -3, -2, -1]]            # -6*0.5 = -3; -3*0.5 = -1.5 -> -2; -0.5 -> -1
    assert oi.prelu_i32(np.array([-3, 3, -2 ** 31], np.int32), 1 << 24).tolist() == [-2, 3, -2 ** 30]
    assert oi.prelu_i32(np.array([-2 ** 31], np.int32), 3 << 25).tolist() == [-2 ** 31]   # 3x: saturates


def test_softmax_properties_and_fallback():
    # uniform logits -> uniform probabilities summing to ~2^32; a dominant logit takes (almost) everything
    p = oi.softmax_i32(np.zeros((1, 255), np.int32)).astype(np.int64)
    # e = LUT[0] = 65536 each, S = 255 * 65536, inv = (2^32 + S/2) / S = 257, p = 65536 * 257
    assert (p == 65536 * 257).all()
    x = np.zeros((1, 4), np.int32)
    x[0, 2] = 13 << 16                                  # 13.0 in Q15.16: the others fall off the 12-wide table
    p = oi.softmax_i32(x).astype(np.int64)
    assert p[0, 2] > 0.99999 * 2 ** 32 and (p[0, [0, 1, 3]] < 2 ** 32 * 1e-5).all()
    cdf = oi.quantize_pmf(np.zeros((3, 255), np.int32))
    assert cdf.dtype == np.uint16 and (np.diff(cdf.astype(np.int64), axis=1) > 0).all() and (cdf[:, -1] == 65535).all()


def test_kernel_offset_order():
    c = np.array([[0, 4, 4, 4]])
    nb = np.array([[0, 4 + dx, 4 + dy, 4 + dz] for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
    t = oi.kernel_table(nb, c, (3, 3, 3), (1, 1, 1))
    assert t[:, 0].tolist() == list(range(27))          # odd kernel: x fastest, centred
    kids = np.array([[0, 8 + dx, 8 + dy, 8 + dz] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)])
    t = oi.kernel_table(kids, c, (2, 2, 2), (2, 2, 2))
    assert t[:, 0].tolist() == list(range(8))           # even kernel: z fastest, anchored at 0
    far = np.array([[0, 16 + d, 16, 16] for d in (-1, 0, 1, 2)])
    t = oi.kernel_table(far, c, (4, 4, 4), (4, 4, 4))   # 4^3 / stride 4: offsets -1..2 (not block aligned)
    hits = {k: int(v) for k, v in enumerate(t[:, 0]) if v >= 0}
    assert hits == {16 * 0 + 4 * 1 + 1: 0, 16 * 1 + 4 * 1 + 1: 1, 16 * 2 + 4 * 1 + 1: 2, 16 * 3 + 4 * 1 + 1: 3}


def test_oracle_codec_lossless_roundtrip():
    xyz = lidar_cloud(3, beams=12, azimuths=256)
    for skip in (0, 2):
        cfg = Config(channels=32, skip_top_scales_num=skip)
        model = Model(cfg, 'cpu')
        randomize_(model, 1)
        o = oi.OracleInt(model.state_dict(), cfg)
        data = o.compress(batched(xyz).astype(np.int64) + np.array([0, 7, 0, 3]))
        rec = o.decompress(data)
        assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, (xyz + np.array([7, 0, 3])).tolist()))
        assert [int.from_bytes(data[i:i + 2], 'little') for i in (0, 2, 4)] == (xyz.min(0) + np.array([7, 0, 3])).tolist()
