"""Device-side rANS decoders (fpcc_rans_binary_decode_dev, fpcc_simple_dec_pop_dev) against the reference's golden streams
(tests/golden/rans.json, produced by the reference's own C++ coders), against libfpcc_host on large seeded streams, and
inside the two codecs: decoding with the device decoders must reconstruct exactly what the host path reconstructs."""
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd.rans_coder import BinaryRansCoder, RansDecoder, RansEncoder
from fastpcc_amd.synthetic import batched, enliven, lidar_cloud, surface_cloud

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), 'golden', 'rans.json')) as f:
    G = json.load(f)


def _state(stream: bytes, ops):
    x = int.from_bytes(stream[:4], 'little')
    return torch.tensor([x - (1 << 32) if x >= 1 << 31 else x, 4, 0, 0], dtype=torch.int32).cuda()


def test_binary_decoder_on_the_reference_golden_streams():
    from fastpcc_amd import hipops as ops
    for case in G['binary']:
        stream = bytes.fromhex(case['stream'])
        prob = torch.tensor(case['prob'], dtype=torch.int32).to(torch.int16).cuda() if max(case['prob']) < 32768 else \
            torch.from_numpy(np.array(case['prob'], dtype=np.uint16).view(np.int16)).cuda()
        bits, ones, status = ops.rans_binary_decode_dev(ops.stream_to_device(stream, 'cuda'), len(stream), prob.view(-1))
        assert int(status.item()) == 0
        assert bits.cpu().numpy().astype(bool).tolist() == [bool(b) for b in np.array(case['bits']).reshape(-1)]
        assert int(ones.item()) == int(np.sum(case['bits']))


@pytest.mark.parametrize('n', [0, 1, 63, 64, 65, 1000, 300_000])
def test_binary_decoder_equals_host_decoder(n):
    from fastpcc_amd import hipops as ops
    rng = np.random.default_rng(n)
    p = np.clip(np.round(rng.beta(.3, .3, n) * 65536), 1, 65535).astype(np.uint16)
    bits = rng.random(n) < p / 65536
    stream = BinaryRansCoder(1).encode(bits[None], p[None].astype(np.uint32))[0]
    got, ones, status = ops.rans_binary_decode_dev(ops.stream_to_device(stream, 'cuda'), len(stream),
                                                   torch.from_numpy(p.view(np.int16)).cuda())
    assert int(status.item()) == 0 and int(ones.item()) == int(bits.sum())
    assert (got.cpu().numpy().astype(bool) == bits).all()


def test_binary_decoder_rejects_a_zero_state_and_survives_truncation():
    from fastpcc_amd import hipops as ops
    prob = torch.full((100,), 1000, dtype=torch.int16).cuda()
    _, _, status = ops.rans_binary_decode_dev(ops.stream_to_device(b'\0' * 16, 'cuda'), 16, prob)
    assert int(status.item()) == -2
    rng = np.random.default_rng(1)
    p = np.clip(np.round(rng.beta(.3, .3, 5000) * 65536), 1, 65535).astype(np.uint16)
    bits = rng.random(5000) < p / 65536
    stream = BinaryRansCoder(1).encode(bits[None], p[None].astype(np.uint32))[0][:9]          # truncated: garbage out, but it ends
    got, _, status = ops.rans_binary_decode_dev(ops.stream_to_device(stream, 'cuda'), len(stream), torch.from_numpy(p.view(np.int16)).cuda())
    host = np.zeros((1, 5000), dtype=bool)
    BinaryRansCoder(1).decode([stream], p[None].astype(np.uint32), host)
    assert int(status.item()) == 0 and (got.cpu().numpy().astype(bool) == host[0]).all()      # the same garbage as the host's


def test_row_decoder_on_the_reference_golden_streams():
    from fastpcc_amd import hipops as ops
    done = 0
    for case in G['simple']:
        if 'bin' in case or any(np.array(b['rows']).shape[-1] > 256 for b in case['blocks']):
            continue
        stream = bytes.fromhex(case['stream'])
        dev_stream, state = ops.stream_to_device(stream, 'cuda'), _state(stream, ops)
        for blk in reversed(case['blocks']):
            rows = np.array(blk['rows'], dtype=np.uint16)
            if rows.shape[0] == 1 and len(blk['symbols']) > 1:
                rows = np.repeat(rows, len(blk['symbols']), 0)
            sym, children = ops.simple_dec_pop_dev(state, dev_stream, len(stream), torch.from_numpy(rows.view(np.int16)).cuda())
            assert sym.cpu().numpy().view(np.uint16).tolist() == blk['symbols']
            assert children.tolist() == [sum(bin((s + 1) & 255).count('1') for s in blk['symbols']), 0]      # count, status word
            done += 1
    assert done > 0


@pytest.mark.parametrize('n', [1, 64, 65, 5000])
def test_row_decoder_continues_a_host_decoder_mid_stream(n):
    """255-entry CDF rows as the integer codec makes them; the stream starts with symbols the HOST decodes (as the codec's
    bottom coordinates are), then the device takes over from fpcc_simple_dec_tell, level after level"""
    from fastpcc_amd import hipops as ops
    rng = np.random.default_rng(n + 7)

    def rows_and_symbols(m):
        f = rng.integers(1, 400, (m, 255)).astype(np.int64)
        c = np.cumsum(f * (65000 // f.sum(1, keepdims=True)), 1)
        c[:, -1] = 65535
        c = c.astype(np.uint16)
        return c, rng.integers(0, 255, m).astype(np.uint16)

    blocks = [rows_and_symbols(n), rows_and_symbols(max(1, n // 3)), rows_and_symbols(7)]
    enc = RansEncoder(1 << 22)
    for rows, sym in blocks:                               # pushed first = decoded last
        enc.encode(rows, sym)
    stream = enc.flush()
    dec = RansDecoder()
    dec.flush(stream)
    head = np.zeros(7, dtype=np.uint16)
    dec.decode(blocks[2][0], head)                         # host decodes the last-pushed block
    assert (head == blocks[2][1]).all()
    x, pos = dec.tell()
    state = torch.tensor([x - (1 << 32) if x >= 1 << 31 else x, pos, 0, 0], dtype=torch.int32).cuda()
    dev_stream = ops.stream_to_device(stream, 'cuda')
    for rows, want in (blocks[1], blocks[0]):
        sym, info = ops.simple_dec_pop_dev(state, dev_stream, len(stream), torch.from_numpy(rows.view(np.int16)).cuda())
        assert (sym.cpu().numpy().view(np.uint16) == want).all() and int(info[1]) == 0
    # a state that is not a rANS state, and a CDF row that does not increase: flagged in the sticky status word, no crash
    bad_state = torch.tensor([5, 4, 0, 0], dtype=torch.int32).cuda()
    _, info = ops.simple_dec_pop_dev(bad_state, dev_stream, len(stream), torch.from_numpy(blocks[0][0].view(np.int16)).cuda())
    assert int(info[1]) & 1 and int(bad_state[3]) & 1
    flat = np.full((200, 255), 20000, dtype=np.uint16)      # a third of the slots fall between two equal edges
    flat[:, 0] = 40000
    _, info = ops.simple_dec_pop_dev(_state(stream, ops), dev_stream, len(stream), torch.from_numpy(flat.view(np.int16)).cuda())
    assert int(info[1]) & 2


def test_v2_decompress_with_the_device_decoder_reconstructs_the_same_cloud():
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    model = model.cuda().eval()
    xyz = surface_cloud(2, 128, 60000)
    data = model.compress(torch.from_numpy(batched(xyz)).to(torch.int32).cuda())
    host = model.decompress(data).cpu().numpy()
    model.em_lossless_based.device_decoder = True
    try:
        dev = model.decompress(data).cpu().numpy()
    finally:
        model.em_lossless_based.device_decoder = False
    assert host.shape == dev.shape and (host == dev).all()


def test_int_codec_decompress_with_the_device_decoder_is_lossless():
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    model = Model(Config(channels=64), 'cuda')
    randomize_(model, 5)
    model = model.cuda().eval()
    xyz = lidar_cloud(3, beams=24, azimuths=512)
    data = model.compress(torch.from_numpy(batched(xyz)).cuda())
    host = model.decompress(data).cpu().numpy()
    model.device_decoder = True
    try:
        dev = model.decompress(data).cpu().numpy()
    finally:
        model.device_decoder = False
    assert (host == dev).all() and sorted(map(tuple, dev.tolist())) == sorted(map(tuple, xyz.tolist()))
