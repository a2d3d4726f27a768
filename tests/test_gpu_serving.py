"""Frames in flight (fastpcc_amd/serving.py) on the GPU: two codec contexts on one stream write exactly the bytes and decode exactly the
points of the single-frame code, for different frames interleaved in any order; the per-context stream form does too."""
import numpy as np
import pytest
import torch

from fastpcc_amd import engine as ME
from fastpcc_amd.serving import FramePipeline, wait_for_my_work
from util import batched, enliven, surface_cloud

pytestmark = pytest.mark.gpu


def _model():
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    return model.cuda().eval()


def _frames():
    return [torch.from_numpy(batched(surface_cloud(20 + i, 128, n))).to(torch.int32).cuda() for i, n in enumerate((30000, 9000, 52000, 17000))]


def _key(points: torch.Tensor) -> np.ndarray:
    p = points.cpu().numpy().astype(np.int64)
    return np.sort((p[:, 0] << 42) | (p[:, 1] << 21) | p[:, 2])


@pytest.mark.parametrize('own_streams', [False, True])
def test_pipelined_frames_equal_the_single_frame_code(own_streams):
    model = _model()
    frames = _frames()
    want = []
    for f in frames:
        data = model.compress(f)
        ME.clear_global_coordinate_manager()
        rec = model.decompress(data)
        torch.cuda.synchronize()
        ME.clear_global_coordinate_manager()
        want.append((data, _key(rec)))

    def step(ctx, i):
        f = frames[i % len(frames)]
        data = ctx.compress(f)
        ME.clear_global_coordinate_manager()
        rec = ctx.decompress(data)
        wait_for_my_work(rec.device)
        ME.clear_global_coordinate_manager()
        return data, _key(rec)

    with FramePipeline(model, depth=2, own_streams=own_streams) as pipe:
        assert pipe.models[1] is not model
        assert all(a.data_ptr() == b.data_ptr() for a, b in zip(model.parameters(), pipe.models[1].parameters()))
        order = [0, 1, 2, 3, 3, 1, 0, 2, 2, 2, 1, 3]
        got = pipe.map(step, order)
    torch.cuda.synchronize()
    for i, (data, key) in zip(order, got):
        assert data == want[i][0], f'frame {i}: bytes differ in the pipeline'
        assert np.array_equal(key, want[i][1])


def test_a_failing_frame_does_not_poison_the_pipeline():
    model = _model()
    frames = _frames()
    data0 = model.compress(frames[0])
    ME.clear_global_coordinate_manager()

    def step(ctx, i):
        if i == 1:
            return ctx.decompress(b'\\x00\\x01garbage')          # a corrupt stream: the decoder raises
        return ctx.compress(frames[0])

    with FramePipeline(model, depth=2) as pipe:
        with pytest.raises(Exception):
            pipe.map(step, [0, 1, 2])
        assert pipe.map(step, [0, 2]) == [data0, data0]
    torch.cuda.synchronize()
