"""fastpcc_amd/serving.py on the CPU: codec contexts share their weights, the pipeline keeps item order, hands exceptions to the
caller and gives every worker thread its own "global" coordinate manager."""
import threading
import time

import pytest
import torch

from fastpcc_amd import engine as ME
from fastpcc_amd.serving import FramePipeline, clone_context


class _Leaf(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.arange(6, dtype=torch.float32).reshape(2, 3))
        self.register_buffer('scale', torch.tensor([2.0]))
        self._overlap = {'cpu': object()}          # per-device state a context creates for itself
        self.cache = {}


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b = _Leaf(), _Leaf()


def test_contexts_share_weights_and_nothing_else():
    m = _Model().eval()
    twin = clone_context(m)
    assert twin is not m and twin.a is not m.a and twin.a.cache is not m.a.cache
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), twin.named_parameters()):
        assert n1 == n2 and p1.data_ptr() == p2.data_ptr()
    for (n1, b1), (n2, b2) in zip(m.named_buffers(), twin.named_buffers()):
        assert n1 == n2 and b1.data_ptr() == b2.data_ptr()
    assert twin.a._overlap == {} and m.a._overlap != {}          # the original keeps its own, the copy starts empty
    assert not twin.training
    with torch.no_grad():
        m.a.w[0, 0] = 41.0
    assert float(twin.a.w[0, 0]) == 41.0                         # one set of weights


def test_pipeline_order_exceptions_and_thread_local_manager():
    m = _Model().eval()
    seen = {}

    def fn(ctx, item):
        assert ME.global_coordinate_manager() is None            # nothing leaks in from another frame
        ME.set_global_coordinate_manager(ME.CoordinateManager(D=3))
        seen.setdefault(threading.current_thread().name, set()).add(id(ctx))
        time.sleep(0.005)                                        # long enough for both workers to take part
        if item == 5:
            raise ValueError('frame 5')
        return item * 10 + (0 if ctx is m else 1) * 0

    with FramePipeline(m, depth=2, device=torch.device('cpu')) as pipe:
        assert pipe.map(fn, range(5)) == [0, 10, 20, 30, 40]
        with pytest.raises(ValueError, match='frame 5'):
            pipe.map(fn, range(8))
        assert pipe.map(fn, [1, 2]) == [10, 20]                   # usable after a failed batch
    assert len(seen) == 2 and all(len(v) == 1 for v in seen.values())          # two workers, each with its own context
    assert ME.global_coordinate_manager() is None                # the caller's thread never saw the workers' managers
    with FramePipeline(m, depth=1, device=torch.device('cpu')) as pipe:
        ME.clear_global_coordinate_manager()
        assert pipe.map(lambda ctx, it: (ctx is m, it), [3]) == [(True, 3)]
    ME.clear_global_coordinate_manager()
    with pytest.raises(ValueError):
        FramePipeline(m, depth=0)


def test_named_stages_exclude_each_other_across_contexts():
    m = _Model().eval()
    inside = {'a': 0, 'b': 0}
    worst = {'a': 0, 'b': 0}
    guard = threading.Lock()

    def visit(pipe, name):
        with pipe.stage(name):
            with guard:
                inside[name] += 1
                worst[name] = max(worst[name], inside[name])
            time.sleep(0.003)
            with guard:
                inside[name] -= 1

    with FramePipeline(m, depth=2, device=torch.device('cpu')) as pipe:
        pipe.map(lambda ctx, it: (visit(pipe, 'a'), visit(pipe, 'b')), range(12))
        assert pipe.stage('a') is pipe.stage('a') and pipe.stage('a') is not pipe.stage('b')
    assert worst == {'a': 1, 'b': 1}                              # never two contexts in the same stage
    with FramePipeline(m, depth=1, device=torch.device('cpu')) as pipe:
        with pipe.stage('a'):                                     # one context: nothing to exclude
            with pipe.stage('a'):
                pass
