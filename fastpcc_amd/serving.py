"""Several frames in flight on one GPU.

One frame at a time leaves the GPU idle while the host does the serial parts of the format -- the single-state rANS tail of
`compress`, the D2H -> host decode -> H2D chain of every occupancy level in `decompress` (3.7 of 22.3 ms per frame in round 3) --
and every idle stretch also drops the shader clock, which then takes milliseconds of uninterrupted load to come back
(profiles/r03/clock_ramp.md).  Frames are independent, so a serving process keeps `depth` of them in flight:

  * one worker thread per frame slot, each with its OWN codec context -- a copy of the model's modules (derived-weight caches,
    coder pool, pinned staging buffers, side stream) over the SAME parameter tensors -- and its own "global" coordinate manager
    (engine.py keeps that per thread);
  * all workers enqueue on the SAME HIP stream: kernels of different frames never overlap (per-kernel timings stay what they are
    alone; no CU is split between two launches), the stream is simply never empty -- while one frame waits for the host the
    other frame's launches run;
  * a worker waits for ITS OWN work through events (never a stream or device synchronise, which would also wait for what the other
    frames queued behind it); the blocking calls (event waits, the coder pool's wait, ctypes calls) release the GIL.

Nothing here changes a byte of a stream: a context runs exactly the single-frame code.  The reference has no counterpart (its test
loop codes one frame at a time, test.py); this is what "whole-job throughput" means for a stream of frames on one MI355X.
"""
import contextlib
import copy
import os
import queue
import sys
import threading
from typing import Callable, List, Optional, Sequence

import torch

from . import engine as ME


def clone_context(model: torch.nn.Module) -> torch.nn.Module:
    """a second context of the same codec: every module copied (so that per-module caches, coder pools and staging buffers are
    separate) over the original's parameter and buffer objects (one set of weights in HBM)"""
    # per-device overlap state (coder pool threads, side stream, pinned flags) is created lazily by every context for itself
    stash = [(m, m._overlap) for m in model.modules() if hasattr(m, '_overlap')]
    for m, _ in stash:
        m._overlap = {}
    try:
        twin = copy.deepcopy(model)
    finally:
        for m, st in stash:
            m._overlap = st
    # the copy's modules hold the ORIGINAL's parameter and buffer objects (weight tying, not just shared storage): one tensor object
    # means one version counter, so caches keyed on (storage, version) -- hipops.packed_weights -- serve both contexts from one entry
    # instead of re-packing a layer whenever the contexts alternate
    for m_src, m_dst in zip(model.modules(), twin.modules()):
        for name, p in m_src._parameters.items():
            if p is not None and name in m_dst._parameters:
                m_dst._parameters[name] = p
        for name, b in m_src._buffers.items():
            if b is not None and name in m_dst._buffers:
                m_dst._buffers[name] = b
    return twin.eval() if not model.training else twin


def wait_for_my_work(device: Optional[torch.device] = None) -> None:
    """block until everything THIS thread has enqueued on its current stream is done (an event, not a stream synchronise)"""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ev.synchronize()


class FramePipeline:
    """`depth` codec contexts driven by `depth` worker threads; `map(fn, items)` runs fn(context_model, item) for every item, at
    most `depth` at a time, results in item order.  depth == 1 runs in the calling thread."""

    def __init__(self, model: torch.nn.Module, depth: int = 2, device: Optional[torch.device] = None, own_streams: bool = False):
        """own_streams: every context enqueues on a stream of its own instead of the caller's current one -- kernels of different
        frames then run side by side (the latency-sized launches of the coarse levels beside another frame's large ones: more
        frames per second, but a launch's duration is no longer its own).  Contexts share weights and the caches derived from
        them: run one frame on context 0 and synchronise before using the others, so that every cached copy exists."""
        if depth < 1:
            raise ValueError('depth >= 1')
        self.depth = depth
        self.own_streams = own_streams and depth > 1
        if device is None:                                  # (the integer codec holds buffers only)
            first = next(model.parameters(), None)
            device = (first if first is not None else next(model.buffers())).device
        self.device = device
        self.models: List[torch.nn.Module] = [model] + [clone_context(model) for _ in range(depth - 1)]
        self._jobs: 'queue.Queue' = queue.Queue()
        self._threads: List[threading.Thread] = []
        self._closed = False
        self._stages: dict = {}
        self._stages_guard = threading.Lock()
        if depth > 1:
            # the host-side chains of two frames interleave at the interpreter's switch interval: a thread that returns from a blocking
            # wait (an event, the coder pool) gets the interpreter back only after that interval when the other thread is busy
            # enqueueing -- CPython hands the lock over on request only, a thread that releases it around a C call usually has it
            # back before the waiter wakes.  A frame has ~30 such returns on its critical path (every occupancy level of the decoder):
            # at the default 5 ms the pipeline would crawl, at 0.2 ms it was 44 Mpoints/s on a slower host, at 0.1 ms 49.5 (as at
            # 0.05 ms; 0.02 ms costs more in switches than it buys: tools/r04/s47.sh, s49.sh).  FPCC_SWITCH_INTERVAL overrides.
            self._old_switch = sys.getswitchinterval()
            sys.setswitchinterval(min(self._old_switch, float(os.environ.get('FPCC_SWITCH_INTERVAL', '1e-4'))))
            for i in range(depth):
                t = threading.Thread(target=self._worker, args=(self.models[i],), name=f'fpcc-frame-{i}', daemon=True)
                t.start()
                self._threads.append(t)

    def _worker(self, model: torch.nn.Module) -> None:
        if self.device.type == 'cuda':
            torch.cuda.set_device(self.device)
            if self.own_streams:
                torch.cuda.set_stream(torch.cuda.Stream(device=self.device))
        while True:
            job = self._jobs.get()
            if job is None:
                return
            fn, item, out, idx, done = job
            try:
                out[idx] = (True, fn(model, item))
            except BaseException as e:                   # noqa: BLE001 -- handed to the caller of map()
                out[idx] = (False, e)
            finally:
                ME.clear_global_coordinate_manager()
                done.release()

    def stage(self, name: str):
        """`with pipe.stage('compress'): ...` -- stages of the same name exclude each other across the contexts.  Two frames that
        run the SAME stage at the same time want the same resource at the same time: two decodes both wait for the host while the
        GPU idles, two encodes queue their launches alternately and reach their host tails together (measured: frames in lock-step
        take 24 ms each, frames in opposite phase 19 ms; which of the two a free-running pipeline falls into is chance).  With the
        stages named, a host-paced stage of one frame always runs beside a GPU-heavy stage of the other.  depth 1: no-op."""
        if self.depth == 1:
            return contextlib.nullcontext()
        with self._stages_guard:
            lock = self._stages.get(name)
            if lock is None:
                lock = self._stages[name] = threading.Lock()
        return lock

    def map(self, fn: Callable, items: Sequence) -> list:
        items = list(items)
        if self.depth == 1:
            return [fn(self.models[0], it) for it in items]
        out: list = [None] * len(items)
        done = threading.Semaphore(0)
        for idx, it in enumerate(items):
            self._jobs.put((fn, it, out, idx, done))
        for _ in items:
            done.acquire()
        res = []
        for ok, v in out:
            if not ok:
                raise v
            res.append(v)
        return res

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join(timeout=10)
        if self.depth > 1:
            sys.setswitchinterval(self._old_switch)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
