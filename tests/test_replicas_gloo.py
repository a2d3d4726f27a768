"""The N>1 path of bench.py on CPU: two gloo processes exercise rank discovery, the barrier, the max/sum aggregation that
produces the whole-job figure, frame sharding and the variable-length byte gather used for partition bitstreams."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from fastpcc_amd import replicas
    import torch.distributed as dist
    assert replicas.env_rank() == (rank, world, rank)
    replicas.init('gloo')
    dev = torch.device('cpu')
    replicas.barrier(dev)
    # rank r "codes" frames r, r+2, ... of 5; takes 1.0 + r seconds for 1000*(r+1) points each
    mine = replicas.frames_of_rank(list(range(5)), rank, world)
    elapsed, units = replicas.aggregate(1.0 + rank, 1000.0 * (rank + 1) * len(mine), dev)
    blobs = replicas.gather_bytes(bytes([rank + 1]) * (3 + 4 * rank) if rank else b'', dev)
    spread = replicas.rank_spread(10.0 + rank, dev)
    replicas.barrier(dev)
    q.put((rank, mine, elapsed, units, blobs, spread))
    dist.destroy_process_group()


def test_two_rank_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][1] == [0, 2, 4] and got[1][1] == [1, 3]
    for rank, mine, elapsed, units, blobs, spread in got:
        assert spread == {'ranks': 2, 'backend': 'gloo', 'min': 10.0, 'max': 11.0}   # what the bench line's rccl_ranks / per_rank_value carry
        assert elapsed == 2.0                                   # max over ranks
        assert units == 1000.0 * 3 + 2000.0 * 2                 # sum over ranks
        assert blobs == [b'', bytes([2]) * 7]                   # ragged, one of them empty


def test_single_process_is_identity():
    from fastpcc_amd import replicas
    assert replicas.aggregate(0.5, 7.0, torch.device('cpu')) == (0.5, 7.0)
    assert replicas.gather_bytes(b'abc', torch.device('cpu')) == [b'abc']
    assert replicas.rank_spread(3.0, torch.device('cpu')) == {'ranks': 1, 'backend': 'none', 'min': 3.0, 'max': 3.0}
    assert replicas.frames_of_rank('abcd', 0, 1) == list('abcd')


def test_cpulist_parsing_and_binding_is_a_noop_without_a_gpu(monkeypatch):
    """bind_to_device_numa_node: sysfs cpulist syntax; nothing happens (None) where the device or its topology cannot be read"""
    from fastpcc_amd import replicas
    assert replicas._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert replicas._parse_cpulist('') == []
    monkeypatch.setenv('FPCC_NUMA_BIND', '1')
    import os
    before = os.sched_getaffinity(0)
    assert replicas.bind_to_device_numa_node(0) is None or os.sched_getaffinity(0) <= before
    os.sched_setaffinity(0, before)
    monkeypatch.setenv('FPCC_NUMA_BIND', '0')
    assert replicas.bind_to_device_numa_node(0) is None
