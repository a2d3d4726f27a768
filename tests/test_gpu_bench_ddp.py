"""`bench.py --gpus 2` end to end as the driver launches it (torch.distributed.run, one process per rank), on ONE device over gloo: the
replica timing of the headline metric plus the cfg#5 DDP leg (global batch 8 split over the ranks, gradients all-reduced) must come
back in ONE JSON line from rank 0 -- a dry run of the multi-GPU code path (/root/reference/train.py:139,210-217,382-383), not a
measurement (FPCC_BENCH_ONE_DEVICE puts both ranks on cuda:0, FPCC_BENCH_BACKEND replaces RCCL, which refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device_report_replicas_and_the_ddp_step():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FPCC_BENCH_ONE_DEVICE='1', FPCC_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--ddp-steps', '3',
           '--resolution', '512']
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, proc.stdout[-2000:]                    # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 4 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert 'replicas x2' in rec['config']['parallelism']
    assert rec['roofline']['bound'] == 'mfma' and 0 < rec['roofline']['frac'] < 1
    ddp = rec['config']['secondary']['cfg5_training_ddp']
    assert 'error' not in ddp, ddp
    assert ddp['ranks'] == 2 and ddp['parallelism'] == 'ddp2' and ddp['steps'] == 3
    assert ddp['ms_per_step'] > 0 and ddp['clouds_per_s'] > 0
    assert ddp['exposed_allreduce_ms_per_step'] is not None and ddp['exposed_allreduce_ms_per_step'] >= 0
    assert ddp['gradient_bytes'] == 4 * ddp['parameters'] and ddp['parameters'] > 1e7
