"""MI355X: the RCCL branch of the sharded renderer on the hardware there is -- a ONE-rank `nccl` process group.
Kept in a file of its own that sorts LAST: every other GPU test has run by the time a communicator is created (the suite is
run with -x; RCCL's bootstrap is the one piece here that depends on the box's network stack)."""
import os

import pytest
import torch

from tests.gpu_util import _torchrun

pytestmark = pytest.mark.gpu


def test_one_rank_rccl_group_runs_the_gather_path():
    """VERDICT r03 #1a: the RCCL collective executes on the hardware there is.  A ONE-rank `nccl` process group with
    force_collective takes the N > 1 branch of ShardedRenderer (occnerf_amd/parallel.py: Morton-block plan + checksum
    all-gather, padded device send buffer, asynchronous dist.gather into the list-of-views receive buffer, work.wait() stream
    ordering, un-permutation) -- what replaces the reference's DataParallel scatter/gather (network.py:68-72,142-146).
    Three pipelined frames (host frames, cost-aware plans) and three movement frames (device rays, named camera, cached
    plan) bit-identical to single=True; bench.py reports the leg with backend nccl."""
    got = _torchrun(['tools/sharded_check.py', '--force-collective'], 1)
    assert got['world_size_formed'] == 1 and got['backend'] == 'nccl' and got['collective'], got
    assert got['gathers_issued'] == 3 and got['plans_verified'] == 3, got
    assert got['bit_identical'] and got['max_abs_diff'] == 0.0, got
    got = _torchrun(['tools/sharded_check.py', '--force-collective', '--kind', 'movement', '--frames', '3'], 1)
    assert got['gathers_issued'] == 3 and got['plans_verified'] >= 1 and got['bit_identical'], got
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                          '--only', 'rccl_world1'], cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    # ONE line on stdout, the JSON line: RCCL's version banner (printed through C stdio at communicator creation) and any
    # other library output go to stderr
    assert len([l for l in res.stdout.splitlines() if l.strip()]) == 1, res.stdout[-1000:]
    line = json.loads(res.stdout.strip())
    leg = line['rccl_world1']
    assert leg.get('backend') == 'nccl' and leg['world_size_formed'] == 1 and leg['collective'], leg
    assert leg['gathers_issued'] >= 4 and leg['plans_verified'] == 1 and leg['bit_identical_to_headline'], leg


def test_run_py_through_a_one_rank_rccl_group(tmp_path):
    """`run.py --type movement` (device rays, named camera, one frame of lag, device image assembly, PNG writer) with the
    sharded renderer's N > 1 branch over a ONE-rank `nccl` group (OCC_FORCE_COLLECTIVE=1 under the launcher) writes the
    same PNG bytes as the plain single-process run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = ['--cfg', os.path.join(root, 'configs/occnerf/synthetic/occnerf.yaml'), '--type', 'movement', 'render_frames', '3',
           'render_size', '256', 'N_samples', '64']
    one, rccl = tmp_path / 'one', tmp_path / 'rccl'
    one.mkdir()
    rccl.mkdir()
    env = {**{k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}, 'PYTHONPATH': root,
           'HSA_ENABLE_IPC_MODE_LEGACY': '0'}
    subprocess.check_call([sys.executable, os.path.join(root, 'run.py')] + cli, cwd=str(one), env=env, timeout=600)
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    subprocess.check_call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr',
                           '127.0.0.1', '--master-port', str(port), os.path.join(root, 'run.py')] + cli, cwd=str(rccl),
                          env={**env, 'OCC_FORCE_COLLECTIVE': '1'}, timeout=600)
    sub = os.path.join('experiments', 'occnerf', 'synthetic', 'capsule_body', 'occnerf', 'seeded', 'movement')
    names = sorted(os.listdir(one / sub))
    assert len(names) == 3 and names == sorted(os.listdir(rccl / sub))
    for n in names:
        assert (one / sub / n).read_bytes() == (rccl / sub / n).read_bytes(), n


