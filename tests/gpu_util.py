"""Helpers shared by the GPU parity tests: the seeded network together with its oracle-side model context, golden
frames on the device.  (The model / frame plumbing itself lives in occnerf_amd/seeded.py, the CPU render assembled from
the oracle's stages in oracle/chain.py.)"""
import os

import numpy as np
import torch

from occnerf_amd import seeded
from occnerf_amd.seeded import FRAME_KEYS  # noqa: F401
from oracle.chain import golden_frame, per_frame_cpu, stagewise_oracle_render  # noqa: F401
from tests import util


def build_network(seed=0, amplify=False, S=128, non_rigid=False, device='cuda:0', mlp_precision='fp32'):
    """(Network with the seeded checkpoint, the oracle-side model context of the same checkpoint)."""
    ctx = util.model_context(seed, amplify)
    net = seeded.build_network(seed, amplify, S=S, non_rigid=non_rigid, device=device, mlp_precision=mlp_precision,
                               state_dict=ctx['sd'])
    return net, ctx


def frame_to_device(g_or_frame, device):
    frame = golden_frame(g_or_frame) if 'meta.S' in g_or_frame else g_or_frame
    return seeded.frame_to_device(frame, device)

DEV = 'cuda:0'


def same(got, want, name):
    """Bit-exact comparison with a useful failure message."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    bad = got != want
    if bad.any():
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        i = np.unravel_index(np.argmax(diff), diff.shape)
        raise AssertionError(f'{name}: {int(bad.sum())}/{bad.size} entries differ, max |diff| = {diff.max():.3e} '
                             f'at {i}: got {got[i]!r} want {want[i]!r}')


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def _dev_model(ctx, ops):
    """Device-side constants the way Network._context builds them."""
    base = T(ctx['point_base'])
    normals = T(ctx['normals'])
    sets = [np.arange(base.shape[0])] + [np.asarray(f) for f in ctx['fps']]
    rows, imap, begin = [], [], [0]
    for idx in sets:
        pts = ctx['point_base'][idx]
        pad = (-len(idx)) % 4
        rows.append(np.concatenate([pts, np.full((pad, 3), np.inf, np.float32)]))
        imap.append(np.concatenate([idx, np.zeros(pad, idx.dtype)]))
        begin.append(begin[-1] + len(idx) + pad)
    p4 = np.concatenate(rows)
    p4 = np.concatenate([p4, np.zeros((p4.shape[0], 1), np.float32)], 1)
    seed = [int(l + 1 < len(sets) and set(sets[l + 1].tolist()) <= set(sets[l].tolist()))
            for l in range(len(sets))]
    return {'base': base, 'normals': normals, 'unit': ops.unit_normals(normals), 'points': T(p4),
            'imap': T(np.concatenate(imap).astype(np.int32)), 'begin': begin, 'seed': seed,
            'b32': float(np.float32(ctx['bound'])),
            'tb32': float(np.float32(2 * np.float64(ctx['bound']))),
            'emb': T(ctx['embeddings']), 'off': T(ctx['offsets'])}


def _clusters(ctx):
    from occnerf_amd import geometry
    sets = [np.arange(len(ctx['point_base']))] + [np.asarray(f) for f in ctx['fps']]
    cl = geometry.build_knn_clusters(ctx['point_base'], sets)
    return {k: (T(v) if k in ('points', 'index_map', 'centers', 'ranges', 'radius', 'group_centers', 'group_ranges', 'group_radius') else v)
            for k, v in cl.items()}


def stagewise_table(ctx, oracle):
    """The per-point feature table [P,35] of a model context (oracle side)."""
    kb, sdf = oracle.point_sdf(ctx['point_cloud'], ctx['point_base'], ctx['normals'])
    return oracle.point_table(kb, sdf, ctx['point_cloud'], ctx['bound'], ctx['embeddings'], ctx['offsets'], ctx['S'], ctx['H'])


def _torchrun(script_args, nproc, timeout=900, extra_env=None):
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    import json
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert lines, res.stdout[-2000:] + res.stderr[-2000:]
    return json.loads(lines[-1])
