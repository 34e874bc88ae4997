"""Helpers shared by the GPU parity tests: the seeded network together with its oracle-side model context, golden
frames on the device.  (The model / frame plumbing itself lives in occnerf_amd/seeded.py, the CPU render assembled from
the oracle's stages in oracle/chain.py.)"""
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
