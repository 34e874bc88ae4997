"""Sub-plug-in loaders with the reference's names (component_factory.py:3-26)."""
from core.nets.create_network import load_plugin


def load_positional_embedder(module_name):
    return load_plugin(module_name, 'get_embedder')


def load_canonical_mlp(module_name):
    return load_plugin(module_name, 'CanonicalMLP')


def load_mweight_vol_decoder(module_name):
    return load_plugin(module_name, 'MotionWeightVolumeDecoder')


def load_pose_decoder(module_name):
    return load_plugin(module_name, 'BodyPoseRefiner')


def load_non_rigid_motion_mlp(module_name):
    return load_plugin(module_name, 'NonRigidMotionMLP')
