from occnerf_amd.modules import BodyPoseRefiner  # noqa: F401
