from occnerf_amd.modules import MotionWeightVolumeDecoder  # noqa: F401
