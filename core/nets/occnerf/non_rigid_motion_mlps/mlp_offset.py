from occnerf_amd.modules import NonRigidMotionMLP  # noqa: F401
