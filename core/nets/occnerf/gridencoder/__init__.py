from occnerf_amd.gridencoder import GridEncoder  # noqa: F401
