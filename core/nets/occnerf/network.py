"""cfg.network_module = 'core.nets.occnerf.network' resolves here (drop-in seam)."""
from occnerf_amd.network import Network  # noqa: F401
