"""`from configs import cfg, args`: the reference builds its config singleton at import time
from sys.argv (configs/config.py:65-72); this does the same through occnerf_amd.config."""
import sys

from occnerf_amd.config import make_cfg

cfg, args = make_cfg(sys.argv[1:])
