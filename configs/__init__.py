from .config import cfg, args  # noqa: F401  (reference import path: `from configs import cfg, args`)
