from occnerf_amd.canonical_mlp import CanonicalMLP  # noqa: F401
