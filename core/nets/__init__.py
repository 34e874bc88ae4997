from .create_network import create_network  # noqa: F401
