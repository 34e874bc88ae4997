"""Plug-in seam of the reference (core/nets/create_network.py:5-15): the dotted module name in
cfg.network_module is resolved to a file relative to the working directory and must expose
`Network`.  importlib replaces the deprecated `imp.load_source`."""
import importlib.util
import os
import sys

from configs import cfg


def load_plugin(module, attr):
    path = module.replace('.', '/') + '.py'
    if not os.path.exists(path):                      # launched from another cwd: repo-relative
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), path)
    spec = importlib.util.spec_from_file_location(module, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault(module, mod)
    spec.loader.exec_module(mod)
    return getattr(mod, attr)


def create_network():
    return load_plugin(cfg.network_module, 'Network')()
