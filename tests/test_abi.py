"""CPU: the C-ABI library loads and exports exactly what include/occnerf_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'occnerf_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(occnerf_\w+)\s*\(', src)))


def test_header_symbols_are_exported():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, 'occnerf_amd', 'liboccnerf_hip.so'))
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f'{n} declared in occnerf_hip.h but not exported'


def test_python_binding_covers_the_header():
    from occnerf_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.lib()
    assert lib.occnerf_abi_version() == 3
    assert lib.occnerf_canonical_mlp_packed_floats() == 479812 + 0 or lib.occnerf_canonical_mlp_packed_floats() > 461568


def test_argument_errors_are_reported_without_a_gpu():
    from occnerf_amd import _lib
    lib = _lib.lib()
    # null tensors -> error code + message, no launch attempted
    rc = lib.occnerf_canonical_mlp(None, 128, None, None, None)
    assert rc != 0 and b'null' in lib.occnerf_last_error()
    rc = lib.occnerf_grad_total_variation(None, None, None, None, 0.0, 0, 0, 0, 0, 0.0, 0, 0, 0, None)
    assert rc != 0 and b'not implemented' in lib.occnerf_last_error()


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle (prompt rule 3)."""
    pkg = os.path.join(ROOT, 'occnerf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert 'liboccnerf_oracle' not in text, f
