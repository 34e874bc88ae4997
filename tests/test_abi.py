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
    assert lib.occnerf_abi_version() == 5
    assert lib.occnerf_canonical_mlp_packed_floats() == 479812 + 0 or lib.occnerf_canonical_mlp_packed_floats() > 461568


def test_argument_errors_are_reported_without_a_gpu():
    from occnerf_amd import _lib
    lib = _lib.lib()
    # null tensors -> error code + message, no launch attempted
    rc = lib.occnerf_canonical_mlp(None, 128, None, None, None)
    assert rc != 0 and b'null' in lib.occnerf_last_error()
    rc = lib.occnerf_grad_total_variation(None, None, None, None, 0.0, 0, 0, 0, 0, 0.0, 0, 0, 0, None)
    assert rc != 0 and b'not implemented' in lib.occnerf_last_error()


def test_reference_native_module_names_import():
    """the reference binds its operators as `import _gridencoder` (gridencoder/grid.py:9) and `import _shencoder`
    (shencoder/sphere_harmonics.py:9); both names resolve from the repo root, the first to the three pybind names of
    src/bindings.cpp:5-9 bound to the HIP library, the second to two by-name refusals."""
    import inspect
    import _gridencoder
    import _shencoder
    from occnerf_amd import ops
    assert _gridencoder.grid_encode_forward is ops.grid_encode_forward
    assert _gridencoder.grid_encode_backward is ops.grid_encode_backward
    assert _gridencoder.grad_total_variation is ops.grad_total_variation
    # positional order of the reference's calls (grid.py:55,83; shencoder.h:9-10)
    assert list(inspect.signature(_gridencoder.grid_encode_forward).parameters) == [
        'inputs', 'embeddings', 'offsets', 'outputs', 'B', 'D', 'Cc', 'L', 'S', 'H', 'dy_dx', 'gridtype', 'align_corners', 'interp']
    assert list(inspect.signature(_gridencoder.grid_encode_backward).parameters) == [
        'grad', 'inputs', 'embeddings', 'offsets', 'grad_embeddings', 'B', 'D', 'Cc', 'L', 'S', 'H', 'dy_dx', 'grad_inputs',
        'gridtype', 'align_corners', 'interp']
    assert list(inspect.signature(_shencoder.sh_encode_forward).parameters) == ['inputs', 'outputs', 'B', 'D', 'C', 'dy_dx']
    assert list(inspect.signature(_shencoder.sh_encode_backward).parameters) == ['grad', 'inputs', 'B', 'D', 'C', 'dy_dx',
                                                                                 'grad_inputs']


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle (prompt rule 3)."""
    pkg = os.path.join(ROOT, 'occnerf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, f
                assert 'liboccnerf_oracle' not in text, f


def test_product_and_benchmark_do_not_import_the_test_package():
    """bench.py, smoke(), run.py, train.py, the package and the tools must run without `tests/`: the seeded model
    and frame plumbing live in occnerf_amd/seeded.py, the oracle chain in oracle/chain.py.  (tools/debug_train.py is a debugging aid
    for the tests themselves.)"""
    import glob
    files = ['bench.py', '__graft_entry__.py', 'run.py', 'train.py'] + glob.glob(os.path.join(ROOT, 'occnerf_amd', '*.py')) + \
        [f for f in glob.glob(os.path.join(ROOT, 'tools', '*.py')) if not f.endswith('debug_train.py')] + \
        glob.glob(os.path.join(ROOT, 'oracle', '*.py')) + glob.glob(os.path.join(ROOT, 'core', '**', '*.py'), recursive=True)
    for f in files:
        text = open(f if os.path.isabs(f) else os.path.join(ROOT, f)).read()
        assert 'from tests' not in text and 'import tests' not in text, f


def test_synthetic_frame_source_types():
    """The frame source behind core/data/create_dataset.py: frame counts per type as the reference's datasets define them
    (tpose 1, allview 23 = allview.py:69, progress <= 300 = create_dataset.py:40-42) and the per-frame dict's keys."""
    from occnerf_amd.sequence import SyntheticFrames
    assert len(SyntheticFrames('tpose', img_size=16)) == 1
    assert len(SyntheticFrames('allview', img_size=16)) == 23
    assert len(SyntheticFrames('progress', img_size=16, render_frames=1000)) == 300
    assert len(SyntheticFrames('movement', img_size=16, render_frames=7)) == 7
    src = SyntheticFrames('movement', img_size=16, render_frames=3, device_rays=False)
    batches = list(src)
    assert len(batches) == 3
    for k in ('rays', 'near', 'far', 'ray_mask', 'dst_Rs', 'dst_Ts', 'cnl_gtfms', 'motion_weights_priors', 'dst_posevec',
              'cnl_bbox_min_xyz', 'cnl_bbox_scale_xyz', 'bgcolor', 'img_width', 'frame_name'):
        assert k in batches[0], k
    assert batches[0]['rays'].shape[0] == 1 and batches[0]['rays'].shape[1] == 2          # leading batch dimension
    dev = SyntheticFrames('freeview', img_size=16, render_frames=2, device_rays=True)
    b = next(iter(dev))
    assert 'rays' not in b and 'camera_K' in b and 'dst_bbox_min' in b
