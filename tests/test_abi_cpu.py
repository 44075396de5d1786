"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what
include/mvp_hip.h declares; the host mirror refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from tests.conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'mvp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mvp_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    from mvpnet_amd import _lib
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 26
    for name in names:
        assert hasattr(lib, name), 'libmvp_hip.so does not export ' + name
    assert sorted(_lib.EXPORTS) == names, 'python signature table and header disagree'
    assert b'gfx950' in lib.mvp_version()
    assert lib.mvp_strerror(-1).startswith(b'invalid argument')


def test_signature_table_matches_the_header_argument_counts():
    """Every ctypes signature in mvpnet_amd/_lib.py has as many entries as the header's prototype has parameters (stream included):
    ctypes accepts extra positional arguments silently, so a short table would go unnoticed."""
    import re
    from mvpnet_amd import _lib
    text = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'mvp_hip.h')).read(), flags=re.S)
    for name, sig in _lib._SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;]*?)\)\s*;', text, flags=re.S)
        assert m, name
        n = len([a for a in m.group(1).split(',') if a.strip()])
        assert n == len(sig), (name, n, len(sig))


def test_argument_errors_do_not_launch():
    """Precondition failures return MVP_E* before any HIP call (safe without a GPU)."""
    import ctypes
    from mvpnet_amd import _lib
    lib = _lib.lib()
    dummy = ctypes.c_void_p(16)
    assert lib.mvp_fps_f32(None, 1, 8, 3, 4, dummy, None) == -3                    # MVP_ENULL
    assert lib.mvp_fps_f32(dummy, 1, 8, 3, 9, dummy, None) == -1                   # N >= M   (fps_kernel.cu:156)
    assert lib.mvp_fps_f32(dummy, 1, 8, 4, 4, dummy, None) == -1                   # D in {2,3}
    assert lib.mvp_knn_distance_f32(dummy, dummy, 1, 4, 8, 5, dummy, dummy, None) == -2   # k must be 3 (:171)
    assert lib.mvp_knn_distance_f32(dummy, dummy, 1, 4, 2, 3, dummy, dummy, None) == -1   # N2 >= k
    assert lib.mvp_pixel_knn_bruteforce_f32(dummy, dummy, dummy, 1, 8, 8, 9, dummy, None, None) == -1


@pytest.mark.parametrize('fn', ['fps', 'ball', 'knn', 'group', 'interp'])
def test_no_cpu_fallback(fn):
    import mvpnet_amd.ops as ops
    x = torch.rand(1, 3, 16)
    with pytest.raises(RuntimeError):
        if fn == 'fps':
            ops.farthest_point_sample(x, 4)
        elif fn == 'ball':
            ops.ball_query(x, x, 0.1, 4)
        elif fn == 'knn':
            ops.knn_distance(x, x, 3)
        elif fn == 'group':
            ops.group_points(x, torch.zeros(1, 2, 2, dtype=torch.long))
        else:
            ops.feature_interpolate(x, torch.zeros(1, 2, 3, dtype=torch.long), torch.rand(1, 2, 3))


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, 'mvpnet_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), f
                assert 'mvp_oracle' not in text and 'c_oracle' not in text, f


def test_thread_local_precision_scope():
    """mvp_mlp_precision_scope: a thread-local override of the contraction precision (nothing process-wide is written): another thread
    keeps seeing the process default while this one is inside a scope."""
    import threading
    from mvpnet_amd import _lib as L
    lib = L.lib()
    default = lib.mvp_get_mlp_precision()
    seen = {}
    with L.mlp_precision('fp32'):
        assert lib.mvp_get_mlp_precision() == 0
        t = threading.Thread(target=lambda: seen.setdefault('other', lib.mvp_get_mlp_precision()))
        t.start()
        t.join()
        with L.mlp_precision('bf16x3', backward='bf16x6'):
            assert lib.mvp_get_mlp_precision() == 3 and lib.mvp_get_mlp_precision_backward() == 6
        assert lib.mvp_get_mlp_precision() == 0
    assert seen['other'] == default and lib.mvp_get_mlp_precision() == default
    assert lib.mvp_mlp_precision_scope(5, -1) == -1  # MVP_EINVAL
    with L.mlp_precision('bf16', backward='bf16'):  # plain bf16 operands, one product (opt-in)
        assert lib.mvp_get_mlp_precision() == 1 and lib.mvp_get_mlp_precision_backward() == 1
    assert lib.mvp_get_mlp_precision() == default
