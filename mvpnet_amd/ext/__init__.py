"""Drop-in replacements of the reference's six pybind11 extension modules.

Each module here has the NAME and the FUNCTION SIGNATURES of one of
`mvpnet.ops.{fps,ball_query,ball_query_distance,group_points,knn_distance,interpolate}_cuda`
(reference: mvpnet/ops/cuda/*.cpp), implemented over the C ABI of libmvp_hip.so.
`install(package)` registers them under another package name so that the reference's own
`mvpnet/ops/*.py` wrappers (`from . import fps_cuda`) import them unmodified
(see INTEGRATION.md).
"""
import importlib
import sys

NAMES = ['fps_cuda', 'ball_query_cuda', 'ball_query_distance_cuda', 'group_points_cuda', 'knn_distance_cuda',
         'interpolate_cuda']


def install(package='mvpnet.ops'):
    """Make `import <package>.<name>_cuda` resolve to the HIP-backed modules."""
    for name in NAMES:
        mod = importlib.import_module('mvpnet_amd.ext.' + name)
        sys.modules[package + '.' + name] = mod
        pkg = sys.modules.get(package)
        if pkg is not None:
            setattr(pkg, name, mod)
