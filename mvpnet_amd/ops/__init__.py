"""Host-side mirror of the reference's `mvpnet.ops` package (same function names, argument
meaning, defaults and autograd behaviour; reference: mvpnet/ops/*.py), running on
libmvp_hip.so through mvpnet_amd.ext.  GPU only -- no CPU fallback."""
from .fps import farthest_point_sample
from .ball_query import ball_query, ball_query_distance
from .group_points import group_points
from .knn_distance import knn_distance
from .interpolate import feature_interpolate
from .lifting import unproject, pixel_knn, lift_gather, lift, rotate_rows

__all__ = ['farthest_point_sample', 'ball_query', 'ball_query_distance', 'group_points', 'knn_distance',
           'feature_interpolate', 'unproject', 'pixel_knn', 'lift_gather', 'lift', 'rotate_rows']


def as_point_major(x, transpose):
    """(B,3,N) -> contiguous (B,N,3) when `transpose` (the reference wrappers' convention,
    e.g. mvpnet/ops/fps.py:28-30); otherwise just make it contiguous."""
    return (x.transpose(1, 2) if transpose else x).contiguous()
