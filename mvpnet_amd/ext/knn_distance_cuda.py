"""knn_distance_cuda: reference mvpnet/ops/cuda/knn_distance.cpp:8-15."""
import torch

from .. import _lib as L
from .ball_query_cuda import _check


def knn_distance(query, key, k):
    """query (B,N1,3), key (B,N2,3), k == 3 -> [int64 (B,N1,3), squared distance (B,N1,3)]
    (knn_distance_kernel.cu:154-196; k != 3 is rejected at :171)."""
    _check(query, key)
    if int(k) != 3:
        raise RuntimeError('Only support 3-NN.')
    B, N1, _ = query.shape
    N2 = key.size(1)
    if N2 < 3:
        raise RuntimeError('Expected num_key >= k, got {}'.format(N2))
    index = torch.empty((B, N1, 3), dtype=torch.int64, device=query.device)
    distance = torch.empty((B, N1, 3), dtype=query.dtype, device=query.device)
    from .ball_query_cuda import BALL_GRID
    nbytes = int(L.lib().mvp_knn3_grid_workspace(B, N1, N2)) if (BALL_GRID and query.dtype == torch.float32) else 0
    if nbytes > 0:  # many pairs: through the cell grid (csrc/ball_grid.hip), same index and distance
        ws = torch.empty(nbytes, dtype=torch.uint8, device=query.device)
        L.call('mvp_knn3_grid_f32', query, L.ptr(query), L.ptr(key), B, N1, N2, 1.0, L.ptr(index), None, L.ptr(distance), L.ptr(ws), nbytes)
    else:
        L.call('mvp_knn_distance_' + L.suffix(query), query, L.ptr(query), L.ptr(key), B, N1, N2, 3, L.ptr(index), L.ptr(distance))
    return index, distance
