"""ball_query_distance_cuda: reference mvpnet/ops/cuda/ball_query_distance.cpp:7-15."""
import torch

from .. import _lib as L
from .ball_query_cuda import _check, grid_workspace


def ball_query_distance(query, key, radius, max_neighbors):
    """-> [int64 (B,N1,K), T (B,N1,K)]; padded distance slots are -1 (ball_query_distance_kernel.cu:171)."""
    _check(query, key)
    B, N1, _ = query.shape
    N2 = key.size(1)
    index = torch.empty((B, N1, int(max_neighbors)), dtype=torch.int64, device=query.device)
    distance = torch.empty((B, N1, int(max_neighbors)), dtype=query.dtype, device=query.device)
    ws = grid_workspace(query, key) if B * N1 > 0 else None
    if ws is not None:
        L.call('mvp_ball_query_grid_f32', query, L.ptr(query), L.ptr(key), B, N1, N2, float(radius), int(max_neighbors), L.ptr(index),
               L.ptr(distance), L.ptr(ws), ws.numel())
    else:
        L.call('mvp_ball_query_distance_' + L.suffix(query), query, L.ptr(query), L.ptr(key), B, N1, N2, float(radius),
               int(max_neighbors), L.ptr(index), L.ptr(distance))
    return index, distance
