"""ball_query_cuda: reference mvpnet/ops/cuda/ball_query.cpp:7-15."""
import os

import torch

from .. import _lib as L

# large float32 clouds go through the cell grid (csrc/ball_grid.hip: same rows, ~20x fewer pair tests); MVP_BALL_GRID=0: always the sweep kernel
BALL_GRID = os.environ.get('MVP_BALL_GRID', '1') != '0'


def grid_workspace(query, key):
    """uint8 scratch for mvp_ball_query_grid_f32, or None when the sweep kernel takes this shape."""
    if not (BALL_GRID and key.dtype == torch.float32):
        return None
    nbytes = int(L.lib().mvp_ball_query_grid_workspace(key.size(0), query.size(1), key.size(1)))
    return torch.empty(nbytes, dtype=torch.uint8, device=key.device) if nbytes > 0 else None


def _check(query, key):
    L.require_gpu(query, key)
    if query.dim() != 3 or key.dim() != 3 or query.size(2) != 3 or key.size(2) != 3:
        raise RuntimeError('query and key must be (batch_size, num_points, 3)')
    if query.size(0) != key.size(0):
        raise RuntimeError('Mismatched batch size: {} vs {}'.format(query.size(0), key.size(0)))
    if query.dtype != key.dtype:
        raise RuntimeError('query and key must have the same dtype')


def ball_query(query, key, radius, max_neighbors):
    """query (B,N1,3), key (B,N2,3) -> int64 (B,N1,max_neighbors)  (ball_query_kernel.cu:147-187)."""
    _check(query, key)
    B, N1, _ = query.shape
    N2 = key.size(1)
    index = torch.empty((B, N1, int(max_neighbors)), dtype=torch.int64, device=query.device)
    ws = grid_workspace(query, key) if B * N1 > 0 else None
    if ws is not None:
        L.call('mvp_ball_query_grid_f32', query, L.ptr(query), L.ptr(key), B, N1, N2, float(radius), int(max_neighbors), L.ptr(index), None,
               L.ptr(ws), ws.numel())
    else:
        L.call('mvp_ball_query_' + L.suffix(query), query, L.ptr(query), L.ptr(key), B, N1, N2, float(radius),
               int(max_neighbors), L.ptr(index))
    return index
