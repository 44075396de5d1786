"""PointNet++ (single-scale grouping) on the HIP ops: mirror of the reference's
`mvpnet.models.pn2` (modules.py:13-186, pn2ssg.py:22-137) -- same class names, constructor
arguments, data-dict keys (`points`, `feature` -> `seg_logit`) and `state_dict` keys.
All geometry (FPS, ball query, 3-NN, gathers, interpolation) runs in libmvp_hip.so.
"""
import torch
import torch.nn as nn

from .nn import SharedMLP, SharedMLPDO, batch_index_select, xavier_uniform
from . import ops


class QueryGrouper(nn.Module):
    """Ball query + grouping around centroids (modules.py:13-41)."""

    def __init__(self, radius, max_neighbors):
        super().__init__()
        assert radius > 0.0 and max_neighbors > 0
        self.radius, self.max_neighbors = radius, max_neighbors

    def forward(self, new_xyz, xyz, feature, use_xyz):
        with torch.no_grad():
            index = ops.ball_query(new_xyz, xyz, self.radius, self.max_neighbors)
        # neighbour coordinates relative to their centroid: (B,3,M,K)
        group_xyz = ops.group_points(xyz, index) - new_xyz.unsqueeze(-1)
        if feature is None:
            return group_xyz, group_xyz
        group_feature = ops.group_points(feature, index)
        if use_xyz:
            group_feature = torch.cat([group_feature, group_xyz], dim=1)  # features first, then xyz (:33)
        return group_feature, group_xyz

    def extra_repr(self):
        return 'radius={}, max_neighbors={}'.format(self.radius, self.max_neighbors)


class SetAbstraction(nn.Module):
    """FPS -> ball query -> grouped shared MLP -> max over neighbours (modules.py:44-113)."""

    def __init__(self, in_channels, mlp_channels, num_centroids, radius, max_neighbors, use_xyz):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = mlp_channels[-1]
        self.num_centroids, self.radius, self.max_neighbors, self.use_xyz = num_centroids, radius, max_neighbors, use_xyz
        if self.use_xyz or self.in_channels == 0:
            self.in_channels += 3
        self.mlp = SharedMLP(self.in_channels, mlp_channels, ndim=2, bn=True)
        self.grouper = None if num_centroids == 0 else QueryGrouper(radius, max_neighbors)

    def forward(self, xyz, feature=None):
        """xyz (B,3,N), feature (B,C,N) or None -> new_xyz (B,3,M), new_feature (B,C_out,M)."""
        if self.num_centroids == 0:  # one global group centred at the origin (modules.py:88-95)
            assert feature is not None
            new_xyz = xyz.new_zeros([xyz.size(0), 3, 1])
            group_feature = feature.unsqueeze(2)
            if self.use_xyz:
                group_feature = torch.cat([group_feature, xyz.unsqueeze(2)], dim=1)
        else:
            if self.num_centroids == -1:  # every point is a centroid
                new_xyz = xyz
            else:
                with torch.no_grad():
                    index = ops.farthest_point_sample(xyz, self.num_centroids)
                new_xyz = batch_index_select(xyz, index, dim=2)
            group_feature, _ = self.grouper(new_xyz, xyz, feature, use_xyz=self.use_xyz)
        new_feature = self.mlp(group_feature)
        return new_xyz, new_feature.max(dim=3)[0]

    def extra_repr(self):
        return 'num_centroids={}, radius={}, max_neighbors={}, use_xyz={}'.format(
            self.num_centroids, self.radius, self.max_neighbors, self.use_xyz)


class FeatureInterpolator(nn.Module):
    """Inverse-(squared-)distance interpolation from 3 nearest keys (modules.py:116-153)."""

    def __init__(self, num_neighbors, eps=1e-10):
        super().__init__()
        self.num_neighbors, self._eps = num_neighbors, eps

    def forward(self, query_xyz, key_xyz, query_feature, key_feature):
        with torch.no_grad():
            index, distance = ops.knn_distance(query_xyz, key_xyz, self.num_neighbors)
            inv = 1.0 / torch.clamp(distance, min=self._eps)  # weights from the SQUARED distance (:135-140)
            weight = inv / torch.sum(inv, dim=2, keepdim=True)
        interpolated = ops.feature_interpolate(key_feature, index, weight)
        if query_feature is None:
            return interpolated
        return torch.cat([interpolated, query_feature], dim=1)

    def extra_repr(self):
        return 'num_neighbors={}'.format(self.num_neighbors)


class FeaturePropagation(nn.Module):
    """Interpolate sparse features onto the dense level, concat the skip, shared MLP (modules.py:156-186)."""

    def __init__(self, in_channels, in_channels_prev, mlp_channels, num_neighbors):
        super().__init__()
        self.in_channels = in_channels + in_channels_prev
        self.out_channels = mlp_channels[-1]
        self.mlp = SharedMLP(self.in_channels, mlp_channels, ndim=1, bn=True)
        if num_neighbors == 0:
            self.interpolator = None
        elif num_neighbors == 3:
            self.interpolator = FeatureInterpolator(num_neighbors)
        else:
            raise ValueError('Expected value 3, but {} given.'.format(num_neighbors))

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature):
        if self.interpolator is None:  # broadcast a single global feature
            assert sparse_xyz.size(2) == 1 and sparse_feature.size(2) == 1
            new_feature = torch.cat([sparse_feature.expand(-1, -1, dense_xyz.size(2)), dense_feature], dim=1)
        else:
            new_feature = self.interpolator(dense_xyz, sparse_xyz, dense_feature, sparse_feature)
        return self.mlp(new_feature)


class PN2SSG(nn.Module):
    """4 x SetAbstraction, 4 x FeaturePropagation, dropout MLP, per-point classifier (pn2ssg.py:22-118)."""

    def __init__(self, in_channels, num_classes,
                 sa_channels=((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 256, 512)),
                 num_centroids=(2048, 512, 128, 32), radius=(0.1, 0.2, 0.4, 0.8), max_neighbors=(32, 32, 32, 32),
                 fp_channels=((256, 256), (256, 256), (256, 128), (128, 128, 128)), fp_neighbors=(3, 3, 3, 3),
                 seg_channels=(128,), dropout_prob=0.5, use_xyz=True):
        super().__init__()
        self.in_channels, self.num_classes, self.use_xyz = in_channels, num_classes, use_xyz
        n_sa = len(sa_channels)
        assert len(num_centroids) == n_sa and len(radius) == n_sa and len(max_neighbors) == n_sa
        assert len(fp_channels) == n_sa and len(fp_neighbors) == n_sa

        self.sa_modules = nn.ModuleList()
        c_in = in_channels
        for ch, m, r, k in zip(sa_channels, num_centroids, radius, max_neighbors):
            self.sa_modules.append(SetAbstraction(c_in, ch, m, r, k, use_xyz))
            c_in = ch[-1]

        skip = [0] + [ch[-1] for ch in sa_channels]  # the input feature is not used as a skip (pn2ssg.py:63-64)
        self.fp_modules = nn.ModuleList()
        c_in = skip[-1]
        for level, (ch, k) in enumerate(zip(fp_channels, fp_neighbors)):
            self.fp_modules.append(FeaturePropagation(c_in, skip[-2 - level], ch, k))
            c_in = ch[-1]

        self.mlp_seg = SharedMLPDO(fp_channels[-1][-1], seg_channels, ndim=1, bn=True, p=dropout_prob)
        self.seg_logit = nn.Conv1d(seg_channels[-1], num_classes, 1, bias=True)
        self.reset_parameters()

    def forward(self, data_batch):
        xyz = data_batch['points']
        feature = data_batch.get('feature', None)
        xyzs, feats = [xyz], [None]
        for sa in self.sa_modules:
            xyz, feature = sa(xyz, feature)
            xyzs.append(xyz)
            feats.append(feature)
        up = feats[-1]
        for level, fp in enumerate(self.fp_modules):
            up = fp(xyzs[-2 - level], xyzs[-1 - level], feats[-2 - level], up)
        return {'seg_logit': self.seg_logit(self.mlp_seg(up))}

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                xavier_uniform(m)
