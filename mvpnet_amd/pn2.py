"""PointNet++ (single-scale grouping) on the HIP ops: mirror of the reference's
`mvpnet.models.pn2` (modules.py:13-186, pn2ssg.py:22-137) -- same class names, constructor
arguments, data-dict keys (`points`, `feature` -> `seg_logit`) and `state_dict` keys.
All geometry (FPS, ball query, 3-NN, gathers, interpolation) runs in libmvp_hip.so.

Internally activations are channels-last rows ((B,N,C): mvpnet_amd/rows.py, csrc/rows.hip): the
reference's (B,C,M,K) tensors, its transposes and its separate conv / BN / ReLU / max kernels are
replaced by coalesced row gathers, one row-major GEMM per layer and fused BN+ReLU(+max) kernels.
The public `forward` signatures still take and return the reference's channel-major tensors.
"""
import os

import torch
import torch.nn as nn

from .nn import SharedMLP, SharedMLPDO, batch_index_select, xavier_uniform
from . import ops
from . import rows as R
from . import _lib as L


# launch shape of the sampling kernels of a TRAINING step's prefetched geometry (1: one wave per SIMD -- beside the backward pass the
# sampler leaves issue slots to the kernels it shares CUs with; 0: the fastest chain)
TRAIN_FPS_SHAPE = int(os.environ.get('MVP_TRAIN_FPS_SHAPE', '1'))
# Sampling a cloud that is itself a sampling result, in sampling order, returns 0, 1, 2, ... (PN2SSG._centroid_run): the deeper levels'
# centroids are prefixes of the first level's.  0 = sample every level (A/B switch; same coordinates either way).
FPS_PREFIX = os.environ.get('MVP_FPS_PREFIX', '1') != '0'
# The whole geometry plan of a network in ONE library call (mvp_pn2_plan_f32) where its shape allows; 0 = level by level from Python (A/B switch)
NATIVE_PLAN = os.environ.get('MVP_NATIVE_PLAN', '1') != '0'
# The segmentation head's SharedMLPDO as the last layer of the last feature-propagation level's chain (it is that level's only consumer);
# 0 = two chains with the level's output activation in between (A/B switch)
MERGE_HEAD = os.environ.get('MVP_MERGE_HEAD', '1') != '0'
HEAD_HANDOVER = os.environ.get('MVP_HEAD_HANDOVER', '1') != '0'  # (A/B: the head's ReLU / dropout mask + column sums in the logit layer's input gradient)


def centroid_levels(xyz, index, counts):
    """xyz (B,N,D), index (B,M) int64 = farthest_point_sample(xyz, M), counts[0] = M >= counts[1] >= ... -> [xyz[b, index[b, :c]] for c in
    counts], each (B,c,D) contiguous: the centroids of a chain of sampling levels from ONE gather launch (mvp_fps_centroid_levels_f32)."""
    B, N, D = xyz.shape
    M = index.size(1)
    assert counts[0] == M and all(0 < b <= a for a, b in zip(counts, counts[1:])) and len(counts) <= 8
    if xyz.is_cuda and xyz.dtype == torch.float32:
        import ctypes
        xyz, index = xyz.contiguous(), index.contiguous()
        outs = [torch.empty((B, c, D), dtype=torch.float32, device=xyz.device) for c in counts]
        L.call('mvp_fps_centroid_levels_f32', xyz, L.ptr(xyz), L.ptr(index), B, N, D, M, len(counts), (ctypes.c_int64 * len(counts))(*counts),
               (ctypes.c_void_p * len(counts))(*[o.data_ptr() for o in outs]))
        return outs
    g = torch.gather(xyz, 1, index.unsqueeze(-1).expand(-1, -1, D))
    return [g if c == M else g[:, :c].contiguous() for c in counts]


def _has_hooks(module):
    """forward (pre-)hooks on `module` or any of its children: a caller that observes a module's output must get THAT module's output, so
    chains are only merged across module boundaries nobody is watching"""
    return any(m._forward_hooks or m._forward_pre_hooks for m in module.modules())


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

    def forward_rows(self, new_xyz, xyz, feature, index=None):
        """new_xyz (B,M,3), xyz (B,N,3), feature (B,N,C) or None -> (B,M,K,round4(C+3)) rows
        [feature | xyz - centre | 0-pad]  (same values as forward(), features first then xyz)."""
        if index is None:
            with torch.no_grad():
                index = ops.ball_query(new_xyz, xyz, self.radius, self.max_neighbors, transpose=False)
        return R.group_rows(feature, xyz, new_xyz, index)

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

    def centroids(self, xyz, fps_shape=None):
        """Farthest point sampling -> centroid coordinates (the sequential part of the geometry).  fps_shape: launch shape of the
        sampling kernel for this call (include/mvp_hip.h: mvp_fps_shape_f32)."""
        with torch.no_grad():
            if self.num_centroids == -1:
                return xyz
            index = ops.farthest_point_sample(xyz, self.num_centroids, transpose=False, shape=fps_shape)
            return torch.gather(xyz, 1, index.unsqueeze(-1).expand(-1, -1, 3))

    def neighbours(self, new_xyz, xyz, with_csr=False):
        """Ball query [+ with_csr: the transposed index (offsets, slots) the backward of the grouping gathers through, and -- for levels the
        training-mode fused path can take (K = 32, fp32) -- the geometry-only sums of its per-point first-layer passes (R.geom_sums)]."""
        with torch.no_grad():
            ball = ops.ball_query(new_xyz, xyz, self.radius, self.max_neighbors, transpose=False)
            if with_csr:
                csr = R.build_csr(ball, xyz.size(1))
                if R.SA_TRAIN_FUSED and self.max_neighbors == 32 and xyz.dtype == torch.float32 and len(self.mlp) == 3 and \
                        R.sa_level_train_widths_ok(*(l.conv.weight.size(0) for l in self.mlp)):
                    csr = csr + R.geom_sums(csr[0], csr[1], xyz.contiguous(), new_xyz.contiguous(), 32)
                return (ball,) + csr
        return (ball,)

    def geometry(self, xyz, with_csr=False):
        """Everything of this layer that depends on coordinates only: centroids + neighbour index [+ transposed index]."""
        new_xyz = self.centroids(xyz)
        return (new_xyz,) + self.neighbours(new_xyz, xyz, with_csr)

    def forward_rows(self, xyz, feature=None, geometry=None):
        """xyz (B,N,3), feature (B,N,C) or None -> new_xyz (B,M,3), new_feature (B,M,C_out)."""
        B, N, _ = xyz.shape
        use_feature = feature is not None
        if self.num_centroids == 0:  # one global group centred at the origin (modules.py:88-95)
            assert use_feature
            new_xyz = xyz.new_zeros([B, 1, 3])
            x = torch.cat([feature, xyz], dim=2) if self.use_xyz else feature
            if x.size(2) % 4:
                x = torch.nn.functional.pad(x, (0, 4 - x.size(2) % 4))
            if N <= 255:  # the fused BN + ReLU + max kernel stores its arg-max in one byte
                return new_xyz, R.shared_mlp_rows(x.reshape(B * N, -1), self.mlp, K=N).view(B, 1, -1)
            return new_xyz, R.shared_mlp_rows(x.reshape(B * N, -1), self.mlp).view(B, N, -1).max(dim=1, keepdim=True)[0]
        geometry = self.geometry(xyz) if geometry is None else geometry
        new_xyz, ball = geometry[0], geometry[1]
        csr = tuple(geometry[2:4]) if len(geometry) >= 4 else None
        csr_geo = tuple(geometry[2:6]) if len(geometry) >= 6 else csr   # (+ the geometry sums of the fused training path)
        M, K = new_xyz.size(1), self.max_neighbors
        ps = R.parts(self.mlp)
        q0 = ps[0]
        w0 = q0.w
        c1 = w0.size(0)
        # the linear-first factorisation hands shared_mlp_rows a chain whose first layer is already applied: only its fused
        # path (R.mlp_chain_is_fused) can take that
        if (self.use_xyz or not use_feature) and R.mlp_chain_is_fused(self.mlp, ps=ps) and K <= 255:
            # The first shared-MLP layer is linear, so its feature columns commute with the grouping:
            #   W1.[f(idx) | xyz(idx) - c] = (W1f.f)(idx) + W1xyz.(xyz(idx) - c)
            # -> the 1x1 conv over the feature runs on the N points instead of the M*K = 8N grouped rows, and the
            #    grouped tensor has C_1 instead of C+3 columns.  The 3 coordinate columns are applied to the difference
            #    inside the grouping kernel (same operation order as modules.py:27 + conv; no cancellation).
            zf = None
            # one gradient buffer for the weight's two column groups (feature columns here, coordinate columns in the grouping kernel)
            sink = R.WeightGradSink(w0, 2) if (use_feature and torch.is_grad_enabled() and w0.requires_grad) else None
            if use_feature:
                cf = feature.size(2)
                pad = (-cf) % 4
                f = torch.nn.functional.pad(feature, (0, pad)) if pad else feature
                zf = R.linear_rows(f.reshape(B * N, -1), w0, cols=(0, cf), sink=sink).view(B, N, c1)
            bn_training = q0.bn.training
            if not bn_training and not torch.is_grad_enabled():
                # inference: gather -> 3 layers -> max in ONE kernel, nothing between the gathered rows and (B,M,C_3) touches HBM
                fused = R.sa_fused_eval(zf, xyz, new_xyz, ball, self.mlp)
                if fused is not None:
                    return new_xyz, fused
            if bn_training and R.sa_level_train_ok(zf, self.mlp, K):
                # training: the whole level without any (B,M,K,C) tensor -- every pass re-creates the ball's rows from zf (csrc/sa_train.hip)
                return new_xyz, R.sa_level_train(zf, xyz, new_xyz, ball, self.mlp, csr=csr_geo, sink=sink)
            # (B,M,K,C_1): conv output of layer 1; in training also its batch statistics AND the BatchNorm finalize, from the same call
            y1 = R.group_lin_rows(zf, xyz, new_xyz, w0, ball, want_stat=q0.bn if bn_training else False, csr=csr, sink=sink)
            stat1 = None
            if bn_training:
                y1, stat1 = y1[0], (y1[1], y1[2])
            new_feature = R.shared_mlp_rows(y1.view(B * M * K, c1), self.mlp, K=K, first_done=True, first_stat=stat1)
            return new_xyz, new_feature.view(B, M, -1)
        if use_feature and feature.size(2) % 4:
            feature = torch.nn.functional.pad(feature, (0, 4 - feature.size(2) % 4))  # cannot happen with reference configs
        group = self.grouper.forward_rows(new_xyz, xyz, feature if use_feature else None, index=ball)  # (B,M,K,ld)
        if use_feature and not self.use_xyz:
            group = group[..., :self.in_channels].contiguous()  # features only (modules.py:32-35 without the concat)
        elif use_feature and group.size(3) != self.in_channels:
            # columns are [feature(C) | xyz(3) | pad]; the conv weight expects [feature(C_true) | xyz(3)]
            c_true = self.in_channels - 3
            if feature.size(2) != c_true:
                group = torch.cat([group[..., :c_true], group[..., feature.size(2):feature.size(2) + 3]], dim=-1)
        new_feature = R.shared_mlp_rows(group.view(B * M * K, group.size(3)), self.mlp, K=K)
        return new_xyz, new_feature.view(B, M, -1)

    def forward(self, xyz, feature=None, rows=False, geometry=None):
        """xyz (B,3,N), feature (B,C,N) or None -> new_xyz (B,3,M), new_feature (B,C_out,M).
        rows=True: channels-last in and out ((B,N,3), (B,N,C) -> (B,M,3), (B,M,C_out))."""
        if rows:
            return self.forward_rows(xyz, feature, geometry=geometry)
        new_xyz, new_feature = self.forward_rows(xyz.transpose(1, 2).contiguous(),
                                                 None if feature is None else feature.transpose(1, 2).contiguous())
        return new_xyz.transpose(1, 2).contiguous(), new_feature.transpose(1, 2).contiguous()

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

    def geometry(self, query_xyz, key_xyz, with_csr=False):
        """3-NN index and inverse-squared-distance weights (coordinates only) [+ with_csr: the transposed index]."""
        with torch.no_grad():
            if query_xyz.is_cuda and query_xyz.dtype == torch.float32 and self.num_neighbors == 3:
                index, weight = R.knn3_weights(query_xyz, key_xyz, self._eps)  # index + weights from ONE kernel (no clamp / 1/x / sum / div)
            else:
                index, distance = ops.knn_distance(query_xyz, key_xyz, self.num_neighbors, transpose=False)
                inv = 1.0 / torch.clamp(distance, min=self._eps)
                weight = inv / torch.sum(inv, dim=2, keepdim=True)
            if with_csr:
                return (index, weight) + R.build_csr(index, key_xyz.size(1))
        return index, weight

    def forward_rows(self, query_xyz, key_xyz, query_feature, key_feature, geometry=None):
        """query_xyz (B,N1,3), key_xyz (B,N2,3), query_feature (B,N1,C1) or None, key_feature (B,N2,C2)
        -> (B,N1,C2[+C1]) rows, interpolated features first (modules.py:145)."""
        geometry = self.geometry(query_xyz, key_xyz) if geometry is None else geometry
        index, weight = geometry[0], geometry[1]
        interpolated = R.interp_rows(key_feature, index, weight)
        if query_feature is None:
            return interpolated
        return torch.cat([interpolated, query_feature], dim=2)

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

    def forward_rows(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, geometry=None, tail=None, sparse_act=None, act_out=None):
        """rows in / rows out: (B,N,3), (B,M,3), (B,N,C1) or None, (B,M,C2) -> (B,N,C_out).
        tail = (single-layer SharedMLPDO, training): the module that is this level's ONLY consumer, run as the last layer of the same chain
        (PN2SSG: mlp_seg behind the last propagation level) -> (B,N,C_tail): the level's output activation is never materialised, its
        BatchNorm-backward column sums come from the tail layer's input-gradient epilogue instead of a pass of their own.  Returns None when
        the chain cannot take it (the caller then runs the two modules one after the other)."""
        B, N, _ = dense_xyz.shape
        if self.interpolator is None:  # broadcast a single global feature
            assert sparse_xyz.size(1) == 1 and sparse_feature.size(1) == 1
            new_feature = torch.cat([sparse_feature.expand(-1, N, -1), dense_feature], dim=2)
        else:
            q0 = R.Parts(self.mlp._modules['0'])
            w0 = q0.w
            c1 = w0.size(0)
            c2 = sparse_feature.size(2)
            if tail is not None:
                # can the chain take the tail?  Decided BEFORE any work: the first layer below updates its BatchNorm's running statistics,
                # so a refusal after it would make the caller's second call update them twice
                chain = list(self.mlp) + list(tail[0])
                if not (len(tail[0]) == 1 and R.mlp_chain_is_fused(chain) and q0.bn is not None and tail[0][0].bn.training == q0.bn.training):
                    return None
            if q0.bn is not None and q0.relu is not None and q0.bias is None and q0.rm is not None and \
                    q0.bn.momentum is not None and c1 % 4 == 0 and 256 % (c1 // 4) == 0 and c2 % 4 == 0:
                # The first shared-MLP layer is linear and so is the interpolation:
                #   W1.[interp(f_sparse) | f_dense] = interp(W1a.f_sparse) + W1b.f_dense
                # -> the wide GEMM runs on the M = N/4 sparse points; the interpolation kernel adds the skip part and emits
                #    the layer's pre-BN output together with its batch statistics (modules.py:135-145,178-186; fp32 rounding only).
                M = sparse_xyz.size(1)
                geometry = self.interpolator.geometry(dense_xyz, sparse_xyz) if geometry is None else geometry
                index, weight = geometry[0], geometry[1]
                csr = tuple(geometry[2:4]) if len(geometry) >= 4 else None
                ctot = w0.numel() // c1                               # columns [interpolated (C2) | skip (C1)]
                sink = None
                if dense_feature is not None and torch.is_grad_enabled() and w0.requires_grad:
                    sink = R.WeightGradSink(w0, 2)  # one gradient buffer for the weight's two column groups
                # sparse_act / act_out (rows.ActivationHandOver): sparse_feature is the output of the level before and read by nobody else; this
                # level's own output goes to the level behind in the same way
                z = R.linear_rows(sparse_feature.reshape(B * M, c2), w0, cols=(0, c2), sink=sink, act_src=sparse_act).view(B, M, c1)
                zs = None
                if dense_feature is not None:
                    zs = R.linear_rows(dense_feature.reshape(B * N, -1), w0, cols=(c2, ctot), sink=sink).view(B, N, c1)
                bn_training = q0.bn.training
                # no skip feature (the level that lands on the input cloud): the first layer's gradient has ONE consumer, the interpolation's gather
                # backward -- its BatchNorm-backward finish is then applied there, on load (rows.DeferredFinish)
                defer = R.DeferredFinish() if (zs is None and bn_training and torch.is_grad_enabled()) else None
                y1 = R.interp_add_rows(z, index, weight, zs, want_stat=q0.bn if bn_training else False, csr=csr, defer=defer)
                stat1 = None
                if bn_training:
                    y1, stat1 = y1[0], (y1[1], y1[2])
                if tail is not None:
                    return R.shared_mlp_rows(y1.view(B * N, c1), chain, first_done=True, first_stat=stat1, dropout_p=tail[0].p, training=tail[1],
                                             dropout_last_only=True, defer=defer, act_out=act_out).view(B, N, -1)
                return R.shared_mlp_rows(y1.view(B * N, c1), self.mlp, first_done=True, first_stat=stat1, defer=defer, act_out=act_out).view(B, N, -1)
            if tail is not None:
                return None
            new_feature = self.interpolator.forward_rows(dense_xyz, sparse_xyz, dense_feature, sparse_feature, geometry=geometry)
        if tail is not None:
            return None
        return R.shared_mlp_rows(new_feature.reshape(B * N, -1), self.mlp).view(B, N, -1)

    def forward(self, dense_xyz, sparse_xyz, dense_feature, sparse_feature, rows=False, geometry=None, tail=None, sparse_act=None, act_out=None):
        if rows:
            return self.forward_rows(dense_xyz, sparse_xyz, dense_feature, sparse_feature, geometry=geometry, tail=tail, sparse_act=sparse_act,
                                     act_out=act_out)
        t = lambda x: None if x is None else x.transpose(1, 2).contiguous()
        return self.forward_rows(t(dense_xyz), t(sparse_xyz), t(dense_feature), t(sparse_feature)).transpose(1, 2).contiguous()


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

    def _centroid_run(self, level, xyz, fps_shape=None):
        """Centroids of set-abstraction level `level` (input cloud xyz (B,N,3)) AND of the levels behind it that sample its output:
        {level: new_xyz, level + 1: ..., ...}.  The reference samples level after level (modules.py:74-87, pn2ssg.py:92-99); level l + 1
        then samples a cloud that IS level l's sampling result in sampling order, and farthest point sampling of such a cloud returns
        0, 1, 2, ...: by induction the second chain's running distances are the first chain's (same centroids so far, same arithmetic),
        the point the first chain took next attains the maximum over the superset and hence over the subset, points taken earlier sit
        at distance 0, and any tied point has a higher index -- so the first maximum is index i.  When the maximum is 0 (fewer distinct
        points than samples) both chains return point 0, whose coordinates the prefix repeats.  Either way the centroid COORDINATES of
        level l + 1 are the first M_{l+1} rows of level l's: one sampling launch and one gather per run of levels instead of one each per
        level (tests/test_model_gpu.py::test_centroid_prefix_equals_the_chained_sampling).  MVP_FPS_PREFIX=0 samples every level."""
        mods = self.sa_modules
        m = mods[level]
        assert m.num_centroids != 0
        if m.num_centroids == -1:
            return {level: xyz}
        run, counts = [level], [m.num_centroids]
        j = level + 1
        while FPS_PREFIX and j < len(mods) and len(run) < 8 and (mods[j].num_centroids == -1 or 0 < mods[j].num_centroids <= counts[-1]):
            counts.append(counts[-1] if mods[j].num_centroids == -1 else mods[j].num_centroids)
            run.append(j)
            j += 1
        with torch.no_grad():
            index = ops.farthest_point_sample(xyz, m.num_centroids, transpose=False, shape=fps_shape)
            return dict(zip(run, centroid_levels(xyz, index, counts)))

    def plan_geometry(self, xyz, stream=None, with_csr=None):
        """All coordinate-only work of the network -- 4 x (FPS, ball query) and 4 x (3-NN + weights) -- as a
        plan that forward() consumes.  FPS is a chain of ~2700 dependent steps that occupies only B CUs; with
        `stream` (a side HIP stream) it runs concurrently with the feature path (lifting, aggregation MLP) on
        the caller's stream.  The plan records an event; forward() makes the current stream wait for it.
        with_csr (default: training with gradients enabled): also the transposed ball / 3-NN indices, so the backward of
        the grouping and of the interpolation gathers instead of scattering with atomics."""
        if with_csr is None:
            with_csr = self.training and torch.is_grad_enabled()
        if NATIVE_PLAN and self._native_plan_ok(xyz):
            return self._plan_geometry_native(xyz, stream, with_csr)
        cur = torch.cuda.current_stream(xyz.device)
        if stream is not None:
            stream.wait_stream(cur)
        # Only the FPS chain is sequential (level l + 1 samples level l's centroids).  Outside graph capture the ball queries, the
        # 3-NN searches and the transposed indices go to a SECOND side stream, each as soon as its centroids exist, so the
        # geometry's critical path is the FPS chain alone (2.8 of 3.7 ms at B = 32: it bounds the eval-mode forward).
        # (Training keeps ONE side stream: its geometry hides under 11 ms of forward + backward anyway, and a third stream of
        # small kernels takes issue slots from the MFMA kernels: 2790 -> 2750 chunks/s.)
        two = stream is not None and not with_csr and not torch.cuda.is_current_stream_capturing()
        s2 = self._second_stream(xyz.device) if two else None

        def on_second(after, fn):
            if not two:
                return fn()
            ev = torch.cuda.Event()
            ev.record(after)
            with torch.cuda.stream(s2):
                s2.wait_event(ev)
                return fn()

        # a training step's prefetched chain hides under forward + backward: sample with half the waves (passed per call, nothing
        # process-wide is touched)
        fps_shape = (TRAIN_FPS_SHAPE if (with_csr and stream is not None) else 0) if xyz.is_cuda else None
        # Inference plans on a side stream also record one event (pair) per level: forward() then starts set-abstraction level l as soon
        # as ITS centroids and neighbours exist instead of after the whole chain (a single chunk: the levels' MLPs run under the rest of
        # the FPS chain).  Inside a graph capture (one side stream) the events become edges of the graph.
        level_events = [] if (stream is not None and not with_csr) else None
        with torch.cuda.stream(stream if stream is not None else cur):
            run = torch.cuda.current_stream(xyz.device)
            sa, xyzs = [], [xyz]
            fp_by_level = {}
            cents = {}
            for level, m in enumerate(self.sa_modules):
                if m.num_centroids == 0:
                    sa.append(None)
                    xyzs.append(xyz.new_zeros([xyz.size(0), 1, 3]))
                else:
                    if level not in cents:  # one sampling launch for this level and every level that samples its output
                        cents.update(self._centroid_run(level, xyzs[-1], fps_shape))
                    new_xyz = cents[level]
                    csr = with_csr and (level > 0 or self.in_channels > 0)
                    sa.append((new_xyz,) + on_second(run, lambda m=m, a=new_xyz, b=xyzs[-1], c=csr: m.neighbours(a, b, c)))
                    xyzs.append(new_xyz)
                if level_events is not None:
                    # recorded BEFORE this level's 3-NN search is queued: set abstraction does not need it (one chunk: 180 us on the
                    # second stream); feature propagation waits for the plan's end event, which covers every 3-NN
                    ev_run, ev_s2 = torch.cuda.Event(), (torch.cuda.Event() if two else None)
                    ev_run.record(run)   # this level's centroids (the FPS chain so far; with ONE side stream also its neighbours)
                    if two:
                        ev_s2.record(s2)  # its ball query
                    level_events.append((ev_run, ev_s2))
                fpm = self.fp_modules[len(self.sa_modules) - 1 - level]  # propagates level + 1 -> level
                if fpm.interpolator is not None:
                    fp_by_level[level] = on_second(run, lambda i=fpm.interpolator, a=xyzs[-2], b=xyzs[-1]: i.geometry(a, b, with_csr=with_csr))
            fp = [fp_by_level.get(len(self.sa_modules) - 1 - k) for k in range(len(self.fp_modules))]
            if two:
                run.wait_stream(s2)
            event = torch.cuda.Event()
            event.record()
        # `xyz` is read by the side stream long after this function returns (ball query / 3-NN of level 1 run after the 2.4 ms
        # FPS): the plan keeps it alive, otherwise the caller's stream may recycle its memory while it is still being read.
        plan = {'sa': sa, 'fp': fp, 'event': event, 'stream': stream, 'xyz': xyz}
        if level_events is not None:
            plan['level_events'] = level_events
        if stream is not None and not torch.cuda.is_current_stream_capturing():
            xyz.record_stream(stream)
        if stream is not None and not torch.cuda.is_current_stream_capturing():  # tensors were allocated on the side stream but are consumed on the caller's
            for g in sa + fp:
                if g is not None:
                    for t in g:
                        t.record_stream(cur)
        return plan

    def _native_plan_ok(self, xyz):
        """The whole plan in ONE library call (mvp_pn2_plan_f32) needs: float32 clouds on the GPU, every level sampling the level above
        (so one sampling launch + prefixes give all centroids), 3-NN interpolation at every propagation level, the 4-level default wiring."""
        ms = [m.num_centroids for m in self.sa_modules]
        return (xyz.is_cuda and xyz.dtype == torch.float32 and FPS_PREFIX and len(ms) <= 8 and len(self.fp_modules) == len(ms) and
                all(m > 0 for m in ms) and ms[0] <= xyz.size(1) and all(b <= a for a, b in zip(ms, ms[1:])) and
                all(f.interpolator is not None and f.interpolator.num_neighbors == 3 for f in self.fp_modules) and min(ms) >= 3)

    def _plan_geometry_native(self, xyz, stream, with_csr):
        """plan_geometry through mvp_pn2_plan_f32 (csrc/plan.hip): the same launches in the same order on ONE stream, from one table of
        buffers allocated here -- ~0.1 ms of host time instead of ~0.65 (25 calls + allocations), which in a training step sits between
        the forward and the backward pass and in inference in front of a single chunk's latency.  Inference plans on a side stream carry
        one event per level (recorded by the library call) so forward() starts level l as soon as its own geometry is queued."""
        import ctypes
        dev = xyz.device
        B, N, _ = xyz.shape
        mods = self.sa_modules
        nl = len(mods)
        ms = [m.num_centroids for m in mods]
        ks = [m.max_neighbors for m in mods]
        cur = torch.cuda.current_stream(dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if stream is not None:
            stream.wait_stream(cur)
        run = stream if stream is not None else cur
        fps_shape = (TRAIN_FPS_SHAPE if (with_csr and stream is not None) else 0)
        i32, i64, f32 = torch.int32, torch.int64, torch.float32
        with torch.cuda.stream(run):
            e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
            fps_index = e((B, ms[0]), i64)
            table = [fps_index]
            sa, fp = [], []
            geom = []
            for l, m in enumerate(mods):
                nin = N if l == 0 else ms[l - 1]
                csr_l = with_csr and (l > 0 or self.in_channels > 0)
                g = (e((B, ms[l], 3), f32), e((B, ms[l], ks[l]), i64))
                scratch = ()
                want_geo = False
                if with_csr:  # (the library builds every level's transposed index when asked for any: level 0 without an input feature keeps none)
                    g += (e((B, nin + 1), i32), e((B, ms[l] * ks[l]), i32))
                    scratch = (e((B, nin), i32),)
                    want_geo = bool(csr_l and R.SA_TRAIN_FUSED and ks[l] == 32 and len(m.mlp) == 3 and
                                    R.sa_level_train_widths_ok(*(x.conv.weight.size(0) for x in m.mlp)))
                    if want_geo:
                        g += (e((B, nin, 4), f32), torch.zeros(16, dtype=torch.float64, device=dev))
                geom.append(1 if want_geo else 0)
                table += list(g[:4]) + list(scratch) + list(g[4:])
                sa.append(g if (csr_l or not with_csr) else g[:2])
            for l in range(nl - 1, -1, -1):
                nq = N if l == 0 else ms[l - 1]
                g = (e((B, nq, 3), i64), e((B, nq, 3), f32))
                scratch = ()
                if with_csr:
                    g += (e((B, ms[l] + 1), i32), e((B, 3 * nq), i32))
                    scratch = (e((B, ms[l]), i32),)
                table += list(g) + list(scratch)
                fp.append(g)
            level_events = None
            ev_arr = None
            if stream is not None and not with_csr:
                level_events = []
                for _ in range(nl):
                    ev = torch.cuda.Event()
                    ev.record(run)  # (creates the handle; the library call records it again, later in stream order)
                    level_events.append((ev, None))
                ev_arr = (ctypes.c_void_p * nl)(*[ev.cuda_event for ev, _ in level_events])
            flags = (1 if with_csr else 0) | (2 if (with_csr and any(geom)) else 0) | (4 if (with_csr and L.DW_WORKSPACE) else 0)
            from .ext.ball_query_cuda import BALL_GRID
            ws_bytes = max(max(int(L.lib().mvp_ball_query_grid_workspace(B, ms[l], N if l == 0 else ms[l - 1])),
                               int(L.lib().mvp_knn3_grid_workspace(B, N if l == 0 else ms[l - 1], ms[l]))) for l in range(nl)) if BALL_GRID else 0
            if ws_bytes > 0:  # large levels: ball query through the cell grid (csrc/ball_grid.hip), scratch at the end of the table
                table.append(e((ws_bytes,), torch.uint8))
                flags |= 8 | (16 if os.environ.get('MVP_KNN_GRID', '1') == '0' else 0)
            radius = (ctypes.c_float * nl)(*[float(m.radius) for m in mods])
            L.call('mvp_pn2_plan_f32', xyz, L.ptr(xyz), B, N, nl, (ctypes.c_int64 * nl)(*ms), radius, (ctypes.c_int64 * nl)(*ks),
                   (ctypes.c_int32 * nl)(*geom), int(fps_shape), flags, float(self.fp_modules[0].interpolator._eps),
                   (ctypes.c_void_p * len(table))(*[t.data_ptr() for t in table]), len(table), ev_arr, L.ptr(L.fps_status(dev)))
            event = torch.cuda.Event()
            event.record()
        plan = {'sa': sa, 'fp': fp, 'event': event, 'stream': stream, 'xyz': xyz}
        if level_events is not None:
            plan['level_events'] = level_events
        if stream is not None and not capturing:
            xyz.record_stream(stream)
            for t in table:
                t.record_stream(cur)
        return plan

    @staticmethod
    def slice_plan(plan, lo, hi):
        """The geometry plan of clouds lo..hi-1 of a plan made for a larger batch (views, no copies).  FPS runs one workgroup
        per cloud for ~2.8 ms whatever the batch size, so planning ALL chunks of a scene in one call and slicing per batch costs
        one FPS chain per scene instead of one per batch (scene.infer_scene)."""
        cut = lambda g: None if g is None else tuple(t[lo:hi] for t in g[:4])  # (the geometry sums behind the transposed index are batch-wide: dropped)
        out = {'sa': [cut(g) for g in plan['sa']], 'fp': [cut(g) for g in plan['fp']], 'event': plan['event'],
               'stream': plan['stream'], 'xyz': plan['xyz']}
        if 'level_events' in plan:  # the per-level events of the whole plan hold for every slice of it
            out['level_events'] = plan['level_events']
        return out

    def _second_stream(self, device):
        if getattr(self, '_geo_stream2', None) is None or self._geo_stream2.device != device:
            self._geo_stream2 = torch.cuda.Stream(device=device)
        return self._geo_stream2

    def forward(self, data_batch):
        """data_batch: 'points' (B,3,N) [+ 'points_rows' (B,N,3): the same, already transposed] [+ 'feature' (B,C,N), or 'feature_rows' (B,N,C) channels-last]
        [+ 'geometry_plan' from plan_geometry()] -> {'seg_logit': (B,num_classes,N)}."""
        with R.zero_pool.step(data_batch['points'].device), R.eval_invstd.scope(self):  # one zero fill per step / one invstd pass per eval forward
            return self._forward(data_batch)

    def _forward(self, data_batch):
        xyz = data_batch['points_rows'] if 'points_rows' in data_batch else data_batch['points'].transpose(1, 2).contiguous()  # (B,N,3)
        plan = data_batch.get('geometry_plan')
        level_events = None if plan is None else plan.get('level_events')
        if plan is not None and plan.get('event') is not None and level_events is None:
            torch.cuda.current_stream(xyz.device).wait_event(plan['event'])
        if 'feature_rows' in data_batch:
            feature = data_batch['feature_rows']
        else:
            feature = data_batch.get('feature', None)
            feature = None if feature is None else feature.transpose(1, 2).contiguous()
        B, N, _ = xyz.shape
        xyzs, feats = [xyz], [None]
        cents = {}
        for level, sa in enumerate(self.sa_modules):
            geometry = None if plan is None else plan['sa'][level]
            if plan is None and sa.num_centroids != 0 and xyz.is_cuda:  # no plan: still ONE sampling launch per run of levels
                if level not in cents:
                    cents.update(self._centroid_run(level, xyz))
                geometry = (cents[level],) + sa.neighbours(cents[level], xyz)
            if level_events is not None:  # this level's geometry only (the deeper levels are still being sampled)
                cur = torch.cuda.current_stream(xyz.device)
                for ev in level_events[level]:
                    if ev is not None:
                        cur.wait_event(ev)
                if level + 1 == len(self.sa_modules) and plan.get('event') is not None:
                    cur.wait_event(plan['event'])  # the plan's own end (nothing is left to run: this joins the side stream, as a capture needs)
            xyz, feature = sa(xyz, feature, rows=True, geometry=geometry)
            xyzs.append(xyz)
            feats.append(feature)
        up = feats[-1]
        x = None
        hand = None  # rows.ActivationHandOver of the level before: its output `up` is read by this level's first linear layer and nobody else
        head_hand = None
        watched = [_has_hooks(fp) for fp in self.fp_modules]
        for level, fp in enumerate(self.fp_modules):
            geo = None if plan is None else plan['fp'][level]
            if MERGE_HEAD and level + 1 == len(self.fp_modules) and up.is_cuda and not _has_hooks(fp) and not _has_hooks(self.mlp_seg):
                # the segmentation head's MLP is the last propagation level's only consumer: one chain (no activation tensor in between, no
                # column-statistics pass of its own in backward)
                # ... and the logit layer is the head's only consumer (rows.ActivationHandOver, HEAD_HANDOVER): its input gradient applies the head's
                # dropout + ReLU mask and sums the two BatchNorm-backward columns in its epilogue
                head_hand = R.ActivationHandOver() if (HEAD_HANDOVER and torch.is_grad_enabled() and not _has_hooks(self.seg_logit)) else None
                x = fp(xyzs[-2 - level], xyzs[-1 - level], feats[-2 - level], up, rows=True, geometry=geo, tail=(self.mlp_seg, self.training),
                       sparse_act=hand, act_out=head_hand)
                if x is not None:
                    x = x.reshape(B * N, -1)
                    break
                head_hand = None
            nxt = None
            if (level + 1 < len(self.fp_modules) and up.is_cuda and torch.is_grad_enabled() and not watched[level] and not watched[level + 1]
                    and self.fp_modules[level + 1].interpolator is not None):
                nxt = R.ActivationHandOver()
            up = fp(xyzs[-2 - level], xyzs[-1 - level], feats[-2 - level], up, rows=True, geometry=geo, sparse_act=hand, act_out=nxt)
            hand = nxt
        if x is None:
            x = R.shared_mlp_rows(up.reshape(B * N, -1), self.mlp_seg, dropout_p=self.mlp_seg.p, training=self.training)
        logit = R.linear_rows(x, self.seg_logit.weight, self.seg_logit.bias, act_src=head_hand)  # (B*N, classes)
        # (B,classes,N) as the reference returns it -- as a transposed VIEW of the rows: the loss and the vote kernels take strided logits,
        # the gradient comes back in the same layout (mvpnet3d._SegLossFn), so neither direction pays a transposing copy
        return {'seg_logit': logit.view(B, N, self.num_classes).transpose(1, 2)}

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Linear)):
                xavier_uniform(m)
