"""Channels-last ("rows") building blocks of the PointNet++ pipeline on libmvp_hip.so (csrc/rows.hip).

A point's C features are one contiguous row.  These functions compute exactly what the
reference's channel-major modules compute (QueryGrouper, feature_interpolate, Conv+BN+ReLU,
torch.max over neighbours: mvpnet/models/pn2/modules.py:20-37,107-108,135-145;
common/nn/modules/conv.py:41-51) but on (rows, C) matrices, so gathers / scatters are
coalesced row accesses and a shared-MLP layer is one row-major GEMM.
"""
import os
import weakref

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _lib as L


# Widest layer (output channels; input channels up to 2x... see below) whose backward runs as ONE kernel (mvp_mlp_layer_backward_f32).
# Measured on MI355X (profiles/r02_*): up to 64 x 96 the one-kernel backward halves the layer's HBM traffic and time; the 128-wide
# variants run at one wave per SIMD and re-read dy_i per c_in slice -- tuning knob, default from the measurements.
FUSE_BWD_MAX_COUT = int(os.environ.get('MVP_BWD_FUSE_MAXC', '64'))
FUSE_BWD_MAX_CIN = int(os.environ.get('MVP_BWD_FUSE_MAXCIN', '96'))
# Column statistics of the tile kernels: below this many rows (R / 128 workgroups) the workgroups add into `stat` with fp64 atomics, from it on
# through per-tile slots + a reduction launch (atomics queue per address: ~10 ns each)
PARTIAL_MIN_ROWS = int(os.environ.get('MVP_PARTIAL_MIN_ROWS', '65536'))
# Layers with 65 .. 128 channels on either side (the last propagation level, the segmentation head, level 3's middle layer): the whole
# backward of a layer -- BatchNorm finish, weight gradient, input gradient + the previous layer's ReLU mask and column sums -- in ONE pass
# with the row tile staged in LDS (mvp_mlp_layer_backward_wide_p_f32, csrc/mlp_bwd_wide.hip) instead of finish pass + input-gradient GEMM +
# reduction launch on the training stream and the weight-gradient GEMM beside them.  MVP_BWD_WIDE=0: the per-layer kernels (A/B switch).
WIDE_BWD = os.environ.get('MVP_BWD_WIDE', '1') != '0'
WIDE_BWD_MIN_ROWS = int(os.environ.get('MVP_BWD_WIDE_MIN_ROWS', '16384'))  # (fewer rows than ~one tile per CU: the tile kernels' narrow variants)


# ... and, in its 64-channel instance, the 64 -> 64 layers over very many rows (the aggregation MLP's inner layers: 786 432 rows), where the
# register-resident one-kernel backward runs at one wave per SIMD.  MVP_BWD_WIDE64=0: those layers stay on mvp_mlp_layer_backward_f32.
WIDE_BWD_64 = os.environ.get('MVP_BWD_WIDE64', '1') != '0'
# ... and as the backward of the layer in front of a SUM over the neighbours (mode 2 with one gradient row per K rows).  MVP_BWD_WIDE_POOLED=0: the
# BatchNorm-backward pass writes dy_L first (A/B switch).
WIDE_BWD_POOLED = os.environ.get('MVP_BWD_WIDE_POOLED', '1') != '0'
WIDE_BWD_64_MIN_ROWS = int(os.environ.get('MVP_BWD_WIDE64_MIN_ROWS', '262144'))


# Layers wider than the one-pass backward takes (256 / 512 channels: set-abstraction levels 3 - 4, propagation levels 1 - 3): the input gradient in
# one pass with the BatchNorm-backward finish on load (mvp_mlp_input_grad_wide_p_f32, csrc/mlp_dx_wide.hip) instead of finish pass + input-gradient
# GEMM + reduction launch, the weight gradient beside it with the same finish on load.  OFF by default: alone on the chip the one-pass form is
# 15 - 40 % faster than the launches it replaces (profiles/r06_dx_wide_alone.txt), inside the training step it is 0.6 % SLOWER (6.295 -> 6.33 ms,
# profiles/r06_dx_wide_step_ab.txt): a persistent workgroup per CU with 123 KB of LDS leaves no room for the kernels of the other two streams
# (weight gradients, the next batch's geometry) that the many small workgroups of the per-layer kernels share the CUs with, and the weight image
# costs a launch per layer.  MVP_DX_WIDE=1 switches it on (tests/test_model_gpu.py runs the network both ways).
DX_WIDE = os.environ.get('MVP_DX_WIDE', '0') == '1'
DX_WIDE_MIN_ROWS = int(os.environ.get('MVP_DX_WIDE_MIN_ROWS', '16384'))
DX_WIDE_MAX_COUT = int(os.environ.get('MVP_DX_WIDE_MAXC', '512'))


def dx_wide_ok(prec, R, cout, cin, ldx, finish):
    """True when a layer's input gradient (R rows, cin -> cout, input row stride ldx, `finish`: its own gradient still needs the
    BatchNorm-backward finish) takes mvp_mlp_input_grad_wide_p_f32: whole 64 x 128 weight blocks, a one- or two-piece backward split, at most
    256 channels with a pending finish (512 without)."""
    return bool(DX_WIDE and prec[0] != 0 and prec[1] in (1, 3) and R >= DX_WIDE_MIN_ROWS and cout % 64 == 0 and cin % 128 == 0 and ldx % 4 == 0 and
                cout <= min(DX_WIDE_MAX_COUT, 256 if finish else 512) and max(cout, cin) >= L.mlp_min_width())


def aligned16(*tensors):
    """True when every tensor (None allowed) starts on a 16-byte boundary: the one-pass backward and the finish-on-load weight gradient read
    whole 16-byte row pieces and answer MVP_EUNSUPPORTED otherwise (a contiguous() view keeps its storage offset) -- their callers then keep
    the per-layer kernels instead of raising in the middle of a backward pass."""
    return all(t is None or t.data_ptr() % 16 == 0 for t in tensors)


def wide_backward_ok(prec, R, cout, cin, ldx):
    """True when a layer (R rows, cin -> cout, input row stride ldx) takes the one-pass wide backward: whole 16-byte quadruples per row, and
    either more than 64 and at most 128 channels (a one- or two-piece backward split) or the 64 -> 64 shape over very many rows (any split:
    bf16, bf16x3, bf16x6)."""
    if not (WIDE_BWD and prec[0] != 0 and prec[1] in (1, 3, 6) and cout % 4 == 0 and cin % 4 == 0 and ldx % 4 == 0):
        return False
    if prec[1] == 6 and max(cout, cin) > 64:  # three pieces per operand fit the LDS for the 64-channel instance only (csrc/mlp_bwd_wide.hip)
        return False
    if max(cout, cin) < L.mlp_min_width():  # (mvp_set_mlp_precision's min_width: such layers run on the fp32 MFMA, which the one-pass kernel has not)
        return False
    if 64 < max(cout, cin) <= 128:
        return R >= WIDE_BWD_MIN_ROWS
    return WIDE_BWD_64 and cout == 64 and cin == 64 and R >= WIDE_BWD_64_MIN_ROWS


# Set-abstraction levels (K = 32 neighbours, max pooling): run the LAST shared-MLP layer without ever storing its (B*M*32, C) output --
# forward leaves per-ball max / min of the pre-BN values (mvp_mlp_forward_pool_f32), backward re-computes the layer from its input inside
# the one-kernel layer backward (POOL front end).  Needs C_out, C_in <= 64 and >= 32768 rows (levels 1 and, with 64-wide MLPs, 2).
POOL_WITHOUT_Y = os.environ.get('MVP_POOL_NO_Y', '1') != '0'
# Dropout behind a single-layer SharedMLPDO chain (the segmentation head) inside the BatchNorm + ReLU kernels, mask regenerated in backward.
FUSE_DROPOUT = os.environ.get('MVP_FUSE_DROPOUT', '1') != '0'


class Parts:
    """What the hot paths read of one Conv + BN + ReLU layer (common/nn/modules/conv.py:4-51), fetched straight from the module's
    parameter / buffer / sub-module dictionaries: `layer.conv.weight` goes through nn.Module.__getattr__ (a Python function with three
    dictionary probes) per dot -- ~1300 such calls per training step were 0.5 ms of its host time.  Nothing is cached: the dictionaries are
    read on every call, so replaced parameters / modules are seen."""
    __slots__ = ('conv', 'w', 'bias', 'bn', 'relu', 'gamma', 'beta', 'rm', 'rv', 'nbt')

    def __init__(self, layer):
        mods = layer._modules
        conv = self.conv = mods['conv']
        cp = conv._parameters
        self.w, self.bias = cp['weight'], cp.get('bias')
        bn = self.bn = mods.get('bn')
        self.relu = mods.get('relu')
        if bn is None:
            self.gamma = self.beta = self.rm = self.rv = self.nbt = None
        else:
            bp, bb = bn._parameters, bn._buffers
            self.gamma, self.beta = bp.get('weight'), bp.get('bias')
            self.rm, self.rv, self.nbt = bb.get('running_mean'), bb.get('running_var'), bb.get('num_batches_tracked')


def parts(mlp):
    """[Parts(layer) for layer in mlp] for a SharedMLP / list of Conv*BNReLU layers."""
    return [Parts(l) for l in (mlp._modules.values() if isinstance(mlp, torch.nn.ModuleList) else mlp)]


def _round4(c):
    return (c + 3) // 4 * 4


class GroupRows(torch.autograd.Function):
    """[feature row | xyz - center | 0-pad] per (centroid, neighbour); grad -> feature only."""

    @staticmethod
    def forward(ctx, feature, xyz, center, index):
        L.require_gpu(feature, xyz, center, index)
        B, M, K = index.shape
        N = xyz.size(1)
        C = 0 if feature is None else feature.size(2)
        ld = _round4(C + 3)
        out = torch.empty((B, M, K, ld), dtype=torch.float32, device=index.device)
        L.call('mvp_group_rows_f32', index, L.ptr(feature), L.ptr(xyz), L.ptr(center), L.ptr(index), B, N, C, M, K, ld, L.ptr(out))
        ctx.save_for_backward(index)
        ctx.dims = (B, N, C, M, K, ld)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        B, N, C, M, K, ld = ctx.dims
        if C == 0:
            return None, None, None, None
        (index,) = ctx.saved_tensors
        g = grad_out.contiguous()
        grad = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
        L.call('mvp_group_rows_backward_f32', g, L.ptr(g), L.ptr(index), B, N, C, M, K, ld, L.ptr(grad))
        return grad, None, None, None


def group_rows(feature, xyz, center, index):
    """feature (B,N,C) or None (C % 4 == 0), xyz (B,N,3), center (B,M,3), index (B,M,K) int64
    -> (B,M,K,ld) with ld = round_up(C+3, 4): columns [0,C) features, [C,C+3) xyz - center, rest 0."""
    if feature is not None and (feature.dtype != torch.float32 or feature.size(2) % 4):
        raise RuntimeError('group_rows: float32 feature with C % 4 == 0 expected')
    return GroupRows.apply(None if feature is None else feature.contiguous(), xyz.contiguous(), center.contiguous(),
                           index.contiguous())


class ZeroPool:
    """One zero fill per training step instead of one per accumulator.

    The statistics / weight-gradient kernels ACCUMULATE into their outputs (include/mvp_hip.h), so every shared-MLP chain
    needs zeroed arenas in forward and in backward: ~90 `torch.zeros` fills of a few KB per step, 4-5 us of device time
    each and all on the critical stream.  `step()` (entered by the top-level model forward) allocates ONE zeroed
    buffer sized by the previous step's demand; `zeros()` hands out disjoint 256-byte aligned views of it, each exactly
    once -- a request that does not fit (first step, changed shapes, a second forward before the first backward) falls
    back to `torch.zeros`.  The buffer is never recycled by the pool: the views (e.g. the parameters' .grad) keep it alive."""

    def __init__(self):
        self.buf = None
        self.cursor = 0
        self.demand = 0
        self.capacity = 0
        self.depth = 0

    class _Step:
        def __init__(self, pool, dev):
            self.pool, self.dev = pool, dev

        def __enter__(self):
            p = self.pool
            if p.depth == 0:
                p.capacity = max(p.demand, p.cursor)  # what the last step asked for in total
                p.buf = torch.zeros(p.capacity // 8, dtype=torch.float64, device=self.dev) if p.capacity else None
                p.cursor = p.demand = 0
                if self.dev.type == 'cuda':
                    weight_slices.refresh(self.dev)
            p.depth += 1

        def __exit__(self, *exc):
            self.pool.depth -= 1

    def step(self, dev):
        return ZeroPool._Step(self, dev)

    def zeros(self, shape, dtype, dev):
        if isinstance(shape, int):
            shape = (shape,)
        numel = 1
        for d in shape:
            numel *= int(d)
        nbytes = (numel * torch.empty(0, dtype=dtype).element_size() + 255) // 256 * 256
        self.demand += nbytes
        buf = self.buf
        if buf is None or buf.device != dev or self.cursor + nbytes > self.capacity or numel == 0:
            return torch.zeros(shape, dtype=dtype, device=dev)
        view = buf[self.cursor // 8:(self.cursor + nbytes) // 8].view(dtype)[:numel].view(shape)
        self.cursor += nbytes
        return view


zero_pool = ZeroPool()


class WeightSlices:
    """Column groups of conv weights as the kernels take them -- contiguous, 16-byte aligned rows, zero padded to the operand's row
    length -- in persistent buffers that ONE launch (mvp_copy_slices_f32) refreshes at the start of a forward, instead of a strided
    copy per slice and forward (~16 per training step; none at all in inference, where the weights do not change).

    get() returns the operand for the weight's CURRENT version (a miss -- first use, or the weight changed after the refresh --
    copies at once).  Every slice has two buffers and a refresh writes the one not handed out last, so the operand a forward saved
    for its backward stays intact until the second optimizer step after it."""

    ENABLED = os.environ.get('MVP_WEIGHT_SLICES', '1') != '0'

    class Entry:
        __slots__ = ('weight', 'c0', 'cols', 'bufs', 'cur', 'version', 'ptr')

    def __init__(self):
        self.entries = {}
        self.tables = {}   # device -> ([table writing bufs[0], table writing bufs[1]] on the device, entries in table order)
        self.retired = []  # superseded tables stay allocated: a captured graph may still launch the refresh kernel with their address

    @staticmethod
    def _src(e):
        return e.weight.detach().reshape(e.weight.size(0), -1)[:, e.c0:e.c0 + e.cols]

    def get(self, weight, c0, c1, ld):
        if not WeightSlices.ENABLED:  # a fresh copy per call (A/B switch)
            w = weight.detach().reshape(weight.size(0), -1)[:, c0:c1]
            return (F.pad(w, (0, ld - (c1 - c0))) if ld != c1 - c0 else w).contiguous()
        key = (id(weight), c0, c1, ld)
        e = self.entries.get(key)
        if e is None or e.weight is not weight or e.bufs[0].device != weight.device:
            e = WeightSlices.Entry()
            e.weight, e.c0, e.cols, e.cur, e.version, e.ptr = weight, c0, c1 - c0, 0, None, None
            e.bufs = [torch.zeros((weight.size(0), ld), dtype=torch.float32, device=weight.device) for _ in range(2)]
            if not torch.cuda.is_current_stream_capturing():
                self.entries[key] = e
                old = self.tables.pop(weight.device, None)
                if old is not None:
                    self.retired.append(old[0])
        if e.version != weight._version or e.ptr != weight.data_ptr():
            e.cur ^= 1
            e.bufs[e.cur][:, :e.cols].copy_(self._src(e))
            e.version, e.ptr = weight._version, weight.data_ptr()
        return e.bufs[e.cur]

    def invalidate(self):
        """Forget what the buffers hold: the next get() / refresh() copies every slice again.  Needed after writes that do not bump
        the weights' version counters -- `.data` assignments, collectives (dist.broadcast_parameters calls this)."""
        for e in self.entries.values():
            e.version = None

    def refresh(self, dev):
        """Top of a forward: bring every slice of a changed weight up to date, all of them in one launch."""
        if not WeightSlices.ENABLED:
            return
        mine = [e for e in self.entries.values() if e.bufs[0].device == dev]
        if not mine:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        tab = self.tables.get(dev)
        if (tab is None or any(e.weight.data_ptr() != p for e, p in zip(tab[1], tab[2]))) and not capturing:
            rows = [[], []]
            for e in mine:
                w2 = e.weight.detach().reshape(e.weight.size(0), -1)
                for par in (0, 1):
                    rows[par].append([w2.data_ptr() + 4 * e.c0, e.bufs[par].data_ptr(), w2.stride(0), e.bufs[par].stride(0), w2.size(0), e.cols])
            if tab is not None:
                self.retired.append(tab[0])
            tab = self.tables[dev] = ([torch.tensor(r, dtype=torch.int64).to(dev) for r in rows], mine, [e.weight.data_ptr() for e in mine])
        if tab is None:
            return  # (inside a capture, before any table exists: get() copies slice by slice)
        stale = [e.version != e.weight._version or e.ptr != e.weight.data_ptr() for e in tab[1]]
        if not capturing and not any(stale):
            return
        # every slice moves to its other buffer together (one table per parity): slices that were up to date are copied again
        # (a slice whose parity drifted -- a miss in get() since the last refresh -- is simply re-aligned here)
        par = tab[1][0].cur ^ 1
        L.call('mvp_copy_slices_f32', tab[0][par], L.ptr(tab[0][par]), len(tab[1]))
        for e in tab[1]:
            e.cur, e.version, e.ptr = par, e.weight._version, e.weight.data_ptr()


weight_slices = WeightSlices()


# Weight gradients of the wide layers (mvp_mlp_weight_grad_f32: one launch per layer) run on a second HIP stream beside the
# input-gradient chain: each of these kernels alone leaves most of the chip's issue slots and HBM bandwidth idle
# (profiles/r02_step_kernel_stats.csv), so the two streams overlap.  Nothing may read such a gradient on the calling stream before the
# join, and autograd does read a returned gradient right away when it has to ADD it (a weight used through several column slices, or
# a parameter whose .grad already exists).  So only the shared-MLP chain does this, for weights that are leaves used exactly once and
# whose .grad was None at forward time (autograd then just stores the tensor); the join is an end-of-backward callback of the autograd
# engine, so .grad is complete on the caller's stream when backward() returns.  Fork and join are plain stream waits: they are captured
# with the step under a hipGraph.  Measured: DESIGN.md section 5.
DW_SIDE_STREAM = os.environ.get('MVP_DW_SIDE_STREAM', '1') != '0'
LINEAR_ASIDE = os.environ.get('MVP_LINEAR_ASIDE', '1') != '0'
DW_FINISH_ON_LOAD = os.environ.get('MVP_DW_FINISH_ON_LOAD', '1') != '0'  # (A/B: the first layer's finish pass inside its weight-gradient launches)
REL_DW_FUSED = os.environ.get('MVP_REL_DW_FUSED', '1') != '0'  # (A/B: FeatureAggregation's first-layer weight gradient, feature + relation columns, in ONE launch)
REL_DW_SPLIT = os.environ.get('MVP_REL_DW_SPLIT', '1') != '0'  # (the relation columns' weight gradient on the calling stream at the end of the backward pass)  # (A/B switch: whole-weight linear layers' weight / bias gradients beside the chain)
_EXP_SKIP_DW = os.environ.get('MVP_EXP_SKIP_DW', '0') == '1'


class WeightUse:
    """One forward use of some conv weights whose gradient kernels may run on the side stream.  That is only safe while autograd has
    nothing to ADD the returned gradient to before the end-of-backward join: the weight is used by exactly ONE pending forward (two
    forwards before one backward, forward-forward-backward-backward micro-batching or a Parameter shared by two layers make the engine
    sum the gradients on the calling stream while the side stream still writes them), its .grad is None when the backward runs, and
    it carries no hooks (post-accumulate hooks such as DDP's read the gradient at once).  Every use registers a weak reference on its
    weights; a use that finds another PENDING one (alive, its backward not run yet) marks BOTH as shared, for good.  Checked again at
    backward time (`aside_ok`); the backward marks its use `done` (the autograd graph, and with it this object, may live on in the
    caller's `preds` long after)."""
    __slots__ = ('weights', 'shared', 'done', '__weakref__')

    def __init__(self, weights):
        self.weights = list(weights)
        self.shared = False
        self.done = False
        me = weakref.ref(self)
        for w in self.weights:
            live = [r for r in w.__dict__.get('_mvp_uses', ()) if r() is not None and not r().done]
            for r in live:
                r().shared = True
            if live:
                self.shared = True
            live.append(me)
            w.__dict__['_mvp_uses'] = live

    def aside_ok(self):
        return not self.shared and all(w.is_leaf and w.grad is None and not w._backward_hooks and not w._post_accumulate_grad_hooks
                                       for w in self.weights)


class SideStream:
    def __init__(self):
        self.streams = {}   # device -> (side stream, fork event)
        self.open = set()
        self.keep = []      # what the side-stream kernels touch, alive until the join (instead of record_stream per tensor)
        self.task = None    # autograd graph task the join callback is queued on

    def _join(self):
        for dev in list(self.open):
            torch.cuda.current_stream(dev).wait_stream(self.streams[dev][0])
        self.open.clear()
        self.keep.clear()  # freed behind the join on the calling stream: the allocator hands the blocks to later work of that stream only

    def run(self, dev, name, args, tensors, prec=None):
        """Library entry point `name(*args, stream)` on the side stream, after everything queued so far on the current one;
        `tensors` = what it touches.  Only inside an autograd backward pass (the join is queued as its final callback).
        prec: (precision, precision_backward) as arguments of the call (_lib.call)."""
        st = self.streams.get(dev)
        if st is None:
            st = self.streams[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Event())
        task = torch._C._current_graph_task_id()
        if task != self.task or not self.open:
            # one join per backward pass (keyed on the graph task: a backward that raised leaves `open` non-empty, and the next
            # pass must still get its own callback; _join is idempotent)
            torch.autograd.Variable._execution_engine.queue_callback(self._join)
            self.task = task
        self.open.add(dev)
        side, fork = st
        fork.record(torch.cuda.current_stream(dev))
        side.wait_event(fork)
        if dev.index == torch.cuda.current_device():
            L.call_on(side, name, *args, prec=prec)
        else:
            with torch.cuda.stream(side):
                L.call(name, tensors[0], *args, prec=prec)
        self.keep.extend(tensors)


side_stream = SideStream()


class WeightGradSink:
    """ONE gradient buffer for a conv weight that is used through several column slices in one forward (the linear-first
    factorisations in pn2.SetAbstraction / FeaturePropagation: feature columns per point, coordinate or skip columns elsewhere).
    Every user adds its columns into the same zeroed (C_out, C_tot) buffer and the user whose backward runs LAST hands it to autograd:
    one gradient per weight instead of one full-size zero-filled tensor per slice plus autograd's additions -- and, as nothing reads
    the buffer before the end of backward, the slice kernels can run on the side stream (`aside`, same conditions as SideStream)."""

    def __init__(self, weight, users):
        self.shape = tuple(weight.shape)
        self.numel = weight.numel()
        self.users = self.left = users
        self.buf = None
        self.use = WeightUse([weight]) if DW_SIDE_STREAM else None
        self.aside = None  # decided by the first run() of a backward pass, the same for all the weight's slices

    def buffer(self, dev):
        if self.buf is None:
            self.buf = zero_pool.zeros(self.numel + 4, torch.float32, dev)  # + 4: slack for the 4-column coordinate operand
        return self.buf

    def run(self, dev, name, args, tensors, prec=None):
        if self.aside is None:
            self.aside = bool(DW_SIDE_STREAM and self.use is not None and self.use.aside_ok())
        if self.aside:
            side_stream.run(dev, name, args, [self.buf] + list(tensors), prec=prec)
        else:
            L.call(name, tensors[0], *args, prec=prec)

    def done(self):
        """-> the gradient when this was the last user, else None."""
        self.left -= 1
        if self.left > 0:
            return None
        g = self.buf[:self.numel].view(self.shape)
        self.buf, self.left, self.aside = None, self.users, None
        if self.use is not None:
            self.use.done = True
        return g


class EvalInvStd:
    """1/sqrt(running_var + eps) of ALL BatchNorm layers of a model in two multi-tensor launches per eval-mode forward,
    instead of an add and an rsqrt launch per layer (50 launches of ~4 us each in PN2SSG, 5 % of the forward).  Filled by
    the top-level model forward (`with eval_invstd.scope(model)`), looked up by running_var storage address; a miss falls
    back to the per-layer computation.  Same arithmetic (fp32 add, fp32 rsqrt), so results are bit-identical."""

    def __init__(self):
        self.table = {}
        self.depth = 0

    class _Scope:
        def __init__(self, owner, model):
            self.owner, self.model = owner, model

        def __enter__(self):
            o = self.owner
            if o.depth == 0 and not self.model.training:
                by_eps = {}
                for m in self.model.modules():
                    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.running_var is not None and m.running_var.is_cuda:
                        by_eps.setdefault(float(m.eps), []).append(m.running_var)
                for eps, rvs in by_eps.items():
                    inv = torch._foreach_add(rvs, eps)
                    torch._foreach_rsqrt_(inv)
                    for rv, t in zip(rvs, inv):
                        o.table[(rv.data_ptr(), eps)] = t
            o.depth += 1

        def __exit__(self, *exc):
            o = self.owner
            o.depth -= 1
            if o.depth == 0:
                o.table.clear()

    def scope(self, model):
        return EvalInvStd._Scope(self, model)

    def get(self, running_var, eps):
        t = self.table.get((running_var.data_ptr(), float(eps)))
        return t if t is not None else torch.rsqrt(running_var + eps)


eval_invstd = EvalInvStd()


def build_csr(index, N, sorted=None):
    """index (B, ...) int64 positions into N points -> (offsets (B,N+1) int32, slots (B,E) int32): for every point the list of
    flattened positions that read it (mvp_csr_build_i64).  Turns the scatter-add backward of a gather into a gather.
    sorted (default: the reproducible mode, _lib.DW_WORKSPACE): every list ascending (mvp_csr_build_sorted_i64) -- the gather then adds in
    the same order in every run; unsorted the build is ~45 % shorter."""
    B = index.size(0)
    flat = index.reshape(B, -1)
    E = flat.size(1)
    dev = index.device
    offsets = torch.empty((B, N + 1), dtype=torch.int32, device=dev)
    slots = torch.empty((B, E), dtype=torch.int32, device=dev)
    cursor = torch.empty((B, N), dtype=torch.int32, device=dev)
    name = 'mvp_csr_build_sorted_i64' if (L.DW_WORKSPACE if sorted is None else sorted) else 'mvp_csr_build_i64'
    L.call(name, flat, L.ptr(flat), B, E, N, L.ptr(offsets), L.ptr(slots), L.ptr(cursor))
    return offsets, slots


def geom_sums(offsets, slots, xyz, centre, K):
    """Geometry-only sums of a set-abstraction level for the per-point first-layer passes (mvp_sa_geom_sums_f32, csrc/sa_train.hip):
    offsets (B,N+1), slots (B,M*K) = build_csr(ball index, N), xyz (B,N,3), centre (B,M,3) -> dsum (B,N,4) float32 = per point (sum of the
    centred coordinates of the rows that gathered it, their count), gsum (16) float64 = first and second moments of the centred
    coordinates over all rows.  Coordinates only: part of the geometry plan."""
    L.require_gpu(offsets, slots, xyz, centre)
    B, N, _ = xyz.shape
    dsum = torch.empty((B, N, 4), dtype=torch.float32, device=xyz.device)
    gsum = torch.zeros(16, dtype=torch.float64, device=xyz.device)
    L.call('mvp_sa_geom_sums_f32', xyz, L.ptr(offsets), L.ptr(slots), L.ptr(xyz), L.ptr(centre), B, N, centre.size(1), K, L.ptr(dsum), L.ptr(gsum))
    return dsum, gsum


def knn3_weights(query, key, eps=1e-10):
    """query (B,N1,3), key (B,N2,3) float32 -> index (B,N1,3) int64, weight (B,N1,3): the 3 nearest keys and FeatureInterpolator's
    inverse-squared-distance weights (modules.py:135-140) from one launch (mvp_knn3_weights_f32)."""
    L.require_gpu(query, key)
    q, k = query.contiguous(), key.contiguous()
    B, N1, _ = q.shape
    if k.size(1) < 3:
        raise RuntimeError('knn3_weights: at least 3 keys expected')
    index = torch.empty((B, N1, 3), dtype=torch.int64, device=q.device)
    weight = torch.empty((B, N1, 3), dtype=torch.float32, device=q.device)
    from .ext.ball_query_cuda import BALL_GRID
    nbytes = int(L.lib().mvp_knn3_grid_workspace(B, N1, k.size(1))) if BALL_GRID else 0
    if nbytes > 0:  # many pairs: through the cell grid (csrc/ball_grid.hip), same triples and weights
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        L.call('mvp_knn3_grid_f32', q, L.ptr(q), L.ptr(k), B, N1, k.size(1), float(eps), L.ptr(index), L.ptr(weight), None, L.ptr(ws), nbytes)
    else:
        L.call('mvp_knn3_weights_f32', q, L.ptr(q), L.ptr(k), B, N1, k.size(1), float(eps), L.ptr(index), L.ptr(weight), None)
    return index, weight


class GroupLinRows(torch.autograd.Function):
    """out[b,m,k,:] = zf[b,j,:] + wxyz . (xyz[b,j] - centre[b,m]),  j = index[b,m,k]   (zf may be None).
    want_stat: also return the float64 column sums [sum out | sum out^2] (the BatchNorm batch statistics of this layer,
    computed by the same kernel).  Gradients: zf (row scatter-add) and wxyz (grad_out^T . diff rows); coordinates carry no
    gradient on this path (the reference computes them under no_grad too: fps.py:11-13, modules.py:22-27 on leaf points)."""

    @staticmethod
    def forward(ctx, zf, xyz, centre, wxyz, index, want_stat, offsets=None, slots=None, sink=None):
        ctx.sink = sink
        ctx.prec = L.current_precision()  # backward runs on autograd's thread: it gets the forward's precision as arguments
        # wxyz: a weight whose LAST three columns multiply the coordinates -- a (C,3) matrix, or the layer's whole conv weight
        # (C, C_in + 3 [,1,1]): the slice is taken here and its gradient written into the full-size gradient below, so autograd
        # sees no slicing (which costs a zero fill and a strided copy per slice in backward).
        w_full = wxyz
        ctot_ = w_full.numel() // w_full.size(0)
        if w_full.is_cuda and w_full.dtype == torch.float32:
            wxyz = weight_slices.get(w_full, ctot_ - 3, ctot_, 3)
        else:
            wxyz = w_full.detach().reshape(w_full.size(0), -1)[:, -3:].contiguous()
        L.require_gpu(xyz, centre, wxyz, index)
        B, N, _ = xyz.shape
        _, M, K = index.shape
        C = wxyz.size(0)
        need_w = w_full.requires_grad
        ctx.w_shape = tuple(w_full.shape)
        dev = xyz.device
        out = torch.empty((B, M, K, C), dtype=torch.float32, device=dev)
        diff = torch.empty((B, M, K, 4), dtype=torch.float32, device=dev) if need_w else None
        # want_stat: False / True / the layer's BatchNorm module (training): its finalize (mean | invstd, running statistics) then rides
        # on the statistics reduction of this call and the result carries a third tensor (2C floats: mean | invstd)
        bn = want_stat if isinstance(want_stat, torch.nn.Module) else None
        stat = zero_pool.zeros(2 * C + (1 if bn is not None else 0), torch.float64, dev) if want_stat else None
        partial = torch.empty(L.lib().mvp_group_lin_partial_count(B, C, M, K), dtype=torch.float64, device=dev) if want_stat else None
        mi = torch.empty(2 * C, dtype=torch.float32, device=dev) if bn is not None else None
        if bn is not None:
            L.call('mvp_group_lin_rows_bn_f32', xyz, L.ptr(zf), L.ptr(xyz), L.ptr(centre), L.ptr(wxyz), L.ptr(index), B, N, C, M, K,
                   L.ptr(out), L.ptr(diff), L.ptr(stat), L.ptr(partial), float(bn.eps), 0.1 if bn.momentum is None else float(bn.momentum),
                   L.ptr(mi), L.ptr_at(mi, C), L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked))
        else:
            L.call('mvp_group_lin_rows_f32', xyz, L.ptr(zf), L.ptr(xyz), L.ptr(centre), L.ptr(wxyz), L.ptr(index), B, N, C, M, K,
                   L.ptr(out), L.ptr(diff), L.ptr(stat), L.ptr(partial))
        ctx.has_zf = zf is not None
        if offsets is None and ctx.has_zf and zf.requires_grad:  # not supplied by the geometry plan: build it here
            offsets, slots = build_csr(index, N)
        ctx.save_for_backward(index, diff, offsets, slots)
        ctx.dims = (B, N, C, M, K)
        if want_stat:
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics output in backward
            if mi is not None:
                ctx.mark_non_differentiable(stat, mi)
                return out, stat, mi
            ctx.mark_non_differentiable(stat)
            return out, stat
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, *unused):
        index, diff, offsets, slots = ctx.saved_tensors
        B, N, C, M, K = ctx.dims
        g = grad_out.contiguous()
        gz = gw = None
        if ctx.has_zf and ctx.needs_input_grad[0]:
            gz = torch.empty((B, N, C), dtype=torch.float32, device=g.device)
            if offsets is not None:  # gather through the transposed index: no atomics, no zero fill
                L.call('mvp_gather_rows_backward_csr_f32', g, L.ptr(g), L.ptr(offsets), L.ptr(slots), None, B, N, C, M * K, 1, C, L.ptr(gz))
            else:
                L.call('mvp_group_rows_backward_f32', g, L.ptr(g), L.ptr(index), B, N, C, M, K, C, L.ptr(gz))
        if diff is not None and ctx.needs_input_grad[3]:
            # accumulated into; the other columns stay zero.  The diff rows are [dx,dy,dz,0]: as a 4-column operand (16-byte loads)
            # the zero column adds 0.0 to the first element of the next row -- and, for the last row, to one slack element.
            numel = 1
            for d in ctx.w_shape:
                numel *= d
            ctot = numel // C
            sink = ctx.sink
            if sink is not None:  # the weight's other slices add their columns into the same buffer (WeightGradSink)
                buf = sink.buffer(g.device)
                sink.run(g.device, 'mvp_mlp_weight_grad_f32', (L.ptr(g), L.ptr(diff), B * M * K, C, 4, 4, None, None, None, None,
                                                               L.ptr_at(buf, ctot - 3), ctot), (g, diff), prec=ctx.prec)
                gw = sink.done()
            elif ctot >= 4:
                buf = zero_pool.zeros(numel + 4, torch.float32, g.device)
                gw = buf[:numel].view(ctx.w_shape)
                L.call('mvp_mlp_weight_grad_f32', g, L.ptr(g), L.ptr(diff), B * M * K, C, 4, 4, None, None, None, None,
                       L.ptr_at(buf, ctot - 3), ctot, prec=ctx.prec)
            else:  # the weight IS the (C,3) coordinate part (no input feature)
                gw4 = zero_pool.zeros((C, 4), torch.float32, g.device)
                L.call('mvp_mlp_weight_grad_f32', g, L.ptr(g), L.ptr(diff), B * M * K, C, 4, 4, None, None, None, None, L.ptr(gw4), 4, prec=ctx.prec)
                gw = gw4[:, :3].contiguous().view(ctx.w_shape)
        return gz, None, None, gw, None, None, None, None, None


def group_lin_rows(zf, xyz, centre, wxyz, index, want_stat=False, csr=None, sink=None):
    """zf (B,N,C) or None, xyz (B,N,3), centre (B,M,3), wxyz (C,3) -- or the whole conv weight (C, C_in+3[,1,1]) whose last
    three columns are used --, index (B,M,K) -> (B,M,K,C) [, stat (2C) float64].  want_stat = the layer's BatchNorm module (training):
    its finalize runs on the same call's reduction and a third tensor (2C floats: mean | invstd) is returned.
    csr: (offsets, slots) of build_csr(index, N) when the geometry plan already holds it."""
    offsets, slots = csr if csr is not None else (None, None)
    return GroupLinRows.apply(None if zf is None else zf.contiguous(), xyz.contiguous(), centre.contiguous(), wxyz,
                              index.contiguous(), want_stat if isinstance(want_stat, torch.nn.Module) else bool(want_stat), offsets, slots, sink)


class InterpRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature, index, weight):
        L.require_gpu(feature, index, weight)
        B, N1, C = feature.shape
        N2 = index.size(1)
        out = torch.empty((B, N2, C), dtype=torch.float32, device=feature.device)
        L.call('mvp_interp_rows_f32', feature, L.ptr(feature), L.ptr(index), L.ptr(weight), B, N1, C, N2, C, L.ptr(out))
        ctx.save_for_backward(index, weight)
        ctx.dims = (B, N1, C, N2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        index, weight = ctx.saved_tensors
        B, N1, C, N2 = ctx.dims
        g = grad_out.contiguous()
        grad = torch.empty((B, N1, C), dtype=torch.float32, device=g.device)
        L.call('mvp_interp_rows_backward_f32', g, L.ptr(g), L.ptr(index), L.ptr(weight), B, N1, C, N2, C, L.ptr(grad))
        return grad, None, None


def interp_rows(feature, index, weight):
    """feature (B,N1,C), index (B,N2,3) int64, weight (B,N2,3) -> (B,N2,C)."""
    if feature.dtype != torch.float32 or feature.size(2) % 4:
        raise RuntimeError('interp_rows: float32 feature with C % 4 == 0 expected')
    return InterpRows.apply(feature.contiguous(), index.contiguous(), weight.contiguous())


class DeferredFinish:
    """Hand-over between two autograd nodes: a shared-MLP chain whose first layer ran BEFORE it (the linear-first factorisation) ends its
    backward with the BatchNorm-backward finish of that layer, dz_0 -> dy_0 -- a pass over the largest tensor of the level whose only
    consumer, when the level has no skip feature, is the interpolation's gather backward.  With one of these shared by the two nodes the
    chain returns dz_0 UNFINISHED and leaves (y_0, mean, invstd, gamma, the two column sums) here; the interpolation node's backward then
    gathers through mvp_gather_rows_backward_csr_finish_f32, which forms dy_0 while it loads the rows (csrc/rows.hip).
    MVP_DEFER_FINISH=0: the finish pass stays (A/B switch)."""
    ENABLED = os.environ.get('MVP_DEFER_FINISH', '1') != '0'
    __slots__ = ('info', 'accepts')

    def __init__(self):
        self.info = None      # set by the chain's backward: (y0, mean, invstd, gamma, stat, training)
        self.accepts = False  # set by the interpolation node's forward when its backward can take the hand-over (gather path, no skip term)


class ActivationHandOver:
    """Hand-over between two autograd nodes, the other way round from DeferredFinish: a shared-MLP chain (K = 1, no dropout) whose OUTPUT
    a = relu(bn(y_L)) has exactly one consumer, a LinearRows node (the next propagation level's linear-first factorisation applies its first
    layer to this level's output: pn2.FeaturePropagation).  The chain's backward starts with the gradient of that activation and needs,
    before anything else, dz_L = g * [a > 0] and its two BatchNorm-backward column sums -- a pass over (g, y_L) and a reduction launch.
    The input-gradient kernel of the consumer can do both in its epilogue (mvp_mlp_input_grad_f32 with y_prev: the form the chain uses
    BETWEEN its own layers): the chain leaves (y_L, mean, invstd, gamma, beta) here in forward, the consumer's backward takes them, returns
    dz_L instead of g and leaves the sums here; the chain's backward finds them and skips its own pass.  Only for outputs nobody else
    reads (the caller guarantees it: PN2SSG wires the propagation levels, and not when anybody hooks the modules).
    MVP_ACT_HANDOVER=0: the chain's own pass (A/B switch)."""
    ENABLED = os.environ.get('MVP_ACT_HANDOVER', '1') != '0'
    __slots__ = ('info', 'stat')

    def __init__(self):
        self.info = None   # set by the chain's forward: (y_L, mean, invstd, gamma, beta)
        self.stat = None   # set by the consumer's backward: the two column sums of the dz_L it returned


class InterpAddRows(torch.autograd.Function):
    """out = interp(feature; index, weight) (+ add); want_stat: also the float64 column sums [sum out | sum out^2]."""

    @staticmethod
    def forward(ctx, feature, index, weight, add, want_stat, offsets=None, slots=None, defer=None):
        L.require_gpu(feature, index, weight, add)
        B, N1, C = feature.shape
        N2 = index.size(1)
        dev = feature.device
        out = torch.empty((B, N2, C), dtype=torch.float32, device=dev)
        bn = want_stat if isinstance(want_stat, torch.nn.Module) else None  # as GroupLinRows: the BatchNorm finalize rides on the reduction
        stat = zero_pool.zeros(2 * C + (1 if bn is not None else 0), torch.float64, dev) if want_stat else None
        partial = torch.empty(L.lib().mvp_group_lin_partial_count(B, C, N2, 1), dtype=torch.float64, device=dev) if want_stat else None
        mi = torch.empty(2 * C, dtype=torch.float32, device=dev) if bn is not None else None
        if bn is not None:
            L.call('mvp_interp_add_rows_bn_f32', feature, L.ptr(feature), L.ptr(index), L.ptr(weight), L.ptr(add), B, N1, C, N2, L.ptr(out),
                   L.ptr(stat), L.ptr(partial), float(bn.eps), 0.1 if bn.momentum is None else float(bn.momentum), L.ptr(mi), L.ptr_at(mi, C),
                   L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked))
        else:
            L.call('mvp_interp_add_rows_f32', feature, L.ptr(feature), L.ptr(index), L.ptr(weight), L.ptr(add), B, N1, C, N2, L.ptr(out),
                   L.ptr(stat), L.ptr(partial))
        if offsets is None and feature.requires_grad:
            offsets, slots = build_csr(index, N1)
        ctx.save_for_backward(index, weight, offsets, slots)
        ctx.dims = (B, N1, C, N2)
        ctx.has_add = add is not None
        ctx.defer = defer
        if defer is not None:
            defer.accepts = bool(DeferredFinish.ENABLED and add is None and offsets is not None and feature.requires_grad)
        if want_stat:
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics output in backward
            if mi is not None:
                ctx.mark_non_differentiable(stat, mi)
                return out, stat, mi
            ctx.mark_non_differentiable(stat)
            return out, stat
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, *unused):
        index, weight, offsets, slots = ctx.saved_tensors
        B, N1, C, N2 = ctx.dims
        g = grad_out.contiguous()
        grad = None
        if ctx.needs_input_grad[0]:
            grad = torch.empty((B, N1, C), dtype=torch.float32, device=g.device)
            info = None if ctx.defer is None else ctx.defer.info
            if info is not None:
                # grad_out is dz_0, handed over unfinished by the chain behind this node: dy_0 is formed while the rows are gathered
                y0, mean, invstd, gamma, stat, training = info
                ctx.defer.info = None
                L.call('mvp_gather_rows_backward_csr_finish_f32', g, L.ptr(g), L.ptr(y0), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), L.ptr(stat),
                       int(training), L.ptr(offsets), L.ptr(slots), L.ptr(weight), B, N1, C, 3 * N2, 3, C, L.ptr(grad))
            elif offsets is not None:
                L.call('mvp_gather_rows_backward_csr_f32', g, L.ptr(g), L.ptr(offsets), L.ptr(slots), L.ptr(weight), B, N1, C, 3 * N2, 3, C,
                       L.ptr(grad))
            else:
                L.call('mvp_interp_rows_backward_f32', g, L.ptr(g), L.ptr(index), L.ptr(weight), B, N1, C, N2, C, L.ptr(grad))
        elif ctx.defer is not None:
            ctx.defer.info = None
        return grad, None, None, (g if ctx.has_add and ctx.needs_input_grad[3] else None), None, None, None, None


def interp_add_rows(feature, index, weight, add=None, want_stat=False, csr=None, defer=None):
    """feature (B,N1,C), index / weight (B,N2,3), add (B,N2,C) or None -> (B,N2,C) [, stat (2C) float64].
    csr: (offsets, slots) of build_csr(index, N1) when the geometry plan already holds it.
    defer: a DeferredFinish shared with the shared-MLP chain that consumes the output (shared_mlp_rows(..., defer=))"""
    if feature.dtype != torch.float32 or feature.size(2) % 4:
        raise RuntimeError('interp_add_rows: float32 feature with C % 4 == 0 expected')
    offsets, slots = csr if csr is not None else (None, None)
    return InterpAddRows.apply(feature.contiguous(), index.contiguous(), weight.contiguous(), None if add is None else add.contiguous(),
                               want_stat if isinstance(want_stat, torch.nn.Module) else bool(want_stat), offsets, slots, defer)


class BNActRows(torch.autograd.Function):
    """BatchNorm (+ReLU) (+max over K consecutive rows) on y (G*K, C); one fused forward and backward."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, momentum, eps, relu, K):
        L.require_gpu(y, gamma, beta)
        R, C = y.shape
        G = R // K
        dev = y.device
        stat = torch.empty(2 * C, dtype=torch.float64, device=dev)
        out = torch.empty((G, C), dtype=torch.float32, device=dev)
        arg = torch.empty((G, C), dtype=torch.uint8, device=dev) if K > 1 else None
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean = running_mean
            invstd = eval_invstd.get(running_var, eps)
        L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(gamma), L.ptr(beta), G, K, C, int(training), float(eps),
               float(momentum), int(relu), L.ptr(running_mean) if training else None,
               L.ptr(running_var) if training else None, L.ptr(stat), L.ptr(mean), L.ptr(invstd), L.ptr(out), L.ptr(arg),
               L.ptr(_cs_partial(R, C, dev)) if training else None)
        ctx.save_for_backward(y, gamma, beta, mean, invstd, out, arg)
        ctx.cfg = (G, K, C, bool(relu), bool(training))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        y, gamma, beta, mean, invstd, out, arg = ctx.saved_tensors
        G, K, C, relu, training = ctx.cfg
        g = grad_out.contiguous()
        if not training:
            # eval: statistics are constants -> plain affine backward (no batch terms)
            scale = gamma * invstd
            z = ((y - mean) * invstd) * gamma + beta
            if K > 1:
                z = z.view(G, K, C)
                mask = torch.zeros_like(z)
                mask.scatter_(1, arg.long().unsqueeze(1), 1.0)
                dz = g.unsqueeze(1) * mask
                if relu:
                    dz = dz * (out.unsqueeze(1) > 0)
                dz = dz.reshape(G * K, C)
            else:
                dz = g * (z > 0) if relu else g
            xhat = (y - mean) * invstd
            return dz * scale, (dz * xhat).sum(0), dz.sum(0), None, None, None, None, None, None, None
        stat = torch.empty(2 * C, dtype=torch.float64, device=y.device)
        dy = torch.empty_like(y)
        dgb = torch.empty((2, C), dtype=torch.float32, device=y.device)
        L.call('mvp_bn_rows_backward_f32', y, L.ptr(g), L.ptr(out), L.ptr(arg), L.ptr(y), L.ptr(mean), L.ptr(invstd),
               L.ptr(gamma), L.ptr(beta), G, K, C, int(relu), 1, L.ptr(stat), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]),
               L.ptr(_cs_partial(G * K, C, y.device)))
        return dy, dgb[0], dgb[1], None, None, None, None, None, None, None


def bn_act_rows(y, bn, relu=True, K=1):
    """y (G*K, C) float32 -> (G, C): BatchNorm `bn` (an nn.BatchNorm1d/2d module: its weight, bias, running
    statistics, momentum, eps and train/eval state) + optional ReLU + max over each K consecutive rows."""
    if y.dtype != torch.float32 or y.dim() != 2:
        raise RuntimeError('bn_act_rows: (rows, C) float32 expected')
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if bn.momentum is not None:
        momentum = bn.momentum
    elif bn.training and bn.num_batches_tracked is not None:
        momentum = 1.0 / float(bn.num_batches_tracked)  # momentum None = cumulative moving average (nn.BatchNorm); one host read on this rare path
    else:
        momentum = 0.0
    return BNActRows.apply(y.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps,
                           relu, K)


def _partial(R, cols, device):
    """Scratch for the atomics-free statistics reduction of the MFMA kernels (ceil(R/128) x 2 x cols float64)."""
    if R < PARTIAL_MIN_ROWS:
        return None  # few workgroups: fp64 atomics are cheaper than a second launch
    return torch.empty(((R + 127) // 128) * 2 * cols, dtype=torch.float64, device=device)


def _cs_partial(R, C, device):
    """Scratch of the column-statistics passes (mvp_colstats_partial_count): lets them run on up to 2048 workgroups."""
    return torch.empty(L.lib().mvp_colstats_partial_count(R, C), dtype=torch.float64, device=device)


def _bn_backward(dsrc, out, arg, y, mean, invstd, gamma, beta, G, K, C, relu, training):
    """-> dy (G*K,C), dgamma (C), dbeta (C) through mvp_bn_rows_backward_f32."""
    stat = torch.empty(2 * C, dtype=torch.float64, device=y.device)
    dy = torch.empty_like(y)
    dgb = torch.empty((2, C), dtype=torch.float32, device=y.device)
    L.call('mvp_bn_rows_backward_f32', y, L.ptr(dsrc), L.ptr(out), L.ptr(arg), L.ptr(y), L.ptr(mean), L.ptr(invstd),
           L.ptr(gamma), L.ptr(beta), G, K, C, int(relu), int(training), L.ptr(stat), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]),
           L.ptr(_cs_partial(G * K, C, y.device)))
    return dy, dgb[0], dgb[1]


def _bn_apply(y, mean, invstd, gamma, beta, G, K, C, relu, pool_sum=False):
    """act(bn(y)) with GIVEN statistics (+ max, or with pool_sum the sum, over K): the apply-only mode of
    mvp_bn_rows_forward_f32 (no arg-max buffer = sum)."""
    out = torch.empty((G, C), dtype=torch.float32, device=y.device)
    arg = torch.empty((G, C), dtype=torch.uint8, device=y.device) if (K > 1 and not pool_sum) else None
    L.call('mvp_bn_rows_forward_f32', y, L.ptr(y), L.ptr(gamma), L.ptr(beta), G, K, C, 0, 0.0, 0.0, int(relu), None, None, None,
           L.ptr(mean), L.ptr(invstd), L.ptr(out), L.ptr(arg), None)
    return out, arg


class MLPChainRows(torch.autograd.Function):
    """A whole SharedMLP (conv1x1 + BN + ReLU per layer, optional max over K after the last) as ONE autograd
    node on rows.  Forward: one fp32-MFMA kernel per layer (mvp_mlp_forward_f32) that applies the previous
    layer's BN+ReLU while loading and emits this layer's batch statistics from its epilogue, so only the
    pre-BN outputs y_i ever reach HBM.  Backward re-creates each activation from y_i on the fly."""

    @staticmethod
    def forward(ctx, x0, training, K, eps_mom, bn_buffers, first_stat, pool_sum, *params):
        # params = (W_1, gamma_1, beta_1, ..., W_L, gamma_L, beta_L); bn_buffers = [(running_mean, running_var, num_batches_tracked or None)] * L
        nl = len(params) // 3
        R = x0.size(0)
        dev = x0.device
        # the contraction precision of THIS node: what the calling thread's scope / the process default says now, passed to every launch
        # as arguments -- also to the backward's, which autograd issues from its own thread (a forward under
        # _lib.mlp_precision(...) used to get the process default in backward)
        prec = ctx.prec = L.current_precision()
        # pool_sum: plain bool, or a dict of options {'sum', 'drop_p', 'drop_seed' (dropout behind the last layer, K = 1: SharedMLPDO),
        # 'use' (WeightUse: the weight gradients may run on the side stream, see SideStream)}
        opts = pool_sum if isinstance(pool_sum, dict) else {'sum': bool(pool_sum)}
        drop_p, drop_seed = opts.get('drop_p', 0.0), opts.get('drop_seed', 0)
        ctx.dw_use = opts.get('use')
        ctx.defer = opts.get('defer')
        pool_sum = bool(opts.get('sum', False))
        # 'rel': (rel (R,4), first conv weight): the first layer's input is [x0 | rel] without the concatenated tensor -- its weight has
        # x0.size(1) + 4 columns, the last four meet `rel` in the kernel's epilogue (mvp_mlp_forward_rel_bn_f32)
        rel, w0_param = opts.get('rel', (None, None))
        ctx.rel = (rel, w0_param)
        ctx.dropout = (float(drop_p), int(drop_seed))
        ys, means, invstds = [], [], []
        act = (None, None, None, None)
        x = x0
        couts = [x0.size(1) if params[3 * i] is None else params[3 * i].size(0) for i in range(nl)]
        # the kernels ADD their column sums to `stat`: one zeroed arena for the whole chain instead of a memset per layer
        # (+ 1 per layer: the completion counter of the "last workgroup finalizes" reduction lives behind the layer's sums)
        arena = zero_pool.zeros(2 * sum(couts) + nl, torch.float64, dev) if training else None
        wl = params[3 * (nl - 1)]
        pooled = bool(POOL_WITHOUT_Y and training and K == 32 and not pool_sum and nl >= 2 and wl is not None and R % 32 == 0 and R >= 32768 and
                      prec[0] != 0 and wl.size(0) <= min(64, FUSE_BWD_MAX_COUT) and wl.size(1) <= 64 and
                      wl.size(0) % 4 == 0 and wl.size(1) % 4 == 0)
        off = 0
        for i in range(nl):
            w, gamma, beta = params[3 * i], params[3 * i + 1], params[3 * i + 2]
            eps, mom = eps_mom[i]
            stat = arena[off:off + 2 * couts[i] + 1] if training else None
            off += 2 * couts[i] + 1
            if w is None:  # x0 already IS this layer's pre-BN output (the linear part ran before the grouping)
                assert i == 0
                cout = x0.size(1)
                y = x0
                if training and isinstance(first_stat, tuple):  # ... and finalized the BatchNorm on its reduction: (stat, mean | invstd)
                    first_mi = first_stat[1]
                    ys.append(y)
                    means.append(first_mi[:cout])
                    invstds.append(first_mi[cout:])
                    act = (means[-1], invstds[-1], gamma, beta)
                    x = y
                    continue
                if training and first_stat is not None:
                    stat = first_stat  # the grouping kernel already summed the columns
                elif training:
                    L.call('mvp_colstats_f32', y, L.ptr(y), R, cout, L.ptr(stat), L.ptr(_cs_partial(R, cout, dev)))
            else:
                cout, cin = w.size(0), w.size(1)
                if i == 0 and rel is not None:
                    cin_f = x0.size(1)
                    assert cin == cin_f + 4 and act[0] is None
                    wrel = weight_slices.get(w0_param, cin_f, cin, 4)
                    rm, rv, nbt = bn_buffers[i]
                    y = torch.empty((R, cout), dtype=torch.float32, device=dev)
                    if training:
                        mean = torch.empty(cout, dtype=torch.float32, device=dev)
                        invstd = torch.empty(cout, dtype=torch.float32, device=dev)
                        L.call('mvp_mlp_forward_rel_bn_f32', x, L.ptr(x), R, cin_f, cin_f, L.ptr(w), cin, cout, L.ptr(rel), L.ptr(wrel), L.ptr(y),
                               L.ptr(stat), L.ptr(_partial(R, cout, dev)), float(eps), float(mom), L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv),
                               L.ptr(nbt), prec=prec)
                    else:
                        L.call('mvp_mlp_forward_rel_bn_f32', x, L.ptr(x), R, cin_f, cin_f, L.ptr(w), cin, cout, L.ptr(rel), L.ptr(wrel), L.ptr(y),
                               None, None, 0.0, 0.0, None, None, None, None, None, prec=prec)
                        mean, invstd = rm, eval_invstd.get(rv, eps)
                    ys.append(y)
                    means.append(mean)
                    invstds.append(invstd)
                    act = (mean, invstd, gamma, beta)
                    x = y
                    continue
                if pooled and i == nl - 1:
                    # last layer of a set-abstraction MLP: batch statistics + per-ball max / min of the pre-BN output, no (R, cout) tensor
                    G = R // K
                    rm, rv, nbt = bn_buffers[i]
                    mean = torch.empty(cout, dtype=torch.float32, device=dev)
                    invstd = torch.empty(cout, dtype=torch.float32, device=dev)
                    ymax = torch.empty((G, cout), dtype=torch.float32, device=dev)
                    ymin = torch.empty((G, cout), dtype=torch.float32, device=dev)
                    amax = torch.empty((G, cout), dtype=torch.uint8, device=dev)
                    amin = torch.empty((G, cout), dtype=torch.uint8, device=dev)
                    L.call('mvp_mlp_forward_pool_f32', x, L.ptr(x), R, cin, x.size(1), L.ptr(w), cin, cout, L.ptr(act[0]), L.ptr(act[1]),
                           L.ptr(act[2]), L.ptr(act[3]), L.ptr(ymax), L.ptr(ymin), L.ptr(amax), L.ptr(amin), L.ptr(stat),
                           L.ptr(torch.empty(((R + 127) // 128) * 2 * cout, dtype=torch.float64, device=dev)), float(eps), float(mom),
                           L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.ptr(nbt), prec=prec)
                    out = torch.empty((G, cout), dtype=torch.float32, device=dev)
                    arg = torch.empty((G, cout), dtype=torch.uint8, device=dev)
                    ysel = torch.empty((G, cout), dtype=torch.float32, device=dev)
                    L.call('mvp_pool_finalize_f32', ymax, L.ptr(ymax), L.ptr(ymin), L.ptr(amax), L.ptr(amin), L.ptr(mean), L.ptr(invstd),
                           L.ptr(gamma), L.ptr(beta), G, cout, 1, L.ptr(out), L.ptr(arg), L.ptr(ysel))
                    ys.append(ysel)  # stands in for y_L in the saved list: (G, cout), the pre-BN value behind each pooled output
                    means.append(mean)
                    invstds.append(invstd)
                    break
                y = torch.empty((R, cout), dtype=torch.float32, device=dev)
                if training and R > 0:
                    # forward + batch statistics + BatchNorm finalize (mean / invstd / running statistics) in one call: the finalize
                    # rides on the last workgroup of the statistics reduction
                    rm, rv, nbt = bn_buffers[i]
                    mean = torch.empty(cout, dtype=torch.float32, device=dev)
                    invstd = torch.empty(cout, dtype=torch.float32, device=dev)
                    L.call('mvp_mlp_forward_bn_f32', x, L.ptr(x), R, cin, x.size(1), L.ptr(w), cin, cout, L.ptr(act[0]), L.ptr(act[1]),
                           L.ptr(act[2]), L.ptr(act[3]), L.ptr(y), L.ptr(stat), L.ptr(_partial(R, cout, dev)), float(eps), float(mom),
                           L.ptr(mean), L.ptr(invstd), L.ptr(rm), L.ptr(rv), L.ptr(nbt), prec=prec)
                    ys.append(y)
                    means.append(mean)
                    invstds.append(invstd)
                    act = (mean, invstd, gamma, beta)
                    x = y
                    continue
                L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, cin, x.size(1), L.ptr(w), cin, cout, L.ptr(act[0]), L.ptr(act[1]),
                       L.ptr(act[2]), L.ptr(act[3]), None, L.ptr(y), L.ptr(stat), L.ptr(_partial(R, cout, dev) if training else None), prec=prec)
            rm, rv, nbt = bn_buffers[i]
            if training:
                mean = torch.empty(cout, dtype=torch.float32, device=dev)
                invstd = torch.empty(cout, dtype=torch.float32, device=dev)
                L.call('mvp_bn_finalize_f32', y, L.ptr(stat), R, cout, float(eps), float(mom), L.ptr(mean), L.ptr(invstd),
                       L.ptr(rm), L.ptr(rv), L.ptr(nbt))
            else:
                mean, invstd = rm, eval_invstd.get(rv, eps)
            ys.append(y)
            means.append(mean)
            invstds.append(invstd)
            act = (mean, invstd, gamma, beta)
            x = y
        cl = ys[-1].size(1)
        G = R // K
        if not pooled and drop_p > 0:
            assert K == 1
            out, arg = torch.empty((R, cl), dtype=torch.float32, device=dev), None
            L.call('mvp_bn_rows_forward_dropout_f32', x0, L.ptr(ys[-1]), L.ptr(params[-2]), L.ptr(params[-1]), R, cl, 0, 0.0, 0.0, 1, None, None,
                   None, L.ptr(means[-1]), L.ptr(invstds[-1]), L.ptr(out), None, float(drop_p), int(drop_seed))
        elif not pooled:
            out, arg = _bn_apply(ys[-1], means[-1], invstds[-1], params[-2], params[-1], G, K, cl, True, pool_sum)
        ao = opts.get('act_out')
        ctx.act_out = None
        if ao is not None and ActivationHandOver.ENABLED and not pooled and K == 1 and training is not None:
            # (drop_p > 0: the dropout folded into this chain's BatchNorm + ReLU pass -- the consumer regenerates the keep mask from (p, seed))
            ao.info, ao.stat = (ys[-1], means[-1], invstds[-1], params[-2], params[-1], float(drop_p), int(drop_seed)), None
            ctx.act_out = ao
        ctx.first_linear = params[0] is not None
        ctx.pooled = pooled
        ctx.save_for_backward(x0, out, arg, *ys, *means, *invstds, *[p for p in params if p is not None])
        ctx.cfg = (nl, training, K, R)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        nl, training, K, R = ctx.cfg
        saved = ctx.saved_tensors
        x0, out, arg = saved[0], saved[1], saved[2]
        ys = saved[3:3 + nl]
        means = saved[3 + nl:3 + 2 * nl]
        invstds = saved[3 + 2 * nl:3 + 3 * nl]
        params = list(saved[3 + 3 * nl:])
        if not ctx.first_linear:
            params.insert(0, None)
        grads = [None] * (3 * nl)
        g = grad_out.contiguous()
        G = R // K
        # last layer: through max-over-K + ReLU + BN
        cl = ys[-1].size(1)
        pool = None
        wl = params[3 * (nl - 1)]
        # the consumer of this chain's output already applied the ReLU mask and summed the columns (ActivationHandOver): g IS dz_L
        handed = None
        if ctx.act_out is not None and ctx.act_out.stat is not None:
            handed, ctx.act_out.stat = ctx.act_out.stat, None
        last_wide = bool(handed is None and not ctx.pooled and K == 1 and nl >= 2 and wl is not None and g.is_cuda and
                         wide_backward_ok(ctx.prec, R, wl.size(0), wl.size(1), ys[nl - 2].size(1)) and R * cl < 2 ** 32 and
                         aligned16(g, ys[-1], ys[nl - 2]))
        # ... and the last layer in front of a SUM over K (FeatureAggregation, arg is None): the same one-pass backward takes the gradient of the
        # POOLED output and reads row r / K of it (mvp_mlp_layer_backward_wide_pooled_p_f32): no (R, C) gradient tensor, no pass that writes it
        last_wide_sum = bool(WIDE_BWD_POOLED and not ctx.pooled and K > 1 and arg is None and ctx.dropout[0] == 0 and nl >= 2 and wl is not None and
                             g.is_cuda and wide_backward_ok(ctx.prec, R, wl.size(0), wl.size(1), ys[nl - 2].size(1)) and R * cl < 2 ** 32 and
                             aligned16(g, ys[-1], ys[nl - 2]))
        if handed is not None:
            dy, dgam, dbet = None, None, None
        elif ctx.pooled:
            # the last layer's (R, cl) output does not exist: its BatchNorm-backward column sums come from the (G, cl) tensors, dy_L is
            # formed inside the one-kernel layer backward from the re-computed y_L
            ysel = ys[-1]
            stat_l = torch.empty(2 * cl, dtype=torch.float64, device=g.device)
            L.call('mvp_pool_backward_stats_f32', g, L.ptr(g), L.ptr(out), L.ptr(ysel), L.ptr(means[-1]), L.ptr(invstds[-1]), G, cl, 1,
                   L.ptr(stat_l), L.ptr(_cs_partial(G, cl, g.device)))
            pool = (g, out, arg)
            dy, dgam, dbet = None, None, None
        elif last_wide_sum:
            stat_d = torch.empty(2 * cl, dtype=torch.float64, device=g.device)
            L.call('mvp_bn_rows_backward_f32', g, L.ptr(g), None, None, L.ptr(ys[-1]), L.ptr(means[-1]), L.ptr(invstds[-1]), L.ptr(params[-2]),
                   L.ptr(params[-1]), G, K, cl, 1, int(training), L.ptr(stat_d), None, None, None, L.ptr(_cs_partial(R, cl, g.device)))
            dy, dgam, dbet = None, None, None
        elif last_wide:
            # the last layer goes through the one-pass wide backward (mode 2): only the two column sums of dz_L = g * keep * relu'(bn(y_L)) are
            # computed here; the finish, the masks and dy_L happen while that kernel loads its rows (no dy_L tensor, no finish pass)
            stat_d = torch.empty(2 * cl, dtype=torch.float64, device=g.device)
            L.call('mvp_bn_rows_backward_dropout_f32', g, L.ptr(g), L.ptr(ys[-1]), L.ptr(means[-1]), L.ptr(invstds[-1]), L.ptr(params[-2]),
                   L.ptr(params[-1]), R, cl, 1, int(training), L.ptr(stat_d), None, None, None, L.ptr(_cs_partial(R, cl, g.device)),
                   ctx.dropout[0], ctx.dropout[1])
            dy, dgam, dbet = None, None, None
        elif ctx.dropout[0] > 0:  # the keep mask is regenerated from (p, seed) inside the BatchNorm-backward passes
            stat_d = torch.empty(2 * cl, dtype=torch.float64, device=g.device)
            dy = torch.empty_like(ys[-1])
            dgb = torch.empty((2, cl), dtype=torch.float32, device=g.device)
            L.call('mvp_bn_rows_backward_dropout_f32', g, L.ptr(g), L.ptr(ys[-1]), L.ptr(means[-1]), L.ptr(invstds[-1]), L.ptr(params[-2]),
                   L.ptr(params[-1]), R, cl, 1, int(training), L.ptr(stat_d), L.ptr(dy), L.ptr(dgb[0]), L.ptr(dgb[1]),
                   L.ptr(_cs_partial(R, cl, g.device)), ctx.dropout[0], ctx.dropout[1])
            dgam, dbet = dgb[0], dgb[1]
        else:
            dy, dgam, dbet = _bn_backward(g, out, arg, ys[-1], means[-1], invstds[-1], params[-2], params[-1], G, K, cl, True, training)
        dx0 = None
        none4 = (None, None, None, None)
        # `dW` and `stat` are accumulated into by the kernels: two zeroed arenas for the whole chain
        w_numel = [0 if params[3 * i] is None else params[3 * i].numel() for i in range(nl)]
        dw_arena = zero_pool.zeros(sum(w_numel), torch.float32, g.device)
        st_arena = zero_pool.zeros(2 * sum(params[3 * i].size(1) for i in range(1, nl)), torch.float64, g.device)
        dw_off, st_off = 0, 0
        dev = g.device
        split = ctx.prec[0] != 0   # the one-kernel layer backward (mvp_mlp_layer_backward_f32) contracts in split-bf16 only
        # State while walking the layers backwards: `gcur` is either dy_i itself (pending is None) or dz_i = the gradient w.r.t.
        # layer i's ACTIVATION already masked by its ReLU, with `pending` = its two BatchNorm-backward column sums: the "finish"
        # step (dz_i -> dy_i) then happens INSIDE the fused layer kernel, or as its own pass when the layer cannot be fused.
        gcur, pending = dy, None
        if handed is not None:
            gcur, pending = g, handed
        elif pool is not None:
            gcur, pending = None, stat_l
        elif last_wide or last_wide_sum:
            gcur, pending = g, stat_d
        dw_aside = bool(DW_SIDE_STREAM and ctx.dw_use is not None and ctx.dw_use.aside_ok())
        for i in range(nl - 1, -1, -1):
            w = params[3 * i]
            cout = ys[i].size(1)
            need_dz = (i > 0 or ctx.needs_input_grad[0]) and w is not None
            cin = 0 if w is None else w.size(1)
            src = None if w is None else (x0 if i == 0 else ys[i - 1])
            pool_here = pool is not None and i == nl - 1
            rel, w0_param = ctx.rel if i == 0 else (None, None)
            wide = bool(i > 0 and w is not None and not pool_here and rel is None and need_dz and wide_backward_ok(ctx.prec, R, cout, cin, src.size(1)) and
                        aligned16(gcur, ys[i], src))
            fuse = wide or pool_here or (split and w is not None and cout <= FUSE_BWD_MAX_COUT and cin <= FUSE_BWD_MAX_CIN and rel is None and
                                         (not need_dz or cin % 4 == 0) and (i > 0 or src.size(1) == cin or not need_dz))
            dxw = bool(not fuse and i > 0 and w is not None and rel is None and need_dz and not L.DW_WORKSPACE and
                       dx_wide_ok(ctx.prec, R, cout, cin, src.size(1), pending is not None) and aligned16(gcur, ys[i], src) and w.is_contiguous())
            assert wide or not ((last_wide or last_wide_sum) and i == nl - 1)
            if pending is not None and w is None and ctx.defer is not None and ctx.defer.accepts and ctx.defer.info is None:
                # i == 0, x0 was this layer's pre-BN output and its ONLY consumer gathers it (DeferredFinish): dz_0 goes back unfinished, the
                # gather forms dy_0 on load.  The BatchNorm parameter gradients are the two column sums themselves.
                ctx.defer.info = (ys[0], means[0], invstds[0], params[1], pending, training)
                dgb = pending.view(2, cout).to(torch.float32)
                grads[1], grads[2] = dgb[1], dgb[0]
                dx0 = gcur
                break
            if (pending is not None and not fuse and DW_FINISH_ON_LOAD and i == 0 and rel is not None and not need_dz and ctx.prec[0] != 0 and
                    ctx.prec[1] in (1, 3) and cout % 4 == 0 and cout > 32 and src.size(1) > 32 and src.size(1) == w.size(1) - 4 and
                    aligned16(gcur, ys[0]) and max(cout, src.size(1)) >= L.mlp_min_width()):
                # FIRST layer over [x0 | rel] whose input needs no gradient (FeatureAggregation on a frozen 2D branch): dy_0 is needed by the two
                # weight-gradient launches only, and they form it from (dz_0, y_0) while they load it (mvp_mlp_weight_grad_finish_p_f32): no finish
                # pass (95 us at the very end of the training stream, where nothing overlaps it), no (R, C) dy tensor.  The BatchNorm parameter
                # gradients are the two column sums.
                cin_f = src.size(1)
                dw = dw_arena[dw_off:dw_off + cout * cin].view(cout, cin)
                dw_off += cout * cin
                grads[0] = dw
                dgb0 = pending.view(2, cout).to(torch.float32)
                grads[1], grads[2] = dgb0[1], dgb0[0]
                ws_ptr, ws_floats = L.current_dw_workspace(dev)
                if REL_DW_FUSED and ws_ptr is None and 32 < cin_f <= 64 and aligned16(src, rel):
                    # ... and both column groups from ONE launch (mvp_mlp_weight_grad_finish_rel_p_f32: the relation columns as a third operand block):
                    # the two launches below each stream dz_0 and y_0 -- 1.0 GB for what is 0.6 GB, HBM-bound, with nothing left to overlap them.
                    # On the calling stream: nothing else is queued behind it.
                    L.call('mvp_mlp_weight_grad_finish_rel_p_f32', gcur, L.ptr(gcur), L.ptr(ys[0]), L.ptr(means[0]), L.ptr(invstds[0]), L.ptr(params[1]),
                           L.ptr(pending), int(training), L.ptr(src), R, cout, cin_f, cin_f, L.ptr(rel), L.ptr(dw), L.ptr_at(dw, cin_f), cin, *ctx.prec)
                    break
                for xs, ncol, c0 in ((src, cin_f, 0), (rel, 4, cin_f)):
                    fargs = (L.ptr(gcur), L.ptr(ys[0]), L.ptr(means[0]), L.ptr(invstds[0]), L.ptr(params[1]), L.ptr(pending), int(training), L.ptr(xs), R,
                             cout, ncol, ncol, L.ptr_at(dw, c0), cin)
                    if dw_aside and xs is not rel and ws_ptr is None:
                        # (everything the side-stream kernel reads stays alive until the join -- also the statistics and gamma, which this node's saved
                        # tensors would otherwise release to the calling stream's allocator when the node is freed: ADVICE r5)
                        side_stream.run(dev, 'mvp_mlp_weight_grad_finish_p_f32', fargs + (None, 0) + tuple(ctx.prec),
                                        (gcur, ys[0], xs, dw, pending, means[0], invstds[0], params[1]))
                    else:
                        L.call('mvp_mlp_weight_grad_finish_p_f32', gcur, *(fargs + (ws_ptr, ws_floats) + tuple(ctx.prec)))
                break
            if pending is not None and not fuse and not dxw:
                # dz_i -> dy_i as its own pass (also hands back the BatchNorm parameter gradients)
                dyi = torch.empty((R, cout), dtype=torch.float32, device=dev)
                dgb = torch.empty((2, cout), dtype=torch.float32, device=dev)
                L.call('mvp_bn_rows_backward_finish_f32', gcur, L.ptr(gcur), L.ptr(ys[i]), L.ptr(means[i]), L.ptr(invstds[i]), L.ptr(params[3 * i + 1]),
                       L.ptr(params[3 * i + 2]), R, cout, int(training), L.ptr(pending), L.ptr(dyi), L.ptr(dgb[0]), L.ptr(dgb[1]))
                gcur, pending = dyi, None
                dgam, dbet = dgb[0], dgb[1]
            if pending is None:
                grads[3 * i + 1], grads[3 * i + 2] = dgam, dbet
            if dxw:
                # a 256- / 512-wide inner layer: input gradient in one pass (finish of dz_i on load when it is pending, ReLU mask + column sums of
                # layer i - 1 in the epilogue), the weight gradient beside the chain with the same finish on load -- no finish pass, no dy_i tensor
                act = (means[i - 1], invstds[i - 1], params[3 * i - 2], params[3 * i - 1])
                dw = dw_arena[dw_off:dw_off + cout * cin].view(cout, cin)
                dw_off += cout * cin
                grads[3 * i] = dw
                stat = st_arena[st_off:st_off + 2 * cin]
                st_off += 2 * cin
                dz = torch.empty((R, cin), dtype=torch.float32, device=dev)
                fin = pending is not None
                dgb = torch.empty((2, cout), dtype=torch.float32, device=dev) if fin else None
                if fin:
                    wg_name = 'mvp_mlp_weight_grad_finish_act_p_f32'
                    wg_args = (L.ptr(gcur), L.ptr(ys[i]), L.ptr(means[i]), L.ptr(invstds[i]), L.ptr(params[3 * i + 1]), L.ptr(pending), int(training), L.ptr(src),
                               R, cout, cin, src.size(1), L.ptr(act[0]), L.ptr(act[1]), L.ptr(act[2]), L.ptr(act[3]), L.ptr(dw), cin, None, 0) + tuple(ctx.prec)
                    wg_keep = (gcur, ys[i], means[i], invstds[i], params[3 * i + 1], pending, src, dw) + act
                    if dw_aside:
                        side_stream.run(dev, wg_name, wg_args, wg_keep)
                    else:
                        L.call(wg_name, gcur, *wg_args)
                else:
                    wg_args = (L.ptr(gcur), L.ptr(src), R, cout, cin, src.size(1), L.ptr(act[0]), L.ptr(act[1]), L.ptr(act[2]), L.ptr(act[3]), L.ptr(dw), cin)
                    if dw_aside:
                        side_stream.run(dev, 'mvp_mlp_weight_grad_f32', wg_args, (gcur, src, dw) + act, prec=ctx.prec)
                    else:
                        L.call('mvp_mlp_weight_grad_f32', gcur, *wg_args, prec=ctx.prec)
                nbytes = int(L.lib().mvp_mlp_input_grad_wide_workspace_bytes(cout, cin))
                wimg = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                L.call('mvp_mlp_input_grad_wide_f32', gcur, L.ptr(gcur), L.ptr(ys[i]) if fin else None, L.ptr(means[i]) if fin else None,
                       L.ptr(invstds[i]) if fin else None, L.ptr(params[3 * i + 1]) if fin else None, L.ptr(pending), L.ptr(None if dgb is None else dgb[0]),
                       L.ptr(None if dgb is None else dgb[1]), int(training), L.ptr(src), src.size(1), L.ptr(act[0]), L.ptr(act[1]), L.ptr(act[2]), L.ptr(act[3]),
                       L.ptr(w), cin, R, cout, cin, L.ptr(dz), L.ptr(stat), L.ptr(wimg), nbytes, prec=ctx.prec)
                if fin:
                    grads[3 * i + 1], grads[3 * i + 2] = dgb[0], dgb[1]
                gcur, pending = dz, stat
                continue
            if w is None:  # i == 0: x0 was this layer's pre-BN output, its gradient is dy itself
                dx0 = gcur
                break
            act = none4 if i == 0 else (means[i - 1], invstds[i - 1], params[3 * i - 2], params[3 * i - 1])
            dw = dw_arena[dw_off:dw_off + cout * cin].view(cout, cin)
            dw_off += cout * cin
            grads[3 * i] = dw
            stat = None
            if i > 0:
                stat = st_arena[st_off:st_off + 2 * cin]
                st_off += 2 * cin
            dz = torch.empty((R, cin if rel is None else src.size(1)), dtype=torch.float32, device=dev) if need_dz else None
            if rel is not None:
                # first layer over [x0 | rel]: the feature columns and the four relation columns of dW from two launches (beside the chain
                # when allowed), the input gradient (only when x0 needs one) from the feature columns of the weight
                cin_f = src.size(1)
                for xs, ncol, c0 in ((src, cin_f, 0), (rel, 4, cin_f)):
                    wg_args = (L.ptr(gcur), L.ptr(xs), R, cout, ncol, ncol, None, None, None, None, L.ptr_at(dw, c0), cin)
                    # this is the END of the backward pass when x0 needs no gradient (FeatureAggregation on a frozen 2D branch): nothing is left
                    # for the calling stream to do, so only the feature columns go beside it and the four relation columns run ON it -- the
                    # two launches (each streams dy_1 once) overlap instead of queueing on the one side stream (REL_DW_SPLIT: A/B switch)
                    if dw_aside and not (REL_DW_SPLIT and xs is rel and not need_dz):
                        side_stream.run(dev, 'mvp_mlp_weight_grad_f32', wg_args, (gcur, xs, dw), prec=ctx.prec)
                    else:
                        L.call('mvp_mlp_weight_grad_f32', gcur, *wg_args, prec=ctx.prec)
                if need_dz:
                    wf = weight_slices.get(w0_param, 0, cin_f, cin_f)
                    L.call('mvp_mlp_input_grad_f32', gcur, L.ptr(gcur), R, cout, L.ptr(wf), cin_f, None, None, None, None, None, L.ptr(dz), None, None, prec=ctx.prec)
                    dx0 = dz
                break
            if wide:
                # mode 0: gcur is dy_i; 1: dz_i (finish inside); 2: the gradient of the layer's (dropped-out) activation (last layer)
                mode = 0 if pending is None else (2 if ((last_wide or last_wide_sum) and i == nl - 1) else 1)
                pool_k = K if (last_wide_sum and i == nl - 1) else 1
                dgb = torch.empty((2, cout), dtype=torch.float32, device=dev) if pending is not None else None
                ticket = zero_pool.zeros(2, torch.int32, dev)
                ws_ptr, ws_floats = L.current_dw_workspace(dev)
                fin = pending is not None
                L.call('mvp_mlp_layer_backward_wide_pooled_f32', src, L.ptr(gcur), L.ptr(ys[i]) if fin else None, L.ptr(means[i]) if fin else None,
                       L.ptr(invstds[i]) if fin else None, L.ptr(params[3 * i + 1]) if fin else None, L.ptr(params[3 * i + 2]) if mode == 2 else None,
                       L.ptr(pending), L.ptr(None if dgb is None else dgb[0]), L.ptr(None if dgb is None else dgb[1]), int(training), mode, pool_k,
                       float(ctx.dropout[0]) if mode == 2 else 0.0, int(ctx.dropout[1]) if mode == 2 else 0, L.ptr(src), src.size(1), L.ptr(act[0]),
                       L.ptr(act[1]), L.ptr(act[2]), L.ptr(act[3]), L.ptr(w), cin, R, cout, cin, L.ptr(dw), cin, L.ptr(dz), L.ptr(stat), L.ptr(ticket),
                       ws_ptr, ws_floats, prec=ctx.prec)
                if pending is not None:
                    grads[3 * i + 1], grads[3 * i + 2] = dgb[0], dgb[1]
            elif fuse:
                dgb = torch.empty((2, cout), dtype=torch.float32, device=dev) if pending is not None else None
                part = torch.empty(L.lib().mvp_mlp_layer_backward_partial_count(R, cin), dtype=torch.float64, device=dev) if (i > 0 and need_dz) else None
                L.call('mvp_mlp_layer_backward_f32', src, L.ptr(gcur), L.ptr(ys[i]) if (pending is not None and not pool_here) else None,
                       L.ptr(means[i]) if pending is not None else None, L.ptr(invstds[i]) if pending is not None else None,
                       L.ptr(params[3 * i + 1]) if pending is not None else None, L.ptr(pending), L.ptr(None if dgb is None else dgb[0]),
                       L.ptr(None if dgb is None else dgb[1]), int(training), L.ptr(src), src.size(1), L.ptr(act[0]), L.ptr(act[1]), L.ptr(act[2]),
                       L.ptr(act[3]), L.ptr(w), cin, R, cout, cin, L.ptr(dw), cin, L.ptr(dz), L.ptr(stat), L.ptr(part),
                       L.ptr(pool[0]) if pool_here else None, L.ptr(pool[1]) if pool_here else None, L.ptr(pool[2]) if pool_here else None, prec=ctx.prec)
                if pending is not None:
                    grads[3 * i + 1], grads[3 * i + 2] = dgb[0], dgb[1]
            else:
                wg_args = (L.ptr(gcur), L.ptr(src), R, cout, cin, src.size(1), L.ptr(act[0]), L.ptr(act[1]), L.ptr(act[2]), L.ptr(act[3]),
                           L.ptr(dw), cin)
                if _EXP_SKIP_DW:
                    pass  # (timing experiment only: tools/exp/README.md, "what the graph step would cost with the weight gradients for free")
                elif dw_aside:
                    side_stream.run(dev, 'mvp_mlp_weight_grad_f32', wg_args, (gcur, src, dw) + tuple(t for t in act if t is not None), prec=ctx.prec)
                else:
                    L.call('mvp_mlp_weight_grad_f32', gcur, *wg_args, prec=ctx.prec)
                if need_dz and i > 0:
                    # d(input) = dy . W with the previous layer's ReLU mask and BN-backward column sums in the epilogue
                    pm, pi, pg, pb = act
                    L.call('mvp_mlp_input_grad_f32', gcur, L.ptr(gcur), R, cout, L.ptr(w), cin, L.ptr(ys[i - 1]), L.ptr(pm), L.ptr(pi),
                           L.ptr(pg), L.ptr(pb), L.ptr(dz), L.ptr(stat), L.ptr(_partial(R, cin, dev)), prec=ctx.prec)
                elif need_dz:
                    L.call('mvp_mlp_input_grad_f32', gcur, L.ptr(gcur), R, cout, L.ptr(w), cin, None, None, None, None, None, L.ptr(dz), None, None, prec=ctx.prec)
            if i > 0:
                gcur, pending = dz, stat
            elif need_dz:
                dx0 = dz if x0.size(1) == cin else F.pad(dz, (0, x0.size(1) - cin))
        if ctx.dw_use is not None:
            ctx.dw_use.done = True
        return (dx0, None, None, None, None, None, None) + tuple(grads)


class RelationRows(torch.autograd.Function):
    """cat[feature (B,N,k,C), src_xyz (B,N,k,3) - tgt_xyz (B,N,3), squared length] -> (B,N,k,C+4) in one kernel
    (mvp_relation_rows_f32).  Gradient reaches the feature columns only (coordinates carry none on this path)."""

    @staticmethod
    def forward(ctx, feature, src_xyz, tgt_xyz):
        L.require_gpu(feature, src_xyz, tgt_xyz)
        B, N, k, C = feature.shape
        out = torch.empty((B, N, k, C + 4), dtype=torch.float32, device=feature.device)
        L.call('mvp_relation_rows_f32', feature, L.ptr(feature), L.ptr(src_xyz), L.ptr(tgt_xyz), B * N, k, C, L.ptr(out))
        ctx.C = C
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g[..., :ctx.C].contiguous(), None, None


def relation_rows(feature, src_xyz, tgt_xyz):
    return RelationRows.apply(feature.contiguous(), src_xyz.contiguous(), tgt_xyz.contiguous())


class LinearRows(torch.autograd.Function):
    """y = x . W^T (+ bias) on rows with the fp32-MFMA kernels (forward, input gradient, weight gradient)."""

    @staticmethod
    def forward(ctx, x, w_full, bias, c0=0, c1=None, sink=None, act_src=None):
        ctx.sink = sink
        # x is the output of a chain that left its last layer here (ActivationHandOver): the input gradient applies that layer's ReLU mask
        # and sums the columns in its epilogue
        info = None if act_src is None else act_src.info
        ctx.act_src = act_src if (info is not None and ActivationHandOver.ENABLED and x.is_cuda and tuple(info[0].shape) == tuple(x.shape)) else None
        prec = ctx.prec = L.current_precision()
        # w_full (C_out, C_tot[,1[,1]]); columns [c0, c1) multiply x (R, >= c1 - c0 columns; extra columns are zero padding).
        # The slice is copied here (one small kernel) and its gradient is written straight into a full-size zeroed gradient
        # (lddw), so autograd sees no slicing: that costs a zero fill and a strided copy per slice in backward.
        w2 = w_full.detach().reshape(w_full.size(0), -1)
        c1 = w2.size(1) if c1 is None else c1
        R, ldx = x.shape
        cin = c1 - c0
        if w2.is_cuda and w2.dtype == torch.float32:
            w = weight_slices.get(w_full, c0, c1, ldx)  # persistent operand buffer, refreshed once per forward for all slices
        else:
            w = w2[:, c0:c1]
            if ldx != cin:
                w = torch.nn.functional.pad(w, (0, ldx - cin))
            w = w.contiguous()
        L.require_gpu(x, w, bias)
        cout = w.size(0)
        y = torch.empty((R, cout), dtype=torch.float32, device=x.device)
        L.call('mvp_mlp_forward_f32', x, L.ptr(x), R, ldx, ldx, L.ptr(w), ldx, cout, None, None, None, None, L.ptr(bias), L.ptr(y), None, None, prec=prec)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.slice = (c0, cin, tuple(w_full.shape))
        # a layer that uses the WHOLE weight once (the segmentation head's last layer: 262 144 x 128 -> 20) may put its weight gradient
        # beside the chain like the shared-MLP layers do (34 us of the training stream)
        ctx.bias_param = bias if (bias is not None and bias.is_leaf) else None
        ctx.dw_use = WeightUse([w_full]) if (DW_SIDE_STREAM and LINEAR_ASIDE and sink is None and c0 == 0 and c1 == w2.size(1) and ctx.needs_input_grad[1]) else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        R, cin = x.shape
        cout = w.size(0)
        gy = gy.contiguous()
        gx = gw = gb = None
        aside = False
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            src = ctx.act_src
            if src is not None and src.info is not None:
                yp, pm, pi, pg, pb, drop_p, drop_seed = src.info
                stat = zero_pool.zeros(2 * cin, torch.float64, gy.device)
                part = L.ptr(_partial(R, cin, gy.device) if R > 65536 else None)  # (<= 512 tiles: fp64 atomics, no reduction launch)
                if drop_p > 0:  # x is the DROPPED-OUT activation (the segmentation head in front of the logit layer): keep mask regenerated in the epilogue
                    L.call('mvp_mlp_input_grad_dropout_f32', gy, L.ptr(gy), R, cout, L.ptr(w), cin, L.ptr(yp), L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb),
                           drop_p, drop_seed, L.ptr(gx), L.ptr(stat), part, prec=ctx.prec)
                else:
                    L.call('mvp_mlp_input_grad_f32', gy, L.ptr(gy), R, cout, L.ptr(w), cin, L.ptr(yp), L.ptr(pm), L.ptr(pi), L.ptr(pg), L.ptr(pb), L.ptr(gx),
                           L.ptr(stat), part, prec=ctx.prec)
                src.stat = stat
            else:
                L.call('mvp_mlp_input_grad_f32', gy, L.ptr(gy), R, cout, L.ptr(w), cin, None, None, None, None, None, L.ptr(gx), None, None, prec=ctx.prec)
        if ctx.needs_input_grad[1]:
            c0, ncol, shape = ctx.slice
            sink = ctx.sink
            if sink is not None:  # the weight's other slices add their columns into the same buffer (WeightGradSink)
                buf = sink.buffer(gy.device)
                sink.run(gy.device, 'mvp_mlp_weight_grad_f32', (L.ptr(gy), L.ptr(x), R, cout, ncol, cin, None, None, None, None,
                                                                L.ptr_at(buf, c0), sink.numel // cout), (gy, x), prec=ctx.prec)
                gw = sink.done()
            else:
                # (on the calling stream: without a sink, autograd adds the gradients of a weight's several slices straight away)
                gw = zero_pool.zeros(shape, torch.float32, w.device)  # accumulated into; columns outside the slice stay zero
                wg_args = (L.ptr(gy), L.ptr(x), R, cout, ncol, cin, None, None, None, None, L.ptr_at(gw, c0), gw.numel() // cout)
                use = ctx.dw_use
                aside = use is not None and DW_SIDE_STREAM and use.aside_ok()
                if aside:
                    # (gw itself must not be kept: AccumulateGrad takes a gradient over as it is only while nobody else holds the tensor --
                    # otherwise it CLONES it on the calling stream, before the side stream has written it; the parameter's .grad keeps the memory)
                    side_stream.run(gy.device, 'mvp_mlp_weight_grad_f32', wg_args, (gy, x), prec=ctx.prec)
                else:
                    L.call('mvp_mlp_weight_grad_f32', gy, *wg_args, prec=ctx.prec)
                if use is not None:
                    use.done = True
        if ctx.has_bias and ctx.needs_input_grad[2]:
            st = side_stream.streams.get(gy.device)
            bp = ctx.bias_param
            if (aside and st is not None and gy.device in side_stream.open and bp is not None and bp.grad is None and not bp._backward_hooks
                    and not bp._post_accumulate_grad_hooks):
                # the bias gradient (a column sum of the logits' gradient: 20 us + two dispatch gaps) behind the weight gradient on the side
                # stream: output allocated here, on the calling stream, which is also where it is read after the join
                gb = torch.empty(cout, dtype=gy.dtype, device=gy.device)
                with torch.cuda.stream(st[0]):
                    torch.sum(gy, 0, out=gb)
            else:
                gb = gy.sum(0)
        return gx, gw, gb, None, None, None, None


def linear_rows(x, weight, bias=None, cols=None, sink=None, act_src=None):
    """x (R, C_in) float32, weight (C_out, C_in[,1[,1]]), bias (C_out) or None -> (R, C_out).
    cols=(c0, c1): use only those columns of `weight` (x then has c1 - c0 columns, plus optional zero padding).
    act_src: an ActivationHandOver filled by the chain whose output x is (and whose ONLY consumer this call is)."""
    c0, c1 = (0, None) if cols is None else cols
    return LinearRows.apply(x.contiguous(), weight, bias, c0, c1, sink, act_src)


def linear_rows_bf16(x, weight, bias=None, scale=None, shift=None, relu=False):
    """x (R, C_in) bfloat16 rows (row stride a multiple of 8), weight (C_out, C_in[,1[,1]]) float32 master weights -> (R, C_out) bfloat16:
    act((x . bf16(weight)^T + bias) * scale + shift) with fp32 accumulation (mvp_mlp_forward_bf16, csrc/mlp_bf16.hip) -- the reference's
    conv -> BatchNorm -> ReLU layer (common/nn/modules/conv.py:41-51) on bfloat16 activations in inference, the running-statistics
    BatchNorm folded into scale / shift by the caller.  No autograd, not part of the fp32 parity path."""
    if not (x.is_cuda and weight.is_cuda):
        raise RuntimeError('mvpnet_amd ops run on the GPU only; there is no CPU fallback')
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1:
        raise RuntimeError('linear_rows_bf16: x must be a (R, C_in) bfloat16 tensor with unit column stride')
    w = weight.detach().reshape(weight.size(0), -1)
    if w.dtype != torch.float32 or not w.is_contiguous() or w.size(1) != x.size(1):
        raise RuntimeError('linear_rows_bf16: weight must be contiguous float32 (C_out, C_in)')
    R, cin = x.shape
    cout = w.size(0)
    f = lambda t: None if t is None else t.detach().float().contiguous()
    bias, scale, shift = f(bias), f(scale), f(shift)
    y = torch.empty((R, cout), dtype=torch.bfloat16, device=x.device)
    L.call('mvp_mlp_forward_bf16', x, L.ptr(x), R, cin, x.stride(0), L.ptr(w), cin, cout, L.ptr(bias), L.ptr(scale), L.ptr(shift), int(bool(relu)),
           L.ptr(y), cout)
    return y


# Training-mode set-abstraction levels without their (B*M*32, C) tensors (csrc/sa_train.hip): every pass re-creates the ball's rows from
# the per-point tensor zf instead of loading stored activations.  MVP_SA_TRAIN=0: the per-layer path (A/B switch).
SA_TRAIN_FUSED = os.environ.get('MVP_SA_TRAIN', '1') != '0'


# replicas of the per-point statistics pass' fp64 sums (the workgroups' closing atomics queue per address); MVP_SA_STATS1_REPLICAS=0: none (A/B)
STATS1_REPLICAS = int(os.environ.get('MVP_SA_STATS1_REPLICAS', '16'))


def sa_level_train_widths_ok(c1, c2, c3):
    """Widths csrc/sa_train.hip instantiates: C1, C2 <= 64 with C3 <= 64, or the (33..64, 33..64, 65..128) shape of the reference network's level 2."""
    return c1 >= 4 and max(c1, c2) <= 64 and (c3 <= 64 or (c3 <= 128 and min(c1, c2) > 32))


def sa_level_train_ok(zf, mlp, K):
    """True when a level (zf (B,N,C1) per-point first-layer output, 3-layer SharedMLP `mlp` in training mode, K neighbours) runs as
    SALevelTrain: K = 32, widths sa_level_train_widths_ok and multiples of 4, BatchNorm + ReLU everywhere, a split-bf16 contraction."""
    if not (SA_TRAIN_FUSED and zf is not None and zf.is_cuda and zf.dtype == torch.float32 and len(mlp) == 3 and K == 32 and mlp_chain_is_fused(mlp)):
        return False
    if L.DW_WORKSPACE:  # the reproducible mode keeps the per-layer path: its weight gradients go through a workspace + ordered reduction,
        return False    # the fused passes flush theirs (and the coordinate-column sums) with fp32 atomics
    ps = parts(mlp)
    c1, c2, c3 = (q.w.size(0) for q in ps)
    return sa_level_train_widths_ok(c1, c2, c3) and all(q.bn.training for q in ps) and L.current_precision()[0] != 0 and \
        all(q.w.is_contiguous() for q in ps)


class SALevelTrain(torch.autograd.Function):
    """One set-abstraction level in training mode (QueryGrouper + SharedMLP(ndim=2) + max over the neighbours: mvpnet/models/pn2/modules.py:
    20-37,100-108, conv.py:41-51 with batch statistics) WITHOUT any (B*M*32, C) activation tensor: three forward passes (statistics of
    y_1; of y_2; of y_3 + per-ball max / min) and three backward passes (layer 3; layer 2; per point through the transposed index) that
    each re-create the ball's rows from zf, the coordinates and the ball index (csrc/sa_train.hip).  Only dz_2 and dz_1 are stored.
    Gradients: zf, the first layer's coordinate columns (into the weight's gradient sink), W2, W3, the three BatchNorms' parameters."""

    @staticmethod
    def forward(ctx, zf, xyz, centre, index, offsets, slots, dsum, gsum, w1_full, sink, bns, W2, W3, g1, b1, g2, b2, g3, b3):
        L.require_gpu(zf, xyz, centre, index, W2, W3)
        prec = ctx.prec = L.current_precision()
        ctx.sink = sink
        B, N, C1 = zf.shape
        M, K = index.size(1), index.size(2)
        C2, C3 = W2.size(0), W3.size(0)
        dev = zf.device
        ctot = w1_full.numel() // w1_full.size(0)
        ctx.w_shape = tuple(w1_full.shape)
        wxyz = weight_slices.get(w1_full, ctot - 3, ctot, 3)
        nrep = STATS1_REPLICAS
        arena = zero_pool.zeros(2 * (C1 + C2 + C3) + 3 + 3 * C1 + nrep * 5 * C1, torch.float64, dev)
        stat1, stat2, stat3 = arena[:2 * C1 + 1], arena[2 * C1 + 1:2 * (C1 + C2) + 2], arena[2 * (C1 + C2) + 2:2 * (C1 + C2 + C3) + 3]
        zsum = arena[2 * (C1 + C2 + C3) + 3:2 * (C1 + C2 + C3) + 3 + 3 * C1]
        rep1 = arena[2 * (C1 + C2 + C3) + 3 + 3 * C1:]   # replicas of the first pass' sums (mvp_sa_train_stats1_ws_f32)
        mi = torch.empty(2 * (C1 + C2 + C3), dtype=torch.float32, device=dev)
        m1, i1 = mi[:C1], mi[C1:2 * C1]
        m2, i2 = mi[2 * C1:2 * C1 + C2], mi[2 * C1 + C2:2 * (C1 + C2)]
        m3, i3 = mi[2 * (C1 + C2):2 * (C1 + C2) + C3], mi[2 * (C1 + C2) + C3:]
        (rm1, rv1, nb1, e1, mo1), (rm2, rv2, nb2, e2, mo2), (rm3, rv3, nb3, e3, mo3) = bns
        if offsets is None:
            offsets, slots = build_csr(index, N)
        if dsum is None:  # (not supplied by the geometry plan)
            dsum, gsum = geom_sums(offsets, slots, xyz, centre, K)
        # pass 1: statistics of y_1 (+ BatchNorm-1 finalize) per POINT: y_1 is affine in per-point data (csrc/sa_train.hip)
        L.call('mvp_sa_train_stats1_ws_f32', zf, L.ptr(zf), L.ptr(dsum), L.ptr(wxyz), L.ptr(gsum), B, N, M, K, C1, L.ptr(stat1), L.ptr(zsum), float(e1),
               float(mo1), L.ptr(m1), L.ptr(i1), L.ptr(rm1), L.ptr(rv1), L.ptr(nb1), L.ptr(rep1) if nrep else None, rep1.numel())
        level = (L.ptr(zf), L.ptr(xyz), L.ptr(centre), L.ptr(index), L.ptr(wxyz), B, N, M, K, C1, L.ptr(m1), L.ptr(i1), L.ptr(g1), L.ptr(b1), L.ptr(W2), C2)
        # pass 2: statistics of y_2
        L.call('mvp_sa_train_forward_f32', xyz, 2, *level, None, None, None, None, None, 0, L.ptr(stat2), float(e2), float(mo2), L.ptr(m2), L.ptr(i2),
               L.ptr(rm2), L.ptr(rv2), L.ptr(nb2), None, None, None, None, prec=prec)
        # pass 3: statistics of y_3 + per ball max / min of the pre-BN values
        G = B * M
        ymax = torch.empty((G, C3), dtype=torch.float32, device=dev)
        ymin = torch.empty((G, C3), dtype=torch.float32, device=dev)
        amax = torch.empty((G, C3), dtype=torch.uint8, device=dev)
        amin = torch.empty((G, C3), dtype=torch.uint8, device=dev)
        L.call('mvp_sa_train_forward_f32', xyz, 3, *level, L.ptr(m2), L.ptr(i2), L.ptr(g2), L.ptr(b2), L.ptr(W3), C3, L.ptr(stat3), float(e3), float(mo3),
               L.ptr(m3), L.ptr(i3), L.ptr(rm3), L.ptr(rv3), L.ptr(nb3), L.ptr(ymax), L.ptr(ymin), L.ptr(amax), L.ptr(amin), prec=prec)
        out = torch.empty((G, C3), dtype=torch.float32, device=dev)
        arg = torch.empty((G, C3), dtype=torch.uint8, device=dev)
        ysel = torch.empty((G, C3), dtype=torch.float32, device=dev)
        L.call('mvp_pool_finalize_f32', ymax, L.ptr(ymax), L.ptr(ymin), L.ptr(amax), L.ptr(amin), L.ptr(m3), L.ptr(i3), L.ptr(g3), L.ptr(b3), G, C3, 1,
               L.ptr(out), L.ptr(arg), L.ptr(ysel))
        ctx.save_for_backward(zf, xyz, centre, index, offsets, slots, dsum, gsum, zsum, wxyz, mi, g1, b1, g2, b2, g3, W2, W3, out, arg, ysel)
        ctx.dims = (B, N, M, K, C1, C2, C3)
        return out.view(B, M, C3)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        zf, xyz, centre, index, offsets, slots, dsum, gsum, zsum, wxyz, mi, g1, b1, g2, b2, g3, W2, W3, out, arg, ysel = ctx.saved_tensors
        B, N, M, K, C1, C2, C3 = ctx.dims
        prec = ctx.prec
        dev = zf.device
        G, R = B * M, B * M * K
        g = grad_out.contiguous().view(G, C3)
        m1, i1 = mi[:C1], mi[C1:2 * C1]
        m2, i2 = mi[2 * C1:2 * C1 + C2], mi[2 * C1 + C2:2 * (C1 + C2)]
        m3, i3 = mi[2 * (C1 + C2):2 * (C1 + C2) + C3], mi[2 * (C1 + C2) + C3:]
        stat3 = torch.empty(2 * C3, dtype=torch.float64, device=dev)
        L.call('mvp_pool_backward_stats_f32', g, L.ptr(g), L.ptr(out), L.ptr(ysel), L.ptr(m3), L.ptr(i3), G, C3, 1, L.ptr(stat3),
               L.ptr(_cs_partial(G, C3, dev)))
        st = zero_pool.zeros(2 * (C2 + C1), torch.float64, dev)
        stat2, stat1 = st[:2 * C2], st[2 * C2:]
        dw = zero_pool.zeros(C3 * C2 + C2 * C1 + 16 * 4 * C1, torch.float32, dev)
        dW3, dW2, tsum = dw[:C3 * C2].view(C3, C2), dw[C3 * C2:C3 * C2 + C2 * C1].view(C2, C1), dw[C3 * C2 + C2 * C1:]
        dgb = torch.empty(2 * (C1 + C2 + C3), dtype=torch.float32, device=dev)
        dg1, db1 = dgb[:C1], dgb[C1:2 * C1]
        dg2, db2 = dgb[2 * C1:2 * C1 + C2], dgb[2 * C1 + C2:2 * (C1 + C2)]
        dg3, db3 = dgb[2 * (C1 + C2):2 * (C1 + C2) + C3], dgb[2 * (C1 + C2) + C3:]
        level = (L.ptr(zf), L.ptr(xyz), L.ptr(centre), L.ptr(index), L.ptr(wxyz), B, N, M, K, C1, L.ptr(m1), L.ptr(i1), L.ptr(g1), L.ptr(b1), L.ptr(W2), C2,
                 L.ptr(m2), L.ptr(i2), L.ptr(g2), L.ptr(b2))
        dz2 = torch.empty((R, C2), dtype=torch.float32, device=dev)
        L.call('mvp_sa_train_backward_f32', zf, 3, *level, L.ptr(W3), C3, L.ptr(m3), L.ptr(i3), L.ptr(g3), L.ptr(stat3), L.ptr(dg3), L.ptr(db3), 1, None,
               L.ptr(g), L.ptr(out), L.ptr(arg), L.ptr(dW3), C2, L.ptr(dz2), L.ptr(stat2), None, prec=prec)
        dz1 = torch.empty((R, C1), dtype=torch.float32, device=dev)
        L.call('mvp_sa_train_backward_f32', zf, 2, *level, None, 0, L.ptr(m2), L.ptr(i2), L.ptr(g2), L.ptr(stat2), L.ptr(dg2), L.ptr(db2), 1, L.ptr(dz2),
               None, None, None, L.ptr(dW2), C1, L.ptr(dz1), L.ptr(stat1), L.ptr(tsum), prec=prec)
        del dz2
        # pass 1: per point through the transposed index; the coordinate columns' gradient goes straight into the weight's full-size gradient
        numel = 1
        for d in ctx.w_shape:
            numel *= d
        ctot = numel // C1
        sink = ctx.sink
        buf = sink.buffer(dev) if sink is not None else zero_pool.zeros(numel + 4, torch.float32, dev)
        gz = torch.empty((B, N, C1), dtype=torch.float32, device=dev)
        L.call('mvp_sa_train_backward1_f32', zf, L.ptr(dz1), L.ptr(offsets), L.ptr(slots), L.ptr(zf), L.ptr(dsum), L.ptr(wxyz), L.ptr(tsum), L.ptr(zsum),
               L.ptr(gsum), B, N, M, K, C1, L.ptr(m1), L.ptr(i1), L.ptr(g1), L.ptr(stat1), 1, L.ptr(dg1), L.ptr(db1), L.ptr(gz), L.ptr_at(buf, ctot - 3), ctot)
        gw1 = sink.done() if sink is not None else buf[:numel].view(ctx.w_shape)
        return (gz if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None, gw1, None, None, dW2, dW3, dg1, db1, dg2, db2, dg3, db3)


def sa_level_train(zf, xyz, centre, index, mlp, csr=None, sink=None):
    """zf (B,N,C1) = the level's first-layer feature columns applied per point, xyz (B,N,3), centre (B,M,3), index (B,M,32) int64,
    mlp = the level's 3-layer SharedMLP in training mode (sa_level_train_ok) -> pooled feature (B,M,C3).  csr = (offsets, slots[, dsum, gsum])
    of build_csr(index, N) [and geom_sums] when the geometry plan holds them; sink = the WeightGradSink of the first layer's weight."""
    offsets, slots = (csr[0], csr[1]) if csr is not None else (None, None)
    dsum, gsum = (csr[2], csr[3]) if csr is not None and len(csr) >= 4 else (None, None)
    q1, q2, q3 = parts(mlp)
    bns = [(q.rm, q.rv, q.nbt, q.bn.eps, 0.1 if q.bn.momentum is None else q.bn.momentum) for q in (q1, q2, q3)]
    W2 = q2.w.reshape(q2.w.size(0), -1)
    W3 = q3.w.reshape(q3.w.size(0), -1)
    return SALevelTrain.apply(zf.contiguous(), xyz.contiguous(), centre.contiguous(), index.contiguous(), offsets, slots, dsum, gsum, q1.w, sink, bns,
                              W2, W3, q1.gamma, q1.beta, q2.gamma, q2.beta, q3.gamma, q3.beta)


SA_FUSED_EVAL = os.environ.get('MVP_SA_FUSED', '1') != '0'


def sa_fused_eval(zf, xyz, centre, index, mlp):
    """A whole set-abstraction level in ONE kernel, inference mode (mvp_sa_fused_forward_f32, csrc/sa_fused.hip): zf (B,N,C1) = the first
    layer's feature columns applied per point (or None), xyz (B,N,3), centre (B,M,3), index (B,M,32), mlp = the level's 3-layer
    SharedMLP in eval mode -> (B,M,C3), or None when the level does not qualify (the caller then runs the per-layer kernels)."""
    if not (SA_FUSED_EVAL and len(mlp) == 3 and mlp_chain_is_fused(mlp) and index.size(2) == 32 and L.get_mlp_precision() != 'fp32'):
        return None
    c1, c2, c3 = (l.conv.weight.size(0) for l in mlp)
    if c1 > 64 or c2 > 64 or c3 > 128 or c1 % 4 or c2 % 4 or c3 % 4 or any(l.bn.training for l in mlp):
        return None
    blocks = lambda c: 1 if c <= 32 else 2 if c <= 64 else 4
    if (blocks(c1), blocks(c2), blocks(c3)) not in {(1, 1, 1), (1, 1, 2), (1, 2, 1), (1, 2, 2), (1, 2, 4), (2, 1, 1), (2, 1, 2), (2, 2, 1), (2, 2, 2),
                                                     (2, 2, 4)}:
        return None  # not an instantiated (C1, C2, C3) block shape of csrc/sa_fused.hip
    L.require_gpu(xyz, centre, index)
    B, N, _ = xyz.shape
    M = centre.size(1)
    bn = []
    for l in mlp:
        bn += [l.bn.running_mean, eval_invstd.get(l.bn.running_var, l.bn.eps), l.bn.weight, l.bn.bias]
    w1 = mlp[0].conv.weight
    ctot_ = w1.numel() // c1
    wxyz = weight_slices.get(w1, ctot_ - 3, ctot_, 3)
    w2 = mlp[1].conv.weight.detach().reshape(c2, c1).contiguous()
    w3 = mlp[2].conv.weight.detach().reshape(c3, c2).contiguous()
    out = torch.empty((B, M, c3), dtype=torch.float32, device=xyz.device)
    zfc = None if zf is None else zf.contiguous()
    L.call('mvp_sa_fused_forward_f32', xyz, L.ptr(zfc), L.ptr(xyz.contiguous()), L.ptr(centre.contiguous()), L.ptr(index.contiguous()), L.ptr(wxyz),
           B, N, M, 32, c1, *[L.ptr(t.detach().contiguous()) for t in bn[0:4]], L.ptr(w2), c2, *[L.ptr(t.detach().contiguous()) for t in bn[4:8]],
           L.ptr(w3), c3, *[L.ptr(t.detach().contiguous()) for t in bn[8:12]], L.ptr(out), None, prec=L.current_precision())
    return out


def mlp_chain_is_fused(mlp, dropout_p=0.0, ps=None):
    """True when `mlp` (a SharedMLP) runs as ONE MLPChainRows node: conv without bias + BatchNorm with running statistics + ReLU
    in every layer, widths the rows kernels tile (C % 4 == 0 and C / 4 divides 256), dropout only behind a single layer.
    ps: parts(mlp) when the caller already has them."""
    ps = parts(mlp) if ps is None else ps
    if not (dropout_p == 0 or len(ps) == 1):
        return False
    for q in ps:
        # (bn.momentum None = PyTorch's cumulative moving average, factor 1 / num_batches_tracked: the kernels take one fixed factor, so such
        # layers keep the per-layer torch path -- ADVICE r4)
        if q.bn is None or q.relu is None or q.bias is not None or q.rm is None or q.bn.momentum is None:
            return False
        c = q.w.size(0)
        if c % 4 or 256 % (c // 4):
            return False
    return True


def relation4_rows(src_xyz, tgt_xyz):
    """src_xyz (B,N,k,3), tgt_xyz (B,N,3) -> (B,N,k,4) = [src - tgt | squared length]: FeatureAggregation's relation columns alone
    (mvpnet_3d.py:55-56; coordinates carry no gradient on this path)."""
    L.require_gpu(src_xyz, tgt_xyz)
    B, N, k, _ = src_xyz.shape
    with torch.no_grad():
        s, t = src_xyz.contiguous(), tgt_xyz.contiguous()
        out = torch.empty((B, N, k, 4), dtype=torch.float32, device=s.device)
        L.call('mvp_relation4_rows_f32', s, L.ptr(s), L.ptr(t), B * N, k, L.ptr(out))
    return out


def shared_mlp_rows(x, mlp, K=1, dropout_p=0.0, training=False, first_done=False, first_stat=None, reduce='max', rel=None, dropout_last_only=False,
                    defer=None, act_out=None):
    """Apply a SharedMLP (stack of pointwise conv + BN + ReLU, common/nn/modules/mlp.py:38-75) to a row
    matrix x (R, ld >= C_in; extra columns are zero padding).  The last layer also takes the max over each
    K consecutive rows when K > 1 (SetAbstraction, pn2/modules.py:107-108), or their sum with reduce='sum'
    (FeatureAggregation, mvpnet_3d.py:40-41,59).
    first_done=True: x already is the first layer's conv output (the linear part was applied per point before
    the grouping, see SetAbstraction.forward_rows); only its BatchNorm + ReLU and the remaining layers run here.
    rel (R,4): the first layer's input is [x | rel] -- its weight has x.size(1) + 4 columns -- without the concatenated tensor (fused
    chains only: FeatureAggregation)."""
    n = len(mlp)
    # dropout follows EVERY layer of a SharedMLPDO (mlp.py:86-92): a single-layer chain can still be fused, dropout on its output.
    # dropout_last_only: `mlp` is a SharedMLP followed by a single-layer SharedMLPDO run as ONE chain (PN2SSG: the last feature-propagation
    # MLP + the segmentation head, whose only consumer it is) -- the dropout belongs to the last layer alone
    ps = parts(mlp)
    fused = mlp_chain_is_fused(mlp, 0.0 if dropout_last_only else dropout_p, ps) and K <= 255
    assert fused or not dropout_last_only, 'dropout_last_only needs the fused chain'
    if fused:
        bn_training = ps[0].bn.training
        params, buffers, eps_mom = [], [], []
        for li, q in enumerate(ps):
            w = None if (first_done and li == 0) else (q.w if q.w.dim() == 2 else q.w.reshape(q.w.size(0), -1))
            params += [w, q.gamma, q.beta]
            buffers.append((q.rm, q.rv, q.nbt if bn_training else None))
            eps_mom.append((q.bn.eps, q.bn.momentum))
        opts = {'sum': bool(reduce == 'sum' and K > 1)}
        if rel is not None:
            assert not first_done and rel.dim() == 2 and rel.size(1) == 4 and rel.size(0) == x.size(0)
            opts['rel'] = (rel.contiguous(), ps[0].w)
        if defer is not None and first_done and bn_training:
            opts['defer'] = defer   # (rows.DeferredFinish: the finish of the first layer's gradient may be left to the node in front)

        if DW_SIDE_STREAM and torch.is_grad_enabled():  # (a first layer that ran before the grouping has its own use: WeightGradSink)
            opts['use'] = WeightUse([q.w for li, q in enumerate(ps) if not (first_done and li == 0)])
        # dropout behind the (single) layer: folded into the BatchNorm + ReLU passes (mvp_bn_rows_forward_dropout_f32) unless a graph is
        # being captured (the seed would be baked into the capture; torch's own dropout advances its Philox offset per replay)
        fold = dropout_p > 0 and training and K == 1 and 0 < dropout_p < 1 and FUSE_DROPOUT and x.size(0) * ps[-1].w.size(0) < 2 ** 32 and \
            not torch.cuda.is_current_stream_capturing()
        if fold:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())  # torch's CPU generator: follows torch.manual_seed
            opts['drop_p'], opts['drop_seed'] = float(dropout_p), seed & 0x7fffffffffffffff
        # (rows.ActivationHandOver: the output's only consumer masks and sums its gradient -- with a dropout behind the last layer only when it is
        # folded into this chain's kernels, i.e. when the consumer's input IS this node's output; F.dropout below would sit between the two)
        if act_out is not None and K == 1 and torch.is_grad_enabled() and (dropout_p == 0 or not training or fold):
            opts['act_out'] = act_out
        out = MLPChainRows.apply(x.contiguous(), bn_training, K, eps_mom, buffers, first_stat if bn_training else None, opts, *params)
        return F.dropout(out, p=dropout_p, training=training, inplace=False) if (dropout_p > 0 and not fold) else out
    assert not first_done and rel is None, 'first_done / rel need the fused path (BN + ReLU, no bias, no dropout)'
    if K > 255:  # the pooled BatchNorm kernel keeps its arg-max in one byte: pool with torch after a K = 1 pass
        x = shared_mlp_rows(x, mlp, 1, dropout_p, training)
        x = x.view(-1, K, x.size(1))
        return x.sum(dim=1) if reduce == 'sum' else x.max(dim=1)[0]
    for i, layer in enumerate(mlp):
        w = layer.conv.weight.reshape(layer.conv.weight.size(0), -1)  # (C_out, C_in)
        if x.size(1) != w.size(1):
            w = F.pad(w, (0, x.size(1) - w.size(1)))
        y = linear_rows(x, w)
        if layer.bn is not None:
            x = bn_act_rows(y, layer.bn, relu=layer.relu is not None, K=K if (i == n - 1 and reduce != 'sum') else 1)
            if K > 1 and i == n - 1 and reduce == 'sum':
                x = x.view(-1, K, x.size(1)).sum(dim=1)
        else:
            if layer.conv.bias is not None:
                y = y + layer.conv.bias
            x = F.relu(y) if layer.relu is not None else y
            if K > 1 and i == n - 1:
                x = x.view(-1, K, x.size(1)).sum(dim=1) if reduce == 'sum' else x.view(-1, K, x.size(1)).max(dim=1)[0]
        if dropout_p > 0:
            x = F.dropout(x, p=dropout_p, training=training, inplace=False)
    return x
