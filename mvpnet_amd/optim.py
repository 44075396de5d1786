"""Adam with the whole model's update in ONE launch (csrc/adam.hip, mvp_adam_step_f32).

`FusedAdam` IS a torch.optim.Adam -- same constructor, same param_groups, same per-parameter state ('step', 'exp_avg', 'exp_avg_sq'), so
`state_dict()` / `load_state_dict()` interchange with torch.optim.Adam checkpoints (the reference saves the optimizer through its
Checkpointer, common/utils/checkpoint.py:48-53) and learning-rate schedulers drive it as they drive the reference's
(common/solver/build.py:7-41) -- whose `step()` hands the four pointer lists of all float32 GPU parameters to one kernel instead of
ATen's three multi_tensor_apply launches at the end of the training step's critical stream."""
import ctypes

import torch

from . import _lib as L


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kwargs):
        if amsgrad or kwargs.get('maximize') or kwargs.get('capturable') or kwargs.get('differentiable'):
            raise ValueError('FusedAdam covers plain Adam (no amsgrad / maximize / capturable / differentiable)')
        kwargs.pop('fused', None)
        kwargs.pop('foreach', None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False, **kwargs)
        self._plans = {}  # group index -> cached pointer arrays (parameters and moments do not move; gradients are re-read every step)
        self._step_bufs = {}  # group index -> (host buffer of the group's step counts, its 0-dim views = the states' 'step' tensors)

    def _plan(self, gi, plist):
        key = tuple(id(p) for p in plist)
        plan = self._plans.get(gi)
        if plan is not None and plan['key'] == key and all(p.data_ptr() == q for p, q in zip(plist, plan['pptr'])) \
                and all(self.state[p]['exp_avg'] is m and self.state[p]['exp_avg_sq'] is v for p, m, v in zip(plist, *plan['keep'])):
            return plan  # (load_state_dict replaces the moment tensors: the identity check above then rebuilds the lists)
        n = len(plist)
        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
        moments1 = [self.state[p]['exp_avg'] for p in plist]
        moments2 = [self.state[p]['exp_avg_sq'] for p in plist]
        plan = {'key': key, 'pptr': [p.data_ptr() for p in plist], 'p': arr([p.data_ptr() for p in plist]),
                'm': arr([t.data_ptr() for t in moments1]), 'v': arr([t.data_ptr() for t in moments2]), 'g': (ctypes.c_void_p * n)(),
                'numel': (ctypes.c_int64 * n)(*[p.numel() for p in plist]), 'keep': (moments1, moments2)}
        self._plans[gi] = plan
        return plan

    def _advance_steps(self, gi, plist):
        """+1 on every parameter's step count -> the counts as floats.  Every parameter keeps its OWN host-side float32 step tensor, as
        torch.optim.Adam does (state_dict / checkpoints interchange), but the tensors of a group are 0-dim VIEWS of one host buffer: one
        in-place add and one tolist() per step instead of a 77-tensor foreach add and 77 float() calls (0.3 ms of the step's host time).
        load_state_dict / anybody replacing a state's 'step' is noticed by identity and the buffer rebuilt from the values found."""
        cache = self._step_bufs.get(gi)
        state = self.state
        if cache is None or len(cache[1]) != len(plist) or any(state[p]['step'] is not v for p, v in zip(plist, cache[1])):
            base = torch.tensor([float(state[p]['step']) for p in plist], dtype=torch.float32)
            views = [base[i] for i in range(len(plist))]
            for p, v in zip(plist, views):
                state[p]['step'] = v
            cache = self._step_bufs[gi] = (base, views)
        cache[0].add_(1.0)
        return cache[0].tolist()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) or p.grad.is_sparse:
                    raise RuntimeError('FusedAdam: parameters must be dense contiguous float32 tensors on the GPU (use torch.optim.Adam otherwise)')
                st = self.state[p]
                if len(st) == 0:  # torch.optim.Adam's own lazy state: a float32 step count on the host, zero moments
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            counts = self._advance_steps(gi, plist)
            plan = self._plan(gi, plist)
            grads = []
            g_arr = plan['g']
            for i, p in enumerate(plist):
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                grads.append(g)
                g_arr[i] = g.data_ptr()
            beta1, beta2 = group['betas']
            dev = plist[0].device
            if any(p.device != dev for p in plist):
                raise RuntimeError('FusedAdam: the parameters of a group must live on one device')
            # The bias corrections depend on the parameter's OWN step count (torch.optim.Adam keeps one per parameter): parameters that
            # skipped an iteration (grad None), were added later or came from a checkpoint with unequal counts get their own launch --
            # one launch per DISTINCT count, i.e. one launch in the usual case.
            if all(c == counts[0] for c in counts):
                buckets = [(counts[0], plan['p'], g_arr, plan['m'], plan['v'], plan['numel'], len(plist))]
            else:
                buckets = []
                for c in sorted(set(counts)):
                    ids = [i for i, ci in enumerate(counts) if ci == c]
                    sub = lambda arr, typ: (typ * len(ids))(*[arr[i] for i in ids])
                    buckets.append((c, sub(plan['p'], ctypes.c_void_p), sub(g_arr, ctypes.c_void_p), sub(plan['m'], ctypes.c_void_p),
                                    sub(plan['v'], ctypes.c_void_p), sub(plan['numel'], ctypes.c_int64), len(ids)))
            with torch.cuda.device(dev):
                for count, p_arr, gg_arr, m_arr, v_arr, n_arr, n in buckets:
                    code = L._fn('mvp_adam_step_f32')(p_arr, gg_arr, m_arr, v_arr, n_arr, n, float(group['lr']), float(beta1), float(beta2),
                                                      float(group['eps']), float(group['weight_decay']), count,
                                                      torch.cuda.current_stream(dev).cuda_stream)
                    if code != 0:
                        L.check(code, 'mvp_adam_step_f32')
            del grads
        return loss
