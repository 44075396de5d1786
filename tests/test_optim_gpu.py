"""mvpnet_amd.optim.FusedAdam (csrc/adam.hip, mvp_adam_step_f32): the reference's optimizer step (torch.optim.Adam built by
common/solver/build.py:7-22, stepped at train_mvpnet_3d.py:176) as ONE launch over all parameter tensors -- checked against
torch.optim.Adam itself, state and checkpoints included."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _params(dev, seed, many):
    g = torch.Generator(device='cpu').manual_seed(seed)
    shapes = [(64, 64, 1), (64,), (3,), (1,), (1000, 33), (5, 7, 3), (128, 259), (4099,), (2048,), (256, 128, 1, 1)]
    if many:
        shapes += [(17 + i,) for i in range(120)]  # more tensors than one launch holds (96)
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
    base = torch.randn(1001, generator=g).to(dev)
    ps.append(torch.nn.Parameter(base[1:]))  # contiguous, but 4 bytes off a 16-byte boundary: the scalar path
    ps.append(torch.nn.Parameter(torch.randn(10, generator=g).to(dev)))  # never receives a gradient
    return ps


@pytest.mark.parametrize('many', [False, True])
@pytest.mark.parametrize('weight_decay', [0.0, 1e-2])
def test_fused_adam_equals_torch_adam(dev, weight_decay, many):
    from mvpnet_amd.optim import FusedAdam
    pa, pb = _params(dev, 5, many), _params(dev, 5, many)
    assert pa[-2].data_ptr() % 16 != 0
    oa = FusedAdam(pa, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    ob = torch.optim.Adam(pb, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    assert isinstance(oa, torch.optim.Adam)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, milestones=[3], gamma=0.1)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[3], gamma=0.1)
    gen = torch.Generator(device='cpu').manual_seed(9)
    for it in range(6):
        for a, b in zip(pa[:-1], pb[:-1]):
            gr = (torch.randn(a.shape, generator=gen) * (10.0 ** (it % 3 - 1))).to(dev)
            a.grad, b.grad = gr.clone(), gr.clone()
        if it == 4:  # a non-contiguous gradient
            pa[4].grad = pa[4].grad.t().contiguous().t()
            assert not pa[4].grad.is_contiguous()
        oa.step()
        ob.step()
        sa.step()
        sb.step()
    assert oa.param_groups[0]['lr'] == ob.param_groups[0]['lr'] == pytest.approx(2e-4)
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    for a, b in zip(pa[:-1], pb[:-1]):
        for k in ('exp_avg', 'exp_avg_sq'):
            np.testing.assert_allclose(oa.state[a][k].cpu().numpy(), ob.state[b][k].cpu().numpy(), rtol=2e-6, atol=2e-6)  # gradients up to ~40: one fp32 rounding of the lerp
        assert float(oa.state[a]['step']) == float(ob.state[b]['step']) == 6.0
    assert len(oa.state[pa[-1]]) == 0  # no gradient, no state -- as in torch


def test_fused_adam_checkpoints_interchange_with_torch_adam(dev):
    """state_dict() of either optimizer loads into the other and training continues on the same trajectory (the reference's Checkpointer
    stores optimizer.state_dict(): common/utils/checkpoint.py:48-53)."""
    from mvpnet_amd.optim import FusedAdam
    pa, pb = _params(dev, 6, False), _params(dev, 6, False)
    oa, ob = FusedAdam(pa, lr=1e-3, weight_decay=1e-3), torch.optim.Adam(pb, lr=1e-3, weight_decay=1e-3)
    gen = torch.Generator(device='cpu').manual_seed(1)

    def run(opts, plists, steps):
        for _ in range(steps):
            grads = [torch.randn(p.shape, generator=gen).to(dev) for p in plists[0][:-1]]
            for opt, ps in zip(opts, plists):
                for p, gr in zip(ps[:-1], grads):
                    p.grad = gr.clone()
                opt.step()

    run((oa, ob), (pa, pb), 3)
    sd_a, sd_b = copy.deepcopy(oa.state_dict()), copy.deepcopy(ob.state_dict())
    assert sd_a['param_groups'][0].keys() == sd_b['param_groups'][0].keys()
    assert set(sd_a['state'][0].keys()) == set(sd_b['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    # cross-load: the fused optimizer continues from torch's state and vice versa
    pc, pd = [torch.nn.Parameter(p.detach().clone()) for p in pb], [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oc, od = FusedAdam(pc, lr=1e-3, weight_decay=1e-3), torch.optim.Adam(pd, lr=1e-3, weight_decay=1e-3)
    oc.load_state_dict(sd_b)
    od.load_state_dict(sd_a)
    run((oa, ob, oc, od), (pa, pb, pc, pd), 2)
    oc.load_state_dict(copy.deepcopy(oc.state_dict()))  # a load in the middle of training replaces the moment tensors the step had cached
    run((oa, ob, oc, od), (pa, pb, pc, pd), 1)
    for a, b, c, d in zip(pa, pb, pc, pd):
        for other in (b, c, d):
            np.testing.assert_allclose(other.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert float(oc.state[pc[0]]['step']) == 6.0


def test_build_optimizer_gives_the_fused_adam_on_the_gpu(dev):
    from mvpnet_amd import config as C
    from mvpnet_amd.optim import FusedAdam
    cfg = C.get_cfg_mvpnet_3d()  # OPTIMIZER.TYPE 'Adam', BASE_LR as in the reference's defaults
    cfg.OPTIMIZER.TYPE = 'Adam'
    model = torch.nn.Linear(8, 8).to(dev)
    opt = C.build_optimizer(cfg, model)
    assert isinstance(opt, FusedAdam) and opt.defaults['lr'] == cfg.OPTIMIZER.BASE_LR


def test_fused_adam_keeps_one_step_count_per_parameter(dev):
    """torch.optim.Adam applies each parameter's OWN 'step' in the bias corrections; a parameter whose gradient was None on
    some iterations (or one that joined later) has a smaller count than the others.  FusedAdam launches once per distinct count."""
    from mvpnet_amd.optim import FusedAdam
    pa, pb = _params(dev, 11, False), _params(dev, 11, False)
    oa = FusedAdam(pa, lr=2e-3, weight_decay=1e-3)
    ob = torch.optim.Adam(pb, lr=2e-3, weight_decay=1e-3)
    gen = torch.Generator(device='cpu').manual_seed(3)
    for it in range(5):
        for k, (a, b) in enumerate(zip(pa[:-1], pb[:-1])):
            if (k == 2 and it in (1, 3)) or (k == 5 and it < 2):  # these two skip iterations: grad None
                a.grad, b.grad = None, None
                continue
            gr = torch.randn(a.shape, generator=gen).to(dev)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    assert float(oa.state[pa[2]]['step']) == float(ob.state[pb[2]]['step']) == 3.0
    assert float(oa.state[pa[5]]['step']) == float(ob.state[pb[5]]['step']) == 3.0
    assert float(oa.state[pa[0]]['step']) == 5.0
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
