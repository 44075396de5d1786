"""rows.ZeroPool: disjoint zeroed views of one per-step buffer, with `torch.zeros` as the fallback (host logic, no GPU)."""
import torch

from mvpnet_amd.rows import ZeroPool


def test_first_step_falls_back_then_pools():
    pool, dev = ZeroPool(), torch.device('cpu')
    with pool.step(dev):
        a = pool.zeros(10, torch.float64, dev)
        b = pool.zeros((3, 5), torch.float32, dev)
    assert pool.buf is None and a.shape == (10,) and b.shape == (3, 5)         # nothing known yet: plain zeros
    with pool.step(dev):                                                        # sized by the first step's demand
        assert pool.capacity == 512 and pool.buf is not None
        a = pool.zeros(10, torch.float64, dev)
        b = pool.zeros((3, 5), torch.float32, dev)
        c = pool.zeros(7, torch.float32, dev)                                   # does not fit: fallback, still zeros
    base = pool.buf.untyped_storage().data_ptr()
    assert a.untyped_storage().data_ptr() == base and b.untyped_storage().data_ptr() == base
    assert c.untyped_storage().data_ptr() != base
    assert a.dtype == torch.float64 and b.dtype == torch.float32 and b.is_contiguous()
    assert b.data_ptr() - a.data_ptr() == 256                                   # 256-byte slots, no overlap
    a.fill_(1.0)
    b.fill_(2.0)
    assert float(a.sum()) == 10.0 and float(b.sum()) == 30.0 and float(c.sum()) == 0.0
    with pool.step(dev):                                                        # grew to the larger demand; a NEW buffer each step
        assert pool.capacity == 768
        assert pool.buf.untyped_storage().data_ptr() != base or float(pool.buf.sum()) == 0.0
        assert float(pool.zeros(10, torch.float64, dev).sum()) == 0.0
    assert float(a.sum()) == 10.0                                               # earlier views stay intact (grads keep their buffer)


def test_nested_steps_share_one_buffer():
    pool, dev = ZeroPool(), torch.device('cpu')
    for _ in range(2):
        with pool.step(dev):
            with pool.step(dev):                                                # MVPNet3D.forward -> PN2SSG.forward
                x = pool.zeros(4, torch.float32, dev)
            y = pool.zeros(4, torch.float32, dev)                              # backward-side request after the inner forward
    assert x.untyped_storage().data_ptr() == y.untyped_storage().data_ptr() and x.data_ptr() != y.data_ptr()
