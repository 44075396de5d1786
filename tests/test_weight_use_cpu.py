"""rows.WeightUse: when may a weight's gradient kernel leave the calling stream?  Host logic only."""
import torch

from mvpnet_amd.rows import WeightUse


def _w():
    return torch.nn.Parameter(torch.zeros(4, 4))


def test_single_use_may_fork_and_later_uses_after_it_is_done():
    w = _w()
    a = WeightUse([w])
    assert a.aside_ok()
    a.done = True          # its backward ran; the graph (and `a`) may live on in the caller's preds
    b = WeightUse([w])
    assert b.aside_ok() and not a.shared


def test_two_pending_uses_mark_each_other_shared():
    w, v = _w(), _w()
    a = WeightUse([w, v])
    b = WeightUse([w])     # second forward before the first backward (or a Parameter shared by two layers)
    assert a.shared and b.shared and not a.aside_ok() and not b.aside_ok()
    a.done = b.done = True
    c = WeightUse([v])
    assert c.aside_ok()


def test_dropped_graph_does_not_block_forever():
    w = _w()
    a = WeightUse([w])     # e.g. a forward under grad mode whose output was thrown away
    del a
    assert WeightUse([w]).aside_ok()


def test_existing_grad_and_hooks_keep_the_gradient_on_the_calling_stream():
    w = _w()
    a = WeightUse([w])
    w.grad = torch.zeros_like(w)          # accumulation: autograd will ADD
    assert not a.aside_ok()
    w.grad = None
    assert a.aside_ok()
    h = w.register_post_accumulate_grad_hook(lambda p: None)   # DDP-style hooks read the gradient at once
    assert not a.aside_ok()
    h.remove()
    h2 = w.register_hook(lambda g: g)
    assert not a.aside_ok()
    h2.remove()
    assert not WeightUse([w * 1.0]).aside_ok()  # non-leaf tensors never fork
