"""mvpnet_amd.chunks against the reference's chunker / frame selection (golden vectors from the imported
`scene2chunks_legacy` and the re-typed `select_frames`).  Pure torch: runs on CPU here and on the GPU in tests/test_model_gpu.py."""
import numpy as np
import pytest
import torch

from mvpnet_amd.chunks import scene2chunks_legacy, select_frames, crop_pad_choice
from tests.conftest import load_golden


def check_chunker(dev):
    g = load_golden('chunker')
    for ci in range(3):
        stride, thresh = g['c%d_args' % ci]
        pts = torch.from_numpy(g['c%d_points' % ci]).to(dev)
        idx, boxes = scene2chunks_legacy(pts, (1.5, 1.5), float(stride), thresh=int(thresh), margin=(0.2, 0.2), return_bbox=True)
        assert [len(i) for i in idx] == g['c%d_lengths' % ci].tolist()
        got = torch.cat(idx).cpu().numpy() if idx else np.zeros(0, np.int64)
        np.testing.assert_array_equal(got, g['c%d_indices' % ci])
        np.testing.assert_array_equal(torch.stack(boxes).cpu().numpy() if boxes else np.zeros((0, 6)), g['c%d_boxes' % ci])
        assert len(scene2chunks_legacy(pts, (1.5, 1.5), float(stride), thresh=int(thresh))) == len(idx)
    for ci in range(3):
        ov = torch.from_numpy(g['f%d_overlap' % ci]).to(dev)
        assert select_frames(ov, 3) == g['f%d_selected' % ci].tolist()
        assert torch.equal(ov.cpu(), torch.from_numpy(g['f%d_overlap' % ci]))  # the input is not modified


def test_chunker_cpu():
    check_chunker(torch.device('cpu'))


def test_crop_pad_choice():
    gen = torch.Generator().manual_seed(1)
    a = crop_pad_choice(100, 256, generator=gen)
    assert a.shape == (256,) and torch.equal(a[:100], torch.arange(100)) and int(a.max()) < 100
    b = crop_pad_choice(1000, 256, generator=gen)
    assert b.shape == (256,) and b.unique().numel() == 256 and int(b.max()) < 1000


@pytest.mark.gpu
def test_chunker_gpu():
    check_chunker(torch.device('cuda:0'))
