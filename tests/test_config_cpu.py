"""The reference's experiment YAMLs (stored parsed in tests/golden/configs.json) drive mvpnet_amd unmodified."""
import json
import os

import torch
import yaml

from mvpnet_amd import config as C
from tests.conftest import GOLDEN


def yaml_text(name):
    with open(os.path.join(GOLDEN, 'configs.json')) as f:
        return yaml.safe_dump(json.load(f)[name])


def _plain(node):
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (tuple, list)):
        return [_plain(v) for v in node]
    return node


def test_default_trees_equal_the_reference_config_modules():
    """common/config/base.py:10-137 + mvpnet/config/{mvpnet_3d,sem_seg_3d}.py, dumped from the imported modules."""
    with open(os.path.join(GOLDEN, 'config_defaults.json')) as f:
        ref = json.load(f)
    assert _plain(C.get_cfg_mvpnet_3d()) == ref['mvpnet_3d']
    assert _plain(C.get_cfg_sem_seg_3d()) == ref['sem_seg_3d']


def test_mvpnet_3d_yaml():
    cfg = C.load_cfg(text=yaml_text('mvpnet_3d_unet_resnet34_pn2ssg'), opts=['TRAIN.LOG_PERIOD', '10'])
    assert cfg.TASK == 'mvpnet_3d' and cfg.MODEL_3D.TYPE == 'PN2SSG' and cfg.MODEL_2D.TYPE == 'UNetResNet34'
    assert cfg.DATASET.ScanNet2D3DChunks.resize == (160, 120) and cfg.DATASET.ScanNet2D3DChunks.k == 3
    assert cfg.DATASET.ScanNet2D3DChunks.num_rgbd_frames == 3 and cfg.DATASET.ScanNet2D3DChunks.nb_pts == 8192
    assert cfg.TRAIN.BATCH_SIZE == 32 and cfg.TRAIN.LOG_PERIOD == 10
    assert cfg.SCHEDULER.MultiStepLR.milestones == (24000, 32000) and 'StepLR' not in cfg.SCHEDULER  # purged
    assert cfg.MODEL_3D.PN2SSG.in_channels == 64 and cfg.MODEL_3D.PN2SSG.num_classes == 20
    model = C.build_model_mvpnet_3d(cfg, torch.nn.Identity(), load_2d_ckpt=False)
    n3d = sum(p.numel() for p in model.net_3d.parameters())
    nag = sum(p.numel() for p in model.feat_aggreg.parameters())
    assert (n3d, nag) == (967092, 12928)
    opt = C.build_optimizer(cfg, model)
    sched = C.build_scheduler(cfg, opt)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults['lr'] == 0.002 and opt.defaults['betas'] == (0.9, 0.999)
    assert isinstance(sched, torch.optim.lr_scheduler.MultiStepLR) and sorted(sched.milestones) == [24000, 32000]
    # the whole model from the YAML alone: MODEL_2D.TYPE UNetResNet34, frozen (folded BatchNorm, channels-last)
    import pytest
    with pytest.raises(FileNotFoundError):       # MODEL_2D.CKPT_PATH is loaded like the reference does (mvpnet_3d.py:78-81)
        C.build_model_mvpnet_3d(cfg)
    with pytest.warns(UserWarning, match='NOT loaded'):
        full = C.build_model_mvpnet_3d(cfg, load_2d_ckpt=False)
    assert type(full.net_2d).__name__ == 'UNetResNet34' and full.net_2d.num_classes == 20
    assert not any(p.requires_grad for p in full.net_2d.parameters())
    trainable = sum(p.numel() for p in full.parameters() if p.requires_grad)
    assert trainable == 967092 + 12928
    # the frozen 2D branch runs from a folded runtime copy but KEEPS the reference's parameter layout: a full MVPNet3D
    # checkpoint of the reference loads, and the copy follows the loaded weights
    from mvpnet_amd.unet_resnet34 import UNetResNet34
    plain = UNetResNet34(20, p=0.5)
    assert list(full.net_2d.state_dict()) == list(plain.state_dict()) and len(full.state_dict()) == 426
    x = {'image': torch.randn(1, 3, 24, 32)}
    torch.manual_seed(1)
    donor = UNetResNet34(20, p=0.5).eval()
    for m in donor.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    sd = {'net_2d.' + k: v for k, v in donor.state_dict().items()}
    sd.update({k: v for k, v in full.state_dict().items() if not k.startswith('net_2d.')})
    full.load_state_dict(sd)
    full.eval()
    with torch.no_grad():
        torch.testing.assert_close(full.net_2d(x)['feature'], donor(x)['feature'], rtol=1e-4, atol=1e-4)


def test_pn2ssg_chunk_yaml():
    cfg = C.load_cfg(text=yaml_text('pn2ssg_chunk'))
    assert cfg.TASK == 'sem_seg_3d' and cfg.MODEL.TYPE == 'PN2SSG' and cfg.OPTIMIZER.BASE_LR == 0.004
    assert cfg.TRAIN.AUGMENTATION == (('CropPad', 8192), 'RandomRotateZ')
    model = C.build_model_sem_seg_3d(cfg)
    assert sum(p.numel() for p in model.parameters()) == 965044 and model.in_channels == 0


def test_yaml_augmentation_is_honoured():
    """DATASET.ScanNet2D3DChunks.augmentation of mvpnet_3d_unet_resnet34_pn2ssg.yaml (flip 0.5, z_rot (-180, 180)) becomes the
    device-side augmentation; its random draws are the reference's (one rand() per view, one uniform angle per chunk)."""
    import numpy as np
    cfg = C.load_cfg(text=yaml_text('mvpnet_3d_unet_resnet34_pn2ssg'))
    aug = C.build_augmentation(cfg, rng=np.random.RandomState(11))
    assert aug.flip == 0.5 and aug.z_rot == (-180, 180)
    images = torch.arange(2 * 3 * 3 * 2 * 4, dtype=torch.float32).view(2, 3, 3, 2, 4)
    batch = aug({'images': images.clone(), 'depth': torch.zeros(2, 3, 2, 4)})
    rs = np.random.RandomState(11)
    flags = np.array([[rs.rand() < 0.5 for _ in range(3)] for _ in range(2)])
    assert np.array_equal(batch['flip'].numpy().astype(bool), flags)
    for b in range(2):
        for v in range(3):
            assert torch.equal(batch['images'][b, v], images[b, v].flip(-1) if flags[b, v] else images[b, v])
    from scipy.spatial.transform import Rotation
    for b in range(2):
        np.testing.assert_array_equal(batch['z_rot'][b].numpy(), Rotation.from_euler('z', rs.uniform(low=-180, high=180), degrees=True).as_matrix())
    assert C.build_augmentation(C.load_cfg(text='TASK: mvpnet_3d')) is None
