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


def test_mvpnet_3d_yaml():
    cfg = C.load_cfg(text=yaml_text('mvpnet_3d_unet_resnet34_pn2ssg'), opts=['TRAIN.LOG_PERIOD', '10'])
    assert cfg.TASK == 'mvpnet_3d' and cfg.MODEL_3D.TYPE == 'PN2SSG' and cfg.MODEL_2D.TYPE == 'UNetResNet34'
    assert cfg.DATASET.ScanNet2D3DChunks.resize == (160, 120) and cfg.DATASET.ScanNet2D3DChunks.k == 3
    assert cfg.DATASET.ScanNet2D3DChunks.num_rgbd_frames == 3 and cfg.DATASET.ScanNet2D3DChunks.nb_pts == 8192
    assert cfg.TRAIN.BATCH_SIZE == 32 and cfg.TRAIN.LOG_PERIOD == 10
    assert cfg.SCHEDULER.MultiStepLR.milestones == (24000, 32000) and 'StepLR' not in cfg.SCHEDULER  # purged
    assert cfg.MODEL_3D.PN2SSG.in_channels == 64 and cfg.MODEL_3D.PN2SSG.num_classes == 20
    model = C.build_model_mvpnet_3d(cfg, torch.nn.Identity())
    n3d = sum(p.numel() for p in model.net_3d.parameters())
    nag = sum(p.numel() for p in model.feat_aggreg.parameters())
    assert (n3d, nag) == (967092, 12928)
    opt = C.build_optimizer(cfg, model)
    sched = C.build_scheduler(cfg, opt)
    assert isinstance(opt, torch.optim.Adam) and opt.defaults['lr'] == 0.002 and opt.defaults['betas'] == (0.9, 0.999)
    assert isinstance(sched, torch.optim.lr_scheduler.MultiStepLR) and sorted(sched.milestones) == [24000, 32000]
    # the whole model from the YAML alone: MODEL_2D.TYPE UNetResNet34, frozen (folded BatchNorm, channels-last)
    full = C.build_model_mvpnet_3d(cfg)
    assert type(full.net_2d).__name__ == 'UNetResNet34' and full.net_2d.num_classes == 20
    assert not any(p.requires_grad for p in full.net_2d.parameters())
    trainable = sum(p.numel() for p in full.parameters() if p.requires_grad)
    assert trainable == 967092 + 12928


def test_pn2ssg_chunk_yaml():
    cfg = C.load_cfg(text=yaml_text('pn2ssg_chunk'))
    assert cfg.TASK == 'sem_seg_3d' and cfg.MODEL.TYPE == 'PN2SSG' and cfg.OPTIMIZER.BASE_LR == 0.004
    assert cfg.TRAIN.AUGMENTATION == (('CropPad', 8192), 'RandomRotateZ')
    model = C.build_model_sem_seg_3d(cfg)
    assert sum(p.numel() for p in model.parameters()) == 965044 and model.in_channels == 0
