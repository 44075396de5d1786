"""Minimal configuration + factory layer so the reference's experiment YAMLs drive this package unmodified
(`configs/scannet/mvpnet_3d_unet_resnet34_pn2ssg.yaml`, `configs/scannet/3d_baselines/pn2ssg_chunk.yaml`).

The reference uses yacs (absent here): defaults in `common/config/base.py:10-137`, task defaults in
`mvpnet/config/mvpnet_3d.py:6-80` / `mvpnet/config/sem_seg_3d.py`, `purge_cfg` in
`common/config/__init__.py:4-17`, factories in `mvpnet/models/build.py:8-47` and
`common/solver/build.py:7-41`.  This is a PyYAML + `ast.literal_eval` work-alike of exactly what those
need: nested attribute access, defaults, `merge_from_file` / `merge_from_list`, tuples written as strings
("(160, 120)") evaluated like yacs does, and TYPE-keyed purge.  The default trees below are pinned key by key
against the reference's own config modules (tests/golden/config_defaults.json, dumped by importing them with a
dict stand-in for yacs: tests/golden/make_golden.py::gen_config_defaults).
"""
import ast
import copy

import yaml


class CfgNode(dict):
    """dict with attribute access (the subset of yacs.config.CfgNode the reference relies on)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _convert(value):
        if isinstance(value, dict):
            node = CfgNode()
            for k, v in value.items():
                node[k] = CfgNode._convert(v)
            return node
        if isinstance(value, str):  # yacs literal_evals strings: "(160, 120)" -> (160, 120)
            try:
                return ast.literal_eval(value)
            except (ValueError, SyntaxError):
                return value
        if isinstance(value, list):
            return tuple(CfgNode._convert(v) for v in value)
        return value

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = v
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            return self.merge(CfgNode._convert(yaml.safe_load(f) or {}))

    def merge_from_text(self, text):
        return self.merge(CfgNode._convert(yaml.safe_load(text) or {}))

    def merge_from_list(self, opts):
        """['OPTIMIZER.BASE_LR', '0.001', ...] like the reference CLI overrides (train_mvpnet_3d.py:301-305)."""
        assert len(opts) % 2 == 0
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            node[parts[-1]] = CfgNode._convert(value)
        return self


def _base():
    """common/config/base.py:10-137 (the keys this package reads; the YAML may add others)."""
    return CfgNode._convert({
        'TASK': '', 'AUTO_RESUME': True, 'RESUME_STATES': True, 'RESUME_PATH': '',
        'MODEL': {'TYPE': ''}, 'DATASET': {'TYPE': ''}, 'DATALOADER': {'NUM_WORKERS': 0, 'DROP_LAST': True},
        'OPTIMIZER': {'TYPE': '', 'BASE_LR': 0.001, 'WEIGHT_DECAY': 0.0, 'MAX_GRAD_NORM': 0.0,
                      'SGD': {'momentum': 0.9, 'dampening': 0.0}, 'Adam': {'betas': '(0.9, 0.999)'}},
        'SCHEDULER': {'TYPE': '', 'MAX_ITERATION': 1, 'CLIP_LR': 0.0, 'StepLR': {'step_size': 0, 'gamma': 0.1},
                      'MultiStepLR': {'milestones': '()', 'gamma': 0.1}},
        'TRAIN': {'BATCH_SIZE': 1, 'CHECKPOINT_PERIOD': 0, 'LOG_PERIOD': 0, 'SUMMARY_PERIOD': 0, 'MAX_TO_KEEP': 0,
                  'AUGMENTATION': '()', 'FROZEN_PATTERNS': '()', 'LABEL_WEIGHTS_PATH': ''},
        'VAL': {'BATCH_SIZE': 1, 'PERIOD': 0, 'LOG_PERIOD': 0, 'METRIC': '', 'AUGMENTATION': '()', 'REPEATS': 1},
        'OUTPUT_DIR': '@', 'RNG_SEED': -1,
    })


_PN2SSG = {'num_classes': 20, 'sa_channels': '((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 256, 512))',
           'num_centroids': '(2048, 512, 128, 32)', 'radius': '(0.1, 0.2, 0.4, 0.8)', 'max_neighbors': '(32, 32, 32, 32)',
           'fp_channels': '((256, 256), (256, 256), (256, 128), (128, 128, 128))', 'fp_neighbors': '(3, 3, 3, 3)',
           'seg_channels': '(128,)', 'dropout_prob': 0.5, 'use_xyz': True}


def get_cfg_mvpnet_3d():
    """mvpnet/config/mvpnet_3d.py:6-80"""
    cfg = _base()
    cfg.merge(CfgNode._convert({
        'TASK': 'mvpnet_3d', 'VAL': {'METRIC': 'seg_iou'},
        'DATASET': {'TRAIN': '', 'VAL': '', 'ScanNet2D3DChunks': {
            'cache_dir': '', 'image_dir': '', 'chunk_size': '(1.5, 1.5)', 'chunk_thresh': 0.3, 'chunk_margin': '(0.2, 0.2)',
            'nb_pts': 8192, 'num_rgbd_frames': 3, 'resize': '(160, 120)', 'k': 3,
            'image_normalizer': '((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))',
            'augmentation': {'z_rot': '()', 'flip': 0.0, 'color_jitter': '()'}}},
        'MODEL_3D': {'TYPE': '', 'PN2SSG': dict(_PN2SSG, in_channels=64)},
        'MODEL_2D': {'TYPE': '', 'CKPT_PATH': '', 'UNetResNet34': {'num_classes': 20, 'p': 0.0}},
        'FEAT_AGGR': {'in_channels': 64, 'mlp_channels': '(64, 64, 64)', 'reduction': 'sum', 'use_relation': True},
    }))
    return cfg


def get_cfg_sem_seg_3d():
    """mvpnet/config/sem_seg_3d.py (PN2SSG baseline: no input feature)"""
    cfg = _base()
    cfg.merge(CfgNode._convert({
        'TASK': 'sem_seg_3d', 'VAL': {'METRIC': 'seg_iou'},
        'DATASET': {'ROOT_DIR': '', 'TRAIN': '', 'VAL': '',
                    'ScanNet3DChunks': {'chunk_size': '(1.5, 1.5)', 'chunk_thresh': 0.3, 'chunk_margin': '(0.2, 0.2)', 'use_color': False},
                    'ScanNet3DScene': {'use_color': False}},
        'MODEL': {'TYPE': '', 'PN2SSG': dict(_PN2SSG, in_channels=0)}}))
    return cfg


def load_cfg(path=None, text=None, opts=()):
    """Defaults of the YAML's TASK + the YAML + `KEY VALUE` overrides, purged like the reference does."""
    raw = CfgNode()
    if path is not None:
        raw.merge_from_file(path)
    if text is not None:
        raw.merge_from_text(text)
    task = raw.get('TASK', 'mvpnet_3d')
    cfg = {'mvpnet_3d': get_cfg_mvpnet_3d, 'sem_seg_3d': get_cfg_sem_seg_3d}[task]()
    cfg.merge(raw)
    cfg.merge_from_list(list(opts))
    purge_cfg(cfg)
    return cfg


def purge_cfg(cfg):
    """common/config/__init__.py:4-17: under a node with TYPE, drop sibling sub-nodes other than cfg[TYPE]."""
    target = cfg.get('TYPE', None)
    for k in [k for k, v in cfg.items() if isinstance(v, CfgNode)]:
        if target is not None and k != target:
            del cfg[k]
        else:
            purge_cfg(cfg[k])


def build_model_sem_seg_3d(cfg):
    """mvpnet/models/build.py:8-20 -> model (loss: mvpnet_amd.mvpnet3d.SegLoss)"""
    from .pn2 import PN2SSG
    assert cfg.TASK == 'sem_seg_3d' and cfg.MODEL.TYPE == 'PN2SSG', (cfg.TASK, cfg.MODEL.TYPE)
    return PN2SSG(**dict(cfg.MODEL.get('PN2SSG', {})))


def build_model_mvpnet_3d(cfg, net_2d=None, load_2d_ckpt=True, freeze_2d=True):
    """mvpnet/models/build.py:23-47.  net_2d=None: built from cfg.MODEL_2D (TYPE UNetResNet34, mvpnet_amd/unet_resnet34.py);
    with freeze_2d (the reference trains MVPNet with the 2D branch frozen, train_mvpnet_3d.py FROZEN_PATTERNS) it gets a folded
    channels-last RUNTIME COPY (unet_resnet34.frozen_inference) while its own parameters keep the reference layout, so full
    MVPNet3D checkpoints of either side load on the other.  MODEL_2D.CKPT_PATH is loaded like the reference does
    (mvpnet_3d.py:78-81) whenever it is non-empty; load_2d_ckpt=False is for synthetic runs without the checkpoint file and
    says so.  A module instance can be supplied instead (e.g. a feature provider); PN2SSG and FeatureAggregation always come
    from cfg."""
    from .pn2 import PN2SSG
    from .mvpnet3d import MVPNet3D
    assert cfg.TASK == 'mvpnet_3d' and cfg.MODEL_3D.TYPE == 'PN2SSG', (cfg.TASK, cfg.MODEL_3D.TYPE)
    built_here = net_2d is None
    if built_here:
        from .unet_resnet34 import UNetResNet34
        assert cfg.MODEL_2D.TYPE == 'UNetResNet34', cfg.MODEL_2D.TYPE
        net_2d = UNetResNet34(**dict(cfg.MODEL_2D.get('UNetResNet34', {})))
    net_3d = PN2SSG(**dict(cfg.MODEL_3D.get('PN2SSG', {})))
    ckpt = cfg.MODEL_2D.get('CKPT_PATH', '')
    if ckpt and not load_2d_ckpt:
        import warnings
        warnings.warn('MODEL_2D.CKPT_PATH ({}) is NOT loaded (load_2d_ckpt=False): the 2D network keeps its random initialisation'.format(ckpt))
    model = MVPNet3D(net_2d, ckpt if load_2d_ckpt else '', net_3d, **dict(cfg.FEAT_AGGR))
    if built_here and freeze_2d:
        net_2d.frozen_inference()
    return model


def build_augmentation(cfg, rng=None):
    """DATASET.ScanNet2D3DChunks.augmentation.{flip, z_rot} of the YAML (mvpnet/data/build.py -> ScanNet2D3DChunks(flip=, z_rot=))
    as the device-side augmentation of mvpnet_amd.augment (None when both are off).  color_jitter stays a loader transform."""
    from .augment import DeviceAugmentation
    aug = cfg.DATASET.get('ScanNet2D3DChunks', {}).get('augmentation', {})
    flip, z_rot = aug.get('flip', 0.0), aug.get('z_rot', ())
    if not flip and not z_rot:
        return None
    return DeviceAugmentation(flip=flip, z_rot=z_rot, rng=rng)


def build_optimizer(cfg, model):
    """common/solver/build.py:7-22"""
    import torch
    name = cfg.OPTIMIZER.TYPE
    if name == '':
        return None
    kwargs = dict(cfg.OPTIMIZER.get(name, {}))
    params = [p for p in model.parameters()]
    if name == 'Adam' and not kwargs.get('amsgrad') and 'fused' not in kwargs and params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        from .optim import FusedAdam  # a torch.optim.Adam (same state, same checkpoints) whose step is ONE launch for all parameter tensors
        return FusedAdam(params, lr=cfg.OPTIMIZER.BASE_LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY, **kwargs)
    if name in ('Adam', 'AdamW') and 'fused' not in kwargs and params and all(p.is_cuda for p in params):
        kwargs['fused'] = True  # ATen's multi-tensor kernel (three launches for ~80 tensors)
    return getattr(torch.optim, name)(params, lr=cfg.OPTIMIZER.BASE_LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY, **kwargs)


def build_scheduler(cfg, optimizer):
    """common/solver/build.py:25-41 (without ClipLR)"""
    import torch
    name = cfg.SCHEDULER.TYPE
    if name == '':
        return None
    return getattr(torch.optim.lr_scheduler, name)(optimizer, **dict(cfg.SCHEDULER.get(name, {})))
