"""UNetResNet34: the 2D network of MVPNet (mvpnet/models/unet_resnet34.py:9-125), SURVEY.md sec.8f rank 2.

Same class name, constructor arguments, forward contract ({'image'} -> {'seg_logit', 'feature'}) and `state_dict` keys as
the reference, so `MODEL_2D.TYPE: UNetResNet34` of mvpnet_3d_unet_resnet34_pn2ssg.yaml builds and `CKPT_PATH` checkpoints load.
torchvision is not in this image: the ResNet-34 encoder (BasicBlock x [3,4,6,3], He et al. 2015; torchvision/models/resnet.py
layout and key names) is restated here.  PARITY: the decoder / padding / crop / concat logic is pinned against the imported
reference class (tests/golden/unet_resnet34.npz, generated with this file's encoder standing in for torchvision's); the encoder is
pinned against known-answer vectors from a second, functional restatement of torchvision's published resnet34 definition on a
torchvision-keyed state_dict (tests/golden/make_golden.py::torchvision_resnet34_forward -> resnet34_encoder.npz: key names, shapes,
order, the published 21 797 672 parameters, stem / pool / first block and output of every stage).  torchvision itself cannot be
imported here, so this is a restatement checked against a restatement, not against the package.

MI355X notes.  The network is frozen inside MVPNet (train_mvpnet_3d.py freezes net_2d; mvpnet_3d.py:99-101 only reads
'feature'), so `frozen_inference()` folds every eval-mode BatchNorm into the preceding convolution (one MIOpen kernel per
conv instead of conv + BN + ReLU passes over HBM) and switches the module to torch.channels_last: the (B*nv, 64, h, w)
feature map then IS the (B, nv, h, w, 64) channels-last tensor the lifting kernel gathers rows from -- no transpose copy."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN + identity (or 1x1-conv/BN downsample) -> ReLU; keys conv1, bn1, conv2, bn2, downsample.{0,1}."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNet34(nn.Module):
    """Attributes conv1, bn1, relu, maxpool, layer1..4 -- what unet_resnet34.py:17-28 takes from torchvision's resnet34."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._layer(64, 64, 3, 1)
        self.layer2 = self._layer(64, 128, 4, 2)
        self.layer3 = self._layer(128, 256, 6, 2)
        self.layer4 = self._layer(256, 512, 3, 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    @staticmethod
    def _layer(inplanes, planes, blocks, stride):
        return nn.Sequential(BasicBlock(inplanes, planes, stride), *[BasicBlock(planes, planes) for _ in range(blocks - 1)])


def resnet34(pretrained=False):
    """Stand-in for torchvision.models.resnet.resnet34 (no pretrained weights without the network)."""
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights need torchvision / the network; load a checkpoint instead')
    return ResNet34()


class UNetResNet34(nn.Module):
    def __init__(self, num_classes, p=0.0, pretrained=False):
        super().__init__()
        self.num_classes = num_classes
        net = resnet34(pretrained)
        self.encoder0 = nn.Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False)  # conv1 without the downsampling (:19-20)
        self.encoder0.weight.data = net.conv1.weight.data
        self.bn, self.relu, self.maxpool = net.bn1, net.relu, net.maxpool
        self.encoder1, self.encoder2, self.encoder3, self.encoder4 = net.layer1, net.layer2, net.layer3, net.layer4
        self.deconv4 = self.get_deconv(512, 256)
        self.decoder3 = self.get_conv(512, 256)
        self.deconv3 = self.get_deconv(256, 128)
        self.decoder2 = self.get_conv(256, 128)
        self.deconv2 = self.get_deconv(128, 64)
        self.decoder1 = self.get_conv(128, 64)
        self.deconv1 = self.get_deconv(64, 64)
        self.decoder0 = self.get_conv(128, 64)
        self.logit = nn.Conv2d(64, num_classes, 1, bias=True)
        self.dropout = nn.Dropout(p=p) if p > 0.0 else None
        self._folded = False

    @staticmethod
    def get_deconv(c_in, c_out):
        return nn.Sequential(nn.ConvTranspose2d(c_in, c_out, kernel_size=2, stride=2), nn.BatchNorm2d(c_out), nn.ReLU(inplace=True))

    @staticmethod
    def get_conv(c_in, c_out):
        return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.BatchNorm2d(c_out), nn.ReLU(inplace=True))

    def train(self, mode=True):
        """A frozen instance (frozen_inference()) STAYS in eval mode: `model.train()` on the enclosing MVPNet3D must neither switch the
        2D branch's BatchNorms to batch statistics nor move their running statistics (they would end up in the next checkpoint).  The
        reference gets the same effect by re-applying its Freezer after every `model.train()` (mvpnet/train_3d.py:142-143,
        common/nn/freezer.py); here the module refuses to leave eval mode until `unfreeze()`."""
        if self.__dict__.get('_fast') is not None:
            mode = False
        return super().train(mode)

    def unfreeze(self):
        """Undo frozen_inference(): drop the folded runtime copy, parameters trainable again, train() works normally."""
        self.__dict__.pop('_fast', None)
        self.__dict__.pop('_fast_dtype', None)
        self.__dict__['_frozen'] = False  # the load_state_dict hook stays registered but no longer re-creates the folded copy
        for p in self.parameters():
            p.requires_grad_(True)
        return self

    def forward(self, data_dict):
        fast = self.__dict__.get('_fast')
        if fast is not None:  # frozen: always the folded, channels-last runtime copy (train() cannot leave eval mode, see above)
            dtype = self.__dict__.get('_fast_dtype')
            if dtype is None:
                return fast(data_dict)
            with torch.autocast(device_type=data_dict['image'].device.type, dtype=dtype):
                out = fast(data_dict)
            return {k: v.float() for k, v in out.items()}  # the lifting kernels take fp32 feature rows
        x = data_dict['image']
        h, w = x.shape[2], x.shape[3]
        pad_h, pad_w = (h + 15) // 16 * 16 - h, (w + 15) // 16 * 16 - w  # zero-pad to multiples of 16 (:66-73)
        if pad_h > 0 or pad_w > 0:
            x = F.pad(x, [0, pad_w, 0, pad_h])
        feats = []
        x = self.relu(self.bn(self.encoder0(x)))
        feats.append(x)
        x = self.encoder1(self.maxpool(x))
        feats.append(x)
        x = self.encoder2(x)
        feats.append(x)
        x = self.encoder3(x)
        if self.dropout is not None:
            x = self.dropout(x)
        feats.append(x)
        x = self.encoder4(x)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.decoder3(torch.cat([self.deconv4(x), feats[3]], dim=1))
        x = self.decoder2(torch.cat([self.deconv3(x), feats[2]], dim=1))
        x = self.decoder1(torch.cat([self.deconv2(x), feats[1]], dim=1))
        x = self.decoder0(torch.cat([self.deconv1(x), feats[0]], dim=1))
        if pad_h > 0 or pad_w > 0:
            x = x[:, :, 0:h, 0:w]
            if self._folded:  # keep the promise of frozen_inference(): dense (N, h, w, C) rows for the lifting kernel
                x = x.contiguous(memory_format=torch.channels_last)
        return {'seg_logit': self.logit(x), 'feature': x}

    # ------------------------------------------------------------------ frozen, folded, channels-last
    @torch.no_grad()
    def frozen_inference(self, compute_dtype=None):
        """compute_dtype (e.g. torch.bfloat16; default None = fp32, the reference's arithmetic): run the frozen convolutions under
        autocast in that type -- an opt-in speed / accuracy trade for the 2D branch only (bench field with_2d_network: 27.4 -> 17.6 ms
        per step at B = 32); the features handed to the lifting kernels stay fp32 tensors.
        Eval mode, requires_grad off, and a FOLDED RUNTIME COPY of the network (every BatchNorm folded into its convolution,
        torch.channels_last) that eval-mode forward() dispatches to.  The module itself keeps the reference's parameter layout:
        `state_dict()` still has the 426 reference keys, a full MVPNet3D checkpoint written by the reference loads, one saved
        here loads there; the copy is rebuilt after every `load_state_dict` and follows `.to()` / `.cuda()`.  Use on the
        frozen 2D branch of MVPNet (train_mvpnet_3d.py freezes net_2d)."""
        self.eval()
        for p in self.parameters():
            p.requires_grad_(False)
        self.__dict__['_fast_dtype'] = compute_dtype
        self.__dict__['_frozen'] = True
        self._refold()
        if not self.__dict__.get('_refold_hooked'):
            # (only while frozen: after unfreeze() a load_state_dict must not quietly freeze the module again)
            self.register_load_state_dict_post_hook(lambda module, incompatible: module._refold() if module.__dict__.get('_frozen') else None)
            self.__dict__['_refold_hooked'] = True
        return self

    @torch.no_grad()
    def _refold(self):
        import copy
        self.__dict__.pop('_fast', None)
        fast = copy.deepcopy(self)
        fast.__dict__.pop('_refold_hooked', None)
        fast.__dict__.pop('_frozen', None)
        fast.__dict__.pop('_fast_dtype', None)
        fast._load_state_dict_post_hooks.clear()
        fast._fold_in_place()
        self.__dict__['_fast'] = fast  # NOT a registered sub-module: invisible to state_dict() / parameters()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        fast = self.__dict__.get('_fast')
        if fast is not None:
            fast._apply(fn, *args, **kwargs)
        return out

    def _fold_in_place(self):
        """Irreversible (the BatchNorm modules become identities): only ever applied to the runtime copy."""
        self.eval()
        if not self._folded:
            _fold(self.encoder0, self.bn)
            self.bn = nn.Identity()
            for layer in (self.encoder1, self.encoder2, self.encoder3, self.encoder4):
                for blk in layer:
                    _fold(blk.conv1, blk.bn1)
                    _fold(blk.conv2, blk.bn2)
                    blk.bn1, blk.bn2 = nn.Identity(), nn.Identity()
                    if blk.downsample is not None:
                        _fold(blk.downsample[0], blk.downsample[1])
                        blk.downsample[1] = nn.Identity()
            for seq in (self.deconv4, self.decoder3, self.deconv3, self.decoder2, self.deconv2, self.decoder1, self.deconv1, self.decoder0):
                _fold(seq[0], seq[1])
                seq[1] = nn.Identity()
            self._folded = True
        return self.to(memory_format=torch.channels_last)


def _fold(conv, bn):
    """conv <- bn(conv(.)) for an eval-mode BatchNorm2d: scale the output channels, fold the shift into the bias."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if isinstance(conv, nn.ConvTranspose2d):  # weight (C_in, C_out, kh, kw)
        conv.weight.mul_(scale.view(1, -1, 1, 1))
    else:                                     # weight (C_out, C_in, kh, kw)
        conv.weight.mul_(scale.view(-1, 1, 1, 1))
    if conv.bias is None:
        conv.bias = nn.Parameter(shift.clone(), requires_grad=False)
    else:
        conv.bias.mul_(scale).add_(shift)
