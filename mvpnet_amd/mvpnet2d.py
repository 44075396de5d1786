"""MVPNet2D: the 2D-only baseline that lifts per-pixel class logits to the points (mvpnet/models/mvpnet_2d.py:7-34).

Same class name, constructor, data-dict keys (`images`, `knn_indices` -> `seg_logit` (B, classes, N)).  The reference
re-lays the logits out as (B, classes, nv*h*w) (a transposed copy), gathers with `group_points` (channel-major: classes x
N*k scattered 4-byte reads) and averages over k.  Here the 2D network's (B*nv, classes, h, w) output is viewed
channels-last -- free when it already runs in torch.channels_last --, the k-NN rows are gathered by the lifting gather
kernel (mvp_lift_gather_f32: one 80-byte row per neighbour) and averaged.  As in MVPNet3D, the k-NN indices may be computed
on the device from depth / intrinsics / pose when the loader does not supply them."""
import torch
from torch import nn

from . import ops


class MVPNet2D(nn.Module):
    def __init__(self, net_2d):
        super(MVPNet2D, self).__init__()
        self.net_2d = net_2d

    def forward(self, data_batch):
        images = data_batch['images']  # (B,nv,3,h,w)
        b, nv, _, h, w = images.shape
        seg_logit_2d = self.net_2d({'image': images.reshape(b * nv, *images.shape[2:])})['seg_logit']  # (B*nv,classes,h,w)
        nc = seg_logit_2d.size(1)
        logit_cl = seg_logit_2d.permute(0, 2, 3, 1).contiguous().view(b, nv, h, w, nc)  # channels-last rows
        knn = data_batch.get('knn_indices')
        if knn is None:
            # device lifting from depth / intrinsics / pose.  The fused kernel gathers rows whose 16-byte chunks divide a 256-lane
            # workgroup (C/4 | 256); 20 class logits do not, so the indices come from the un-project + projective k-NN entry
            # points and the rows from the generic gather below.
            cam = data_batch['cam_matrix']
            kinv = data_batch['kinv'] if 'kinv' in data_batch else torch.linalg.inv(cam)
            xyz, mask = ops.unproject(data_batch['depth'], kinv, data_batch['pose'], data_batch.get('pixel_box'))
            knn = ops.pixel_knn(xyz, mask, data_batch['points'].transpose(1, 2).contiguous(), int(data_batch.get('k', 3)), cam=cam,
                                pose=data_batch['pose'])
        gathered, _ = ops.lift_gather(logit_cl, None, knn)  # (B,N,k,classes)
        return {'seg_logit': gathered.mean(2).transpose(1, 2).contiguous()}
