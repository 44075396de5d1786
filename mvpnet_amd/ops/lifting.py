"""2D->3D lifting ops on the device (new: the reference does these in CPU dataloader workers,
mvpnet/data/scannet_2d3d.py:33-39,254-313, and gathers channel-major in
mvpnet/models/mvpnet_3d.py:99-109)."""
import torch
from torch.autograd.function import once_differentiable

from .. import _lib as L


def unproject(depth, kinv, pose, box=None):
    """depth (B,nv,h,w) float32 metres or int16/uint16 millimetres (raw PNG values),
    kinv (B,nv,3,3) = inv(cam_matrix[:3,:3]), pose (B,nv,4,4) camera-to-world,
    box (B,4) = (x0,y0,x1,y1) already expanded by the 0.1 m pixel margin, or None
    -> image_xyz (B,nv,h,w,3) float32, image_mask (B,nv,h,w) bool."""
    L.require_gpu(depth, kinv, pose, box)
    B, nv, h, w = depth.shape
    if kinv.shape != (B, nv, 3, 3) or pose.shape != (B, nv, 4, 4) or kinv.dtype != torch.float32 or pose.dtype != torch.float32:
        raise RuntimeError('unproject: kinv must be (B,nv,3,3) float32 and pose (B,nv,4,4) float32')
    if box is not None and (box.shape != (B, 4) or box.dtype != torch.float32):
        raise RuntimeError('unproject: box must be (B,4) float32')
    xyz = torch.empty((B, nv, h, w, 3), dtype=torch.float32, device=depth.device)
    mask = torch.empty((B, nv, h, w), dtype=torch.uint8, device=depth.device)
    if depth.dtype == torch.float32:
        name = 'mvp_unproject_f32'
    elif depth.dtype in (torch.int16, torch.uint16):
        name = 'mvp_unproject_u16'  # int16 storage is reinterpreted as uint16 millimetres
    else:
        raise RuntimeError('unproject: depth must be float32 (m) or (u)int16 (mm)')
    L.call(name, depth, L.ptr(depth), L.ptr(kinv), L.ptr(pose), L.ptr(box), B, nv, h, w, L.ptr(xyz), L.ptr(mask))
    return xyz, mask.bool()


def pixel_knn(image_xyz, image_mask, points, k, cam=None, pose=None, return_distance=False):
    """image_xyz (B,nv,h,w,3) f32, image_mask (B,nv,h,w) bool/uint8, points (B,N,3) f32
    -> knn_indices (B,N,k) int64 flat pixel ids view*h*w + row*w + col, nearest first.
    With cam (B,nv,3,3 forward intrinsics) and pose the projective window search is used,
    otherwise the O(N*P) scan; both are exact with lowest-id tie-breaking."""
    mask = image_mask.to(torch.uint8) if image_mask.dtype != torch.uint8 else image_mask
    L.require_gpu(image_xyz, mask, points, cam, pose)
    if image_xyz.dtype != torch.float32 or points.dtype != torch.float32 or points.dim() != 3 or points.size(2) != 3:
        raise RuntimeError('pixel_knn: image_xyz and points (B,N,3) must be float32')
    B, N, _ = points.shape
    k = int(k)
    if not 1 <= k <= 8:
        raise RuntimeError('pixel_knn: 1 <= k <= 8')
    P = image_xyz.numel() // (B * 3)
    index = torch.empty((B, N, k), dtype=torch.int64, device=points.device)
    dist = torch.empty((B, N, k), dtype=torch.float32, device=points.device) if return_distance else None
    if cam is not None and pose is not None:
        _, nv, h, w, _ = image_xyz.shape
        L.call('mvp_pixel_knn_projective_f32', points, L.ptr(image_xyz), L.ptr(mask), L.ptr(points), L.ptr(cam),
               L.ptr(pose), B, nv, h, w, N, k, L.ptr(index), L.ptr(dist))
    else:
        L.call('mvp_pixel_knn_bruteforce_f32', points, L.ptr(image_xyz), L.ptr(mask), L.ptr(points), B, P, N, k,
               L.ptr(index), L.ptr(dist))
    return (index, dist) if return_distance else index


class LiftGatherFunction(torch.autograd.Function):
    """Channels-last replacement of the two group_points calls in MVPNet3D.forward
    (mvpnet/models/mvpnet_3d.py:103,109); gradient flows to the feature map only."""

    @staticmethod
    def forward(ctx, feature, image_xyz, index):
        L.require_gpu(feature, image_xyz, index)
        B, N, k = index.shape
        C = feature.size(-1)
        P = feature.numel() // (B * C)
        ctx.save_for_backward(index)
        ctx.shape = tuple(feature.shape)
        gfeat = torch.empty((B, N, k, C), dtype=torch.float32, device=feature.device)
        gxyz = torch.empty((B, N, k, 3), dtype=torch.float32, device=feature.device) if image_xyz is not None else None
        L.call('mvp_lift_gather_f32', feature, L.ptr(feature), L.ptr(image_xyz), L.ptr(index), B, P, C, N, k,
               L.ptr(gfeat), L.ptr(gxyz))
        if gxyz is None:
            return gfeat
        ctx.mark_non_differentiable(gxyz)
        return gfeat, gxyz

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_gfeat, _grad_gxyz=None):
        (index,) = ctx.saved_tensors
        B, N, k = index.shape
        C = ctx.shape[-1]
        P = 1
        for s in ctx.shape[1:-1]:
            P *= s
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=grad_gfeat.device)
        g = grad_gfeat.contiguous()
        L.call('mvp_lift_gather_backward_f32', g, L.ptr(g), L.ptr(index), B, P, C, N, k, L.ptr(grad))
        return grad, None, None


def lift_gather(feature, image_xyz, knn_indices):
    """feature (B,nv,h,w,C) or (B,P,C) float32 channels-last, image_xyz (B,nv,h,w,3) or (B,P,3) or None,
    knn_indices (B,N,k) -> gathered feature (B,N,k,C), gathered xyz (B,N,k,3) (None without image_xyz)."""
    if feature.dtype != torch.float32 or (image_xyz is not None and image_xyz.dtype != torch.float32) or knn_indices.dtype != torch.int64:
        raise RuntimeError('lift_gather: float32 feature/xyz and int64 indices expected')
    if image_xyz is None:
        return LiftGatherFunction.apply(feature.contiguous(), None, knn_indices.contiguous()), None
    return LiftGatherFunction.apply(feature.contiguous(), image_xyz.contiguous(), knn_indices.contiguous())


class LiftFunction(torch.autograd.Function):
    """Fused un-project + pixel k-NN + channels-last gather (mvp_lift_f32, two launches).  Gradient flows to
    the feature map only (scatter-add by the k-NN index)."""

    @staticmethod
    def forward(ctx, feature, depth, kinv, cam, pose, box, points, k, want_image_xyz, flip=None, rot=None):
        L.require_gpu(feature, depth, kinv, cam, pose, box, points, flip, rot)
        B, nv, h, w = depth.shape
        N, C = points.size(1), feature.size(-1)
        if depth.dtype == torch.float32:
            is_u16 = 0
        elif depth.dtype in (torch.int16, torch.uint16):
            is_u16 = 1
        else:
            raise RuntimeError('lift: depth must be float32 (m) or (u)int16 (mm)')
        ws = torch.empty(L.lib().mvp_lift_workspace_bytes(B, nv, h, w, N), dtype=torch.uint8, device=depth.device)
        knn = torch.empty((B, N, k), dtype=torch.int64, device=depth.device)
        gfeat = torch.empty((B, N, k, C), dtype=torch.float32, device=depth.device)
        gxyz = torch.empty((B, N, k, 3), dtype=torch.float32, device=depth.device)
        xyz = torch.empty((B, nv, h, w, 3), dtype=torch.float32, device=depth.device) if want_image_xyz else None
        mask = torch.empty((B, nv, h, w), dtype=torch.uint8, device=depth.device) if want_image_xyz else None
        points_rot = torch.empty_like(points) if rot is not None else None
        if flip is None and rot is None:
            L.call('mvp_lift_f32', depth, L.ptr(depth), is_u16, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(points),
                   L.ptr(feature), B, nv, h, w, N, C, k, L.ptr(ws), L.ptr(knn), L.ptr(gfeat), L.ptr(gxyz), L.ptr(xyz), L.ptr(mask))
        else:
            L.call('mvp_lift_aug_f32', depth, L.ptr(depth), is_u16, L.ptr(kinv), L.ptr(cam), L.ptr(pose), L.ptr(box), L.ptr(points),
                   L.ptr(feature), B, nv, h, w, N, C, k, L.ptr(ws), L.ptr(knn), L.ptr(gfeat), L.ptr(gxyz), L.ptr(xyz), L.ptr(mask),
                   L.ptr(flip), L.ptr(rot), L.ptr(points_rot))
        ctx.save_for_backward(knn)
        ctx.shape = tuple(feature.shape)
        outs = (gfeat, gxyz, knn)
        if want_image_xyz:
            outs += (xyz, mask)
        if rot is not None:
            outs += (points_rot,)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_gfeat, *unused):
        (knn,) = ctx.saved_tensors
        B, N, k = knn.shape
        C = ctx.shape[-1]
        P = 1
        for s in ctx.shape[1:-1]:
            P *= s
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=grad_gfeat.device)
        g = grad_gfeat.contiguous()
        L.call('mvp_lift_gather_backward_f32', g, L.ptr(g), L.ptr(knn), B, P, C, N, k, L.ptr(grad))
        return (grad,) + (None,) * 10


def lift(feature, depth, kinv, cam, pose, points, k=3, box=None, return_image_xyz=False, flip=None, rot=None):
    """feature (B,nv,h,w,C) f32 channels-last; depth (B,nv,h,w) f32 m / int16 mm; kinv, cam (B,nv,3,3);
    pose (B,nv,4,4); points (B,N,3) -> gathered feature (B,N,k,C), gathered xyz (B,N,k,3), knn_indices (B,N,k)
    [, image_xyz (B,nv,h,w,3), image_mask (B,nv,h,w) uint8] [, rotated points (B,N,3) when `rot` is given].
    flip (B,nv) bool / uint8: views the loader mirrored (scannet_2d3d.py:293-296) -- `feature` comes from the mirrored image, the
        returned indices / image_xyz / mask are in mirrored pixel order;
    rot (B,3,3) float64: rotation applied after the search to the gathered xyz and the points (:400-409); image_xyz is returned
        un-rotated (rotate_rows does it on request)."""
    if feature.dtype != torch.float32 or points.dtype != torch.float32 or feature.size(-1) % 4:
        raise RuntimeError('lift: float32 channels-last feature with C % 4 == 0 expected')
    if flip is not None:
        if tuple(flip.shape) != tuple(depth.shape[:2]):
            raise RuntimeError('lift: flip must be (B, nv)')
        flip = flip.to(torch.uint8).contiguous()
    if rot is not None:
        if rot.dtype != torch.float64 or tuple(rot.shape) != (depth.size(0), 3, 3):
            raise RuntimeError('lift: rot must be (B, 3, 3) float64')
        rot = rot.contiguous()
    return LiftFunction.apply(feature.contiguous(), depth.contiguous(), kinv.contiguous(), cam.contiguous(),
                              pose.contiguous(), None if box is None else box.contiguous(), points.contiguous(), int(k),
                              bool(return_image_xyz), flip, rot)


def rotate_rows(xyz, rot):
    """xyz (B, ..., 3) float32, rot (B,3,3) float64 -> float32( rot[b] . xyz ) per row (scannet_2d3d.py:400-409)."""
    if xyz.dtype != torch.float32 or rot.dtype != torch.float64 or xyz.size(-1) != 3 or tuple(rot.shape) != (xyz.size(0), 3, 3):
        raise RuntimeError('rotate_rows: (B,...,3) float32 rows and (B,3,3) float64 matrices expected')
    L.require_gpu(xyz, rot)
    x = xyz.contiguous()
    out = torch.empty_like(x)
    B = x.size(0)
    L.call('mvp_rotate_rows_f32', x, L.ptr(x), L.ptr(rot.contiguous()), B, x.numel() // (3 * B) if B else 0, L.ptr(out))
    return out
