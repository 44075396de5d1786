"""Device-side counterpart of the loader's geometric augmentation (mvpnet/data/scannet_2d3d.py:293-296 horizontal flip per
view, :400-409 rotation about z), for batches that are lifted on the device (`depth`, `cam_matrix`, `pose` in the data dict
instead of loader-computed `knn_indices` / `image_xyz`).

The random DRAWS stay on the host, as in the reference (numpy RNG: one `rand()` per view for the flip, one `uniform(lo, hi)`
angle per chunk); what they select is applied by the HIP kernels where the reference applies it: the flip changes the flat
pixel ids / feature rows (mvp_lift_aug_f32), the rotation acts on `points` and the gathered `image_xyz` after the k-NN search
(float64 product, one rounding to float32 -- scipy's Rotation.apply).  `color_jitter` is an image-space PIL transform of the
loader and is not part of this path."""
import numpy as np
import torch


def z_rotation_matrix(angle_deg):
    """3x3 float64 matrix of scipy.spatial.transform.Rotation.from_euler('z', angle, degrees=True): taken from scipy itself
    when it is installed (the reference's own dependency), otherwise from the same unit-quaternion formula."""
    try:
        from scipy.spatial.transform import Rotation
        return Rotation.from_euler('z', float(angle_deg), degrees=True).as_matrix().astype(np.float64)
    except ImportError:
        half = np.deg2rad(float(angle_deg)) / 2.0
        z, w = np.sin(half), np.cos(half)
        n = np.sqrt(z * z + w * w)
        z, w = z / n, w / n
        z2, w2, zw = z * z, w * w, z * w
        return np.array([[w2 - z2, -2.0 * zw, 0.0], [2.0 * zw, w2 - z2, 0.0], [0.0, 0.0, z2 + w2]], np.float64)


class DeviceAugmentation(object):
    """flip: probability of mirroring each view; z_rot: () or (low, high) degrees.  __call__(batch) adds
      'flip'  (B, nv) uint8 and mirrors the flagged views of batch['images'] (the 2D network must see the mirrored image),
      'z_rot' (B, 3, 3) float64 rotation matrices,
    which MVPNet3D.forward / ops.lift consume.  rng: a numpy RandomState / Generator-like with rand() and uniform()."""

    def __init__(self, flip=0.0, z_rot=(), rng=None):
        self.flip = float(flip)
        self.z_rot = tuple(z_rot) if z_rot else ()
        if self.z_rot and len(self.z_rot) != 2:
            raise ValueError('z_rot must be () or (low, high) in degrees')
        self.rng = np.random if rng is None else rng

    def __call__(self, batch):
        ref = batch['depth'] if 'depth' in batch else batch['images']
        B, nv = int(ref.shape[0]), int(ref.shape[1])
        dev = ref.device
        if self.flip:
            flags = np.array([[self.rng.rand() < self.flip for _ in range(nv)] for _ in range(B)], dtype=bool)
            flip = torch.from_numpy(flags.astype(np.uint8)).to(dev)
            batch['flip'] = flip
            if 'images' in batch and flags.any():
                images = batch['images']                                   # (B, nv, 3, h, w)
                batch['images'] = torch.where(flip.bool().view(B, nv, 1, 1, 1), images.flip(-1), images)
        if self.z_rot:
            mats = np.stack([z_rotation_matrix(self.rng.uniform(low=self.z_rot[0], high=self.z_rot[1])) for _ in range(B)])
            batch['z_rot'] = torch.from_numpy(mats).to(dev)
        return batch
