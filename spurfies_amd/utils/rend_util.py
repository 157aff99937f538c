"""Camera helpers with the reference's names (spurfies/utils/rend_util.py:14-22,60-95,143-156).
Tiny elementwise work: stays PyTorch on the GPU (rend_util row)."""
import torch
import torch.nn.functional as F


def lift(x, y, z, intrinsics):
    K = intrinsics
    fx, fy = K[:, 0, 0].unsqueeze(-1), K[:, 1, 1].unsqueeze(-1)
    cx, cy = K[:, 0, 2].unsqueeze(-1), K[:, 1, 2].unsqueeze(-1)
    sk = K[:, 0, 1].unsqueeze(-1)
    x_lift = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    y_lift = (y - cy) / fy * z
    return torch.stack((x_lift, y_lift, z, torch.ones_like(z)), dim=-1)


def get_camera_params(uv, pose, intrinsics):
    """uv [B,N,2], pose [B,4,4] (cam-to-world), intrinsics [B,4,4]|[B,3,3] -> (ray_dirs [B,N,3] unit, cam_loc [B,3])."""
    if pose.shape[1] == 7:
        raise NotImplementedError("quaternion poses are not used on the DTU / MipNeRF-360 path")
    cam_loc = pose[:, :3, 3]
    x, y = uv[:, :, 0], uv[:, :, 1]
    cam = lift(x, y, torch.ones_like(x), intrinsics).permute(0, 2, 1)
    world = (torch.bmm(pose[:, :3, :3], cam[:, :3, :]) + pose[:, :3, 3:]).permute(0, 2, 1)
    return F.normalize(world - cam_loc[:, None, :], dim=2), cam_loc


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.0) / 2.0, (img2 + 1.0) / 2.0
    mse = torch.mean((img1 - img2) ** 2)
    # log(10) as the float32 constant the reference forms on the device: building a device tensor from a Python scalar is a
    # synchronous host-to-device copy, i.e. a full stop of the otherwise sync-free optimisation loop
    return -10.0 * torch.log(mse) / _LOG10


_LOG10 = float(torch.log(torch.tensor(10.0)))
