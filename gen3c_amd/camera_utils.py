"""Camera-path generation for the 3D-cache renders (host-side fp32 geometry, 121 4x4 matrices per path).

Same functions, arguments and results as cosmos_predict1/diffusion/inference/camera_utils.py:21-222
(`look_at_matrix`, `create_horizontal_trajectory`, `create_spiral_trajectory`, `generate_camera_trajectory`); the
per-frame Python loops of the reference are kept as plain tensor code - this is not a hot path."""
from __future__ import annotations

import math

import torch


def look_at_matrix(camera_pos: torch.Tensor, target: torch.Tensor, invert_pos: bool = True) -> torch.Tensor:
    """Y-up look-at view matrix; translation column is -camera_pos (camera_utils.py:30-47)."""
    forward = (target - camera_pos).float()
    forward = forward / torch.norm(forward)
    up = torch.tensor([0.0, 1.0, 0.0], device=camera_pos.device)
    right = torch.linalg.cross(up, forward)
    right = right / torch.norm(right)
    up = torch.linalg.cross(forward, right)
    m = torch.eye(4, device=camera_pos.device)
    m[0, :3], m[1, :3], m[2, :3] = right, up, forward
    m[:3, 3] = (-camera_pos) if invert_pos else camera_pos
    return m


def _views(positions, look_at, camera_rotation):
    out = []
    for pos in positions:
        if camera_rotation == "trajectory_aligned":
            tgt = look_at + pos * 2
        elif camera_rotation == "center_facing":
            tgt = look_at
        elif camera_rotation == "no_rotation":
            tgt = look_at + pos
        else:
            raise ValueError("Camera rotation should be center_facing, trajectory_aligned or no_rotation")
        out.append(look_at_matrix(pos, tgt))  # initial camera position is the origin
    return torch.stack(out)


def _apply(traj: torch.Tensor, w2c: torch.Tensor) -> torch.Tensor:
    if w2c.dim() == 2:
        w2c = w2c.unsqueeze(0).expand(traj.shape[0], -1, -1)
    return torch.bmm(traj, w2c)


def create_horizontal_trajectory(world_to_camera_matrix, center_depth, positive=True, n_steps=13, distance=0.1, device="cuda",
                                 axis="x", camera_rotation="center_facing"):
    """Linear dolly along one camera axis, i*distance*center_depth/n_steps per frame (camera_utils.py:49-89)."""
    if axis not in ("x", "y", "z"):
        raise ValueError("Axis should be x, y or z")
    look_at = torch.tensor([0.0, 0.0, center_depth]).to(device)
    sign = 1 if positive else -1
    k = "xyz".index(axis)
    positions = []
    for i in range(n_steps):
        p = [0, 0, 0]
        p[k] = i * distance * center_depth / n_steps * sign
        positions.append(torch.tensor(p, device=device))
    return _apply(_views(positions, look_at, camera_rotation), world_to_camera_matrix)


def create_spiral_trajectory(world_to_camera_matrix, center_depth, radius_x=0.03, radius_y=0.02, radius_z=0.0, positive=True,
                             camera_rotation="center_facing", n_steps=13, device="cuda", start_from_zero=True, num_circles=1):
    """camera_utils.py:92-139."""
    look_at = torch.tensor([0.0, 0.0, center_depth]).to(device)
    theta_max = 2 * math.pi * num_circles
    positions = []
    for i in range(n_steps):
        theta = theta_max * i / (n_steps - 1)
        if start_from_zero:
            x = radius_x * (math.cos(theta) - 1) * (1 if positive else -1) * (center_depth / 1.0)
        else:
            x = radius_x * (math.cos(theta)) * (center_depth / 1.0)
        y = radius_y * math.sin(theta) * (center_depth / 1.0)
        z = radius_z * math.sin(theta) * (center_depth / 1.0)
        positions.append(torch.tensor([x, y, z], device=device))
    return _apply(_views(positions, look_at, camera_rotation), world_to_camera_matrix)


_LINEAR = {"left": (False, "x"), "right": (True, "x"), "up": (False, "y"), "down": (True, "y"), "zoom_in": (True, "z"), "zoom_out": (False, "z")}


def generate_camera_trajectory(trajectory_type: str, initial_w2c: torch.Tensor, initial_intrinsics: torch.Tensor, num_frames: int,
                               movement_distance: float, camera_rotation: str, center_depth: float = 1.0, device: str = "cuda"):
    """-> (w2cs [1,num_frames,4,4], intrinsics [1,num_frames,3,3])  (camera_utils.py:142-222)."""
    if trajectory_type in ("clockwise", "counterclockwise"):
        seq = create_spiral_trajectory(initial_w2c, center_depth, n_steps=num_frames, positive=trajectory_type == "clockwise", device=device,
                                       camera_rotation=camera_rotation, radius_x=movement_distance, radius_y=movement_distance)
    elif trajectory_type in _LINEAR:
        positive, axis = _LINEAR[trajectory_type]
        seq = create_horizontal_trajectory(initial_w2c, center_depth, n_steps=num_frames, positive=positive, axis=axis,
                                           distance=movement_distance, device=device, camera_rotation=camera_rotation)
    else:
        raise ValueError(f"Unsupported trajectory type: {trajectory_type}")
    w2cs = seq.unsqueeze(0)
    if initial_intrinsics.dim() == 2:
        Ks = initial_intrinsics.unsqueeze(0).unsqueeze(0).repeat(1, num_frames, 1, 1)
    else:
        Ks = initial_intrinsics.unsqueeze(0)
    return w2cs, Ks


def align_depth(source_depth: torch.Tensor, target_depth: torch.Tensor, target_mask: torch.Tensor | None, k: torch.Tensor | None = None,
                c2w: torch.Tensor | None = None, alignment_method: str = "rigid", num_iters: int = 100, lambda_arap: float = 0.1,
                smoothing_kernel_size: int = 3) -> torch.Tensor:
    """Align a predicted (H, W) depth map to the depth rendered from the 3D cache (camera_utils.py:275-345).

    "rigid": affine fit in inverse depth after 10 %/90 % quantile outlier rejection (:225-272); "non_rigid": additionally
    `num_iters` Adam steps on a per-pixel scale map (data loss on unprojected points inside target_mask + lambda_arap * 3x3
    ARAP smoothness). One HIP library call on the current stream (csrc/align.hip); fp32; inputs must live on the GPU."""
    from . import _lib
    if alignment_method not in ("rigid", "non_rigid"):
        raise NotImplementedError(alignment_method)
    if source_depth.dim() != 2 or source_depth.shape != target_depth.shape:
        raise ValueError("align_depth expects (H, W) source and target depth maps of the same shape")
    non_rigid = alignment_method == "non_rigid"
    if non_rigid and (k is None or c2w is None):
        raise ValueError("Camera intrinsics (k) and camera-to-world matrix (c2w) are required for non-rigid alignment")
    if non_rigid and smoothing_kernel_size != 3:
        raise NotImplementedError("the HIP ARAP term is the reference's default 3x3 box (smoothing_kernel_size=3)")
    dev = source_depth.device
    H, W = source_depth.shape
    src = source_depth.detach().to(torch.float32).contiguous()
    tgt = target_depth.detach().to(dev, torch.float32).contiguous()
    msk = None if target_mask is None else (target_mask.to(dev) > 0).to(torch.uint8).contiguous()
    out = torch.empty_like(src)
    lib = _lib.load()
    nbytes = lib.g3_align_depth_workspace_bytes(H, W)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    import ctypes as C
    kinv_p = t_p = None
    if non_rigid:
        kinv = torch.linalg.inv(k.detach().to("cpu", torch.float32)).contiguous()
        tmat = torch.linalg.inv(c2w.detach().to("cpu", torch.float32))[:3].contiguous()  # unproject_points inverts its "w2c" argument
        kinv_p, t_p = C.c_void_p(kinv.data_ptr()), C.c_void_p(tmat.data_ptr())
    _lib.check(lib.g3_align_depth_f32(src.data_ptr(), tgt.data_ptr(), None if msk is None else msk.data_ptr(), kinv_p, t_p,
                                      int(non_rigid), int(num_iters), float(lambda_arap), 1e-3, out.data_ptr(), ws_ptr, nbytes, H, W,
                                      torch.cuda.current_stream(dev).cuda_stream), "g3_align_depth_f32")
    return out
