"""tests/golden/cache_rows_f.npz: outputs of the REFERENCE's multi-view / dynamic / persistent-model code (imported read-only from
/root/reference, CPU tensors) for the SURVEY 8(f) rows:
  f2  Cache3D_BufferSelector.render_cache          (cache_3d.py:346-421)  4 key frames, 9 targets, top-2 selection + near-full exclusivity
  f3  Cache4D.render_cache with start_frame_idx     (cache_3d.py:151-236, 424-433)  9 per-frame sources, two 5-frame windows
  f4  Gen3cPersistentModel.seed_model_from_values (multi-frame branch), prepare_camera_for_inference, resize_intrinsics
      (gen3c_persistent.py:35-52, 138-268, 518-536). That module cannot be imported here (it imports MoGe and the whole pipeline at
      module level), so the three functions are compiled from its source text with `ast` and run with the reference's own Cache4D.
"""
from __future__ import annotations

import ast
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"
from gen_golden_warp import look_left, scene  # noqa: E402  (before install(): it pushes the repo root to the front of sys.path)
import ref_shims  # noqa: E402

ref_shims.install()
sys.path.remove(ref_shims.REFERENCE_ROOT)
sys.path.insert(0, ref_shims.REFERENCE_ROOT)  # the reference's cosmos_predict1 must win over this repository's import-path shim
from cosmos_predict1.diffusion.inference.cache_3d import Cache3D_BufferSelector, Cache4D  # noqa: E402

H, W = 48, 64
out = {}


def key_view(i, n):
    depth, img, K = scene(H, W, seed=i)
    depth = depth + 0.15 * i
    img = np.clip(img * (1.0 - 0.1 * i) + 0.05 * i, -1, 1).astype(np.float32)
    w2c = look_left(-0.25 + 0.18 * i, 0.04 * (i - n / 2))
    mask = np.ones((1, H, W), np.float32)
    mask[:, :, : 4 * i] = 0  # each key frame hides a different strip
    return img, depth[None].astype(np.float32), mask, K, w2c


# ---------------------------------------------------------------- f2: buffer selector
N = 4
views = [key_view(i, N) for i in range(N)]
imgs, deps, msks, Ks, w2cs = (np.stack([v[j] for v in views]) for j in range(5))
T = 9
tw2c = np.stack([look_left(-0.3 + 0.08 * t, 0.01 * t) for t in range(T)])
tK = np.broadcast_to(Ks[-1], (T, 3, 3)).copy()
out.update(sel_images=imgs, sel_depth=deps, sel_mask=msks, sel_K=Ks, sel_w2c=w2cs, sel_tw2c=tw2c, sel_tK=tK)
for tag, kw in (("top2", dict(frame_buffer_max=2)), ("top2_nomax", dict(frame_buffer_max=2, mask_for_max_buffer_model=False)),
                ("thr70", dict(frame_buffer_max=2, mask_full_threshold=0.7)), ("all", dict(frame_buffer_max=4))):
    c = Cache3D_BufferSelector(input_image=torch.from_numpy(imgs)[None], input_depth=torch.from_numpy(deps)[None], input_mask=torch.from_numpy(msks)[None],
                               input_w2c=torch.from_numpy(w2cs)[None], input_intrinsics=torch.from_numpy(Ks)[None], filter_points_threshold=0.05,
                               input_format=["B", "N", "C", "H", "W"], foreground_masking=False, device="cpu", **kw)
    px, mk = c.render_cache(torch.from_numpy(tw2c)[None], torch.from_numpy(tK)[None])
    out[f"sel:{tag}:masks"] = mk.numpy()
    if tag in ("top2", "thr70"):  # (file size: pixels only where the selection / exclusivity logic shows)
        out[f"sel:{tag}:pixels"] = px.numpy()
    print("selector", tag, tuple(px.shape), "per-frame mask means", mk.mean(dim=(3, 4, 5))[0, ::4].numpy().round(2).tolist())
d, mk = c.render_cache(torch.from_numpy(tw2c)[None], torch.from_numpy(tK)[None], render_depth=True)
out["sel:all:depth"], out["sel:all:depth_masks"] = d.numpy(), mk.numpy()

# ---------------------------------------------------------------- f3: Cache4D windows
F_ = 9
fv = [key_view(i % 4, 4) for i in range(F_)]
fi, fd, fm, fK, fw = (np.stack([v[j] for v in fv]) for j in range(5))
fw = np.stack([look_left(0.02 * i, 0.005 * i) for i in range(F_)])  # slowly moving source camera
fd = fd + 0.05 * np.arange(F_, dtype=np.float32)[:, None, None, None]
c4 = Cache4D(input_image=torch.from_numpy(fi).clone(), input_depth=torch.from_numpy(fd), input_mask=torch.from_numpy(fm), input_w2c=torch.from_numpy(fw),
             input_intrinsics=torch.from_numpy(fK), filter_points_threshold=0.05, input_format=["F", "C", "H", "W"], foreground_masking=False, device="cpu")
t4w = np.stack([look_left(0.02 * i + 0.2, 0.005 * i + 0.03) for i in range(F_)])
out.update(c4_images=fi, c4_depth=fd, c4_mask=fm, c4_K=fK, c4_w2c=fw, c4_tw2c=t4w)
for start in (0, 4):
    px, mk = c4.render_cache(torch.from_numpy(t4w[start:start + 5])[None], torch.from_numpy(fK[start:start + 5])[None], start_frame_idx=start)
    out[f"c4:{start}:pixels"], out[f"c4:{start}:masks"] = px.numpy(), mk.numpy()
    print("cache4d window", start, tuple(px.shape), float(mk.mean()))

# ---------------------------------------------------------------- f4: persistent-model helpers, compiled from the reference's source
src = (Path(ref_shims.REFERENCE_ROOT) / "cosmos_predict1/diffusion/inference/gen3c_persistent.py").read_text()
tree = ast.parse(src)
keep = []
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name == "resize_intrinsics":
        keep.append(node)
    if isinstance(node, ast.ClassDef) and node.name == "Gen3cPersistentModel":
        node.body = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in ("seed_model_from_values", "prepare_camera_for_inference", "W", "H")]
        node.decorator_list = []
        keep.append(node)
mod = ast.Module(body=keep, type_ignores=[])
ast.fix_missing_locations(mod)


class _IM:
    BICUBIC = "bicubic"


def _tv_resize(img, size, interpolation=None, antialias=None):  # torchvision.transforms.functional.resize on a float tensor
    assert interpolation == _IM.BICUBIC and antialias
    return torch.nn.functional.interpolate(img, size=size, mode="bicubic", antialias=True, align_corners=False)


_tvf = types.SimpleNamespace(resize=_tv_resize, InterpolationMode=_IM)
sys.modules["torchvision.transforms.functional"] = _tvf
sys.modules["torchvision.transforms"].functional = _tvf  # `import torchvision.transforms.functional as F` resolves through the parent's attribute
ns = dict(np=np, torch=torch, Cache4D=Cache4D, device_with_rank=lambda d: d, F=torch.nn.functional)
exec(compile(mod, "<gen3c_persistent.py (reference, selected functions)>", "exec"), ns)
resize_intrinsics, RefModel = ns["resize_intrinsics"], ns["Gen3cPersistentModel"]

Kb = np.stack([fK[0], fK[1] * np.array([[1.1], [0.9], [1.0]], np.float32)])
out["ri_in"] = Kb
out["ri_plain"] = resize_intrinsics(Kb, (H, W), (96, 160))
out["ri_crop"] = resize_intrinsics(torch.from_numpy(Kb), (H, W), (90, 160), crop_size=(88, 152)).numpy()

m = RefModel.__new__(RefModel)
PH, PW = 64, 96   # the model's working resolution differs from the seeding images' -> the bicubic resize branch runs
m.args = types.SimpleNamespace(width=PW, height=PH, filter_points_threshold=0.05, foreground_masking=False, noise_aug_strength=0.0)
m.device_with_rank = "cpu"
n_seed = 3
images01 = (np.transpose(fi[:n_seed], (0, 2, 3, 1)) + 1) / 2
focal = np.stack([fK[:n_seed, 0, 0], fK[:n_seed, 1, 1]], 1)
pp_rel = np.stack([fK[:n_seed, 0, 2] / W, fK[:n_seed, 1, 2] / H], 1).astype(np.float32)
res = np.tile([[W, H]], (n_seed, 1))
ret = m.seed_model_from_values(images01.astype(np.float32), fd[:n_seed, 0], fw[:n_seed], focal, pp_rel, res, masks_np=fm[:n_seed, 0])
out.update(seed_images01=images01.astype(np.float32), seed_depths=fd[:n_seed, 0], seed_w2c=fw[:n_seed], seed_focal=focal, seed_pp_rel=pp_rel, seed_res=res,
           seed_masks=fm[:n_seed, 0], seed_ret_w2c=ret[0], seed_ret_focal=ret[1], seed_ret_pp=ret[2], seed_ret_res=ret[3],
           seed_seeding_image=m.seeding_image.numpy(), seed_cache_points=m.cache.input_points.numpy(), seed_cache_mask=m.cache.input_mask.numpy(),
           seed_cache_image=m.cache.input_image.numpy(), seed_hw=np.array([PH, PW]))
cams, intr = m.prepare_camera_for_inference(t4w[:5], fK[:5], (H, W), (PH, PW))
out["pc_w2c"], out["pc_K"] = cams.numpy(), intr.numpy()
print("persistent: seeding image", tuple(m.seeding_image.shape), "cache points", tuple(m.cache.input_points.shape), "prepared cams", tuple(cams.shape), tuple(intr.shape))

np.savez_compressed(GOLD / "cache_rows_f.npz", **out)
print("file MB", (GOLD / "cache_rows_f.npz").stat().st_size / 1e6)
