"""tests/golden/vipe_small.npz: outputs of the reference's `load_vipe_data` (vipe_utils.py:172-270, imported from /root/reference) on
a small synthetic ViPE folder that the script also stores (as the decoder-free sidecars gen3c_amd.vipe_utils reads). decord and
OpenEXR are absent from this image: the reference's two decoder calls are served by stand-in modules that hand back the same arrays the
sidecars hold, so everything AFTER decoding (index clamping / last-frame repetition, pose inversion, intrinsics adjustment, resize, crop,
value ranges) is the reference's own arithmetic. Two cases: a window inside the clip, and one that runs past its end."""
import io
import sys
import types
import zipfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import ref_shims  # noqa: E402

ref_shims.install()

rs = np.random.RandomState(7)
F_, H, W = 7, 36, 64
rgb = rs.randint(0, 256, size=(F_, H, W, 3)).astype(np.uint8)
depth = (1.0 + 4.0 * rs.rand(F_, H, W)).astype(np.float16).astype(np.float32)  # EXR stores half floats
inds = np.arange(F_) * 1 + 0
c2w = np.tile(np.eye(4, dtype=np.float32), (F_, 1, 1))
for i in range(F_):
    a = 0.05 * i
    c2w[i, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    c2w[i, :3, 3] = [0.1 * i, -0.02 * i, 0.03 * i]
intr = np.stack([np.array([50.0 + i, 52.0 + i, W / 2 - 0.5 * i, H / 2 + 0.25 * i], np.float32) for i in range(F_)])


class _Batch:
    def __init__(self, a):
        self._a = a

    def asnumpy(self):
        return self._a


class VideoReader:
    def __init__(self, path, num_threads=1):
        pass

    def __len__(self):
        return F_

    def get_batch(self, idx):
        return _Batch(rgb[list(idx)])


class _ExrFile:
    def __init__(self, f):
        self._i = int(np.frombuffer(f.read(), np.int32)[0])

    def header(self):
        box = types.SimpleNamespace(min=types.SimpleNamespace(x=0, y=0), max=types.SimpleNamespace(x=W - 1, y=H - 1))
        return {"dataWindow": box}

    def channel(self, name):
        assert name == "Z"
        return depth[self._i].astype(np.float16).tobytes()


sys.modules["decord"] = types.SimpleNamespace(VideoReader=VideoReader)
sys.modules["OpenEXR"] = types.SimpleNamespace(InputFile=_ExrFile)
from cosmos_predict1.diffusion.inference.vipe_utils import load_vipe_data  # noqa: E402

import tempfile
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    for d in ("rgb", "depth", "pose", "intrinsics"):
        (td / d).mkdir()
    (td / "rgb" / "clip.mp4").write_bytes(b"")  # the stand-in VideoReader ignores the content
    with zipfile.ZipFile(td / "depth" / "clip.zip", "w") as zf:
        for i in range(F_):
            zf.writestr(f"{i:05d}.exr", np.array([i], np.int32).tobytes())  # the stand-in OpenEXR reads the frame number back
    np.savez(td / "pose" / "clip.npz", inds=inds, data=c2w.reshape(F_, 16))
    np.savez(td / "intrinsics" / "clip.npz", inds=inds, data=intr)
    out = dict(rgb=rgb, depth=depth, inds=inds, c2w=c2w, intr=intr)
    for tag, start, n in (("inside", 1, 4), ("past_end", 5, 5), ("beyond", 9, 3)):
        fr, dp, mk, w2c, K = load_vipe_data(str(td), start, resize_hw=(45, 80), crop_hw=(40, 72), num_frames=n)
        out.update({f"{tag}:frames": fr.numpy(), f"{tag}:depth": dp.numpy(), f"{tag}:mask": mk.numpy(), f"{tag}:w2c": w2c.numpy(), f"{tag}:K": K.numpy(),
                    f"{tag}:args": np.array([start, n])})
np.savez_compressed(ROOT / "tests" / "golden" / "vipe_small.npz", **out)
print({k: v.shape for k, v in out.items()})
