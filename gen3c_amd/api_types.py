"""Request / response records of the GEN3C serving boundary: the contract between a client (the reference's GUI,
`gui/api/client.py`) and a resident model. Mirrors the reference's dataclasses FIELD FOR FIELD - names, shapes, dtypes, defaults, the
padding / trimming and compression round trips - so that an object built for one side is accepted by the other:

  reference                                            here
  gui/api/api_types.py:30-135   RequestBase            RequestBase        cameras_to_world [n,3,4], focal_lengths [n,2] (pixels),
                                                                          principal_points [n,2] (relative), resolutions [n,2] (w,h)
  gui/api/api_types.py:138-205  SeedingRequest         SeedingRequest     images [n,h,w,3] in 0..1, depths [n,h,w] | None, masks | None
  gui/api/api_types.py:208-252  CompressedSeedingRequest                  *_compressed: list[bytes] + *_format
  gui/api/api_types.py:255-299  SeedingResult          SeedingResult      depths the model estimated when the request had none
  gui/api/api_types.py:302-334  InferenceRequest       InferenceRequest   timestamps [n], framerate, return_depths, ...
  gui/api/api_types.py:337-373  InferenceResult        InferenceResult    result_ids, images [n,h,w,3], depths [n,h,w], runtime_ms
  gui/api/api_types.py:377-452  CompressedInferenceResult
  gui/api/api_types.py:455-474  RequestState, PendingRequest
  gui/api/encoding.py:22-229    CompressionFormat, compress_images, decompress_buffer, pad_or_trim_array, pad_or_trim_encoded_buffers

The wire format the reference's server puts around these (FastAPI routes, msgpack) is control plane and not built. Lossless NPZ
buffers need numpy only; JPG / PNG / EXR / MP4 buffers need OpenCV exactly as in the reference (absent in this image: those
formats raise with that message instead of silently substituting another codec).
"""
from __future__ import annotations

import asyncio
import io
from dataclasses import dataclass, fields
from enum import Enum
from typing import List, Optional, Tuple

import numpy as np


# --------------------------------------------------------------------------------------------------------------- buffers
class CompressionFormat(Enum):
    JPG = "jpg"
    PNG = "png"
    EXR = "exr"
    MP4 = "mp4"
    NPZ = "npz"


IMAGE_COMPRESSION_FORMATS = (CompressionFormat.JPG, CompressionFormat.PNG, CompressionFormat.EXR)


def _cv2():
    try:
        import cv2  # noqa: PLC0415
        return cv2
    except ImportError as e:  # the reference imports it unconditionally (encoding.py:19)
        raise RuntimeError("JPG / PNG / EXR / MP4 buffers need OpenCV (cv2), which is not installed; NPZ buffers do not") from e


def _npz_bytes(arr: np.ndarray) -> bytes:
    with io.BytesIO() as f:
        np.savez_compressed(f, arr)
        return f.getvalue()


def _npz_array(buf: bytes) -> np.ndarray:
    z = np.load(io.BytesIO(buf), allow_pickle=False)
    if hasattr(z, "files"):
        assert len(z.files) == 1, z.files
        return z[z.files[0]]
    return z


def compress_images(images: Optional[np.ndarray], format: CompressionFormat, is_depth: bool = False, is_bool: bool = False) -> Optional[List[bytes]]:
    """[n,h,w,3] colours in 0..1 -> uint8; [n,h,w] depths stay float32 (EXR / NPZ only); [n,h,w] masks stay bool (NPZ only).
    NPZ: ONE buffer for the whole batch; image formats: one buffer per image (encoding.py:31-73)."""
    if images is None:
        return None
    if is_depth or is_bool:
        assert images.ndim == 3, images.shape
    else:
        assert images.ndim == 4 and images.shape[-1] == 3, images.shape
    if is_depth:
        assert format in (CompressionFormat.EXR, CompressionFormat.NPZ), "Depth images must be encoded as EXR or NPZ"
        data = images.astype(np.float32)
    elif is_bool:
        assert format == CompressionFormat.NPZ, "Bool images (e.g. masks) must be encoded as NPZ"
        data = images.astype(bool)
    else:
        data = (images * 255.0).astype(np.uint8)
    if format == CompressionFormat.NPZ:
        return [_npz_bytes(data)]
    assert format in IMAGE_COMPRESSION_FORMATS, f"Unsupported image compression format: {format}"
    cv2 = _cv2()
    flags = [int(cv2.IMWRITE_JPEG_QUALITY), 100] if format == CompressionFormat.JPG else []
    return [cv2.imencode(f".{format.value}", data[i], flags)[1].tobytes() for i in range(data.shape[0])]


def decompress_buffer(buffers: Optional[List[bytes]], format: CompressionFormat, is_depth: bool = False, is_bool: bool = False) -> Optional[np.ndarray]:
    """Inverse of compress_images; colours come back as float32 in 0..1 from the 8-bit formats, NPZ buffers exactly as stored (encoding.py:76-132)."""
    if buffers is None:
        return None
    assert not (is_depth and is_bool), "Cannot be both a depth and a bool buffer."
    if format == CompressionFormat.NPZ:
        return np.concatenate([_npz_array(b) for b in buffers], axis=0)
    cv2 = _cv2()
    out = []
    if format == CompressionFormat.MP4:
        import tempfile
        assert not is_bool and not is_depth, "Cannot decode a mask or depth from a video."
        for b in buffers:
            with tempfile.NamedTemporaryFile(suffix=".mp4") as f:
                f.write(b)
                f.flush()
                cap = cv2.VideoCapture(f.name)
                while True:
                    ok, frame = cap.read()
                    if not ok:
                        break
                    out.append((cv2.cvtColor(frame, cv2.COLOR_BGR2RGB).astype(np.float32) / 255.0)[None])
                cap.release()
        return np.concatenate(out, axis=0)
    for b in buffers:
        img = np.array(cv2.imdecode(np.frombuffer(b, dtype=np.uint8), cv2.IMREAD_ANYDEPTH if is_depth else cv2.IMREAD_ANYCOLOR))
        if is_bool:
            img = img.astype(bool)
        elif img.dtype == np.uint8:
            img = img.astype(np.float32) / 255.0
        out.append(img[None])
    return np.concatenate(out, axis=0)


def pad_or_trim_array(arr: Optional[np.ndarray], target_size: int) -> Optional[np.ndarray]:
    """First axis to `target_size`: cut from the end, or repeat the last entry (encoding.py:136-154)."""
    if arr is None:
        return None
    n = arr.shape[0]
    if n >= target_size:
        return arr if n == target_size else arr[:target_size]
    return np.concatenate([arr, np.repeat(arr[-1:], target_size - n, axis=0)], axis=0)


def pad_or_trim_encoded_buffers(buffers: Optional[List[bytes]], format: CompressionFormat, target_size: int) -> Optional[List[bytes]]:
    """The same on encoded data: per-image formats hold one buffer per entry; an NPZ buffer is re-packed; an MP4 is re-encoded (encoding.py:158-229)."""
    if buffers is None:
        return None
    if format in IMAGE_COMPRESSION_FORMATS:
        n = len(buffers)
        return buffers[:target_size] if n >= target_size else buffers + [buffers[-1]] * (target_size - n)
    if format == CompressionFormat.NPZ:
        assert len(buffers) == 1, "NPZ buffers should be a single buffer"
        return [_npz_bytes(pad_or_trim_array(_npz_array(buffers[0]), target_size))]
    if format == CompressionFormat.MP4:
        import tempfile
        cv2 = _cv2()
        assert len(buffers) == 1, "MP4 buffers should be a single buffer"
        with tempfile.NamedTemporaryFile(suffix=".mp4") as f, tempfile.NamedTemporaryFile(suffix=".mp4") as g:
            f.write(buffers[0])
            f.flush()
            cap = cv2.VideoCapture(f.name)
            size = (int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)))
            out = cv2.VideoWriter(g.name, cv2.VideoWriter_fourcc(*"mp4v"), cap.get(cv2.CAP_PROP_FPS), size)
            last = None
            for _ in range(target_size):
                ok, frame = cap.read()
                frame = frame if ok else last
                if frame is None:
                    break
                out.write(frame)
                last = frame
            out.release()
            cap.release()
            g.seek(0)
            return [g.read()]
    raise ValueError(f"Unsupported compression format: {format}")


# --------------------------------------------------------------------------------------------------------------- records
def _shallow_dict(obj) -> dict:
    """Field name -> value WITHOUT deep-copying the arrays (dataclasses.asdict would copy every image)."""
    return {f.name: getattr(obj, f.name) for f in fields(obj)}


@dataclass(kw_only=True)
class RequestBase:
    """Cameras of a batch of n frames. Intrinsics: focal lengths in pixels of `resolutions` (w, h), principal points relative (0.5 = centre)."""
    request_id: str
    cameras_to_world: np.ndarray            # [n, 3, 4]
    focal_lengths: np.ndarray               # [n, 2]
    principal_points: np.ndarray            # [n, 2]
    resolutions: Optional[np.ndarray] = None  # [n, 2] (width, height); taken from `images` when the record has them
    frame_count_without_padding: Optional[int] = None  # set by pad_to_frame_count()

    def __post_init__(self):
        imgs = getattr(self, "images", None) if hasattr(self, "images") else None
        if hasattr(self, "images"):
            if self.resolutions is None:
                self.resolutions = np.tile([[imgs.shape[2], imgs.shape[1]]], (len(self), 1))
            else:
                assert np.all(self.resolutions == (imgs.shape[2], imgs.shape[1]))
        elif self.resolutions is None:
            raise ValueError("Missing value `resolutions`")
        n = len(self)
        assert self.cameras_to_world.shape == (n, 3, 4)
        assert self.focal_lengths.shape == (n, 2)
        assert self.principal_points.shape == (n, 2)
        assert self.resolutions.shape == (n, 2)

    def __len__(self) -> int:
        return self.cameras_to_world.shape[0]

    def world_to_cameras(self) -> np.ndarray:
        """[n, 4, 4] inverses of the (completed) camera-to-world matrices."""
        c2w = np.zeros((len(self), 4, 4), dtype=self.cameras_to_world.dtype)
        c2w[:, :3] = self.cameras_to_world
        c2w[:, 3, 3] = 1.0
        return np.linalg.inv(c2w)

    def intrinsics_matrix(self, for_resolutions: Optional[np.ndarray]) -> np.ndarray:
        """[n, 3, 3] pinhole matrices in pixels (principal point made absolute), optionally rescaled to other resolutions (float64, like the reference)."""
        K = np.zeros((len(self), 3, 3))
        K[:, 0, 0], K[:, 1, 1], K[:, 2, 2] = self.focal_lengths[:, 0], self.focal_lengths[:, 1], 1.0
        K[:, 0, 2] = self.principal_points[:, 0] * self.resolutions[:, 0]
        K[:, 1, 2] = self.principal_points[:, 1] * self.resolutions[:, 1]
        if for_resolutions is not None:
            assert for_resolutions.shape == self.resolutions.shape
            K[:, 0, :] *= (for_resolutions[:, 0, None] / self.resolutions[:, 0, None])
            K[:, 1, :] *= (for_resolutions[:, 1, None] / self.resolutions[:, 1, None])
        return K

    def resolution(self) -> Tuple[int, int]:
        return self.resolutions[0, 0], self.resolutions[0, 1]

    def pad_to_frame_count(self, n_frames: int) -> None:
        self.frame_count_without_padding = len(self)
        self._adjust_frame_count(n_frames)

    def trim_to_original_frame_count(self, override_frame_count: Optional[int] = None) -> None:
        n = override_frame_count or self.frame_count_without_padding
        if n is not None:
            self._adjust_frame_count(n)

    def _adjust_frame_count(self, n_frames: int) -> None:
        for name in ("cameras_to_world", "focal_lengths", "principal_points", "resolutions"):
            setattr(self, name, pad_or_trim_array(getattr(self, name), n_frames))


@dataclass(kw_only=True)
class SeedingRequest(RequestBase):
    """Images (+ optional depths / masks) that seed the model's 3D cache."""
    images: np.ndarray                     # [n, h, w, 3] float32 in 0..1
    depths: Optional[np.ndarray]           # [n, h, w] float32 | None = estimate it
    masks: Optional[np.ndarray] = None     # [n, h, w] bool

    def __post_init__(self):
        super().__post_init__()
        n = len(self)
        assert self.images.shape[0] == n and self.images.ndim == 4, self.images.shape
        assert self.depths is None or (self.depths.shape[0] == n and self.depths.ndim == 3), self.depths.shape
        assert self.masks is None or (self.masks.shape[0] == n and self.masks.ndim == 3), self.masks.shape

    def _adjust_frame_count(self, n_frames: int) -> None:
        raise RuntimeError("SeedingRequest: _adjust_frame_count() not supported")

    def compress(self, format_rgb: CompressionFormat = CompressionFormat.JPG, format_depth: Optional[CompressionFormat] = None,
                 format_mask: Optional[CompressionFormat] = None) -> "CompressedSeedingRequest":
        format_depth = format_depth or CompressionFormat.EXR
        format_mask = format_mask or CompressionFormat.NPZ
        kw = _shallow_dict(self)
        kw.update(images=None, depths=None, masks=None)
        return CompressedSeedingRequest(images_compressed=compress_images(self.images, format_rgb), images_format=format_rgb,
                                        depths_compressed=compress_images(self.depths, format_depth, is_depth=True), depths_format=format_depth,
                                        masks_compressed=compress_images(self.masks, format_mask, is_bool=True), masks_format=format_mask, **kw)


@dataclass(kw_only=True)
class CompressedSeedingRequest(SeedingRequest):
    """SeedingRequest whose pixels travel encoded; `images` / `depths` / `masks` are 0-length placeholders until decompress()."""
    images_compressed: List[bytes]
    images_format: CompressionFormat
    depths_compressed: Optional[List[bytes]]
    depths_format: Optional[CompressionFormat]
    masks_compressed: Optional[List[bytes]]
    masks_format: Optional[CompressionFormat]

    def __post_init__(self):  # (the parent's shape checks do not apply to placeholders)
        assert (self.resolutions is not None) or (self.images is not None), "CompressedSeedingRequest: at least one of resolutions or images must be provided"
        w, h = self.resolution()
        if self.images is None:
            self.images = np.empty((0, h, w, 3), dtype=np.float32)
        if self.depths is None and self.depths_compressed is not None:
            self.depths = np.empty((0, h, w), dtype=np.float32)
        if self.masks is None and self.masks_compressed is not None:
            self.masks = np.empty((0, h, w), dtype=bool)
        assert self.images.shape[0] == 0, "CompressedSeedingRequest should not have any raw image data in `self.images` upon construction."

    def decompress(self) -> None:
        self.images = decompress_buffer(self.images_compressed, self.images_format)
        self.depths = decompress_buffer(self.depths_compressed, self.depths_format, is_depth=True)
        self.masks = decompress_buffer(self.masks_compressed, self.masks_format, is_bool=True)


@dataclass(kw_only=True)
class SeedingResult(RequestBase):
    """What seeding settled on: cameras as used, and the depths the model estimated if the request brought none."""
    depths: Optional[np.ndarray] = None    # [n, h, w]

    def __post_init__(self):
        super().__post_init__()
        if self.depths is not None:
            if self.depths.ndim == 4 and self.depths.shape[1] == 1:
                self.depths = self.depths.squeeze(1)
            assert self.depths.shape[0] == len(self) and self.depths.ndim == 3

    @staticmethod
    def from_request(req: SeedingRequest, fallback_depths: Optional[np.ndarray]) -> "SeedingResult":
        res = req.resolutions
        if fallback_depths is not None:  # (in place, as the reference does: the request's resolutions follow the estimated depths)
            res[:, 0], res[:, 1] = fallback_depths.shape[2], fallback_depths.shape[1]
        return SeedingResult(request_id=req.request_id, cameras_to_world=req.cameras_to_world, focal_lengths=req.focal_lengths,
                             principal_points=req.principal_points, resolutions=res, depths=None if req.depths is not None else fallback_depths)

    def _adjust_frame_count(self, n_frames: int) -> None:
        raise RuntimeError("SeedingRequest: _adjust_frame_count() not supported")


@dataclass(kw_only=True)
class InferenceRequest(RequestBase):
    """Cameras to generate frames for."""
    timestamps: np.ndarray                 # [n]
    framerate: float = 30.0
    return_depths: bool = False
    video_encoding_quality: int = 8        # 0..10, for compressed results
    show_cache_renderings: bool = False

    def __post_init__(self):
        super().__post_init__()
        n = len(self)
        assert self.timestamps.shape[0] == n and self.timestamps.ndim == 1, f"Timestamps: expected shape ({n},), found: {self.timestamps.shape}"

    def _adjust_frame_count(self, n_frames: int) -> None:
        super()._adjust_frame_count(n_frames)
        self.timestamps = pad_or_trim_array(self.timestamps, n_frames)


@dataclass(kw_only=True)
class InferenceResult(RequestBase):
    """Generated frames; the request's camera fields are repeated because the model may not have honoured them."""
    result_ids: List[Optional[str]]
    timestamps: np.ndarray                 # [n]
    images: np.ndarray                     # [n, h, w, 3]
    depths: Optional[np.ndarray]           # [n, h, w]
    runtime_ms: float

    def __post_init__(self):
        super().__post_init__()
        n = len(self)
        assert self.timestamps.shape[0] == n and self.timestamps.ndim == 1, f"Timestamps: expected shape ({n},), found: {self.timestamps.shape}"
        assert self.images.ndim == 4 and self.images.shape[0] == n, self.images.shape
        # depths = None (a request without return_depths) is accepted here; the reference's uncompressed record dereferences it (api_types.py:359)
        # and is only ever built with depths in its own code paths (compressed results carry None)
        assert self.depths is None or (self.depths.ndim == 3 and self.depths.shape[0] == n), self.depths.shape

    def _adjust_frame_count(self, n_frames: int) -> None:
        super()._adjust_frame_count(n_frames)
        self.timestamps = pad_or_trim_array(self.timestamps, n_frames)
        if self.images.shape[0] == 0:
            return  # placeholders of a compressed result
        self.images = pad_or_trim_array(self.images, n_frames)
        self.depths = pad_or_trim_array(self.depths, n_frames)


@dataclass(kw_only=True)
class CompressedInferenceResult(InferenceResult):
    """InferenceResult whose frames travel as ONE MP4 (or one buffer per image) and whose depths as EXR / NPZ."""
    images_compressed: List[bytes]
    images_format: CompressionFormat
    depths_compressed: Optional[List[bytes]]
    depths_format: Optional[CompressionFormat]

    def __post_init__(self):
        assert (self.resolutions is not None) or (self.images is not None), "CompressedInferenceResult: at least one of resolutions or images must be provided"
        w, h = self.resolution()
        if self.images is None:
            self.images = np.empty((0, h, w, 3), dtype=np.float32)
        if self.depths is None and self.depths_compressed is not None:
            self.depths = np.empty((0, h, w), dtype=np.float32)
        assert self.images.shape[0] == 0, "CompressedInferenceResult should not have any raw image data in `self.images` upon construction."
        if self.images_format == CompressionFormat.MP4:
            assert len(self.images_compressed) == 1, "CompressedInferenceResult: with an MP4 compressed result, there should be only one buffer (the compressed video)."
        elif self.depths_compressed is not None:
            assert len(self.depths_compressed) == len(self.images_compressed)
            assert self.depths_format in IMAGE_COMPRESSION_FORMATS, f"CompressedInferenceResult: depths_format should be an image format, found {self.depths_format}"

    def _adjust_frame_count(self, n_frames: int) -> None:
        super()._adjust_frame_count(n_frames)
        self.images_compressed = pad_or_trim_encoded_buffers(self.images_compressed, self.images_format, n_frames)
        self.depths_compressed = pad_or_trim_encoded_buffers(self.depths_compressed, self.depths_format, n_frames)

    def decompress(self) -> None:
        self.images = decompress_buffer(self.images_compressed, self.images_format, is_depth=False)
        self.depths = decompress_buffer(self.depths_compressed, self.depths_format, is_depth=True)

    def save_images(self, fname_or_directory: str) -> None:
        """Writes the encoded frames as they are: `<base>.<ext>` for a single buffer, `base_<i>.<ext>` per image otherwise."""
        import os
        path = os.path.realpath(fname_or_directory)
        stem, ext = os.path.splitext(path)
        directory, base = (path, "inference_result") if not ext else (os.path.dirname(path), os.path.basename(stem))
        os.makedirs(directory, exist_ok=True)
        single = len(self.images_compressed) == 1
        for i, buf in enumerate(self.images_compressed):
            name = f"{base}.{self.images_format.value}" if single else f"base_{i:05d}.{self.images_format.value}"
            with open(os.path.join(directory, name), "wb") as f:
                f.write(buf)


class RequestState(Enum):
    """State of an inference request (not of an HTTP request)."""
    REQUEST_PENDING = "Request pending"
    REQUEST_SENT = "Request sent"
    RESULT_PENDING = "Result pending"
    COMPLETE = "Completed"
    FAILED = "Created"  # (sic: the reference's value)


@dataclass(kw_only=True)
class PendingRequest:
    request_id: str
    state: RequestState
    message: str = ""
    task: Optional[asyncio.Task] = None
