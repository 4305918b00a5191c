"""TUM RGB-D style dataset writer (PNG pairs + associate.txt + ground truth) for the run_vo driver.

The reference's `app/run_vo.cpp:36-65` reads `<dataset_dir>/associate.txt` (lines
`rgbT rgbFile depthT depthFile`) and decodes the images with cv::imread; this module produces that
layout from the seeded synthetic generator so the same driver loop can be exercised end to end
without the TUM download (no network here).  PNG encoding uses only zlib.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path: str, img: np.ndarray, filter_type: int = 0) -> None:
    """8-bit RGB (H,W,3), 8-bit grey (H,W) or 16-bit grey (H,W uint16).  filter_type 0..2 (None/Sub/Up)."""
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint16:
        h, w = img.shape
        depth, ctype, raw = 16, 0, img.astype(">u2").view(np.uint8).reshape(h, 2 * w)
        bpp = 2
    elif img.ndim == 3:
        h, w, _ = img.shape
        depth, ctype, raw, bpp = 8, 2, img.reshape(h, 3 * w), 3
    else:
        h, w = img.shape
        depth, ctype, raw, bpp = 8, 0, img, 1
    raw = raw.astype(np.int16)
    if filter_type == 1:
        f = raw.copy(); f[:, bpp:] -= raw[:, :-bpp]
    elif filter_type == 2:
        f = raw.copy(); f[1:] -= raw[:-1]
    else:
        f = raw
    rows = np.concatenate([np.full((h, 1), filter_type, np.uint8), (f & 0xFF).astype(np.uint8)], axis=1)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)))
        comp = zlib.compress(rows.tobytes(), 1)
        half = len(comp) // 2
        fh.write(_chunk(b"IDAT", comp[:half]))                 # two IDAT chunks: decoders must concatenate
        fh.write(_chunk(b"IDAT", comp[half:]))
        fh.write(_chunk(b"IEND", b""))


def write_tum_dataset(root: str, bgr: np.ndarray, depth: np.ndarray, stamps, T_wc=None) -> None:
    """root/rgb/*.png (RGB order, as on disk), root/depth/*.png (16-bit), root/associate.txt, root/groundtruth.txt."""
    from . import capi
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    with open(os.path.join(root, "associate.txt"), "w") as fa:
        for i, t in enumerate(stamps):
            name = "%.6f.png" % t
            write_png(os.path.join(root, "rgb", name), bgr[i][:, :, ::-1], filter_type=i % 3)
            write_png(os.path.join(root, "depth", name), depth[i], filter_type=(i + 1) % 3)
            fa.write("%.6f rgb/%s %.6f depth/%s\n" % (t, name, t, name))
    if T_wc is not None:
        with open(os.path.join(root, "groundtruth.txt"), "w") as fg:
            fg.write("# ground truth trajectory\n# timestamp tx ty tz qx qy qz qw\n")
            for i, t in enumerate(stamps):
                fg.write("%.6f %s\n" % (t, " ".join("%.9f" % v for v in capi.pose12_to_tum(T_wc[i]))))


def write_config(path: str, dataset_dir: str, output_file: str, **overrides) -> None:
    """config/default.yaml of the reference with the given dataset / output paths."""
    keys = {"camera.fx": 517.3, "camera.fy": 516.5, "camera.cx": 318.6, "camera.cy": 255.3, "camera.depth_scale": 5000,
            "number_of_features": 500, "scale_factor": 1.2, "level_pyramid": 8, "match_ratio": 2.0, "max_num_lost": 10,
            "min_inliers": 10, "keyframe_rotation": 0.05, "keyframe_translation": 0.05, "enable_local_optimization": 1,
            "chi2_th": 1, "enable_viewer": 0}
    keys.update(overrides)
    with open(path, "w") as f:
        f.write("%YAML:1.0\n# generated\ndataset_dir: " + dataset_dir + "\noutput_file: " + output_file + "\n")
        for k, v in keys.items():
            f.write("%s: %s\n" % (k, v))
