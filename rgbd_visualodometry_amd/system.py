"""Python driver of the libmyslam-compatible host layer (include/myslam_c.h).

``VoSystem`` = one independent RGB-D stream: Camera + FrontEnd + Backend + map, i.e. what the
reference's ``app/run_vo.cpp:73-117`` sets up and loops over.  The product library is
``rgbd_visualodometry_amd/host/libmyslam_amd.so`` (links the HIP C-ABI library); there is no CPU
fallback.  Tests / the CPU baseline pass the oracle library path explicitly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST_LIB = os.path.join(HERE, "host", "libmyslam_amd.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "liboracle_vo.so")


class Options(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("depth_scale", C.c_float), ("number_of_features", C.c_int32),
                ("scale_factor", C.c_float), ("level_pyramid", C.c_int32), ("match_ratio", C.c_float),
                ("max_num_lost", C.c_int32), ("min_inliers", C.c_int32), ("keyframe_rotation", C.c_double),
                ("keyframe_translation", C.c_double), ("enable_local_optimization", C.c_int32), ("chi2_th", C.c_float),
                ("ransac_iterations", C.c_int32), ("backend_lag_frames", C.c_int32), ("max_frames_in_flight", C.c_int32), ("track_batch", C.c_int32), ("map_capacity", C.c_int32),
                ("device", C.c_int32), ("verbose", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("frames", C.c_int32), ("keyframes", C.c_int32), ("lost", C.c_int32), ("state", C.c_int32),
                ("last_keypoints", C.c_int32), ("last_candidates", C.c_int32), ("last_matches", C.c_int32),
                ("last_ransac_inliers", C.c_int32), ("last_lm_inliers", C.c_int32), ("map_points", C.c_int32),
                ("ba_runs", C.c_int32), ("ba_poses", C.c_int32), ("ba_fixed", C.c_int32), ("ba_points", C.c_int32),
                ("ba_edges", C.c_int32), ("ba_outliers", C.c_int32), ("ba_ms", C.c_double),
                ("ms_extract", C.c_double), ("ms_track", C.c_double), ("ms_keyframe", C.c_double), ("ms_backend", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


SYMBOLS = ["myslam_default_options", "myslam_system_create", "myslam_system_destroy", "myslam_prefetch",
           "myslam_add_frame", "myslam_add_prefetched", "myslam_get_stats", "myslam_get_context", "myslam_last_error", "myslam_backend_name"]

_libs = {}


def _load(path: str):
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError("host library not found: %s (run __graft_entry__.build())" % path)
        lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for s in SYMBOLS:
            getattr(lib, s)
        lib.myslam_last_error.restype = C.c_char_p
        lib.myslam_backend_name.restype = C.c_char_p
        lib.myslam_system_destroy.restype = None
        lib.myslam_system_create.argtypes = [C.POINTER(Options), C.c_char_p, C.POINTER(C.c_void_p)]
        lib.myslam_system_destroy.argtypes = [C.c_void_p]
        lib.myslam_prefetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.myslam_add_frame.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]
        lib.myslam_add_prefetched.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        lib.myslam_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.myslam_get_context.argtypes = [C.c_void_p]
        lib.myslam_get_context.restype = C.c_void_p
        _libs[path] = lib
    return _libs[path]


class VoSystem:
    def __init__(self, lib_path: Optional[str] = None, yaml: Optional[str] = None, **options):
        self.lib = _load(lib_path or HOST_LIB)
        self.opt = Options()
        self.lib.myslam_default_options(C.byref(self.opt))
        for k, v in options.items():
            setattr(self.opt, k, v)
        self.h = C.c_void_p()
        rc = self.lib.myslam_system_create(C.byref(self.opt), yaml.encode() if yaml else None, C.byref(self.h))
        self._check(rc, "myslam_system_create")

    @property
    def backend(self) -> str:
        return self.lib.myslam_backend_name().decode()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.myslam_last_error().decode()))

    def close(self):
        if self.h:
            self.lib.myslam_system_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_frame(self, stamp: float, bgr: np.ndarray, depth: np.ndarray):
        """Host-memory frame -> (tracked, T_wc[12])."""
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        ok = C.c_int()
        T = np.zeros(12)
        self._check(self.lib.myslam_add_frame(self.h, stamp, bgr.ctypes.data, depth.ctypes.data, bgr.strides[0], depth.strides[0], 0,
                                              C.byref(ok), T.ctypes.data), "myslam_add_frame")
        return bool(ok.value), T

    def add_frame_device(self, stamp: float, bgr_ptr: int, depth_ptr: int, bgr_stride: int, depth_stride: int):
        ok = C.c_int()
        T = np.zeros(12)
        self._check(self.lib.myslam_add_frame(self.h, stamp, C.c_void_p(bgr_ptr), C.c_void_p(depth_ptr), bgr_stride, depth_stride, 1,
                                              C.byref(ok), T.ctypes.data), "myslam_add_frame")
        return bool(ok.value), T

    def prefetch(self, stamps: Sequence[float], bgr_ptrs: Sequence[int], depth_ptrs: Sequence[int], bgr_stride: int,
                 depth_stride: int, on_device: bool):
        n = len(bgr_ptrs)
        st = np.ascontiguousarray(stamps, dtype=np.float64)
        b = (C.c_void_p * n)(*bgr_ptrs)
        d = (C.c_void_p * n)(*depth_ptrs)
        self._check(self.lib.myslam_prefetch(self.h, n, st.ctypes.data, C.cast(b, C.c_void_p), C.cast(d, C.c_void_p), bgr_stride,
                                             depth_stride, int(on_device)), "myslam_prefetch")

    def add_prefetched(self):
        ok = C.c_int()
        T = np.zeros(12)
        self._check(self.lib.myslam_add_prefetched(self.h, C.byref(ok), T.ctypes.data), "myslam_add_prefetched")
        return bool(ok.value), T

    def context_handle(self) -> int:
        return self.lib.myslam_get_context(self.h)

    def stats(self) -> dict:
        st = Stats()
        self._check(self.lib.myslam_get_stats(self.h, C.byref(st)), "myslam_get_stats")
        return st.asdict()
