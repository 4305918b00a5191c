"""Python driver of the libmyslam-compatible host layer (include/myslam_c.h).

``VoSystem`` = one independent RGB-D stream: Camera + FrontEnd + Backend + map, i.e. what the
reference's ``app/run_vo.cpp:73-117`` sets up and loops over.  The product library is
``rgbd_visualodometry_amd/host/libmyslam_amd.so`` (links the HIP C-ABI library); there is no CPU
fallback.  A different implementation of include/myslam_c.h is bound by passing its path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST_LIB = os.path.join(HERE, "host", "libmyslam_amd.so")


class Options(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("depth_scale", C.c_float), ("number_of_features", C.c_int32),
                ("scale_factor", C.c_float), ("level_pyramid", C.c_int32), ("match_ratio", C.c_float),
                ("max_num_lost", C.c_int32), ("min_inliers", C.c_int32), ("keyframe_rotation", C.c_double),
                ("keyframe_translation", C.c_double), ("enable_local_optimization", C.c_int32), ("chi2_th", C.c_float),
                ("ransac_iterations", C.c_int32), ("backend_lag_frames", C.c_int32), ("max_frames_in_flight", C.c_int32), ("track_batch", C.c_int32), ("map_capacity", C.c_int32),
                ("device", C.c_int32), ("verbose", C.c_int32), ("triangulate_all", C.c_int32), ("ba_device_graph", C.c_int32), ("reobserve_new_mappoints", C.c_int32), ("map_descriptors_on_device", C.c_int32), ("device_keyframes", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("frames", C.c_int32), ("keyframes", C.c_int32), ("lost", C.c_int32), ("state", C.c_int32),
                ("last_keypoints", C.c_int32), ("last_candidates", C.c_int32), ("last_matches", C.c_int32),
                ("last_ransac_inliers", C.c_int32), ("last_lm_inliers", C.c_int32), ("map_points", C.c_int32),
                ("ba_runs", C.c_int32), ("ba_poses", C.c_int32), ("ba_fixed", C.c_int32), ("ba_points", C.c_int32),
                ("ba_edges", C.c_int32), ("ba_outliers", C.c_int32), ("ba_ms", C.c_double),
                ("ms_extract", C.c_double), ("ms_track", C.c_double), ("ms_keyframe", C.c_double), ("ms_backend", C.c_double),
                ("tracked_frames", C.c_int64), ("sum_active", C.c_int64), ("sum_candidates", C.c_int64), ("sum_matches", C.c_int64),
                ("sum_ransac_inliers", C.c_int64), ("sum_lm_inliers", C.c_int64), ("sum_lm_iters", C.c_int64), ("track_launches", C.c_int64),
                ("ba_failed", C.c_int32), ("ba_capped", C.c_int32), ("triangulated", C.c_int64), ("reobserved_matches", C.c_int64),
                ("ba_sum_d3", C.c_double), ("ba_sum_d2", C.c_double), ("ba_sum_edges", C.c_int64),
                ("ba_sum_points", C.c_int64), ("ba_sum_pairs", C.c_int64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


SYMBOLS = ["myslam_default_options", "myslam_system_create", "myslam_system_destroy", "myslam_prefetch", "myslam_preload",
           "myslam_add_frame", "myslam_add_prefetched", "myslam_get_stats", "myslam_flush",
           "myslam_group_create", "myslam_group_destroy", "myslam_group_join", "myslam_group_stats", "myslam_get_context", "myslam_last_error", "myslam_backend_name",
           # taps for parity tests (include/myslam_c.h)
           "myslam_triangulate", "myslam_se3_log", "myslam_se3_exp", "myslam_keyframe_policy", "myslam_scn_add_keyframe", "myslam_scn_add_mappoint",
           "myslam_scn_observe", "myslam_scn_unobserve", "myslam_scn_covisibility", "myslam_scn_local_map", "myslam_scn_ba_graph", "myslam_scn_mappoint",
           "myslam_scn_run_ba", "myslam_scn_keyframe_pose", "myslam_materialize", "myslam_mappoint_ids"]

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
        lib.myslam_preload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.myslam_add_frame.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]
        lib.myslam_add_prefetched.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        lib.myslam_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        lib.myslam_flush.argtypes = [C.c_void_p]
        lib.myslam_group_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.myslam_group_destroy.argtypes = [C.c_void_p]
        lib.myslam_group_destroy.restype = None
        lib.myslam_group_join.argtypes = [C.c_void_p, C.c_void_p]
        lib.myslam_group_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.myslam_triangulate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        lib.myslam_se3_log.argtypes = [C.c_void_p, C.c_void_p]
        lib.myslam_se3_exp.argtypes = [C.c_void_p, C.c_void_p]
        lib.myslam_keyframe_policy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.myslam_scn_add_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.myslam_scn_add_mappoint.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.myslam_scn_observe.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float]
        lib.myslam_scn_unobserve.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
        lib.myslam_scn_covisibility.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.myslam_scn_local_map.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.myslam_scn_ba_graph.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int,
                                            C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.myslam_scn_mappoint.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        lib.myslam_scn_run_ba.argtypes = [C.c_void_p, C.c_int64]
        lib.myslam_scn_keyframe_pose.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        lib.myslam_materialize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.myslam_mappoint_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        lib.myslam_get_context.argtypes = [C.c_void_p]
        lib.myslam_get_context.restype = C.c_void_p
        _libs[path] = lib
    return _libs[path]


def triangulate(lib_path: Optional[str], T_cw: np.ndarray, pts: np.ndarray):
    """Triangulation tap (reference include/myslam/util.h:16-34) -> (xyz, ok)."""
    lib = _load(lib_path or HOST_LIB)
    T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(-1, 12); p = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(3); ok = C.c_int()
    if lib.myslam_triangulate(len(T), T.ctypes.data, p.ctypes.data, out.ctypes.data, C.byref(ok)) != 0:
        raise RuntimeError("myslam_triangulate failed")
    return out, bool(ok.value)


def se3_log(lib_path: Optional[str], T12) -> np.ndarray:
    lib = _load(lib_path or HOST_LIB); T = np.ascontiguousarray(T12, dtype=np.float64).reshape(12); d = np.zeros(6)
    lib.myslam_se3_log(T.ctypes.data, d.ctypes.data)
    return d


def se3_exp(lib_path: Optional[str], d6) -> np.ndarray:
    lib = _load(lib_path or HOST_LIB); d = np.ascontiguousarray(d6, dtype=np.float64).reshape(6); T = np.zeros(12)
    lib.myslam_se3_exp(d.ctypes.data, T.ctypes.data)
    return T


class StreamGroup:
    """Several VoSystems (independent streams) on one GPU whose tracking shares launch chains; drive each member from its own thread."""

    def __init__(self, lib_path: Optional[str] = None, device: int = 0, max_lanes: int = 64):
        self.lib = _load(lib_path or HOST_LIB)
        self.h = C.c_void_p()
        if self.lib.myslam_group_create(device, max_lanes, C.byref(self.h)) != 0:
            raise RuntimeError("myslam_group_create failed: %s" % self.lib.myslam_last_error().decode())

    def join(self, system: "VoSystem"):
        if self.lib.myslam_group_join(self.h, system.h) != 0:
            raise RuntimeError("myslam_group_join failed: %s" % self.lib.myslam_last_error().decode())

    def stats(self) -> dict:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.myslam_group_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"chains": a.value, "lanes": b.value, "requests": c.value}

    def close(self):
        if self.h:
            self.lib.myslam_group_destroy(self.h)
            self.h = C.c_void_p()


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

    def preload(self, bgr_ptrs: Sequence[int], depth_ptrs: Sequence[int], bgr_stride: int, depth_stride: int):
        """Start the uploads of the NEXT prefetch() call's frames (pinned host memory) beside the tracking of the queued ones."""
        n = len(bgr_ptrs)
        if n == 0:
            return
        b = (C.c_void_p * n)(*bgr_ptrs)
        d = (C.c_void_p * n)(*depth_ptrs)
        self._check(self.lib.myslam_preload(self.h, n, C.cast(b, C.c_void_p), C.cast(d, C.c_void_p), bgr_stride, depth_stride), "myslam_preload")

    def add_prefetched(self):
        ok = C.c_int()
        T = np.zeros(12)
        self._check(self.lib.myslam_add_prefetched(self.h, C.byref(ok), T.ctypes.data), "myslam_add_prefetched")
        return bool(ok.value), T

    def flush(self):
        """Wait for a pending overlapped local BA and merge it (Backend::Flush)."""
        self._check(self.lib.myslam_flush(self.h), "myslam_flush")

    # ---- taps for parity tests (include/myslam_c.h): hand-built scenarios, not the product path --------------------
    def scn_add_keyframe(self, T_cw) -> int:
        T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(12); out = C.c_int64()
        self._check(self.lib.myslam_scn_add_keyframe(self.h, T.ctypes.data, C.byref(out)), "myslam_scn_add_keyframe")
        return out.value

    def scn_add_mappoint(self, xyz) -> int:
        p = np.ascontiguousarray(xyz, dtype=np.float64).reshape(3); out = C.c_int64()
        self._check(self.lib.myslam_scn_add_mappoint(self.h, p.ctypes.data, C.byref(out)), "myslam_scn_add_mappoint")
        return out.value

    def scn_observe(self, kf: int, mp: int, u: float, v: float):
        self._check(self.lib.myslam_scn_observe(self.h, kf, mp, u, v), "myslam_scn_observe")

    def scn_unobserve(self, kf: int, mp: int):
        self._check(self.lib.myslam_scn_unobserve(self.h, kf, mp), "myslam_scn_unobserve")

    def scn_covisibility(self, kf: int, cap: int = 4096):
        ids = np.zeros(cap, np.int64); w = np.zeros(cap, np.int32); a = np.zeros(cap, np.uint8); n = C.c_int()
        self._check(self.lib.myslam_scn_covisibility(self.h, kf, ids.ctypes.data, w.ctypes.data, a.ctypes.data, cap, C.byref(n)), "myslam_scn_covisibility")
        return {int(ids[i]): (int(w[i]), bool(a[i])) for i in range(n.value)}

    def scn_local_map(self, kf: int, cap: int = 1 << 16):
        ids = np.zeros(cap, np.int64); n = C.c_int()
        self._check(self.lib.myslam_scn_local_map(self.h, kf, ids.ctypes.data, cap, C.byref(n)), "myslam_scn_local_map")
        return [int(v) for v in ids[:n.value]]

    def scn_ba_graph(self, kf: int, cap: int = 1 << 16):
        pid = np.zeros(cap, np.int64); xid = np.zeros(cap, np.int64); ep = np.zeros(cap, np.int32); el = np.zeros(cap, np.int32)
        uv = np.zeros((cap, 2), np.float32); npz, nf, nx, ne = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.myslam_scn_ba_graph(self.h, kf, pid.ctypes.data, cap, C.byref(npz), C.byref(nf), xid.ctypes.data, cap, C.byref(nx),
                                                 ep.ctypes.data, el.ctypes.data, uv.ctypes.data, cap, C.byref(ne)), "myslam_scn_ba_graph")
        return {"pose_ids": [int(v) for v in pid[:npz.value]], "n_free": nf.value, "point_ids": [int(v) for v in xid[:nx.value]],
                "edge_pose": ep[:ne.value].copy(), "edge_point": el[:ne.value].copy(), "edge_uv": uv[:ne.value].copy()}

    def scn_mappoint(self, mp: int):
        o, n = C.c_int(), C.c_int(); p = np.zeros(3); nr = np.zeros(3)
        self._check(self.lib.myslam_scn_mappoint(self.h, mp, C.byref(o), C.byref(n), p.ctypes.data, nr.ctypes.data), "myslam_scn_mappoint")
        return {"outlier": bool(o.value), "n_obs": n.value, "xyz": p, "normal": nr}

    def scn_run_ba(self, kf: int):
        self._check(self.lib.myslam_scn_run_ba(self.h, kf), "myslam_scn_run_ba")

    def scn_keyframe_pose(self, kf: int) -> np.ndarray:
        T = np.zeros(12)
        self._check(self.lib.myslam_scn_keyframe_pose(self.h, kf, T.ctypes.data), "myslam_scn_keyframe_pose")
        return T

    def keyframe_policy(self, T_ref_cw, T_cur_cw, num_inliers: int) -> int:
        a = np.ascontiguousarray(T_ref_cw, dtype=np.float64).reshape(12); b = np.ascontiguousarray(T_cur_cw, dtype=np.float64).reshape(12); f = C.c_int()
        self._check(self.lib.myslam_keyframe_policy(self.h, a.ctypes.data, b.ctypes.data, num_inliers, C.byref(f)), "myslam_keyframe_policy")
        return f.value

    def materialize(self):
        """device_keyframes: rebuild the host map objects from the device tables -> (keyframe ids in insertion order, on_device)."""
        ids = np.zeros(1 << 16, np.int64); n, on = C.c_int(), C.c_int()
        self._check(self.lib.myslam_materialize(self.h, ids.ctypes.data, len(ids), C.byref(n), C.byref(on)), "myslam_materialize")
        return [int(v) for v in ids[:n.value]], bool(on.value)

    def mappoint_ids(self, cap: int = 1 << 21):
        ids = np.zeros(cap, np.int64); n = C.c_int()
        self._check(self.lib.myslam_mappoint_ids(self.h, ids.ctypes.data, cap, C.byref(n)), "myslam_mappoint_ids")
        return [int(v) for v in ids[:min(n.value, cap)]]

    def context_handle(self) -> int:
        return self.lib.myslam_get_context(self.h)

    def stats(self) -> dict:
        st = Stats()
        self._check(self.lib.myslam_get_stats(self.h, C.byref(st)), "myslam_get_stats")
        return st.asdict()
