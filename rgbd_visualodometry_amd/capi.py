"""ctypes binding of the C-ABI in include/vo_hip.h.

The product library is ``rgbd_visualodometry_amd/csrc/libvo_hip.so`` (HIP, gfx950).  There is
no CPU fallback: ``load()`` raises if that library is missing.  ``load(path)`` binds any other
implementation of the same C-ABI (the test suite passes its CPU checker's path).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIP_LIB = os.path.join(HERE, "csrc", "libvo_hip.so")
SYNTH_LIB = os.path.join(HERE, "synth", "libvo_synth.so")


class VoParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("depth_scale", C.c_float), ("n_features", C.c_int32),
                ("scale_factor", C.c_float), ("n_levels", C.c_int32), ("fast_threshold", C.c_int32),
                ("edge_threshold", C.c_int32), ("max_frames", C.c_int32), ("map_capacity", C.c_int32),
                ("max_hypotheses", C.c_int32), ("max_track_batch", C.c_int32), ("stream_priority", C.c_int32), ("reserved", C.c_int32 * 6)]


class VoTrackParams(C.Structure):
    _fields_ = [("match_ratio", C.c_float), ("match_floor", C.c_float), ("n_hyp", C.c_int32),
                ("reproj_px", C.c_float), ("confidence", C.c_float), ("seed", C.c_uint64),
                ("huber_delta", C.c_double), ("chi2_cut", C.c_double), ("it_robust", C.c_int32),
                ("it_plain", C.c_int32), ("passes", C.c_int32), ("reserved", C.c_int32 * 5)]


class VoTrackResult(C.Structure):
    _fields_ = [("T_cw", C.c_double * 12), ("n_candidates", C.c_int32), ("n_matches", C.c_int32),
                ("n_ransac_inliers", C.c_int32), ("n_lm_inliers", C.c_int32), ("min_distance", C.c_int32),
                ("ransac_iters", C.c_int32), ("best_hypothesis", C.c_int32), ("lm_iters", C.c_int32),
                ("status", C.c_int32), ("reserved", C.c_int32 * 7)]


class VoBaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_free", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("poses", C.c_void_p), ("points", C.c_void_p), ("edge_pose", C.c_void_p), ("edge_point", C.c_void_p),
                ("edge_uv", C.c_void_p), ("huber_delta", C.c_double), ("chi2_th", C.c_double),
                ("it_robust", C.c_int32), ("it_plain", C.c_int32)]


class VoBaResidentResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("point_slots", C.c_void_p), ("points", C.c_void_p), ("culled_obs", C.c_void_p), ("cap_points", C.c_int32),
                ("cap_culled", C.c_int32), ("n_points", C.c_int32), ("n_fixed", C.c_int32), ("n_edges", C.c_int32), ("n_culled", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lm_iters", C.c_int32), ("n_pairs", C.c_int32)]


class VoKfCommitResult(C.Structure):
    _fields_ = [("n_matched", C.c_int32), ("n_new", C.c_int32), ("first_obs", C.c_int64), ("n_covisible", C.c_int32), ("n_tri_candidates", C.c_int32),
                ("triangulated_slot", C.c_int32), ("n_covisible_total", C.c_int32)]


class VoBaResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("edge_flags", C.c_void_p),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lm_iters", C.c_int32),
                ("reserved", C.c_int32 * 3)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4"), ("depth_raw", "<i4")])
MATCH_DTYPE = np.dtype([("map_index", "<i4"), ("kp_index", "<i4"), ("distance", "<i4"), ("flags", "<i4")])

# every symbol include/vo_hip.h declares
SYMBOLS = ["vo_ctx_create", "vo_ctx_destroy", "vo_strerror", "vo_backend_name", "vo_default_params",
           "vo_default_track_params", "vo_frame_upload", "vo_frames_preload", "vo_frame_bind_device", "vo_orb_detect_describe",
           "vo_orb_fetch", "vo_orb_level_size", "vo_orb_fetch_level", "vo_orb_fetch_blur_level", "vo_map_upsert", "vo_map_upsert_from_frame", "vo_map_set_active",
           "vo_match_active_map", "vo_matches_set", "vo_pnp_ransac", "vo_pose_refine_lm", "vo_track_frame", "vo_track_batch", "vo_track_batch_begin", "vo_track_batch_end", "vo_track_fetch_matches",
           "vo_local_ba", "vo_sync", "vo_profile_enable", "vo_profile_read",
           "vo_group_create", "vo_group_destroy", "vo_group_join", "vo_group_leave", "vo_group_set_gather", "vo_group_stats",
           "vo_set_hypothesis_shard", "vo_set_hypothesis_shard_stream", "vo_triangulate_batch", "vo_kf_set_pose", "vo_obs_append", "vo_obs_kill", "vo_local_ba_resident",
           "vo_local_ba_resident_cut", "vo_local_ba_resident_solve", "vo_local_ba_resident_merge", "vo_local_ba_resident_fetch", "vo_ba_resident_graph", "vo_ba_resident_window", "vo_ba_resident_set_slab_budget", "vo_trace_level", "vo_set_ba_shard", "vo_set_ba_shard_stream",
           "vo_keyframe_commit", "vo_kf_covisibility", "vo_map_set_active_covisible", "vo_local_ba_resident_merge_ledger", "vo_tables_fetch", "vo_scan_call_number"]


VO_E_OVERFLOW = -4        # include/vo_hip.h: vo_status

EXCHANGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int32), C.c_int)     # vo_exchange_fn: in-place element-wise sum over the ranks
STREAM_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_size_t, C.c_void_p)      # vo_stream_allreduce_fn: enqueue the sum on a HIP stream
EXCHANGE_F64_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)     # vo_exchange_f64_fn: in-place element-wise sum of doubles over the ranks (e-3)
STREAM_ALLREDUCE_F64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_void_p)      # vo_stream_allreduce_f64_fn


class VoError(RuntimeError):
    pass


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class VoLib:
    """A loaded implementation of the C-ABI (HIP product or CPU oracle)."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise VoError("C-ABI library not found: %s (build it: python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        L = self.lib
        for s in SYMBOLS:
            getattr(L, s)  # raises AttributeError if a declared symbol is missing
        L.vo_strerror.restype = C.c_char_p
        L.vo_backend_name.restype = C.c_char_p
        L.vo_ctx_destroy.restype = None
        for name in SYMBOLS:
            if name not in ("vo_strerror", "vo_backend_name", "vo_ctx_destroy", "vo_group_destroy"):
                getattr(L, name).restype = C.c_int
        L.vo_ctx_create.argtypes = [C.POINTER(VoParams), C.c_int, C.POINTER(C.c_void_p)]
        L.vo_ctx_destroy.argtypes = [C.c_void_p]
        L.vo_frame_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.vo_frames_preload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.vo_frame_bind_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.vo_orb_detect_describe.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vo_orb_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.vo_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_orb_fetch_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vo_orb_fetch_blur_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vo_map_upsert.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        L.vo_map_upsert_from_frame.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int]
        L.vo_map_set_active.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vo_match_active_map.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_matches_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vo_pnp_ransac.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_pose_refine_lm.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_track_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(VoTrackParams), C.POINTER(VoTrackResult),
                                     C.c_void_p, C.c_int]
        L.vo_track_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(VoTrackParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vo_track_batch_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(VoTrackParams), C.c_void_p, C.c_int]
        L.vo_track_batch_end.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_track_fetch_matches.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.vo_local_ba.argtypes = [C.c_void_p, C.POINTER(VoBaProblem), C.POINTER(VoBaResult)]
        L.vo_sync.argtypes = [C.c_void_p]
        L.vo_group_destroy.restype = None
        L.vo_kf_set_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vo_obs_append.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.vo_obs_kill.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vo_local_ba_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(VoBaResidentResult)]
        L.vo_local_ba_resident_cut.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.vo_local_ba_resident_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(VoBaResidentResult)]
        L.vo_local_ba_resident_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_local_ba_resident_fetch.argtypes = [C.c_void_p, C.POINTER(VoBaResidentResult)]
        L.vo_ba_resident_graph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_int,
                                           C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vo_keyframe_commit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(VoKfCommitResult)]
        L.vo_kf_covisibility.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        L.vo_map_set_active_covisible.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int32, C.POINTER(C.c_int32)]
        L.vo_local_ba_resident_merge_ledger.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_int]
        L.vo_tables_fetch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        L.vo_scan_call_number.argtypes = [C.c_longlong]; L.vo_scan_call_number.restype = C.c_longlong
        L.vo_triangulate_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vo_set_hypothesis_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, EXCHANGE_FN, C.c_void_p]
        L.vo_set_hypothesis_shard_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, STREAM_ALLREDUCE_FN, C.c_void_p]
        L.vo_group_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.vo_group_destroy.argtypes = [C.c_void_p]
        L.vo_group_join.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_group_leave.argtypes = [C.c_void_p, C.c_void_p]
        L.vo_group_set_gather.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vo_group_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.vo_profile_enable.argtypes = [C.c_void_p, C.c_int]
        L.vo_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]

    @property
    def backend(self) -> str:
        return self.lib.vo_backend_name().decode()

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            raise VoError("%s failed: %s (%d)" % (what or "vo call", self.lib.vo_strerror(rc).decode(), rc))

    def default_params(self, **kw) -> VoParams:
        p = VoParams()
        self.check(self.lib.vo_default_params(C.byref(p)))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def default_track_params(self, **kw) -> VoTrackParams:
        t = VoTrackParams()
        self.check(self.lib.vo_default_track_params(C.byref(t)))
        for k, v in kw.items():
            setattr(t, k, v)
        return t

    def context(self, params: VoParams, device: int = 0) -> "VoContext":
        return VoContext(self, params, device)


class VoGroup:
    """Stream group (include/vo_hip.h): the vo_track_batch calls of its member contexts share launch chains."""

    def __init__(self, lib: VoLib, device: int = 0, max_lanes: int = 64):
        self.L = lib
        self.h = C.c_void_p()
        lib.check(lib.lib.vo_group_create(device, max_lanes, C.byref(self.h)), "vo_group_create")

    def join(self, ctx: "VoContext"):
        self.L.check(self.L.lib.vo_group_join(self.h, ctx.h), "vo_group_join")

    def leave(self, ctx: "VoContext"):
        self.L.check(self.L.lib.vo_group_leave(self.h, ctx.h), "vo_group_leave")

    def set_gather(self, min_requests: int, timeout_us: int):
        self.L.check(self.L.lib.vo_group_set_gather(self.h, min_requests, timeout_us), "vo_group_set_gather")

    def stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.L.check(self.L.lib.vo_group_stats(self.h, C.byref(a), C.byref(b), C.byref(c)), "vo_group_stats")
        return {"chains": a.value, "lanes": b.value, "requests": c.value}

    def close(self):
        if self.h:
            self.L.lib.vo_group_destroy(self.h)
            self.h = C.c_void_p()


class VoContext:
    def __init__(self, lib: VoLib, params: VoParams, device: int = 0):
        self.L = lib
        self.params = params
        self.h = C.c_void_p()
        lib.check(lib.lib.vo_ctx_create(C.byref(params), device, C.byref(self.h)), "vo_ctx_create")
        self._keep = {}

    def close(self):
        if self.h:
            self.L.lib.vo_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # frames ---------------------------------------------------------------------------
    def upload(self, slot: int, bgr: np.ndarray, depth: np.ndarray):
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        self.L.check(self.L.lib.vo_frame_upload(self.h, slot, _ptr(bgr), bgr.strides[0], _ptr(depth), depth.strides[0]), "vo_frame_upload")

    def preload_ptrs(self, slot0: int, bgr_ptrs, bgr_stride: int, depth_ptrs, depth_stride: int):
        n = len(bgr_ptrs)
        b = (C.c_void_p * n)(*bgr_ptrs); d = (C.c_void_p * n)(*depth_ptrs)
        self.L.check(self.L.lib.vo_frames_preload(self.h, slot0, n, C.cast(b, C.c_void_p), bgr_stride, C.cast(d, C.c_void_p), depth_stride), "vo_frames_preload")

    def upload_ptr(self, slot: int, bgr_ptr: int, bgr_stride: int, depth_ptr: int, depth_stride: int):
        self.L.check(self.L.lib.vo_frame_upload(self.h, slot, C.c_void_p(bgr_ptr), bgr_stride, C.c_void_p(depth_ptr), depth_stride), "vo_frame_upload")

    def bind_device(self, slot: int, bgr_ptr: int, bgr_stride: int, depth_ptr: int, depth_stride: int):
        self.L.check(self.L.lib.vo_frame_bind_device(self.h, slot, C.c_void_p(bgr_ptr), bgr_stride, C.c_void_p(depth_ptr), depth_stride), "vo_frame_bind_device")

    # ORB ------------------------------------------------------------------------------
    def orb(self, slot0: int = 0, nslots: int = 1):
        self.L.check(self.L.lib.vo_orb_detect_describe(self.h, slot0, nslots), "vo_orb_detect_describe")

    def orb_fetch(self, slot: int = 0, cap: Optional[int] = None):
        cap = cap or 4 * self.params.n_features + 64
        kps = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int()
        self.L.check(self.L.lib.vo_orb_fetch(self.h, slot, _ptr(kps), _ptr(desc), cap, C.byref(n)), "vo_orb_fetch")
        k = min(n.value, cap)
        return kps[:k].copy(), desc[:k].copy()

    def level_size(self, level: int):
        w, h, q = C.c_int(), C.c_int(), C.c_int()
        self.L.check(self.L.lib.vo_orb_level_size(self.h, level, C.byref(w), C.byref(h), C.byref(q)))
        return w.value, h.value, q.value

    def fetch_level(self, slot: int, level: int) -> np.ndarray:
        w, h, _ = self.level_size(level)
        out = np.zeros((h, w), dtype=np.uint8)
        self.L.check(self.L.lib.vo_orb_fetch_level(self.h, slot, level, _ptr(out)), "vo_orb_fetch_level")
        return out

    def fetch_blur_level(self, slot: int, level: int) -> np.ndarray:
        w, h, _ = self.level_size(level)
        out = np.zeros((h, w), dtype=np.uint8)
        self.L.check(self.L.lib.vo_orb_fetch_blur_level(self.h, slot, level, _ptr(out)), "vo_orb_fetch_blur_level")
        return out

    # map ------------------------------------------------------------------------------
    def map_upsert(self, idx, xyz=None, normal=None, desc=None, flags=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        xyz = None if xyz is None else np.ascontiguousarray(xyz, dtype=np.float64)
        normal = None if normal is None else np.ascontiguousarray(normal, dtype=np.float64)
        desc = None if desc is None else np.ascontiguousarray(desc, dtype=np.uint8)
        flags = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        self.L.check(self.L.lib.vo_map_upsert(self.h, _ptr(idx), _ptr(xyz), _ptr(normal), _ptr(desc), _ptr(flags), len(idx)), "vo_map_upsert")

    def map_upsert_from_frame(self, slot, kp_index, idx, xyz=None, normal=None, flags=None):
        """Map points whose descriptor is keypoint kp_index[i] of the frame in `slot` (copied on the device)."""
        idx = np.ascontiguousarray(idx, dtype=np.int32); kp = np.ascontiguousarray(kp_index, dtype=np.int32)
        xyz = None if xyz is None else np.ascontiguousarray(xyz, dtype=np.float64)
        normal = None if normal is None else np.ascontiguousarray(normal, dtype=np.float64)
        flags = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
        self.L.check(self.L.lib.vo_map_upsert_from_frame(self.h, slot, _ptr(kp), _ptr(idx), _ptr(xyz), _ptr(normal), _ptr(flags), len(idx)), "vo_map_upsert_from_frame")

    def map_set_active(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self.L.check(self.L.lib.vo_map_set_active(self.h, _ptr(idx), len(idx)), "vo_map_set_active")

    # stages ---------------------------------------------------------------------------
    def match(self, slot, T_cw, ratio=2.0, floor_dist=30.0, cap=None):
        cap = cap or max(1024, self.params.map_capacity)
        T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(12)
        out = np.zeros(cap, dtype=MATCH_DTYPE)
        n, nc, md = C.c_int(), C.c_int(), C.c_int()
        self.L.check(self.L.lib.vo_match_active_map(self.h, slot, _ptr(T), ratio, floor_dist, _ptr(out), cap, C.byref(n), C.byref(nc), C.byref(md)), "vo_match_active_map")
        return out[:min(n.value, cap)].copy(), nc.value, md.value

    def matches_set(self, xyz, uv):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        self.L.check(self.L.lib.vo_matches_set(self.h, _ptr(xyz), _ptr(uv), len(xyz)), "vo_matches_set")

    def pnp_ransac(self, T_cw, n_hyp=100, reproj_px=4.0, confidence=0.99, seed=1, cap=65536):
        T = np.array(T_cw, dtype=np.float64).reshape(12).copy()
        inl = np.zeros(cap, dtype=np.int32)
        counts = np.zeros(n_hyp, dtype=np.int32)
        n, it, best = C.c_int(), C.c_int(), C.c_int()
        self.L.check(self.L.lib.vo_pnp_ransac(self.h, n_hyp, reproj_px, confidence, seed, _ptr(T), _ptr(inl), cap, C.byref(n), _ptr(counts), C.byref(it), C.byref(best)), "vo_pnp_ransac")
        return T, inl[:min(n.value, cap)].copy(), counts, it.value, best.value

    def pose_lm(self, T_cw, huber_delta=7.815 ** 0.5, chi2_cut=1.0, it_robust=10, it_plain=10, cap=65536):
        T = np.array(T_cw, dtype=np.float64).reshape(12).copy()
        mask = np.zeros(cap, dtype=np.uint8)
        n, it = C.c_int(), C.c_int()
        self.L.check(self.L.lib.vo_pose_refine_lm(self.h, _ptr(T), huber_delta, chi2_cut, it_robust, it_plain, _ptr(mask), cap, C.byref(n), C.byref(it)), "vo_pose_refine_lm")
        return T, mask[:min(n.value, cap)].copy(), it.value

    def track(self, slot, T_prior, tp: VoTrackParams, cap=None):
        cap = cap or max(1024, self.params.map_capacity)
        T = np.ascontiguousarray(T_prior, dtype=np.float64).reshape(12)
        res = VoTrackResult()
        m = np.zeros(cap, dtype=MATCH_DTYPE)
        self.L.check(self.L.lib.vo_track_frame(self.h, slot, _ptr(T), C.byref(tp), C.byref(res), _ptr(m), cap), "vo_track_frame")
        return res, m[:min(res.n_matches, cap)].copy()

    def track_batch(self, slots, T_prior, tp: VoTrackParams, seeds, cap=4096):
        n = len(slots)
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        T = np.ascontiguousarray(T_prior, dtype=np.float64).reshape(12)
        res = (VoTrackResult * n)()
        m = np.zeros((n, cap), dtype=MATCH_DTYPE)
        self.L.check(self.L.lib.vo_track_batch(self.h, n, _ptr(sl), _ptr(T), C.byref(tp), _ptr(sd), C.cast(res, C.c_void_p), _ptr(m), cap), "vo_track_batch")
        return [res[i] for i in range(n)], [m[i, :min(res[i].n_matches, cap)].copy() for i in range(n)]

    def track_batch_deferred(self, slots, T_prior, tp: VoTrackParams, seeds, cap=4096):
        """vo_track_batch without the match copy, then vo_track_fetch_matches per lane (what the host layer does on keyframes)."""
        n = len(slots)
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        T = np.ascontiguousarray(T_prior, dtype=np.float64).reshape(12)
        res = (VoTrackResult * n)()
        self.L.check(self.L.lib.vo_track_batch(self.h, n, _ptr(sl), _ptr(T), C.byref(tp), _ptr(sd), C.cast(res, C.c_void_p), None, cap), "vo_track_batch")
        out = []
        for lane in range(n):
            m = np.zeros(cap, dtype=MATCH_DTYPE)
            got = C.c_int()
            self.L.check(self.L.lib.vo_track_fetch_matches(self.h, lane, _ptr(m), cap, C.byref(got)), "vo_track_fetch_matches")
            out.append(m[:got.value].copy())
        return [res[i] for i in range(n)], out

    def track_batch_begin(self, slots, T_prior, tp: VoTrackParams, seeds, cap=4096):
        """vo_track_batch_begin: enqueue the launch chain and return; the arrays may be dropped right away (the library copies them)."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        T = np.ascontiguousarray(T_prior, dtype=np.float64).reshape(12)
        self.L.check(self.L.lib.vo_track_batch_begin(self.h, len(sl), _ptr(sl), _ptr(T), C.byref(tp), _ptr(sd), cap), "vo_track_batch_begin")
        return len(sl)

    def track_batch_end(self, n, cap=4096):
        """vo_track_batch_end + one vo_track_fetch_matches per lane."""
        res = (VoTrackResult * n)()
        self.L.check(self.L.lib.vo_track_batch_end(self.h, C.cast(res, C.c_void_p)), "vo_track_batch_end")
        out = []
        for lane in range(n):
            m = np.zeros(cap, dtype=MATCH_DTYPE)
            got = C.c_int()
            self.L.check(self.L.lib.vo_track_fetch_matches(self.h, lane, _ptr(m), cap, C.byref(got)), "vo_track_fetch_matches")
            out.append(m[:got.value].copy())
        return [res[i] for i in range(n)], out

    def local_ba(self, poses, n_free, points, edge_pose, edge_point, edge_uv, huber_delta=7.815 ** 0.5, chi2_th=1.0,
                 it_robust=10, it_plain=10):
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 12)
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        ep = np.ascontiguousarray(edge_pose, dtype=np.int32)
        el = np.ascontiguousarray(edge_point, dtype=np.int32)
        uv = np.ascontiguousarray(edge_uv, dtype=np.float32).reshape(-1, 2)
        prob = VoBaProblem(len(poses), n_free, len(points), len(ep), _ptr(poses).value, _ptr(points).value, _ptr(ep).value,
                           _ptr(el).value, _ptr(uv).value, huber_delta, chi2_th, it_robust, it_plain)
        po = np.zeros((max(n_free, 1), 12)); pt = np.zeros((max(len(points), 1), 3)); fl = np.zeros(max(len(ep), 1), dtype=np.uint8)
        res = VoBaResult(_ptr(po).value, _ptr(pt).value, _ptr(fl).value)
        self.L.check(self.L.lib.vo_local_ba(self.h, C.byref(prob), C.byref(res)), "vo_local_ba")
        return po[:n_free], pt[:len(points)], fl[:len(ep)], res

    def set_hypothesis_shard(self, rank: int, world: int, all_reduce_sum):
        """RANSAC hypotheses of this context's frames are scored h % world == rank; ``all_reduce_sum(np.int32 array)`` sums in place over the ranks."""
        def _cb(user, ptr, n):
            all_reduce_sum(np.ctypeslib.as_array(ptr, shape=(n,)))
        self._keep["shard_cb"] = EXCHANGE_FN(_cb) if all_reduce_sum is not None else EXCHANGE_FN(0)
        self.L.check(self.L.lib.vo_set_hypothesis_shard(self.h, rank, world, self._keep["shard_cb"], None), "vo_set_hypothesis_shard")

    def set_hypothesis_shard_stream(self, rank: int, world: int, enqueue_allreduce):
        """On-stream form: ``enqueue_allreduce(device_ptr: int, n: int, hip_stream: int) -> int`` enqueues an in-place int32 SUM over the ranks
        on the given HIP stream (RCCL: ncclAllReduce; through torch: shard.Group.stream_allreduce_i32) and returns 0."""
        def _cb(comm, ptr, n, stream):
            return int(enqueue_allreduce(C.cast(ptr, C.c_void_p).value, int(n), int(stream or 0)))
        self._keep["shard_scb"] = STREAM_ALLREDUCE_FN(_cb) if enqueue_allreduce is not None else STREAM_ALLREDUCE_FN(0)
        self.L.check(self.L.lib.vo_set_hypothesis_shard_stream(self.h, rank, world, self._keep["shard_scb"], None), "vo_set_hypothesis_shard_stream")

    def set_ba_shard(self, rank: int, world: int, all_reduce_sum):
        """e-3: vo_local_ba of this context works on the edges of the points k % world == rank; ``all_reduce_sum(np.float64 array)`` sums in place over
        the ranks (three calls per LM step + one for the result).  world <= 1: off."""
        def _cb(user, ptr, n):
            all_reduce_sum(np.ctypeslib.as_array(ptr, shape=(n,)))
        self._keep["ba_shard_cb"] = EXCHANGE_F64_FN(_cb) if all_reduce_sum is not None else EXCHANGE_F64_FN(0)
        self.L.check(self.L.lib.vo_set_ba_shard(self.h, rank, world, self._keep["ba_shard_cb"], None), "vo_set_ba_shard")

    def set_ba_shard_stream(self, rank: int, world: int, enqueue_allreduce):
        """On-stream form: ``enqueue_allreduce(device_ptr: int, n: int, hip_stream: int) -> int`` enqueues an in-place f64 SUM over the ranks on the given HIP
        stream (RCCL: ncclAllReduce with ncclDouble; through torch: shard.Group.stream_allreduce_f64) and returns 0."""
        def _cb(comm, ptr, n, stream):
            return int(enqueue_allreduce(C.cast(ptr, C.c_void_p).value, int(n), int(stream or 0)))
        self._keep["ba_shard_scb"] = STREAM_ALLREDUCE_F64_FN(_cb) if enqueue_allreduce is not None else STREAM_ALLREDUCE_F64_FN(0)
        self.L.check(self.L.lib.vo_set_ba_shard_stream(self.h, rank, world, self._keep["ba_shard_scb"], None), "vo_set_ba_shard_stream")

    def triangulate_batch(self, view_start, T_cw, xy):
        """Batched N-view triangulation -> (xyz [n, 3], ok [n])."""
        vs = np.ascontiguousarray(view_start, dtype=np.int32); T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(-1, 12)
        z = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        n = len(vs) - 1
        xyz = np.zeros((max(n, 1), 3)); ok = np.zeros(max(n, 1), dtype=np.uint8)
        self.L.check(self.L.lib.vo_triangulate_batch(self.h, n, _ptr(vs), _ptr(T), _ptr(z), _ptr(xyz), _ptr(ok)), "vo_triangulate_batch")
        return xyz[:n], ok[:n].astype(bool)

    # device-resident keyframe bookkeeping -------------------------------------------------------
    def kf_set_pose(self, kf, T_cw):
        k = np.ascontiguousarray(kf, dtype=np.int32); T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(-1, 12)
        self.L.check(self.L.lib.vo_kf_set_pose(self.h, _ptr(k), _ptr(T), len(k)), "vo_kf_set_pose")

    def obs_append(self, kf, map_idx, uv) -> int:
        k = np.ascontiguousarray(kf, dtype=np.int32); m = np.ascontiguousarray(map_idx, dtype=np.int32); z = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        first = C.c_int64()
        self.L.check(self.L.lib.vo_obs_append(self.h, _ptr(k), _ptr(m), _ptr(z), len(k), C.byref(first)), "vo_obs_append")
        return first.value

    def obs_kill(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int64)
        self.L.check(self.L.lib.vo_obs_kill(self.h, _ptr(a), len(a)), "vo_obs_kill")

    def resident_set_slab_budget(self, nbytes):
        """Scratch budget of the bound-sized resident cut; a cut that would exceed it waits for the graph's sizes and carves exactly."""
        self.L.check(self.L.lib.vo_ba_resident_set_slab_budget(self.h, C.c_int64(int(nbytes))), "vo_ba_resident_set_slab_budget")

    def resident_window(self):
        """(observations, map slots) the last resident graph cut of this context visited."""
        a, b = C.c_int64(), C.c_int64()
        self.L.check(self.L.lib.vo_ba_resident_window(self.h, C.byref(a), C.byref(b)), "vo_ba_resident_window")
        return a.value, b.value

    def resident_graph(self, tables: "VoContext", free_kf, cap=1 << 18):
        f = np.ascontiguousarray(free_kf, dtype=np.int32)
        npz, nx, ne = C.c_int32(), C.c_int32(), C.c_int32()
        pk = np.zeros(4096, np.int32); ps = np.zeros(cap, np.int32); ep = np.zeros(cap, np.int32); el = np.zeros(cap, np.int32)
        uv = np.zeros((cap, 2), np.float32); eo = np.zeros(cap, np.int64)
        self.L.check(self.L.lib.vo_ba_resident_graph(self.h, tables.h, _ptr(f), len(f), C.byref(npz), _ptr(pk), 4096, C.byref(nx), _ptr(ps), cap, C.byref(ne),
                                                     _ptr(ep), _ptr(el), _ptr(uv), _ptr(eo), cap), "vo_ba_resident_graph")
        return {"pose_kf": pk[:npz.value].copy(), "point_slots": ps[:nx.value].copy(), "edge_pose": ep[:ne.value].copy(), "edge_point": el[:ne.value].copy(),
                "edge_uv": uv[:ne.value].copy(), "edge_obs": eo[:ne.value].copy()}

    def local_ba_resident(self, tables: "VoContext", free_kf, huber_delta=7.815 ** 0.5, chi2_th=1.0, it_robust=10, it_plain=10, cap_points=1 << 18, cap_culled=1 << 16):
        f = np.ascontiguousarray(free_kf, dtype=np.int32)
        po = np.zeros((max(len(f), 1), 12)); sl = np.zeros(cap_points, np.int32); pt = np.zeros((cap_points, 3)); cu = np.zeros(cap_culled, np.int64)
        r = VoBaResidentResult(_ptr(po).value, _ptr(sl).value, _ptr(pt).value, _ptr(cu).value, cap_points, cap_culled)
        self.L.check(self.L.lib.vo_local_ba_resident(self.h, tables.h, _ptr(f), len(f), huber_delta, chi2_th, it_robust, it_plain, C.byref(r)), "vo_local_ba_resident")
        return po[:len(f)], sl[:r.n_points].copy(), pt[:r.n_points].copy(), cu[:r.n_culled].copy(), r

    def local_ba_resident_merged(self, tables: "VoContext", free_kf, huber_delta=7.815 ** 0.5, chi2_th=1.0, it_robust=10, it_plain=10, cap_points=1 << 18, cap_culled=1 << 16):
        """The back-end's sequence: cut, solve with the result left on the device, merge into `tables` on the device, fetch the host's copy."""
        f = np.ascontiguousarray(free_kf, dtype=np.int32)
        nx, nfx, ne = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.check(self.L.lib.vo_local_ba_resident_cut(self.h, tables.h, _ptr(f), len(f), huber_delta, chi2_th, C.byref(nx), C.byref(nfx), C.byref(ne)), "vo_local_ba_resident_cut")
        cu = np.zeros(cap_culled, np.int64)
        r = VoBaResidentResult(None, None, None, _ptr(cu).value, 0, cap_culled)
        self.L.check(self.L.lib.vo_local_ba_resident_solve(self.h, it_robust, it_plain, C.byref(r)), "vo_local_ba_resident_solve")
        n_culled = r.n_culled
        self.L.check(self.L.lib.vo_local_ba_resident_merge(self.h, tables.h), "vo_local_ba_resident_merge")
        po = np.zeros((max(len(f), 1), 12)); sl = np.zeros(cap_points, np.int32); pt = np.zeros((cap_points, 3))
        r2 = VoBaResidentResult(_ptr(po).value, _ptr(sl).value, _ptr(pt).value, _ptr(cu).value, cap_points, cap_culled)
        self.L.check(self.L.lib.vo_local_ba_resident_fetch(self.h, C.byref(r2)), "vo_local_ba_resident_fetch")
        return po[:len(f)], sl[:r2.n_points].copy(), pt[:r2.n_points].copy(), cu[:n_culled].copy(), r2

    # keyframe bookkeeping on the device tables (second half) -------------------------------------------
    def keyframe_commit(self, lane: int, frame_slot: int, kf: int, T_cw, first_new_slot: int, cap_covis: int = 4096):
        """-> (VoKfCommitResult, {partner keyframe: weight})"""
        T = np.ascontiguousarray(T_cw, dtype=np.float64).reshape(12)
        ck = np.zeros(cap_covis, np.int32); cw = np.zeros(cap_covis, np.int32); r = VoKfCommitResult()
        self.L.check(self.L.lib.vo_keyframe_commit(self.h, lane, frame_slot, kf, _ptr(T), first_new_slot, _ptr(ck), _ptr(cw), cap_covis, C.byref(r)), "vo_keyframe_commit")
        return r, {int(ck[i]): int(cw[i]) for i in range(r.n_covisible)}

    def kf_covisibility(self, kf: int, cap: int = 4096):
        ck = np.zeros(cap, np.int32); cw = np.zeros(cap, np.int32); n = C.c_int32()
        self.L.check(self.L.lib.vo_kf_covisibility(self.h, kf, _ptr(ck), _ptr(cw), cap, C.byref(n)), "vo_kf_covisibility")
        return {int(ck[i]): int(cw[i]) for i in range(n.value)}

    def map_set_active_covisible(self, kfs, n_map_points: int, min_points: int = 100) -> int:
        k = np.ascontiguousarray(kfs, dtype=np.int32); n = C.c_int32()
        self.L.check(self.L.lib.vo_map_set_active_covisible(self.h, _ptr(k), len(k), min_points, n_map_points, C.byref(n)), "vo_map_set_active_covisible")
        return n.value

    def merge_ledger(self, tables: "VoContext", n_free: int, cap_pairs: int = 1 << 16):
        """vo_local_ba_resident_merge_ledger -> (sorted list of (kf_a, kf_b) decrements, free poses)"""
        pa = np.zeros(max(cap_pairs, 1), np.int32); pb = np.zeros(max(cap_pairs, 1), np.int32); n = C.c_int32(); po = np.zeros((max(n_free, 1), 12))
        rc = self.L.lib.vo_local_ba_resident_merge_ledger(self.h, tables.h, _ptr(pa), _ptr(pb), cap_pairs, C.byref(n), _ptr(po), n_free)
        if rc == VO_E_OVERFLOW and n.value > cap_pairs:        # the call says how many pairs it needs; the repeated call completes the merge (include/vo_hip.h)
            cap_pairs = n.value
            pa = np.zeros(cap_pairs, np.int32); pb = np.zeros(cap_pairs, np.int32)
            rc = self.L.lib.vo_local_ba_resident_merge_ledger(self.h, tables.h, _ptr(pa), _ptr(pb), cap_pairs, C.byref(n), _ptr(po), n_free)
        self.L.check(rc, "vo_local_ba_resident_merge_ledger")
        return sorted((int(pa[i]), int(pb[i])) for i in range(n.value)), po[:n_free]

    def tables(self, n_map: int):
        """The device tables as numpy arrays (vo_tables_fetch): observations, map columns for slots [0, n_map), active list."""
        no = C.c_int64(); na = C.c_int32()
        self.L.check(self.L.lib.vo_tables_fetch(self.h, 0, 0, None, None, None, None, C.byref(no), 0, 0, None, None, None, None, None, 0, C.byref(na)), "vo_tables_fetch")
        n = no.value
        kf = np.zeros(n, np.int32); mp = np.zeros(n, np.int32); uv = np.zeros((n, 2), np.float32); al = np.zeros(n, np.uint8)
        xyz = np.zeros((n_map, 3)); nr = np.zeros((n_map, 3)); de = np.zeros((n_map, 32), np.uint8); fl = np.zeros(n_map, np.uint8); ac = np.zeros(max(na.value, 1), np.int32)
        self.L.check(self.L.lib.vo_tables_fetch(self.h, 0, n, _ptr(kf), _ptr(mp), _ptr(uv), _ptr(al), C.byref(no), 0, n_map, _ptr(xyz), _ptr(nr), _ptr(de), _ptr(fl),
                                                _ptr(ac), na.value, C.byref(na)), "vo_tables_fetch")
        return {"obs_kf": kf, "obs_mp": mp, "obs_uv": uv, "obs_alive": al, "xyz": xyz, "normal": nr, "desc": de, "flags": fl, "active": ac[:na.value].copy()}

    def sync(self):
        self.L.check(self.L.lib.vo_sync(self.h), "vo_sync")

    def profile_enable(self, on=True):
        self.L.check(self.L.lib.vo_profile_enable(self.h, int(on)))

    def profile_read(self, cap=64):
        names = (C.c_char * 48 * cap)()
        ms = np.zeros(cap); calls = np.zeros(cap, dtype=np.int64)
        n = C.c_int()
        self.L.check(self.L.lib.vo_profile_read(self.h, C.cast(names, C.c_void_p), _ptr(ms), _ptr(calls), cap, C.byref(n)))
        return {names[i].value.decode(): (float(ms[i]), int(calls[i])) for i in range(n.value)}


_cache = {}


def load(path: Optional[str] = None) -> VoLib:
    """Load the HIP product library (default) or an explicit implementation of the C-ABI."""
    path = path or HIP_LIB
    if path not in _cache:
        _cache[path] = VoLib(path)
    return _cache[path]


# --------------------------------------------------------------------------------------- #
# synthetic stream generator binding
# --------------------------------------------------------------------------------------- #
class SynthParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("depth_scale", C.c_float), ("seed", C.c_uint64), ("noise_sigma", C.c_float),
                ("depth_noise_rel", C.c_float), ("invalid_frac", C.c_float), ("supersample", C.c_int32),
                ("fps", C.c_double), ("speed", C.c_double)]


class Synth:
    def __init__(self, path: str = SYNTH_LIB):
        if not os.path.exists(path):
            raise VoError("synthetic generator library not found: %s" % path)
        self.lib = C.CDLL(path)
        self.lib.synth_render_range.argtypes = [C.POINTER(SynthParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]

    def params(self, **kw) -> SynthParams:
        p = SynthParams()
        self.lib.synth_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def render(self, p: SynthParams, i0: int, n: int, threads: Optional[int] = None):
        threads = threads or min(16, os.cpu_count() or 1)
        bgr = np.empty((n, p.height, p.width, 3), dtype=np.uint8)
        depth = np.empty((n, p.height, p.width), dtype=np.uint16)
        T = np.empty((n, 12)); ts = np.empty(n)
        rc = self.lib.synth_render_range(C.byref(p), i0, n, _ptr(bgr), _ptr(depth), _ptr(T), _ptr(ts), threads)
        if rc:
            raise VoError("synth_render_range failed")
        return bgr, depth, T, ts


def pose12_to_tum(T_wc: np.ndarray):
    """12-double pose (R row-major, t) -> (tx,ty,tz,qx,qy,qz,qw)."""
    R = np.asarray(T_wc[:9]).reshape(3, 3)
    t = np.asarray(T_wc[9:12])
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    return [float(t[0]), float(t[1]), float(t[2])] + [float(v) for v in q]
