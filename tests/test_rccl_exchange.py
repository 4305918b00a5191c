"""The native exchange of the sharded paths (rgbd_visualodometry_amd/host/src/rccl_exchange.cpp): librccl.so is resolved with dlopen at run time and the
two functions handed to vo_set_hypothesis_shard_stream / vo_set_ba_shard_stream call ncclAllReduce directly -- no Python in the exchange.
Without N > 1 GPUs only the plumbing can be tested: symbol resolution, the file rendezvous of the RCCL id, and the error paths."""
import ctypes as C
import os
import threading

import pytest

from rgbd_visualodometry_amd import system

NAMES = ["myslam_rccl_load", "myslam_rccl_unique_id", "myslam_rccl_comm_create", "myslam_rccl_comm_destroy", "myslam_rccl_id_via_file",
         "myslam_rccl_allreduce_i32", "myslam_rccl_allreduce_f64", "myslam_rccl_last_error"]


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(system.HOST_LIB)
    for n in NAMES:
        getattr(L, n)
    L.myslam_rccl_last_error.restype = C.c_char_p
    L.myslam_rccl_load.argtypes = [C.c_char_p]
    L.myslam_rccl_id_via_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p]
    L.myslam_rccl_allreduce_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.myslam_rccl_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def test_the_host_library_exports_the_exchange_and_its_types_match_the_abi(lib):
    """The two all-reduce entry points are what include/vo_hip.h's vo_stream_allreduce_fn / vo_stream_allreduce_f64_fn describe: (comm, device pointer, n, stream) -> int."""
    hdr = open(os.path.join(os.path.dirname(system.HOST_LIB), "myslam", "rccl_exchange.h")).read()
    abi = open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(system.HOST_LIB))), "include", "vo_hip.h")).read()
    assert "int myslam_rccl_allreduce_i32(void* comm, int32_t* device_counts, size_t n, void* hip_stream);" in hdr
    assert "typedef int (*vo_stream_allreduce_fn)(void* comm, int32_t* device_counts, size_t n, void* hip_stream);" in abi
    assert "int myslam_rccl_allreduce_f64(void* comm, double* device_data, size_t n, void* hip_stream);" in hdr
    assert "typedef int (*vo_stream_allreduce_f64_fn)(void* comm, double* device_data, size_t n, void* hip_stream);" in abi
    # before anything is loaded (or without a communicator) the exchange fails with a message instead of crashing
    assert lib.myslam_rccl_allreduce_i32(None, None, 4, None) != 0 and b"communicator" in lib.myslam_rccl_last_error()
    assert lib.myslam_rccl_allreduce_f64(None, None, 4, None) != 0


def test_librccl_resolves_where_it_is_installed(lib):
    assert lib.myslam_rccl_load(b"/nonexistent/librccl.so") in (0, -1)      # (falls back to the default names)
    if not os.path.exists("/opt/rocm/lib/librccl.so"):
        pytest.skip("RCCL is not installed here")
    assert lib.myslam_rccl_load(None) == 0, lib.myslam_rccl_last_error()
    assert lib.myslam_rccl_load(None) == 0                                  # idempotent
    assert lib.myslam_rccl_allreduce_i32(None, None, 4, None) != 0         # loaded, but no communicator


def test_the_id_file_rendezvous_hands_rank_zeros_id_to_the_other_ranks(lib, tmp_path):
    if lib.myslam_rccl_load(None) != 0:
        pytest.skip("RCCL is not installed here")
    path = str(tmp_path / "rccl.id").encode()
    ids = [C.create_string_buffer(128) for _ in range(3)]
    rcs = [None] * 3

    def rank(r):
        rcs[r] = lib.myslam_rccl_id_via_file(path, r, 20, ids[r])
    th = [threading.Thread(target=rank, args=(r,)) for r in (1, 2)]
    [t.start() for t in th]
    rank(0)
    [t.join() for t in th]
    if rcs[0] != 0:
        pytest.skip("ncclGetUniqueId needs more than this box has: %s" % lib.myslam_rccl_last_error().decode())
    assert rcs == [0, 0, 0] and ids[0].raw == ids[1].raw == ids[2].raw and any(ids[0].raw)
    assert lib.myslam_rccl_id_via_file(str(tmp_path / "missing.id").encode(), 1, 0, ids[1]) != 0      # a rank that never gets an id gives up
