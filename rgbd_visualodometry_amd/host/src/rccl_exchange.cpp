// rccl_exchange.cpp -- see myslam/rccl_exchange.h.  RCCL's entry points are resolved at run time (dlopen / dlsym): a one-GPU run never loads the library,
// and the host layer builds where RCCL is not installed.
#include "myslam/rccl_exchange.h"

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

namespace {
// the part of rccl.h this file needs (ROCm 7.2: /opt/rocm/include/rccl/rccl.h:40-43,187,220,448-467,611)
struct NcclUniqueId { char internal[MYSLAM_RCCL_ID_BYTES]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclInt32 = 2, kNcclFloat64 = 8 };
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, void*);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char* (*GetErrorStringFn)(int);

std::mutex g_mu;
void* g_lib = nullptr;
GetUniqueIdFn p_get_id = nullptr; CommInitRankFn p_init = nullptr; AllReduceFn p_allreduce = nullptr; CommDestroyFn p_destroy = nullptr; GetErrorStringFn p_errstr = nullptr;
thread_local std::string t_err;

int fail(const std::string& what, int code = 0) {
    t_err = what;
    if (code && p_errstr) { t_err += ": "; t_err += p_errstr(code); }
    return -1;
}
}  // namespace

extern "C" {

const char* myslam_rccl_last_error(void) { return t_err.c_str(); }

int myslam_rccl_load(const char* path) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_lib) return 0;
    const char* cand[3] = {path, "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* c : cand) { if (c && (h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break; }
    if (!h) return fail(std::string("dlopen librccl.so failed: ") + (dlerror() ? dlerror() : "?"));
    p_get_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId"); p_init = (CommInitRankFn)dlsym(h, "ncclCommInitRank"); p_allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    p_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy"); p_errstr = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    if (!p_get_id || !p_init || !p_allreduce || !p_destroy || !p_errstr) { dlclose(h); p_get_id = nullptr; p_init = nullptr; p_allreduce = nullptr; p_destroy = nullptr; p_errstr = nullptr; return fail("librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString"); }
    g_lib = h;
    return 0;
}

int myslam_rccl_unique_id(char id[MYSLAM_RCCL_ID_BYTES]) {
    if (!id) return fail("null id");
    if (myslam_rccl_load(nullptr)) return -1;
    NcclUniqueId u;
    const int rc = p_get_id(&u);
    if (rc != kNcclSuccess) return fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, MYSLAM_RCCL_ID_BYTES);
    return 0;
}

int myslam_rccl_comm_create(const char id[MYSLAM_RCCL_ID_BYTES], int rank, int world, void** comm) {
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return fail("bad rank / world / id");
    if (myslam_rccl_load(nullptr)) return -1;
    NcclUniqueId u;
    memcpy(u.internal, id, MYSLAM_RCCL_ID_BYTES);
    NcclComm c = nullptr;
    const int rc = p_init(&c, world, u, rank);
    if (rc != kNcclSuccess) return fail("ncclCommInitRank", rc);
    *comm = c;
    return 0;
}

void myslam_rccl_comm_destroy(void* comm) { if (comm && p_destroy) (void)p_destroy(comm); }

int myslam_rccl_id_via_file(const char* path, int rank, int timeout_s, char id[MYSLAM_RCCL_ID_BYTES]) {
    if (!path || !id) return fail("null path / id");
    if (rank == 0) {
        if (myslam_rccl_unique_id(id)) return -1;
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, MYSLAM_RCCL_ID_BYTES, f) != MYSLAM_RCCL_ID_BYTES) { if (f) fclose(f); return fail("cannot write " + tmp); }
        fclose(f);
        if (rename(tmp.c_str(), path) != 0) return fail(std::string("cannot rename to ") + path);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t n = fread(id, 1, MYSLAM_RCCL_ID_BYTES, f);
            fclose(f);
            if (n == MYSLAM_RCCL_ID_BYTES) return 0;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) return fail(std::string("no RCCL id in ") + path + " after the timeout");
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

int myslam_rccl_allreduce_i32(void* comm, int32_t* buf, size_t n, void* stream) {
    if (!p_allreduce || !comm) return fail("RCCL not loaded / no communicator");
    const int rc = p_allreduce(buf, buf, n, kNcclInt32, kNcclSum, comm, stream);
    return rc == kNcclSuccess ? 0 : fail("ncclAllReduce(int32)", rc);
}

int myslam_rccl_allreduce_f64(void* comm, double* buf, size_t n, void* stream) {
    if (!p_allreduce || !comm) return fail("RCCL not loaded / no communicator");
    const int rc = p_allreduce(buf, buf, n, kNcclFloat64, kNcclSum, comm, stream);
    return rc == kNcclSuccess ? 0 : fail("ncclAllReduce(f64)", rc);
}

}  // extern "C"
