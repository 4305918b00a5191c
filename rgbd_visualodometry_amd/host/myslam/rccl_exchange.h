// rccl_exchange.h -- the exchanges of the sharded paths as RCCL all-reduces enqueued on a HIP stream, with no Python and no link-time dependency:
// librccl.so is loaded with dlopen when a multi-rank run asks for it.  SURVEY.md 8e: ncclAllReduce of the per-hypothesis inlier counts (item 2, hypotheses of
// reference src/frontend.cpp:238-241 split over ranks) and of the local BA's reduced system S, b_s (item 2's last sentence, src/backend.cpp:19-195).
// The two functions below have exactly the types include/vo_hip.h asks for (vo_stream_allreduce_fn / vo_stream_allreduce_f64_fn): pass them with the
// communicator as `comm` to vo_set_hypothesis_shard_stream / vo_set_ba_shard_stream.
#ifndef MYSLAM_RCCL_EXCHANGE_H
#define MYSLAM_RCCL_EXCHANGE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define MYSLAM_RCCL_ID_BYTES 128
/* dlopen librccl.so (path NULL: "librccl.so", then "/opt/rocm/lib/librccl.so") and resolve the five entry points; 0 = ok.  Idempotent. */
int myslam_rccl_load(const char* path);
/* rank 0: a fresh unique id (ncclGetUniqueId) to hand to the other ranks by whatever channel the job has (a file, the launcher's store) */
int myslam_rccl_unique_id(char id[MYSLAM_RCCL_ID_BYTES]);
/* ncclCommInitRank on the calling thread's current HIP device; every rank of the job calls it with the same id */
int myslam_rccl_comm_create(const char id[MYSLAM_RCCL_ID_BYTES], int rank, int world, void** comm);
void myslam_rccl_comm_destroy(void* comm);
/* file rendezvous for drivers without a launcher (run_vo): rank 0 writes the id to `path` (atomically: temp file + rename), the others wait for it */
int myslam_rccl_id_via_file(const char* path, int rank, int timeout_s, char id[MYSLAM_RCCL_ID_BYTES]);
/* vo_stream_allreduce_fn: in-place int32 SUM of n device-resident counts, enqueued on hip_stream */
int myslam_rccl_allreduce_i32(void* comm, int32_t* device_counts, size_t n, void* hip_stream);
/* vo_stream_allreduce_f64_fn: in-place f64 SUM */
int myslam_rccl_allreduce_f64(void* comm, double* device_data, size_t n, void* hip_stream);
const char* myslam_rccl_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
