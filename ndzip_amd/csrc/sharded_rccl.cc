// libndzip_hip_rccl.so, the RCCL part: the all-gather of include/ndzip_hip_sharded.h's table over a caller's ncclComm_t, and the
// three bootstrap helpers.  RCCL over xGMI is used for exactly two exchanges per compress -- one uint32 per rank, then the header
// segments (at 8 ranks of config 4: 8 x 256 KiB) -- both latency-bound rings; the hypercube runs never leave the GPU that made them.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include "../../include/ndzip_hip_sharded.h"

static_assert(NDZIP_HIP_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id buffer of the C ABI is ncclUniqueId");

namespace {

int rccl_all_gather_u32(void *ctx, const uint32_t *d_send, uint32_t *d_recv, size_t count, void *hip_stream) {
    return static_cast<int>(ncclAllGather(d_send, d_recv, count, ncclUint32, static_cast<ncclComm_t>(ctx), static_cast<hipStream_t>(hip_stream)));
}

const char *rccl_error_string(void *, int code) { return ncclGetErrorString(static_cast<ncclResult_t>(code)); }

}  // namespace

int ndzip_sharded_fail(int status, const char *fmt, ...);  // sharded.cc

namespace {

int rccl_fail(ncclResult_t r, const char *what) { return ndzip_sharded_fail(NDZIP_HIP_ERR_RUNTIME, "%s: %s", what, ncclGetErrorString(r)); }

}  // namespace

extern "C" {

NDZIP_HIP_API int ndzip_hip_rccl_unique_id(void *id_bytes) {
    if (!id_bytes) return ndzip_sharded_fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    ncclUniqueId id;
    if (ncclResult_t r = ncclGetUniqueId(&id); r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(id_bytes, &id, sizeof id);
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_rccl_comm_create(const void *id_bytes, uint32_t rank, uint32_t world, void **nccl_comm) {
    if (!id_bytes || !nccl_comm || rank >= world) return ndzip_sharded_fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "null argument or rank outside the plan");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    if (ncclResult_t r = ncclCommInitRank(&comm, static_cast<int>(world), id, static_cast<int>(rank)); r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    *nccl_comm = comm;
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_rccl_comm_destroy(void *nccl_comm) {
    if (!nccl_comm) return NDZIP_HIP_OK;
    if (ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(nccl_comm)); r != ncclSuccess) return rccl_fail(r, "ncclCommDestroy");
    return NDZIP_HIP_OK;
}

NDZIP_HIP_API int ndzip_hip_sharded_create(int dtype, int dims, const uint32_t *global_extent, uint32_t rank, uint32_t world, void *nccl_comm,
        void *hip_stream, ndzip_hip_sharded **out) {
    if (out) *out = nullptr;
    if (!nccl_comm && world > 1) return ndzip_sharded_fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "no communicator for a plan of %u ranks", world);  // (one shard exchanges nothing and needs none)
    if (nccl_comm) {  // the communicator must be the plan's
        int n = 0, r = -1;
        if (ncclResult_t e = ncclCommCount(static_cast<ncclComm_t>(nccl_comm), &n); e != ncclSuccess) return rccl_fail(e, "ncclCommCount");
        if (ncclResult_t e = ncclCommUserRank(static_cast<ncclComm_t>(nccl_comm), &r); e != ncclSuccess) return rccl_fail(e, "ncclCommUserRank");
        if (n != static_cast<int>(world) || r != static_cast<int>(rank)) {
            return ndzip_sharded_fail(NDZIP_HIP_ERR_INVALID_ARGUMENT, "communicator is rank %d of %d, the plan says rank %u of %u", r, n, rank, world);
        }
    }
    const ndzip_hip_collectives table{nccl_comm, rccl_all_gather_u32, rccl_error_string};
    return ndzip_hip_sharded_create_with_collectives(dtype, dims, global_extent, rank, world, &table, hip_stream, out);
}

}  // extern "C"
