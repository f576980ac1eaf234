// libpls_hip_rccl.so -- RCCL reducer for include/pls_hip.h handles (see include/pls_hip_rccl.h).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include "../../include/pls_hip_rccl.h"

static_assert(sizeof(ncclUniqueId) <= PLS_HIP_RCCL_ID_BYTES, "id buffer too small");

namespace {
// pls_hip_allreduce_fn: in-place fp64 sum on the launch stream; RCCL's ring / tree reductions leave
// identical bits on every rank
int rccl_allreduce(void *user, void *buf, int64_t count, void *stream) {
    ncclComm_t comm = static_cast<ncclComm_t>(user);
    return ncclAllReduce(buf, buf, static_cast<size_t>(count), ncclDouble, ncclSum, comm,
                         static_cast<hipStream_t>(stream)) == ncclSuccess
               ? 0
               : 1;
}
}  // namespace

extern "C" {

int pls_hip_rccl_unique_id(void *id) {
    if (!id) return PLS_HIP_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PLS_HIP_ERR_DEVICE;
    std::memset(id, 0, PLS_HIP_RCCL_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return PLS_HIP_OK;
}

int pls_hip_rccl_attach(pls_hip_handle h, int device, const void *id, int rank, int nranks, void **comm_out) {
    if (!h || !id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return PLS_HIP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return PLS_HIP_ERR_DEVICE;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    if (ncclCommInitRank(&comm, nranks, u, rank) != ncclSuccess) return PLS_HIP_ERR_DEVICE;
    const int rc = pls_hip_set_reducer(h, rccl_allreduce, comm, rank, nranks);
    if (rc != PLS_HIP_OK) {
        ncclCommDestroy(comm);
        return rc;
    }
    *comm_out = comm;
    return PLS_HIP_OK;
}

int pls_hip_rccl_comm_count(void *comm, int *nranks) {
    if (!comm || !nranks) return PLS_HIP_ERR_INVALID;
    return ncclCommCount(static_cast<ncclComm_t>(comm), nranks) == ncclSuccess ? PLS_HIP_OK : PLS_HIP_ERR_DEVICE;
}

int pls_hip_rccl_detach(pls_hip_handle h, void *comm) {
    if (!h) return PLS_HIP_ERR_INVALID;
    (void)pls_hip_synchronize(h);
    const int rc = pls_hip_set_reducer(h, nullptr, nullptr, 0, 1);
    if (comm) ncclCommDestroy(static_cast<ncclComm_t>(comm));
    return rc;
}

}  // extern "C"
