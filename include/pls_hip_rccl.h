/*
 * pls_hip_rccl.h -- optional helper: an RCCL all-reduce as the reducer of a pls_hip handle, issued
 * directly on the handle's stream (no host runtime in the loop).  For C++ hosts that shard a fit over
 * the GPUs of a node, one process (or thread) per GPU:
 *
 *     char id[PLS_HIP_RCCL_ID_BYTES];
 *     if (rank == 0) pls_hip_rccl_unique_id(id);          // then broadcast id to every rank (MPI, sockets, ...)
 *     void *comm;
 *     pls_hip_rccl_attach(handle, device, id, rank, nranks, &comm); // ncclCommInitRank + pls_hip_set_reducer
 *     pls_hip_fit(handle, X_local, ...);                    // partial products are summed over xGMI
 *     pls_hip_rccl_detach(handle, comm);
 *
 * Built as libpls_hip_rccl.so (links librccl); libpls_hip.so itself has no RCCL dependency.
 */
#ifndef PLS_HIP_RCCL_H
#define PLS_HIP_RCCL_H

#include "pls_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PLS_HIP_RCCL_ID_BYTES 128

/* fills id (PLS_HIP_RCCL_ID_BYTES bytes) with a fresh ncclUniqueId; call on one rank only */
PLS_HIP_API int pls_hip_rccl_unique_id(void *id);
/* creates the communicator for (rank, nranks) on the handle's device and installs it as the reducer */
PLS_HIP_API int pls_hip_rccl_attach(pls_hip_handle h, int device, const void *id, int rank, int nranks, void **comm_out);
/* the number of ranks the communicator itself reports (ncclCommCount): evidence that RCCL saw every rank */
PLS_HIP_API int pls_hip_rccl_comm_count(void *comm, int *nranks);
/* removes the reducer (back to a single-rank handle) and destroys the communicator */
PLS_HIP_API int pls_hip_rccl_detach(pls_hip_handle h, void *comm);

#ifdef __cplusplus
}
#endif
#endif
