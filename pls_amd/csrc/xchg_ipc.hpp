// xchg_ipc.hpp -- the device-side exchange across processes (include/pls_hip.h, pls_hip_xchg_*): inboxes opened over IPC.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

// =============================================================================================
// One process per GPU without RCCL: the device-side exchange across processes (include/pls_hip.h, pls_hip_xchg_*)
// =============================================================================================
struct XchgIpc {
    int rank = 0, n = 1, device = 0;
    double *inbox[plsk::XCHG_MAX] = {nullptr};
    unsigned long long *flags[plsk::XCHG_MAX] = {nullptr};
    bool opened[plsk::XCHG_MAX] = {false};
    int *status = nullptr, *host_status = nullptr, *host_status_dev = nullptr;
    unsigned long long seq = 0;
    long long limit = 0;
    bool connected = false;
};

namespace {

struct XchgBlob {  // what a rank publishes (PLS_HIP_XCHG_HANDLE_BYTES)
    hipIpcMemHandle_t inbox, flags;
    int64_t pid;
    int32_t rank, nranks;
    char pad[PLS_HIP_XCHG_HANDLE_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 16];
};
static_assert(sizeof(XchgBlob) == PLS_HIP_XCHG_HANDLE_BYTES, "exchange blob size");

int ipc_allreduce(void *user, void *buf, int64_t count, void *stream) {
    pls_hip_context *c = static_cast<pls_hip_context *>(user);
    XchgIpc *x = c->xchg;
    if (!x || !x->connected || count % plsk::RED_SLICES != 0) return 20;
    const i64 L = count / plsk::RED_SLICES;
    for (i64 j0 = 0; j0 < L; j0 += plsk::XCHG_CAP) {  // (long messages -- X^T X of KERNEL_TYPE2 -- in pieces)
        const int Lc = (int)std::min<i64>(plsk::XCHG_CAP, L - j0);
        const int rc = plsk::xchg_launch_piece((hipStream_t)stream, x->n, x->rank, x->inbox, x->flags, (double *)buf, L, j0, Lc,
                                               plsk::RED_SLICES, ++x->seq, x->status, x->host_status_dev, x->limit);
        if (rc != 0) return rc;
    }
    return 0;
}

void xchg_release(pls_hip_context *c) {
    XchgIpc *x = c->xchg;
    if (!x) return;
    c->xep = pls_hip_context::XchgEndpoint();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int j = 0; j < x->n; ++j)
        if (x->opened[j]) {
            (void)hipIpcCloseMemHandle(x->inbox[j]);
            (void)hipIpcCloseMemHandle(x->flags[j]);
        }
    if (x->inbox[x->rank]) (void)hipFree(x->inbox[x->rank]);
    if (x->flags[x->rank]) (void)hipFree(x->flags[x->rank]);
    if (x->host_status) (void)hipHostFree(x->host_status);
    (void)hipGetLastError();
    delete x;
    c->xchg = nullptr;
}

int check_xchg(pls_hip_context *c) {
    if (c->xchg && c->xchg->host_status && *c->xchg->host_status)
        return fail(c, PLS_HIP_ERR_REDUCER, "device-side exchange: this rank waited longer than the time limit for its peers' partial "
                                            "sums (a rank failed or fell out of step); the exchange is unusable until it is set up again");
    return PLS_HIP_OK;
}

}  // namespace

extern "C" {

int pls_hip_xchg_create(pls_hip_handle h, int rank, int nranks, void *mine) {
    CHK(check_handle(h));
    if (!mine || nranks < 2 || nranks > plsk::XCHG_MAX || rank < 0 || rank >= nranks)
        return fail(h, PLS_HIP_ERR_INVALID, "bad exchange arguments (2..16 ranks)");
    CHK(set_device(h));
    if (h->xchg) xchg_release(h);
    XchgIpc *x = new (std::nothrow) XchgIpc();
    if (!x) return PLS_HIP_ERR_ALLOC;
    x->rank = rank; x->n = nranks; x->device = h->device;
    h->xchg = x;
    const size_t ib = (size_t)2 * nranks * plsk::XCHG_CAP * 8, fb = (size_t)2 * nranks * 8 + 64;
    XchgBlob blob;
    std::memset(&blob, 0, sizeof(blob));
    if (hipExtMallocWithFlags((void **)&x->inbox[rank], ib, hipDeviceMallocFinegrained) != hipSuccess ||
        hipExtMallocWithFlags((void **)&x->flags[rank], fb, hipDeviceMallocFinegrained) != hipSuccess ||
        hipMemset(x->flags[rank], 0, fb) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipHostMalloc((void **)&x->host_status, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&x->host_status_dev, x->host_status, 0) != hipSuccess ||
        hipIpcGetMemHandle(&blob.inbox, x->inbox[rank]) != hipSuccess || hipIpcGetMemHandle(&blob.flags, x->flags[rank]) != hipSuccess) {
        const std::string why = hipGetErrorString(hipGetLastError());
        xchg_release(h);
        return fail(h, PLS_HIP_ERR_DEVICE, "exchange memory could not be set up or exported: " + why);
    }
    *x->host_status = 0;
    x->status = reinterpret_cast<int *>(x->flags[rank] + 2 * nranks);
    x->limit = plsk::xchg_time_limit(h->device);
    blob.pid = (int64_t)getpid();
    blob.rank = rank; blob.nranks = nranks;
    std::memcpy(mine, &blob, sizeof(blob));
    return PLS_HIP_OK;
}

int pls_hip_xchg_connect(pls_hip_handle h, const void *all) {
    CHK(check_handle(h));
    XchgIpc *x = h->xchg;
    if (!x || !all) return fail(h, PLS_HIP_ERR_INVALID, "pls_hip_xchg_create first");
    CHK(set_device(h));
    const XchgBlob *blobs = static_cast<const XchgBlob *>(all);
    for (int j = 0; j < x->n; ++j) {
        if (blobs[j].rank != j || blobs[j].nranks != x->n) return fail(h, PLS_HIP_ERR_INVALID, "exchange handles are not in rank order");
        if (j == x->rank) continue;
        if (hipIpcOpenMemHandle((void **)&x->inbox[j], blobs[j].inbox, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
            hipIpcOpenMemHandle((void **)&x->flags[j], blobs[j].flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            const std::string why = hipGetErrorString(hipGetLastError());
            return fail(h, PLS_HIP_ERR_DEVICE, "inbox of rank " + std::to_string(j) + " could not be opened: " + why);
        }
        x->opened[j] = true;
    }
    x->connected = true;
    h->xep.on = true;
    h->xep.n = x->n; h->xep.rank = x->rank;
    h->xep.inbox = x->inbox; h->xep.flags = x->flags;
    h->xep.seq = &x->seq; h->xep.status = x->status; h->xep.host_status = x->host_status_dev; h->xep.limit = &x->limit;
    return pls_hip_set_reducer(h, ipc_allreduce, h, x->rank, x->n);
}

int pls_hip_xchg_selftest(pls_hip_handle h) {
    CHK(check_handle(h));
    XchgIpc *x = h->xchg;
    if (!x || !x->connected) return fail(h, PLS_HIP_ERR_INVALID, "pls_hip_xchg_connect first");
    CHK(set_device(h));
    CHK(ensure(h, h->guard, (size_t)plsk::RED_SLICES * 8 * 8));
    double host[plsk::RED_SLICES * 8] = {0};
    for (int j = 0; j < 8; ++j) host[j] = (x->rank + 1.0) * (j + 1);  // slice 0; the other slices stay zero
    HIPCHK(h, hipMemcpyAsync(h->guard.p, host, sizeof(host), hipMemcpyHostToDevice, h->stream));
    const long long keep = x->limit;
    x->limit = std::min(keep, plsk::xchg_time_limit(h->device) / 6 + 1);  // a short limit for the probe round (5 s by default)
    const int rc = ipc_allreduce(h, h->guard.p, plsk::RED_SLICES * 8, (void *)h->stream);
    x->limit = keep;
    if (rc != 0) return fail(h, PLS_HIP_ERR_REDUCER, "exchange launch failed");
    HIPCHK(h, hipMemcpyAsync(host, h->guard.p, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(check_xchg(h));
    for (int j = 0; j < 8; ++j)
        if (host[j] != 0.5 * x->n * (x->n + 1) * (j + 1)) return fail(h, PLS_HIP_ERR_REDUCER, "exchange self-test: wrong sums");
    return PLS_HIP_OK;
}

int pls_hip_xchg_destroy(pls_hip_handle h) {
    CHK(check_handle(h));
    if (h->xchg) {
        xchg_release(h);
        return pls_hip_set_reducer(h, nullptr, nullptr, 0, 1);
    }
    return PLS_HIP_OK;
}

}  // extern "C"
