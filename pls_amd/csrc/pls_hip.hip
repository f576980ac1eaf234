// pls_hip.hip -- the C-ABI of include/pls_hip.h over the gfx950 kernels in this directory.
// Host side of the library: argument checks, workspace, launch geometry, the A-loop of
// Model::plsr (src/pls.cpp:390-437) enqueued on one HIP stream with no host round trip,
// the injected all-reduce for row-sharded fits, HIP-event profiling.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pls_hip.h"
#include "fused_kernels.hpp"
#include "defer_kernels.hpp"
#include "small_kernels.hpp"
#include "coop_update.hpp"
#include "wide1_update.hpp"
#include "largem_kernels.hpp"
#include "tiny_kernels.hpp"
#include "resident_kernels.hpp"
#include "resident_gram.hpp"
#include "stream_kernels.hpp"
#include "xb_mfma4.hpp"
#include "xb_mfma4w.hpp"
#include "syrk_kernels.hpp"
#include "cv_kernels.hpp"
#include "synth_kernels.hpp"
#include "host_pipeline.hpp"
#include "exchange_kernels.hpp"

using plsk::i64;

#include "ctx.hpp"
#include "launch_products.hpp"
#include "launch_update.hpp"
#include "plan_common.hpp"
#include "plan_fit.hpp"
#include "host_entry.hpp"

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int pls_hip_abi_version(void) { return PLS_HIP_ABI_VERSION; }

#ifdef PLS_HIP_TESTING
// testing/libpls_hip.so only (not in include/pls_hip.h): 4 wall-clock stamps per workgroup of the fused passes that follow go
// to `buf` (device memory, 64 bytes per workgroup (4 stamps, XCC_ID, HW_ID); nullptr: off)
__attribute__((visibility("default"))) int pls_hip_test_set_pass_stamps(void *buf) {
    unsigned long long *p = (unsigned long long *)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(plsk::g_pass_stamps), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

int pls_hip_create(pls_hip_handle *out, int device, void *stream) {
    if (!out) return PLS_HIP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return PLS_HIP_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PLS_HIP_ERR_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PLS_HIP_ERR_DEVICE;  // no other target
    pls_hip_context *c = new (std::nothrow) pls_hip_context();
    if (!c) return PLS_HIP_ERR_ALLOC;
    {
        auto off = [](const char *name) { const char *e = getenv(name); return e && atoi(e) == 0; };
        auto on = [](const char *name) { const char *e = getenv(name); return e && atoi(e) != 0; };
        c->env.tiny = !off("PLS_HIP_TINY");
        c->env.cv_refit = on("PLS_HIP_CV_REFIT");
        c->env.tail = !off("PLS_HIP_TAIL");
        {
            const char *e = getenv("PLS_HIP_TAIL");
            c->env.tail_update = e ? (atoi(e) >= 2 ? 2 : 0) : 1;
        }
        c->env.replica_guard = !off("PLS_HIP_REPLICA_GUARD");
        c->env.resident = !off("PLS_HIP_RESIDENT");
        if (const char *e = getenv("PLS_HIP_XB4")) c->env.xb4 = atoi(e);
        if (const char *e = getenv("PLS_HIP_RESIDENT_GRAM")) c->env.resident_gram = atoi(e);
    }
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipSetDevice(device) != hipSuccess) { delete c; return PLS_HIP_ERR_DEVICE; }
    c->stream = (hipStream_t)stream;  // NULL = the device's default (null) stream
    *out = c;
    return PLS_HIP_OK;
}

int pls_hip_destroy(pls_hip_handle h) {
    if (!h) return PLS_HIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->tailcnt, &h->resident, &h->rgflags, &h->zeros, &h->part, &h->sspart, &h->xbpart, &h->wide1, &h->red, &h->red2, &h->xx, &h->xyp, &h->praw, &h->xy, &h->v, &h->cs, &h->coop, &h->lm, &h->gxx, &h->gxy, &h->tab,
                      &h->cvidx, &h->cvx, &h->cvy, &h->cvws, &h->cve, &h->cvtx, &h->cvty, &h->cvtt, &h->cvm, &h->cvkeep, &h->work, &h->hX, &h->hY,
                      &h->hT, &h->hW, &h->hP, &h->hQ, &h->hR, &h->hB, &h->hIn, &h->hOut};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (void *q : h->graveyard) (void)hipFree(q);
    if (h->xchg) xchg_release(h);
    if (h->guard.p) (void)hipFree(h->guard.p);
    if (h->diverged) (void)hipHostFree(h->diverged);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->stager.release();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    delete h;
    return PLS_HIP_OK;
}

int pls_hip_set_stream(pls_hip_handle h, void *stream) {
    CHK(check_handle(h));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)stream;
    return PLS_HIP_OK;
}

int pls_hip_set_option(pls_hip_handle h, int option, int64_t value) {
    CHK(check_handle(h));
    switch (option) {
        case PLS_HIP_OPT_ALGO:
            if (value != PLS_HIP_ALGO_KERNEL && value != PLS_HIP_ALGO_NIPALS && value != PLS_HIP_ALGO_GRAM &&
                value != PLS_HIP_ALGO_AUTO)
                return fail(h, PLS_HIP_ERR_INVALID, "unknown algo");
            h->opt_algo = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSE: h->opt_fuse = value ? 1 : 0; return PLS_HIP_OK;
        case PLS_HIP_OPT_PROFILE: h->opt_profile = value < 0 ? 0 : (value > 2 ? 2 : value); return PLS_HIP_OK;
        case PLS_HIP_OPT_POWER_ITERS:
            if (value < 1 || value > 4096) return fail(h, PLS_HIP_ERR_INVALID, "power iters out of range");
            h->opt_power_iters = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSED_GRID:
            if (value < 0 || value > (1 << 20)) return fail(h, PLS_HIP_ERR_INVALID, "fused grid out of range");
            h->opt_fused_grid = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_WORK_LAYOUT: h->opt_work_layout = value ? 1 : 0; return PLS_HIP_OK;
        case PLS_HIP_OPT_DEFER:
            if (value < 1 || value > plsk::DEFER_MAX) return fail(h, PLS_HIP_ERR_INVALID, "defer out of range");
            h->opt_defer = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_GRAPH: h->opt_graph = value ? 1 : 0; return PLS_HIP_OK;
        default: return fail(h, PLS_HIP_ERR_INVALID, "unknown option");
    }
}

int pls_hip_get_option(pls_hip_handle h, int option, int64_t *value) {
    CHK(check_handle(h));
    if (!value) return PLS_HIP_ERR_INVALID;
    switch (option) {
        case PLS_HIP_OPT_ALGO: *value = h->opt_algo; return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSE: *value = h->opt_fuse; return PLS_HIP_OK;
        case PLS_HIP_OPT_PROFILE: *value = h->opt_profile; return PLS_HIP_OK;
        case PLS_HIP_OPT_POWER_ITERS: *value = h->opt_power_iters; return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSED_GRID: *value = h->opt_fused_grid; return PLS_HIP_OK;
        case PLS_HIP_OPT_GRAPH: *value = h->opt_graph; return PLS_HIP_OK;
        case PLS_HIP_OPT_WORK_LAYOUT: *value = h->opt_work_layout; return PLS_HIP_OK;
        case PLS_HIP_OPT_DEFER: *value = h->opt_defer; return PLS_HIP_OK;
        default: return fail(h, PLS_HIP_ERR_INVALID, "unknown option");
    }
}

int pls_hip_set_reducer(pls_hip_handle h, pls_hip_allreduce_fn fn, void *user, int rank, int nranks) {
    CHK(check_handle(h));
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn))
        return fail(h, PLS_HIP_ERR_INVALID, "bad reducer arguments");
    h->reducer = fn;
    h->reducer_user = user;
    h->rank = rank;
    h->nranks = nranks;
    return PLS_HIP_OK;
}

int pls_hip_set_reduce_buffer(pls_hip_handle h, void *buf, int64_t count) {
    CHK(check_handle(h));
    if ((buf && count <= 0) || (!buf && count != 0)) return fail(h, PLS_HIP_ERR_INVALID, "bad reduce buffer");
    h->user_red = (double *)buf;
    h->user_red_count = count;
    return PLS_HIP_OK;
}

int pls_hip_synchronize(pls_hip_handle h) {
    CHK(check_handle(h));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_diverged(h);
}

const char *pls_hip_last_error(pls_hip_handle h) { return h ? h->err.c_str() : "null handle"; }

int pls_hip_get_timing(pls_hip_handle h, pls_hip_timing *out) {
    CHK(check_handle(h));
    if (!out) return PLS_HIP_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (const Launch &l : h->fits) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, l.e0, l.e1));
        out->fit_ms += ms;
        out->fits += 1;
    }
    for (const Launch &l : h->launches) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, l.e0, l.e1));
        out->fam_ms[l.fam] += ms;
        out->fam_launches[l.fam] += 1;
        out->fam_bytes[l.fam] += l.bytes;
    }
    h->launches.clear();  // harvested: recycle the event pool
    h->fits.clear();
    h->ev_used = 0;
    return PLS_HIP_OK;
}

int pls_hip_fit(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                int64_t K, int64_t M, int64_t A, int method, int dtype, int mem, double *W, double *P,
                double *Q, double *R, void *T, int64_t ldt, double *B) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    if (method != PLS_HIP_KERNEL_TYPE1 && method != PLS_HIP_KERNEL_TYPE2)
        return fail(h, PLS_HIP_ERR_INVALID, "bad method");
    const bool sharded = h->nranks > 1;
    if (N < 0 || (N == 0 && !sharded) || K < 1 || M < 1 || A < 1 || A > K)
        return fail(h, PLS_HIP_ERR_INVALID, "bad shape: need N>=1, K>=1, M>=1, 1<=A<=K");
    if (K > (1 << 30) || M > plsk::LM_MAX)
        return fail(h, PLS_HIP_ERR_UNSUPPORTED, "more than 1024 responses (or K > 2^30) not supported on the device");
    if (N > 0 && (!X || !Y || (!T && method == PLS_HIP_KERNEL_TYPE1))) return fail(h, PLS_HIP_ERR_INVALID, "null X/Y/T");
    if ((method == PLS_HIP_KERNEL_TYPE2 || h->opt_algo == PLS_HIP_ALGO_GRAM) && K > 32768)  // (AUTO never picks GRAM there)
        return fail(h, PLS_HIP_ERR_UNSUPPORTED, "KERNEL_TYPE2 keeps a K x K matrix and 8 reduction slices of it (77 GB at K = 32768): K <= 32768");
    if (!W || !P || !Q || !R) return fail(h, PLS_HIP_ERR_INVALID, "null W/P/Q/R");
    if (ldx < std::max<i64>(N, 1) || ldy < std::max<i64>(N, 1) || (T && ldt < std::max<i64>(N, 1)))
        return fail(h, PLS_HIP_ERR_INVALID, "leading dimension smaller than N");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const int Ki = (int)K, Mi = (int)M, Ai = (int)A;

    const void *dX = X, *dY = Y;
    void *dT = T;
    double *dW = W, *dP = P, *dQ = Q, *dR = R, *dB = B;
    i64 dldx = ldx, dldy = ldy, dldt = ldt;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = std::max<i64>(N, 1) + (std::max<i64>(N, 1) & 1);  // even ld keeps 16-B columns
        CHK(ensure(h, h->hX, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(ensure(h, h->hT, (size_t)ldn * A * es));
        CHK(ensure(h, h->hW, (size_t)K * A * 8));
        CHK(ensure(h, h->hP, (size_t)K * A * 8));
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)K * M * 8));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        // A plan that works from X^T X (AUTO, GRAM, KERNEL_TYPE2) gets it for free: accumulated on the matrix cores
        // row block by row block while X crosses PCIe (upload_accumulate).  Single rank only: the ranks of a sharded
        // fit must not differ in their plan.
        bool pre = false;
        const bool wants_gram = h->opt_algo == PLS_HIP_ALGO_AUTO || h->opt_algo == PLS_HIP_ALGO_GRAM || method == PLS_HIP_KERNEL_TYPE2;
        const bool single_launch = method == PLS_HIP_KERNEL_TYPE1 && h->opt_algo == PLS_HIP_ALGO_AUTO && h->opt_fuse &&
                                   (plsk::tiny_fit_covers(N, Ki, Mi, Ai, ldn, es) || plsk::tiny_fit_m_covers(N, Ki, Mi, Ai, ldn, es) ||
                                    plsk::micro_fit_covers(N, Ki, Mi, Ai, ldn, es));  // (no use for X^T X there)
        if (wants_gram && !single_launch && !h->reducer && N > 0 && K <= 4096 && ensure(h, h->gxx, (size_t)K * K * 8) == PLS_HIP_OK &&
            ensure(h, h->gxy, (size_t)K * M * 8) == PLS_HIP_OK) {
            if (dtype == PLS_HIP_F64)
                CHK(upload_accumulate<double>(h, (double *)h->hX.p, ldn, (const double *)X, ldx, N, Ki, (const double *)h->hY.p, ldn,
                                              Mi, (double *)h->gxx.p, (double *)h->gxy.p, &pre));
            else
                CHK(upload_accumulate<float>(h, (float *)h->hX.p, ldn, (const float *)X, ldx, N, Ki, (const float *)h->hY.p, ldn, Mi,
                                             (double *)h->gxx.p, (double *)h->gxy.p, &pre));
        } else {
            h->err.clear();
            CHK(h2d(h, h->hX.p, ldn, X, ldx, N, K, es));
        }
        if (pre) {
            h->pre_xx = (const double *)h->gxx.p;
            h->pre_xy = (const double *)h->gxy.p;
        }
        dX = h->hX.p; dY = h->hY.p; dT = h->hT.p;
        dW = (double *)h->hW.p; dP = (double *)h->hP.p; dQ = (double *)h->hQ.p; dR = (double *)h->hR.p;
        dB = B ? (double *)h->hB.p : nullptr;
        dldx = dldy = dldt = ldn;
    }
    // PLS_HIP_OPT_GRAPH: a repeated device-memory fit is replayed as one graph launch.  First occurrence of a call: eager (it
    // also sizes every workspace); second: the same enqueue sequence under stream capture (nothing allocates any more),
    // instantiated and launched; from the third on: hipGraphLaunch.  The kernel arguments are baked into the graph, so the key
    // is everything they derive from.
    const bool graphable = h->opt_graph && mem == PLS_HIP_MEM_DEVICE && !h->reducer && h->opt_profile == 0 && !h->user_red &&
                           h->stream != nullptr;  // (the legacy default stream cannot be captured)
    std::vector<uint64_t> key;
    if (graphable) {
        key = {(uint64_t)(uintptr_t)X, (uint64_t)ldx, (uint64_t)(uintptr_t)Y, (uint64_t)ldy, (uint64_t)N, (uint64_t)K, (uint64_t)M,
               (uint64_t)A, (uint64_t)method, (uint64_t)dtype, (uint64_t)(uintptr_t)W, (uint64_t)(uintptr_t)P, (uint64_t)(uintptr_t)Q,
               (uint64_t)(uintptr_t)R, (uint64_t)(uintptr_t)T, (uint64_t)ldt, (uint64_t)(uintptr_t)B, (uint64_t)h->opt_algo,
               (uint64_t)h->opt_fuse, (uint64_t)h->opt_power_iters, (uint64_t)h->opt_fused_grid, (uint64_t)h->opt_work_layout,
               (uint64_t)h->opt_defer, (uint64_t)(uintptr_t)h->stream, (uint64_t)(uintptr_t)h->pre_xx, (uint64_t)(uintptr_t)h->pre_xy};
        if (h->graph_exec && key == h->graph_key) {
            HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
            return PLS_HIP_OK;
        }
    }
    const bool capture = graphable && key == h->graph_seen;
    if (capture) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; h->graph_key.clear(); }
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    }
    begin_fit_timing(h);
    int rc;
    if (dtype == PLS_HIP_F64)
        rc = fit_device<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, Ki, Mi, Ai, method,
                                dW, dP, dQ, dR, (double *)dT, dldt, dB);
    else
        rc = fit_device<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, Ki, Mi, Ai, method, dW,
                               dP, dQ, dR, (float *)dT, dldt, dB);
    end_fit_timing(h);
    if (capture) {
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
        if (rc == PLS_HIP_OK && ce == hipSuccess && graph && hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            h->graph_key = key;
            (void)hipGraphDestroy(graph);
            HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));  // this call's work
        } else {
            if (graph) (void)hipGraphDestroy(graph);
            h->graph_exec = nullptr;
            h->graph_seen.clear();
            (void)hipGetLastError();
            if (rc != PLS_HIP_OK) return rc;
            return fail(h, PLS_HIP_ERR_DEVICE, "stream capture of the fit failed");
        }
    } else if (graphable) {
        h->graph_seen = key;
    }
    if (mem == PLS_HIP_MEM_HOST) h->pre_xx = h->pre_xy = nullptr;
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, W, K, dW, K, K, A, 8));
        CHK(d2h(h, P, K, dP, K, K, A, 8));
        CHK(d2h(h, R, K, dR, K, K, A, 8));
        CHK(d2h(h, Q, M, dQ, M, M, A, 8));
        if (B) CHK(d2h(h, B, K, dB, K, K, M, 8));
        if (method == PLS_HIP_KERNEL_TYPE1) CHK(d2h(h, T, ldt, dT, dldt, N, A, es));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        CHK(check_diverged(h));
    }
    return PLS_HIP_OK;
}

int pls_hip_coefficients(pls_hip_handle h, const double *R, const double *Q, int64_t K, int64_t M,
                         int64_t A, int64_t cc, int mem, double *B) {
    CHK(check_handle(h));
    if (!R || !Q || !B || K < 1 || M < 1 || A < 1 || cc < 0 || cc > A || K > (1 << 30))
        return fail(h, PLS_HIP_ERR_INVALID, "bad coefficients arguments");  // comp <= A: src/pls.cpp:445
    CHK(set_device(h));
    const double *dR = R, *dQ = Q;
    double *dB = B;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)K * M * 8));
        CHK(h2d(h, h->hR.p, K, R, K, K, A, 8));
        CHK(h2d(h, h->hQ.p, M, Q, M, M, A, 8));
        dR = (double *)h->hR.p; dQ = (double *)h->hQ.p; dB = (double *)h->hB.p;
    } else if (mem != PLS_HIP_MEM_DEVICE) {
        return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    }
    const int nblk = (int)((K * M + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, h->stream, dR, dQ,
                       (int)K, (int)M, (int)cc, dB);
    LAUNCH_CHECK(h);
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, B, K, dB, K, K, M, 8));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_xb(pls_hip_handle h, const void *X, int64_t ldx, int64_t N, int64_t K, const double *Bm,
               int64_t ldb, int64_t C, int dtype, int mem, void *out, int64_t ldo) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 0 || K < 1 || C < 1 || K > (1 << 30) || C > (1 << 20) || ldb < K ||
        ldx < std::max<i64>(N, 1) || ldo < std::max<i64>(N, 1) || !Bm || (N > 0 && (!X || !out)))
        return fail(h, PLS_HIP_ERR_INVALID, "bad xb arguments");
    if (N == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const void *dX = X;
    const double *dBm = Bm;
    void *dO = out;
    i64 dldx = ldx, dldo = ldo, dldb = ldb;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = N + (N & 1);
        CHK(ensure(h, h->hIn, (size_t)ldn * K * es));
        CHK(ensure(h, h->hOut, (size_t)ldn * C * es));
        CHK(ensure(h, h->hB, (size_t)K * C * 8));
        CHK(h2d(h, h->hIn.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hB.p, K, Bm, ldb, K, C, 8));
        dX = h->hIn.p; dO = h->hOut.p; dBm = (const double *)h->hB.p;
        dldx = dldo = ldn; dldb = K;
    } else if (mem != PLS_HIP_MEM_DEVICE) {
        return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    }
    int nss = 0;
    if (dtype == PLS_HIP_F64)
        CHK(launch_xb<double>(h, (const double *)dX, dldx, N, (int)K, dBm, dldb, (int)C, (double *)dO, dldo, nullptr, &nss));
    else
        CHK(launch_xb<float>(h, (const float *)dX, dldx, N, (int)K, dBm, dldb, (int)C, (float *)dO, dldo, nullptr, &nss));
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, out, ldo, dO, dldo, N, C, es));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_xty(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                int64_t K, int64_t M, int dtype, double *XY) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || K < 1 || M < 1 || K > (1 << 30) || M > (1 << 20) || !X || !Y || !XY || ldx < N || ldy < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad xty arguments");
    CHK(set_device(h));
    CHK(ensure(h, h->part, (size_t)max_partial_rows(h, N, (int)K) * (size_t)(K * M) * 8));
    int nb = 0;
    if (dtype == PLS_HIP_F64)
        CHK(launch_xty<double>(h, (const double *)X, ldx, (const double *)Y, ldy, N, (int)K, (int)M, (double *)h->part.p, &nb));
    else
        CHK(launch_xty<float>(h, (const float *)X, ldx, (const float *)Y, ldy, N, (int)K, (int)M, (double *)h->part.p, &nb));
    CHK(ensure(h, h->red, (size_t)plsk::RED_SLICES * (size_t)(K * M) * 8));
    CHK(launch_reduce(h, (const double *)h->part.p, nb, (int)(K * M), nullptr, 0, (double *)h->red.p));
    const int nblk = (int)((K * M + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3(nblk), dim3(plsk::WG), 0, h->stream,
                       (const double *)h->red.p, (int)(K * M), XY);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

int pls_hip_deflate(pls_hip_handle h, const void *src, int64_t lds, void *dst, int64_t ldd, int64_t N,
                    int64_t K, const void *t, const double *p, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || K < 1 || K > (1 << 30) || !src || !dst || !t || !p || lds < N || ldd < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad deflate arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return launch_deflate<double>(h, (const double *)src, lds, (double *)dst, ldd, N, (int)K, (const double *)t, p);
    return launch_deflate<float>(h, (const float *)src, lds, (float *)dst, ldd, N, (int)K, (const float *)t, p);
}

}  // extern "C"

#include "plan_stats.hpp"


extern "C" {

int pls_hip_colwise_z_scores(pls_hip_handle h, const void *X, int64_t ldx, int64_t N, int64_t n_total,
                             int64_t K, int dtype, void *Z, int64_t ldz, double *mean, double *sd) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 0 || n_total < N || n_total < 1 || K < 1 || K > (1 << 30) || !mean || !sd || (N > 0 && !X) ||
        ldx < std::max<i64>(N, 1) || (Z && ldz < std::max<i64>(N, 1)))
        return fail(h, PLS_HIP_ERR_INVALID, "bad z-score arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return zscores_device<double>(h, (const double *)X, ldx, N, n_total, (int)K, (double *)Z, ldz, mean, sd);
    return zscores_device<float>(h, (const float *)X, ldx, N, n_total, (int)K, (float *)Z, ldz, mean, sd);
}

int pls_hip_sse_by_components(pls_hip_handle h, const void *S, int64_t lds, const void *Y, int64_t ldy,
                              int64_t N, int64_t A, int64_t M, const double *Q, int dtype, double *SSE) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || A < 1 || M < 1 || M > 1024 || A > (1 << 20) || !S || !Y || !Q || !SSE || lds < N || ldy < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad sse arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return sse_device<double>(h, (const double *)S, lds, (const double *)Y, ldy, N, (int)A, (int)M, Q, SSE);
    return sse_device<float>(h, (const float *)S, lds, (const float *)Y, ldy, N, (int)A, (int)M, Q, SSE);
}

}  // extern "C"

#include "plan_cv.hpp"


extern "C" {

int pls_hip_cv_folds(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                     int64_t K, int64_t M, int64_t A, const int64_t *test_idx, int64_t test_size,
                     int64_t num_folds, int dtype, int mem, double *E) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    if (h->reducer) return fail(h, PLS_HIP_ERR_UNSUPPORTED, "cv_folds: not available on a sharded handle");
    if (N < 2 || K < 1 || M < 1 || A < 1 || A > K || K > (1 << 30) || !X || !Y ||
        !test_idx || !E || test_size < 1 || test_size >= N || num_folds < 1 || ldx < N || ldy < N ||
        num_folds > (1 << 22) || test_size > (1 << 20))
        return fail(h, PLS_HIP_ERR_INVALID, "bad cv_folds arguments");
    if (M > plsk::LM_MAX) return fail(h, PLS_HIP_ERR_UNSUPPORTED, "more than 1024 responses not supported on the device");
    const i64 nobs = num_folds * test_size;
    for (i64 j = 0; j < nobs; ++j)
        if (test_idx[j] < 0 || test_idx[j] >= N) return fail(h, PLS_HIP_ERR_INVALID, "cv_folds: test index out of range");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const void *dX = X, *dY = Y;
    i64 dldx = ldx, dldy = ldy;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = N + ((-N) & 3);  // 16-byte columns for either type
        CHK(ensure(h, h->hX, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(h2d(h, h->hX.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        dX = h->hX.p; dY = h->hY.p;
        dldx = dldy = ldn;
    }
    CHK(ensure(h, h->cve, (size_t)nobs * A * M * 8));
    double *dE = (mem == PLS_HIP_MEM_HOST) ? (double *)h->cve.p : E;
    int rc = PLS_HIP_ERR_ALLOC;
    const bool tiny = M == 1 && plsk::tiny_fit_covers(N, (int)K, 1, (int)A, dldx, es) && !h->env.cv_refit && h->env.tiny;
    const bool tiny_m = plsk::tiny_fit_m_covers(N, (int)K, (int)M, (int)A, dldx, es) && !h->env.cv_refit && h->env.tiny;
    const bool micro = plsk::micro_fit_covers(N, (int)K, (int)M, (int)A, dldx, es) && !h->env.cv_refit && h->env.tiny;
    if (micro) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_micro<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                        (int)test_size, num_folds, dE);
        else
            rc = cv_folds_micro<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                       (int)test_size, num_folds, dE);
    } else if (tiny) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_tiny<double>(h, (const double *)dX, dldx, (const double *)dY, N, (int)K, (int)A, test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_tiny<float>(h, (const float *)dX, dldx, (const float *)dY, N, (int)K, (int)A, test_idx, (int)test_size, num_folds, dE);
    } else if (tiny_m) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_tiny_m<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                         (int)test_size, num_folds, dE);
        else
            rc = cv_folds_tiny_m<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                        (int)test_size, num_folds, dE);
    } else if (cv_batched_covers(h, K, M, A)) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_device<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A,
                                         test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_device<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A,
                                        test_idx, (int)test_size, num_folds, dE);
    }
    if (rc == PLS_HIP_ERR_ALLOC) {  // declined, or the per-fold workspaces of the batched form do not fit: one refit per fold
        h->err.clear();
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_refit<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A,
                                        test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_refit<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A,
                                       test_idx, (int)test_size, num_folds, dE);
    }
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST)
        HIPCHK(h, hipMemcpyAsync(E, dE, (size_t)nobs * A * M * 8, hipMemcpyDeviceToHost, h->stream));
    // the index list is host memory of the caller: the copy above must have consumed it before we return
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PLS_HIP_OK;
}

int pls_hip_model_sse(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                      int64_t K, int64_t M, int64_t A, const double *R, const double *Q, int dtype, int mem,
                      double *SSE) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    const bool empty_member = (N == 0 && h->nranks > 1);  // an empty shard still takes part in the reduction
    if (N < 0 || (N == 0 && !empty_member) || K < 1 || M < 1 || A < 1 || M > 1024 || A > (1 << 20) || K > (1 << 30) ||
        (N > 0 && (!X || !Y)) || !R || !Q || !SSE || ldx < std::max<i64>(N, 1) || ldy < std::max<i64>(N, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad model_sse arguments");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const i64 ldn = std::max<i64>(N, 1) + (std::max<i64>(N, 1) & 1);
    const void *dX = X, *dY = Y;
    const double *dR = R, *dQ = Q;
    double *dE = SSE;
    i64 dldx = ldx, dldy = ldy;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(ensure(h, h->hIn, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)M * A * 8));
        CHK(h2d(h, h->hIn.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        CHK(h2d(h, h->hR.p, K, R, K, K, A, 8));
        CHK(h2d(h, h->hQ.p, M, Q, M, M, A, 8));
        dX = h->hIn.p; dY = h->hY.p; dR = (const double *)h->hR.p; dQ = (const double *)h->hQ.p;
        dE = (double *)h->hB.p;
        dldx = dldy = ldn;
    }
    CHK(ensure(h, h->hOut, (size_t)ldn * A * es));  // the scores S = X R stay on the device
    int nss = 0, rc;
    if (dtype == PLS_HIP_F64) {
        if (N > 0) CHK(launch_xb<double>(h, (const double *)dX, dldx, N, (int)K, dR, K, (int)A, (double *)h->hOut.p, ldn, nullptr, &nss));
        rc = sse_device<double>(h, (const double *)h->hOut.p, ldn, (const double *)dY, dldy, N, (int)A, (int)M, dQ, dE);
    } else {
        if (N > 0) CHK(launch_xb<float>(h, (const float *)dX, dldx, N, (int)K, dR, K, (int)A, (float *)h->hOut.p, ldn, nullptr, &nss));
        rc = sse_device<float>(h, (const float *)h->hOut.p, ldn, (const float *)dY, dldy, N, (int)A, (int)M, dQ, dE);
    }
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, SSE, M, dE, M, M, A, 8));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_synth_x(pls_hip_handle h, void *X, int64_t ldx, int64_t row0, int64_t nrows, int64_t K,
                    uint64_t seed, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (nrows < 0 || row0 < 0 || K < 1 || K > (1 << 28) || (nrows > 0 && !X) || ldx < std::max<i64>(nrows, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad synth_x arguments");
    if (nrows == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    CHK(ensure(h, h->tab, (size_t)K * (plsk::SYN_F + 1) * 8));
    const uint64_t sE = plsk::mix64(seed), sZ = plsk::mix64(seed + 1), sL = plsk::mix64(seed + 2);
    const int ntab = (int)((K * plsk::SYN_F + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::synth_table_kernel, dim3(ntab), dim3(plsk::WG), 0, h->stream,
                       (double *)h->tab.p, (int)K, sL, 0, plsk::mix64(seed + 5));
    LAUNCH_CHECK(h);
    constexpr int KC = 64;
    const dim3 grid((unsigned)((nrows + plsk::WG - 1) / plsk::WG), (unsigned)((K + KC - 1) / KC));
    if (dtype == PLS_HIP_F64)
        hipLaunchKernelGGL((plsk::synth_x_kernel<double, KC>), grid, dim3(plsk::WG), 0, h->stream,
                           (double *)X, ldx, row0, nrows, (int)K, sE, sZ, (const double *)h->tab.p);
    else
        hipLaunchKernelGGL((plsk::synth_x_kernel<float, KC>), grid, dim3(plsk::WG), 0, h->stream,
                           (float *)X, ldx, row0, nrows, (int)K, sE, sZ, (const double *)h->tab.p);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

int pls_hip_synth_y(pls_hip_handle h, void *Y, int64_t ldy, int64_t row0, int64_t nrows, int64_t M,
                    uint64_t seed, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (nrows < 0 || row0 < 0 || M < 1 || M > (1 << 20) || (nrows > 0 && !Y) || ldy < std::max<i64>(nrows, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad synth_y arguments");
    if (nrows == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    // the Y table shares the workspace with the X table: generate Y before or after X, both
    // are stream-ordered
    CHK(ensure(h, h->tab, (size_t)std::max<i64>(M, 1) * plsk::SYN_F * 8));
    const uint64_t sZ = plsk::mix64(seed + 1), sC = plsk::mix64(seed + 3), sN = plsk::mix64(seed + 4);
    const int ntab = (int)((M * plsk::SYN_F + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::synth_table_kernel, dim3(ntab), dim3(plsk::WG), 0, h->stream,
                       (double *)h->tab.p, (int)M, sC, 1, (uint64_t)0);
    LAUNCH_CHECK(h);
    const dim3 grid((unsigned)((nrows + plsk::WG - 1) / plsk::WG));
    if (dtype == PLS_HIP_F64)
        hipLaunchKernelGGL((plsk::synth_y_kernel<double>), grid, dim3(plsk::WG), 0, h->stream,
                           (double *)Y, ldy, row0, nrows, (int)M, sZ, sN, (const double *)h->tab.p);
    else
        hipLaunchKernelGGL((plsk::synth_y_kernel<float>), grid, dim3(plsk::WG), 0, h->stream, (float *)Y,
                           ldy, row0, nrows, (int)M, sZ, sN, (const double *)h->tab.p);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

}  // extern "C"

#include "xchg_ipc.hpp"

#include "group_impl.hpp"
