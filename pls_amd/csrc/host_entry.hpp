// host_entry.hpp -- handle checks, fit timing brackets, host <-> device staging and the upload that accumulates X^T X / X^T Y on the way.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

int check_handle(pls_hip_handle h) { return h ? PLS_HIP_OK : PLS_HIP_ERR_INVALID; }

int set_device(pls_hip_context *c) {
    HIPCHK(c, hipSetDevice(c->device));
    return PLS_HIP_OK;
}

void begin_fit_timing(pls_hip_context *c) {
    c->fit_timed = false;
    if (!c->opt_profile) return;
    c->cur_fit.e0 = take_event(c);
    c->cur_fit.e1 = take_event(c);
    if (c->cur_fit.e0 && c->cur_fit.e1) {
        (void)hipEventRecord(c->cur_fit.e0, c->stream);
        c->fit_timed = true;
    }
}
void end_fit_timing(pls_hip_context *c) {
    if (!c->fit_timed) return;
    (void)hipEventRecord(c->cur_fit.e1, c->stream);
    c->fits.push_back(c->cur_fit);
    c->fit_timed = false;
}

// host <-> device staging of a column-major matrix with leading dimension.  Matrices of a few MB and more go
// through the pinned double-buffer pipeline of host_pipeline.hpp (the caller's pageable memory is repacked by host
// threads while the DMA engine moves the previous tile); small ones are one plain copy.
constexpr size_t PIPELINE_MIN_BYTES = (size_t)4 << 20;
int h2d(pls_hip_context *c, void *dst, i64 ldd, const void *src, i64 lds, i64 rows, i64 cols, size_t es) {
    if (rows == 0 || cols == 0) return PLS_HIP_OK;
    if ((size_t)rows * (size_t)cols * es >= PIPELINE_MIN_BYTES) {
        HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
        HIPCHK(c, plsh::upload(c->stager, c->stream, dst, ldd, src, lds, rows, cols, es));
        return PLS_HIP_OK;
    }
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ldd * es, src, (size_t)lds * es, (size_t)rows * es,
                               (size_t)cols, hipMemcpyHostToDevice, c->stream));
    return PLS_HIP_OK;
}
int d2h(pls_hip_context *c, void *dst, i64 ldd, const void *src, i64 lds, i64 rows, i64 cols, size_t es) {
    if (rows == 0 || cols == 0) return PLS_HIP_OK;
    if ((size_t)rows * (size_t)cols * es >= PIPELINE_MIN_BYTES) {
        HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
        HIPCHK(c, plsh::download(c->stager, c->stream, dst, ldd, src, lds, rows, cols, es));
        return PLS_HIP_OK;
    }
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ldd * es, src, (size_t)lds * es, (size_t)rows * es,
                               (size_t)cols, hipMemcpyDeviceToHost, c->stream));
    return PLS_HIP_OK;
}

// Upload of X (host, rows x K) in ROW blocks with X^T X and X^T Y accumulated block by block on the compute stream
// while the DMA engine already moves the next block: the matrix-core SYRK of a block (2 rb K^2 flops) takes a
// fraction of the block's PCIe time (K * 2e-4 of it), so by the time the last rows have arrived the Gram matrix of
// the whole shard is complete and a Gram-plan fit needs no further pass over X for its component loop.
// dY: the member's rows of Y, already on the device.  *ok = false: the layout does not allow it (plain upload done).
template <typename T>
int upload_accumulate(pls_hip_context *c, T *dX, i64 ldd, const T *hX, i64 ldx, i64 N, int K, const T *dY, i64 ldy,
                      int M, double *XXacc, double *XYacc, bool *ok) {
    constexpr int FV = 16 / sizeof(T);
    const size_t es = sizeof(T);
    *ok = false;
    const i64 KK = (i64)K * K, L0 = (i64)K * M;
    i64 rb = (i64)(plsh::STAGE_BYTES / ((size_t)K * es)) & ~(i64)63;
    const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
    const i64 S = std::max<i64>(1, (2 * (i64)c->num_cu) / (nbk * (nbk + 1) / 2));
    if (N < 1 || K > 4096 || M > plsk::LM_MAX || rb < 64 || !vec_ok<T>(dX, ldd, FV) || !vec_ok<T>(dY, ldy, FV) ||
        ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8) != PLS_HIP_OK ||
        ensure(c, c->part, std::max<size_t>((size_t)S * KK, (size_t)max_partial_rows(c, rb, K) * L0) * 8) != PLS_HIP_OK ||
        ensure(c, c->red, (size_t)plsk::RED_SLICES * std::max<i64>(L0, K + 1) * 8) != PLS_HIP_OK) {
        c->err.clear();
        return h2d(c, dX, ldd, hX, ldx, N, K, es);
    }
    HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->zeros.p) {
        CHK(ensure(c, c->zeros, 256));
        HIPCHK(c, hipMemsetAsync(c->zeros.p, 0, 256, c->stream));
    }
    HIPCHK(c, hipMemsetAsync(XXacc, 0, (size_t)KK * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(XYacc, 0, (size_t)L0 * 8, c->stream));
    plsh::Stager &st = c->stager;
    double *part = (double *)c->part.p, *red2 = (double *)c->red2.p, *red = (double *)c->red.p;
    bool acc = true;
    Range r_up("upload + X^T X / X^T Y accumulation");
    for (i64 r0 = 0; r0 < N; r0 += rb) {
        const i64 rbn = std::min(rb, N - r0);
        const int s = st.slot;
        if (st.busy[s]) HIPCHK(c, hipEventSynchronize(st.ev[s]));
        plsh::repack(*st.pool, (char *)st.buf[s], (char *)const_cast<T *>(hX), ldx, r0, 0, rbn, K, es, true);
        HIPCHK(c, hipMemcpy2DAsync(dX + r0, (size_t)ldd * es, st.buf[s], (size_t)rbn * es, (size_t)rbn * es, (size_t)K,
                                   hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipEventRecord(st.ev[s], c->copy_stream));
        st.busy[s] = true;
        st.slot ^= 1;
        HIPCHK(c, hipStreamWaitEvent(c->stream, st.ev[s], 0));  // kernels of this block (and of the fit) after its rows
        if (!acc) continue;
        int nb = 0;
        if (plsk::launch_syrk<T>(c->stream, c->num_cu, dX + r0, ldd, rbn, K, part, S * KK, &nb, c->zeros.p) != 0) {
            acc = false;  // ragged block the matrix-core kernel declines: the fit forms X^T X itself
            continue;
        }
        LAUNCH_CHECK(c);
        CHK(launch_reduce(c, part, nb, (int)KK, nullptr, 0, red2));
        hipLaunchKernelGGL(plsk::accumulate_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, (const double *)red2, (int)KK, XXacc);
        LAUNCH_CHECK(c);
        CHK(launch_xty<T>(c, dX + r0, ldd, dY + r0, ldy, rbn, K, M, part, &nb));
        CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
        hipLaunchKernelGGL(plsk::accumulate_slices_kernel, dim3((unsigned)((L0 + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, (const double *)red, (int)L0, XYacc);
        LAUNCH_CHECK(c);
    }
    *ok = acc;
    return PLS_HIP_OK;
}

}  // namespace
