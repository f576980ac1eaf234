// plan_cv.hpp -- cross-validation folds (src/pls.cpp:469-549): all folds in one launch, the single-launch fold kernels, one refit per fold.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

// device part of pls_hip_cv_folds on storage type T (X, Y device pointers; E device pointer)
template <typename T>
int cv_folds_device(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                    const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    const i64 K = Ki, M = Mi, A = Ai;
    const plsk::CvLayout L(Ki, Mi, Ai, ts);
    CHK(ensure(h, h->xx, (size_t)K * K * 8));
    CHK(ensure(h, h->xy, (size_t)K * M * 8));
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    CHK(ensure(h, h->cvx, (size_t)nobs * K * 8));
    CHK(ensure(h, h->cvy, (size_t)nobs * M * 8));
    CHK(ensure(h, h->cvws, (size_t)num_folds * (size_t)L.total * 8));
    double *XX = (double *)h->xx.p, *XYd = (double *)h->xy.p;
    // XX and XY of the whole matrix, once (or taken from the upload that already formed them)
    if (h->pre_xx && h->pre_xy) {
        HIPCHK(h, hipMemcpyAsync(XX, h->pre_xx, (size_t)K * K * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(XYd, h->pre_xy, (size_t)K * M * 8, hipMemcpyDeviceToDevice, h->stream));
    } else {
        CHK(compute_xx<T>(h, dX, dldx, N, Ki, XX));
    }
    if (!(h->pre_xx && h->pre_xy)) {
        CHK(ensure(h, h->part, (size_t)max_partial_rows(h, N, Ki) * (size_t)(K * M) * 8));
        CHK(ensure(h, h->red, (size_t)plsk::RED_SLICES * std::max<i64>(K * M, K + 1) * 8));
        int nb = 0;
        CHK(launch_xty<T>(h, dX, dldx, dY, dldy, N, Ki, Mi, (double *)h->part.p, &nb));
        CHK(launch_reduce(h, (const double *)h->part.p, nb, Ki * Mi, nullptr, 0, (double *)h->red.p));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((Ki * Mi + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0,
                           h->stream, (const double *)h->red.p, Ki * Mi, XYd);
        LAUNCH_CHECK(h);
    }
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((plsk::cv_gather_kernel<T>), dim3((unsigned)nobs), dim3(plsk::WG), 0, h->stream, dX, dldx, dY,
                       dldy, Ki, Mi, (const i64 *)h->cvidx.p, (double *)h->cvx.p, (double *)h->cvy.p);
    LAUNCH_CHECK(h);
    {
        Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * A * ((i64)K * K + 4 * K) * 8);
        hipLaunchKernelGGL(plsk::cv_folds_kernel, dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), (size_t)A * 8,
                           h->stream, (const double *)XX, (const double *)XYd, (const double *)h->cvx.p,
                           (const double *)h->cvy.p, Ki, Mi, Ai, ts, (double *)h->cvws.p, dE, (int)h->opt_power_iters);
        LAUNCH_CHECK(h);
    }
    return PLS_HIP_OK;
}

// Small single-response data (the reference's examples): every fold is a single-launch fit (tiny_kernels.hpp) on the whole X
// with its held-out rows masked, one workgroup per fold -- no X^T X at all, which for N < K is the smaller object anyway.
template <typename T>
int cv_folds_tiny(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 N, int Ki, int Ai, const int64_t *test_idx, int ts,
                  i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    const size_t lds = (size_t)2 * Ki * Ai * 8;
    if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_kernel<T>, (int)plsk::TINY_LDS_MAX)  /* raised once per device: to the most any fit asks for */)
        return fail(h, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");
    Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * N * Ki * (i64)sizeof(T));
    hipLaunchKernelGGL((plsk::tiny_fit_kernel<T>), dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), lds, h->stream, dX, dldx, dY,
                       (int)N, Ki, Ai, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (T *)nullptr,
                       (i64)0, (double *)nullptr, (const i64 *)h->cvidx.p, ts, nobs, dE);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

// ... the same for 2..8 responses (tiny_fit_m_kernel in fold mode)
template <typename T>
int cv_folds_tiny_m(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                    const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    const size_t lds = (size_t)(2 * Ki + Mi) * Ai * 8;
    Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * N * Ki * (i64)sizeof(T));
#define TINY_M(MM_)                                                                                                          \
    do {                                                                                                                     \
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_m_kernel<T, MM_>, (int)plsk::TINY_LDS_MAX))               \
            return fail(h, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");            \
        hipLaunchKernelGGL((plsk::tiny_fit_m_kernel<T, MM_>), dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), lds, h->stream, dX, \
                           dldx, dY, dldy, (int)N, Ki, Mi, Ai, (int)h->opt_power_iters, (double *)nullptr, (double *)nullptr,  \
                           (double *)nullptr, (double *)nullptr, (T *)nullptr, (i64)0, (double *)nullptr,                    \
                           (const i64 *)h->cvidx.p, ts, nobs, dE);                                                           \
    } while (0)
    if (Mi <= 2) TINY_M(2); else if (Mi <= 4) TINY_M(4); else TINY_M(8);
#undef TINY_M
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

// ... and the smallest data (N <= 64, K <= 32): one WAVE per fold (micro_fit_kernel)
template <typename T>
int cv_folds_micro(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                   const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * N * Ki * (i64)sizeof(T));
#define MICRO(MM_)                                                                                                           \
    hipLaunchKernelGGL((plsk::micro_fit_kernel<T, MM_>), dim3((unsigned)num_folds), dim3(plsk::WAVE), 0, h->stream, dX, dldx, dY, \
                       dldy, (int)N, Ki, Mi, Ai, (int)h->opt_power_iters, (double *)nullptr, (double *)nullptr, (double *)nullptr, \
                       (double *)nullptr, (T *)nullptr, (i64)0, (double *)nullptr, (const i64 *)h->cvidx.p, ts, nobs, dE)
    if (Mi <= 2) MICRO(2); else if (Mi <= 4) MICRO(4); else MICRO(8);
#undef MICRO
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

// The general form of the same call: one refit per fold on the rows that are not in its test set -- what the reference
// does (src/pls.cpp:478-488, :524-545), with the training rows gathered on the device and the fit running under the
// handle's own plan.  Serves the shapes the batched kernel declines (M > 32, A > 4096, K > 16384, a workspace that does
// not fit); costs num_folds fits.
template <typename T>
int cv_folds_refit(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                   const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    const i64 K = Ki, M = Mi, A = Ai;
    const i64 ldtr = (N + 3) & ~(i64)3;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    CHK(ensure(h, h->cvx, (size_t)nobs * K * 8));
    CHK(ensure(h, h->cvy, (size_t)nobs * M * 8));
    CHK(ensure(h, h->cvkeep, (size_t)N * 8));
    CHK(ensure(h, h->cvtx, (size_t)ldtr * K * sizeof(T)));
    CHK(ensure(h, h->cvty, (size_t)ldtr * M * sizeof(T)));
    CHK(ensure(h, h->cvtt, (size_t)ldtr * A * sizeof(T)));
    CHK(ensure(h, h->cvm, (size_t)(3 * K * A + M * A + (i64)ts * A) * 8));
    double *Wf = (double *)h->cvm.p, *Pf = Wf + K * A, *Rf = Pf + K * A, *Qf = Rf + K * A, *us = Qf + M * A;
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((plsk::cv_gather_kernel<T>), dim3((unsigned)nobs), dim3(plsk::WG), 0, h->stream, dX, dldx, dY,
                       dldy, Ki, Mi, (const i64 *)h->cvidx.p, (double *)h->cvx.p, (double *)h->cvy.p);
    LAUNCH_CHECK(h);
    const double *saved_xx = h->pre_xx, *saved_xy = h->pre_xy;  // products of ALL rows: not a fold's
    h->pre_xx = h->pre_xy = nullptr;
    std::vector<char> held(N, 0);
    std::vector<int64_t> keep(N);
    int rc = PLS_HIP_OK;
    for (i64 f = 0; f < num_folds && rc == PLS_HIP_OK; ++f) {
        for (int i = 0; i < ts; ++i) held[test_idx[f * ts + i]] = 1;
        i64 ntr = 0;
        for (i64 r = 0; r < N; ++r)
            if (!held[r]) keep[ntr++] = r;
        for (int i = 0; i < ts; ++i) held[test_idx[f * ts + i]] = 0;
        if (ntr < 1 || A > K) { rc = fail(h, PLS_HIP_ERR_INVALID, "cv_folds: a fold leaves no training rows"); break; }
        if (hipMemcpyAsync(h->cvkeep.p, keep.data(), (size_t)ntr * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            rc = fail(h, PLS_HIP_ERR_DEVICE, "cv_folds: upload of the training row list failed");
            break;
        }
        const unsigned gx = (unsigned)((ntr + plsk::WG - 1) / plsk::WG);
        hipLaunchKernelGGL((plsk::gather_rows_kernel<T>), dim3(gx, (unsigned)std::min<i64>(K, 1024)), dim3(plsk::WG), 0, h->stream,
                           dX, dldx, (const i64 *)h->cvkeep.p, ntr, Ki, (T *)h->cvtx.p, ldtr);
        hipLaunchKernelGGL((plsk::gather_rows_kernel<T>), dim3(gx, (unsigned)std::min<i64>(M, 1024)), dim3(plsk::WG), 0, h->stream,
                           dY, dldy, (const i64 *)h->cvkeep.p, ntr, Mi, (T *)h->cvty.p, ldtr);
        rc = fit_device<T>(h, (const T *)h->cvtx.p, ldtr, (const T *)h->cvty.p, ldtr, ntr, Ki, Mi, Ai, PLS_HIP_KERNEL_TYPE1,
                           Wf, Pf, Qf, Rf, (T *)h->cvtt.p, ldtr, nullptr);
        if (rc != PLS_HIP_OK) break;
        hipLaunchKernelGGL(plsk::cv_refit_residuals_kernel, dim3((unsigned)ts), dim3(plsk::WG), 0, h->stream,
                           (const double *)h->cvx.p + f * ts * K, (const double *)h->cvy.p + f * ts * M, (const double *)Rf,
                           (const double *)Qf, Ki, Mi, Ai, ts, f, nobs, us, dE);
        // `keep` is rewritten for the next fold: its copy must have been consumed
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            rc = fail(h, PLS_HIP_ERR_DEVICE, "cv_folds: a fold's refit failed on the device");
    }
    h->pre_xx = saved_xx;
    h->pre_xy = saved_xy;
    return rc;
}

// the batched kernel's shapes (cv_kernels.hpp): everything M-sized in one workgroup's LDS, X^T X resident
bool cv_batched_covers(const pls_hip_context *c, i64 K, i64 M, i64 A) {
    return !c->env.cv_refit && A <= 4096 && K <= 16384 && (M == 1 || M <= plsk::MMAX);
}

}  // namespace
