// launch_update.hpp -- the K-sized component update (src/pls.cpp:403-416, :427-433): which kernel family a shape takes, and its launches.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

// n*K (values of P and of R the r update must read) above which it is split over many workgroups
constexpr i64 ROTATE_SPLIT_MIN = 16384;

// The component update for M > 32 responses (largem_kernels.hpp): M-sized data in global memory, plain multi-workgroup
// kernels, the eigenvector by `power_iters` squarings without early exit.  Scratch in c->lm.
int launch_update_large(pls_hip_context *c, const double *red, double *XY, double *W, double *P, double *Q, double *R,
                        double *v, int K, int M, int A, int a, int nip) {
    const i64 MM = (i64)M * M;
    const int nparts = (K + plsk::WG - 1) / plsk::WG;
    const i64 prows = max_partial_rows(c, K, M);
    const size_t need = (size_t)(3 * MM + 2 * M + nparts + 8 + K + prows * MM + (i64)plsk::RED_SLICES * MM) * 8;
    CHK(ensure(c, c->lm, need));
    double *G = (double *)c->lm.p, *Bm = G + MM, *Cm = Bm + MM, *qe = Cm + MM, *qv = qe + M, *ssp = qv + M;
    double *tr = ssp + nparts, *wraw = tr + 8, *xpart = wraw + K, *xred = xpart + prows * MM;
    const dim3 blk(plsk::WG);
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 4 + MM * 3 * c->opt_power_iters) * 8);
    if (a >= 0) {
        hipLaunchKernelGGL(plsk::lm_pq_kernel, dim3(M + nparts), blk, 0, c->stream, red, (const double *)XY, (const double *)R, P,
                           Q, qv, K, M, a);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::lm_deflate_kernel, dim3((unsigned)(((i64)K * M + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           red, XY, (const double *)P, (const double *)qv, K, M, a);
        LAUNCH_CHECK(c);
    } else {
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)(((i64)K * M + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           red, K * M, XY);
        LAUNCH_CHECK(c);
    }
    const int n = a + 1;
    if (n >= A) return PLS_HIP_OK;
    {  // G = XY^T XY (:405) with the column-reduction kernels: "X" = XY (K rows, M columns), "Y" = XY
        int nb = 0;
        CHK(launch_xty<double>(c, XY, K, XY, K, K, M, M, xpart, &nb));
        CHK(launch_reduce(c, xpart, nb, (int)MM, nullptr, 0, xred));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((MM + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           (const double *)xred, (int)MM, G);
        LAUNCH_CHECK(c);
    }
    if (M <= plsk::MMAX) {
        // up to 32 responses: the whole direction solve in one workgroup's LDS, one launch
        hipLaunchKernelGGL(plsk::lm_eig_lds_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, (const double *)G, M,
                           (int)c->opt_power_iters, qe);
        LAUNCH_CHECK(c);
    } else if (M <= plsk::LM_LDS_MAX &&
               plsk::raise_dynamic_lds(reinterpret_cast<const void *>(&plsk::lm_eig_lds_big_kernel), (2 * M * M + M) * 8)) {
        // up to 48 responses: still one launch, B and C in LDS (the fixed point ends the squarings; beyond, one CU is too slow)
        hipLaunchKernelGGL(plsk::lm_eig_lds_big_kernel, dim3(1), dim3(plsk::UPD_THREADS), (size_t)(2 * M * M + M) * 8, c->stream,
                           (const double *)G, M, (int)c->opt_power_iters, qe);
        LAUNCH_CHECK(c);
    } else {
        // dominant eigenvector by repeated squaring, X_0 = G, X_{j+1} = (X_j / tr X_j)^2 -> a multiple of v1 v1^T: one launch
        // per squaring (the trace is formed inside; the finish kernel takes any scale).  The host cannot see the fixed point,
        // so all power_iters squarings run.
        const dim3 sq((M + 15) / 16, (M + 15) / 16), sqb(16, 16);
        const double *src = G;
        double *dst = Bm;
        for (int it = 0; it < std::max(1, (int)c->opt_power_iters); ++it) {
            hipLaunchKernelGGL(plsk::lm_square_normalised_kernel, sq, sqb, 0, c->stream, src, M, dst);
            src = dst;
            dst = (dst == Bm) ? Cm : Bm;
        }
        LAUNCH_CHECK(c);
        const double *Bfin = src;
        hipLaunchKernelGGL(plsk::lm_eig_finish_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, (const double *)G,
                           Bfin, M, qe);
        LAUNCH_CHECK(c);
    }
    hipLaunchKernelGGL(plsk::lm_w_kernel, dim3(nparts), blk, 0, c->stream, (const double *)XY, (const double *)qe, K, M, wraw, ssp);
    LAUNCH_CHECK(c);
    double *wn = W + (i64)n * K;
    hipLaunchKernelGGL(plsk::lm_normalize_kernel, dim3(nparts), blk, 0, c->stream, (const double *)wraw, (const double *)ssp,
                       nparts, K, wn, n == 0 ? R : (double *)nullptr, n == 0 ? v : (double *)nullptr);
    LAUNCH_CHECK(c);
    if (n > 0) {  // r = w - sum_j (p_j^T w) r_j (:412-416)
        double *cs = (double *)c->cs.p;
        hipLaunchKernelGGL(plsk::rotate_dots_kernel, dim3(n), blk, 0, c->stream, P, W, K, n, cs);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::rotate_apply_kernel, dim3(nparts), blk, 0, c->stream, W, R, cs, v, K, n, nip);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

// columns from which the one-response update runs on many workgroups (up to 4096 the one-workgroup kernel keeps XY in
// registers), and from which 2..8 responses leave the cooperative kernel's reach (wide1_update.hpp)
constexpr int WIDE1_MIN = 4097;
constexpr int WIDEM_MIN = plsk::COOP_MAXG * plsk::COOP_WG + 1;

// Will launch_update run the ONE-workgroup kernel for this shape (the form that can take the gather of a sharded fit's
// collective as its prologue)?  The conditions of the branches in launch_update, in their order.
bool update_is_single(int K, int M, int A, int a) {
    int g = 0, e = 0;
    if (M > plsk::MMAX || (M > 8 && (i64)K * M >= 16384)) return false;
    if (M == 1 && K >= WIDE1_MIN && plsk::wide1_geometry(K, &g, &e)) return false;
    if (M >= 2 && M <= 8 && K >= WIDEM_MIN && (i64)K * M >= 16384 && A <= 4096 && plsk::wide1_geometry(K, &g, &e)) return false;
    if (plsk::coop_update_covers(K, M) && A <= 4096) return false;
    const int n = a + 1;
    return !(n < A && n > 0 && (i64)n * K >= ROTATE_SPLIT_MIN);  // (the r recurrence on several workgroups: two more launches)
}

// nip: 0 = KERNEL algo (next pass is X r), 1 = NIPALS (next pass X_a w)
// gx (only where update_is_single says yes): the gather of the component's collective as the kernel's prologue; red is
// then written (slice 0) instead of read
int launch_update(pls_hip_context *c, double *red, double *XY, double *W, double *P,
                  double *Q, double *R, double *v, int K, int M, int A, int a, int nip, const plsk::XchgGather *gx = nullptr) {
    const int n = a + 1;
    // 9 <= M <= 32 responses on many columns: the one-workgroup kernel walks K x M values several times and forms the
    // M (M + 1) / 2 Gram entries one wave per pair (208 us per component at K = 4096, M = 16; 630 us at M = 32 -- more
    // than the 0.32 ms pass); the multi-workgroup kernels of the many-response path with the LDS eigen solve: ~12 launches
    if (M > plsk::MMAX || (M > 8 && (i64)K * M >= 16384))
        return launch_update_large(c, (const double *)red, XY, W, P, Q, R, v, K, M, A, a, nip);
    // One response on very many columns: element-wise work and K-long sums on up to 128 workgroups, two launches
    // (wide1_update.hpp) instead of one workgroup walking K (+ one workgroup per p_j^T w of the r recurrence)
    int w1g = 0, w1e = 0;
    if (M == 1 && K >= WIDE1_MIN && plsk::wide1_geometry(K, &w1g, &w1e)) {
        CHK(ensure(c, c->wide1, (size_t)((i64)(A + 2) * w1g + A + 1) * 8));
        double *w1part = (double *)c->wide1.p, *w1q = w1part + (i64)(A + 1) * w1g, *w1tot = w1q + w1g;
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
        hipLaunchKernelGGL(plsk::wide1_a_kernel, dim3(w1g), dim3(plsk::W1_WG), (size_t)plsk::W1_WG * w1e * 8, c->stream, red, XY, P, Q,
                           K, A, a, w1e, (const double *)w1q, w1part);
        LAUNCH_CHECK(c);
        if (n < A && n + 1 <= 4096) {
            hipLaunchKernelGGL(plsk::wide1_b_kernel<false>, dim3(w1g), dim3(plsk::W1_WG), (size_t)(n + 1) * 8, c->stream,
                               (const double *)XY, W, R, v, K, n, w1e, nip, (const double *)w1part, w1q, (const double *)nullptr);
            LAUNCH_CHECK(c);
        } else if (n < A) {  // more totals than a workgroup's LDS holds: formed once, read from global memory
            hipLaunchKernelGGL(plsk::wide1_totals_kernel, dim3((n + 4) / 4), dim3(plsk::W1_WG), 0, c->stream, (const double *)w1part, n,
                               w1g, w1tot);
            hipLaunchKernelGGL(plsk::wide1_b_kernel<true>, dim3(w1g), dim3(plsk::W1_WG), 0, c->stream, (const double *)XY, W, R, v, K,
                               n, w1e, nip, (const double *)w1part, w1q, (const double *)w1tot);
            LAUNCH_CHECK(c);
        }
        return PLS_HIP_OK;
    }
    // 2..8 responses beyond the cooperative kernel's 16,384 columns: the same arithmetic cut at its two exchanges, three launches
    if (M >= 2 && M <= 8 && K >= WIDEM_MIN && (i64)K * M >= 16384 && A <= 4096 && plsk::wide1_geometry(K, &w1g, &w1e)) {
        CHK(ensure(c, c->wide1, (size_t)((i64)(A + plsk::WM_GSTRIDE + plsk::WM_QSTRIDE) * w1g) * 8));
        double *gp = (double *)c->wide1.p, *cp = gp + (i64)plsk::WM_GSTRIDE * w1g, *qp = cp + (i64)A * w1g;
        const dim3 g(w1g), b(plsk::W1_WG);
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
#define WM_CASE(MM_)                                                                                                              \
    do {                                                                                                                          \
        hipLaunchKernelGGL((plsk::widem_a_kernel<MM_>), g, b, 0, c->stream, red, XY, P, Q, K, M, A, a, w1e, (const double *)qp, gp);  \
        if (n < A) {                                                                                                              \
            hipLaunchKernelGGL((plsk::widem_b_kernel<MM_>), g, b, (size_t)plsk::W1_WG * w1e * 8, c->stream, (const double *)XY, W,      \
                               (const double *)P, K, M, n, w1e, (int)c->opt_power_iters, (const double *)gp, cp);                  \
            hipLaunchKernelGGL((plsk::widem_c_kernel<MM_>), g, b, (size_t)(n + 1) * 8, c->stream, (const double *)XY,                  \
                               (const double *)W, R, v, K, M, n, w1e, nip, (const double *)cp, qp);                                \
        }                                                                                                                         \
    } while (0)
        if (M <= 2) WM_CASE(2); else if (M <= 4) WM_CASE(4); else WM_CASE(8);
#undef WM_CASE
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    if (plsk::coop_update_covers(K, M) && A <= 4096) {
        // several workgroups, two in-launch exchanges, r included (coop_update.hpp): one launch per component
        const size_t need = plsk::coop_scratch_bytes(A);
        if (c->coop.bytes < need) {  // the exchange counters start from zero
            CHK(ensure(c, c->coop, need));
            HIPCHK(c, hipMemsetAsync(c->coop.p, 0, c->coop.bytes, c->stream));
        }
        unsigned *cnt = (unsigned *)c->coop.p;
        // every fit starts from zeroed exchange counters, whatever an earlier (failed) fit left behind
        if (a < 0) HIPCHK(c, hipMemsetAsync(cnt, 0, 256, c->stream));
        double *qraw = (double *)((char *)c->coop.p + 256), *gpart = qraw + plsk::COOP_MAXG * plsk::COOP_QSTRIDE;
        double *cpart = gpart + plsk::COOP_MAXG * plsk::COOP_GSTRIDE;
        const dim3 grid((K + plsk::COOP_WG - 1) / plsk::COOP_WG), blk(plsk::COOP_WG);
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
#define COOP_CASE(MM_) hipLaunchKernelGGL((plsk::coop_update_kernel<MM_>), grid, blk, (size_t)A * sizeof(double), c->stream, \
                                          red, XY, W, P, Q, R, v, K, M, A, a, nip, (int)c->opt_power_iters, cnt, qraw, gpart, cpart)
        if (M <= 2) COOP_CASE(2); else if (M <= 4) COOP_CASE(4); else COOP_CASE(8);
#undef COOP_CASE
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // (the in-kernel r recurrence stages its p_j^T w products in min(A, 4096) doubles of LDS: beyond 4096 components
    // K > 4096 as well, so the multi-workgroup form below takes over from the fourth component on)
    const bool split = n < A && n > 0 && (i64)n * K >= ROTATE_SPLIT_MIN;
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
    if (gx) {
        if (split) return fail(c, PLS_HIP_ERR_DEVICE, "internal: gather prologue on a split update");
        hipLaunchKernelGGL(plsk::component_update_gather_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                           (size_t)std::min(A, 4096) * sizeof(double), c->stream, *gx, red, XY, W, P, Q, R, v, K, M, A, a, nip,
                           (int)c->opt_power_iters, 0);
    } else {
        hipLaunchKernelGGL(plsk::component_update_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                           (size_t)std::min(A, 4096) * sizeof(double), c->stream, (const double *)red, XY, W, P, Q, R, v, K, M, A, a, nip,
                           (int)c->opt_power_iters, (int)split);
    }
    LAUNCH_CHECK(c);
    if (split) {
        double *cs = (double *)c->cs.p;
        hipLaunchKernelGGL(plsk::rotate_dots_kernel, dim3(n), dim3(plsk::WG), 0, c->stream, P, W, K, n, cs);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::rotate_apply_kernel, dim3((K + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0,
                           c->stream, W, R, cs, v, K, n, nip);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

}  // namespace
