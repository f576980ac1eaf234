// plan_stats.hpp -- column statistics / z-scores (src/pls.cpp:69-111) and SSE for every component count (:457-467, :551-562) on the device.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

// column mean / sd / z-scores on the device (two column-reduction passes + one scale pass)
template <typename T>
int zscores_device(pls_hip_context *c, const T *X, i64 ldx, i64 N, i64 n_total, int K, T *Z, i64 ldz,
                   double *mean, double *sd) {
    constexpr int FV = 16 / sizeof(T);
    constexpr int KC = 16;
    const bool wide = vec_ok<T>(X, ldx, FV) && (!Z || vec_ok<T>(Z, ldz, FV));
    const int vec = wide ? FV : 1;
    const int nkg = (K + KC - 1) / KC;
    const i64 nch = std::max<i64>(1, (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec));
    const int G = (int)std::min<i64>(nch, std::max<i64>(1, (8 * c->num_cu) / nkg));
    CHK(ensure(c, c->part, (size_t)G * K * 8));
    CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * K * 8));
    double *part = (double *)c->part.p, *red = (double *)c->red2.p;
    const dim3 grid(G, nkg), blk(plsk::WG);
    // Mean and sd from ONE sweep (colmoments_kernel: per-wave shifted sums merged pairwise) -- two sweeps of X
    // (statistics, scale) for the z-scores instead of the reference's three.  Row-sharded: the shards' (count, mean, M2)
    // triples meet in two all-reduces of K sums (colmoments_shard_kernel), as many as the two-pass form needs.  Every
    // rank takes this branch or none (the environment decides, not the shard), an empty shard contributes zeros.
    const bool shifted = c->reducer || (N > 0 && n_total == N);
    if (shifted) {
        // all workgroups resident at once (5 per CU at 84 VGPRs): one round, no tail
        const int G1 = (int)std::min<i64>(nch, std::max<i64>(1, (5 * c->num_cu) / nkg));
        const bool sharded = c->reducer != nullptr;
        // (the shards' sums travel in the layout every collective of the library has -- RED_SLICES slices of K values, the
        // values in slice 0, zeros behind: the device-side exchanges sum the slices of a message into slice 0)
        CHK(ensure(c, c->part, (size_t)(G1 * (3 * (i64)K + 1) + (sharded ? (3 + (i64)plsk::RED_SLICES) * K + 1 : 0)) * 8));
        part = (double *)c->part.p;
        double *cnt = part + (i64)G1 * 3 * K;
        double *tri = sharded ? cnt + G1 : nullptr, *buf = sharded ? tri + 3 * (i64)K + 1 : nullptr;
        if (sharded) HIPCHK(c, hipMemsetAsync(buf + K, 0, (size_t)(plsk::RED_SLICES - 1) * K * 8, c->stream));
        const dim3 gk((K + plsk::WG - 1) / plsk::WG);
        if (N > 0) {
            {
                Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + 2 * (i64)K * 8);
                const dim3 g1(G1, nkg);
                if (wide) hipLaunchKernelGGL((plsk::colmoments_kernel<T, FV, KC>), g1, blk, 0, c->stream, X, ldx, N, K, part, cnt);
                else hipLaunchKernelGGL((plsk::colmoments_kernel<T, 1, KC>), g1, blk, 0, c->stream, X, ldx, N, K, part, cnt);
                LAUNCH_CHECK(c);
            }
            hipLaunchKernelGGL(plsk::colmoments_finish_kernel, gk, blk, 0, c->stream, (const double *)part, (const double *)cnt, G1, K,
                               mean, sd, tri);
            LAUNCH_CHECK(c);
        } else {
            HIPCHK(c, hipMemsetAsync(tri, 0, (size_t)(3 * (i64)K + 1) * 8, c->stream));
        }
        if (sharded) {
            for (int step = 0; step < 3; ++step) {
                hipLaunchKernelGGL(plsk::colmoments_shard_kernel, gk, blk, 0, c->stream, (const double *)tri, K, (double)n_total, step,
                                   buf, mean, sd);
                LAUNCH_CHECK(c);
                if (step < 2) CHK(do_allreduce(c, buf, (i64)plsk::RED_SLICES * K));
            }
        }
    }
    for (int mode = shifted ? 2 : 0; mode < 2; ++mode) {
        if (N > 0) {
            Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + (i64)K * 8);
#define CS_CASE(V_, M_) hipLaunchKernelGGL((plsk::colstat_kernel<T, V_, KC, M_>), grid, blk, 0, c->stream, X, ldx, N, K, mean, part)
            if (wide) { if (mode == 0) CS_CASE(FV, 0); else CS_CASE(FV, 1); }
            else { if (mode == 0) CS_CASE(1, 0); else CS_CASE(1, 1); }
#undef CS_CASE
            LAUNCH_CHECK(c);
            CHK(launch_reduce(c, part, G, K, nullptr, 0, red));
        } else {
            HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * K * 8, c->stream));
        }
        CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * K));
        hipLaunchKernelGGL(plsk::colstat_finish_kernel, dim3((K + plsk::WG - 1) / plsk::WG), blk, 0, c->stream,
                           (const double *)red, K, (double)n_total, mode, mode == 0 ? mean : sd);
        LAUNCH_CHECK(c);
    }
    if (Z && N > 0) {
        Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T) + 2 * (i64)K * 8);
        const dim3 g2((unsigned)std::min<i64>(nch, std::max<i64>(1, (16 * c->num_cu) / nkg)), nkg);
        if (wide && K <= 65535)
            hipLaunchKernelGGL((plsk::zscale_piece_kernel<T, FV>), dim3((unsigned)nch, K), blk, 0, c->stream, X, ldx, Z, ldz, N, mean, sd);
        else if (wide) hipLaunchKernelGGL((plsk::zscale_kernel<T, FV, KC>), g2, blk, 0, c->stream, X, ldx, Z, ldz, N, K, mean, sd);
        else hipLaunchKernelGGL((plsk::zscale_kernel<T, 1, KC>), g2, blk, 0, c->stream, X, ldx, Z, ldz, N, K, mean, sd);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

template <typename T>
int sse_device(pls_hip_context *c, const T *S, i64 lds, const T *Y, i64 ldy, i64 N, int A, int M,
               const double *Q, double *SSE) {
    // ranges of component counts with at most 1024 running sums each (the sweep keeps them per wave in LDS)
    const int step = std::max(1, 1024 / M);
    const int G = (int)std::min<i64>(std::max<i64>(1, (N + plsk::WG - 1) / plsk::WG), 4 * (i64)c->num_cu);
    for (int c_lo = 0; c_lo < A; c_lo += step) {
        const int c_hi = std::min(A, c_lo + step), AM = (c_hi - c_lo) * M;
        CHK(ensure(c, c->part, (size_t)G * AM * 8));
        CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * AM * 8));
        {
            Scope s(c, PLS_HIP_FAM_XB, (i64)N * (c_hi + M) * sizeof(T) + (i64)AM * 8);
            hipLaunchKernelGGL((plsk::sse_components_kernel<T>), dim3(G), dim3(plsk::WG),
                               (size_t)(plsk::WG / plsk::WAVE) * AM * 8, c->stream, S, lds, Y, ldy, N, c_lo, c_hi, M, Q,
                               (double *)c->part.p);
            LAUNCH_CHECK(c);
        }
        CHK(launch_reduce(c, (const double *)c->part.p, G, AM, nullptr, 0, (double *)c->red2.p));
        CHK(do_allreduce(c, (double *)c->red2.p, (i64)plsk::RED_SLICES * AM));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((AM + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0, c->stream,
                           (const double *)c->red2.p, AM, SSE + (i64)c_lo * M);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

}  // namespace
