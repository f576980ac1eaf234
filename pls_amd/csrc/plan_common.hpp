// plan_common.hpp -- what every plan shares: the sliced collective, X^T X (matrix-core SYRK or column blocks), the replica guard of sharded fits.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

// Every collective of the library is a SLICED message: RED_SLICES slices of count / RED_SLICES values each, whose
// consumers add the slices in index order (an element-wise all-reduce keeps that layout; the device-side exchanges
// leave the total in slice 0 and zeros behind it) -- include/pls_hip.h, pls_hip_allreduce_fn.
int do_allreduce(pls_hip_context *c, double *buf, i64 count) {
    if (!c->reducer) return PLS_HIP_OK;  // an installed reducer is called even for one rank
    if (count % plsk::RED_SLICES != 0) return fail(c, PLS_HIP_ERR_REDUCER, "internal: collective of " + std::to_string(count) + " values is not sliced");
    const int rc = c->reducer(c->reducer_user, buf, count, (void *)c->stream);
    if (rc != 0) return fail(c, PLS_HIP_ERR_REDUCER, "all-reduce callback returned " + std::to_string(rc));
    return PLS_HIP_OK;
}

// XX(K x K, fp64) = X^T X summed over ranks: matrix-core SYRK when the layout allows it, otherwise the
// column-reduction kernel in 32-column blocks.  Uses c->part / c->red2 as scratch.
// Whichever kernels the LOCAL shard takes (its row count, alignment and leading dimension decide, and an
// empty shard runs none), the exchange is always ONE all-reduce of RED_SLICES*K*K values in the same layout:
// the ranks of a sharded fit can never disagree on the sequence of collectives.
// compute_xx_local: this rank's X^T X into the slices of c->red2 (no collective).  With Y and xy_red given, the matrix-core
// path forms X^T Y in the same sweep (its diagonal workgroups, syrk_kernels.hpp) and leaves it, reduced into slices, in
// xy_red; *xy_done says whether it did (the caller runs the separate X^T Y kernel otherwise).
// compute_xx_finish: the one all-reduce of the slices and their sum -> XX.
template <typename T>
int compute_xx_local(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const T *Y = nullptr, i64 ldy = 0, int M = 0,
                     double *xy_red = nullptr, bool *xy_done = nullptr) {
    constexpr int CB = 32;
    const i64 KK = (i64)K * K;
    if (xy_done) *xy_done = false;
    CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8));
    double *red2 = (double *)c->red2.p;
    bool have = false;
    if (N > 0) {
        // matrix-core path: 128 x 128 blocks on v_mfma_f64_16x16x4_f64, row-split partial blocks
        const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
        const i64 S = std::max<i64>(1, (16 * (i64)c->num_cu + nbk * (nbk + 1) / 2 - 1) / (nbk * (nbk + 1) / 2));  // capacity bound
        if (ensure(c, c->part, (size_t)S * KK * 8) == PLS_HIP_OK) {
            double *part = (double *)c->part.p;
            int nb = 0, nb_xy = 0;
            int rc;
            {
                Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) * ((nbk + 1)) + KK * 8);
                if (!c->zeros.p) {  // source of out-of-range rows for the LDS-DMA panels
                    CHK(ensure(c, c->zeros, 256));
                    HIPCHK(c, hipMemsetAsync(c->zeros.p, 0, 256, c->stream));
                }
                // partial rows of X^T Y = row splits of a diagonal block: up to 512 when the blocks are few (K <= 128: every workgroup
                // of the launch is a diagonal one), up to 8 x 512 / 45 = 92 with eight waves of workgroups; launch_syrk checks
                const i64 xycap = (K <= 512 ? 512 : 128) * (i64)K * std::max(M, 1);
                const bool want_xy = Y && xy_red && M >= 1 && M <= 8 &&
                                     ensure(c, c->xyp, (size_t)xycap * 8) == PLS_HIP_OK;
                if (!want_xy) c->err.clear();
                rc = plsk::launch_syrk<T>(c->stream, c->num_cu, X, ldx, N, K, part, S * KK, &nb,
                                          c->zeros.p, want_xy ? Y : nullptr, ldy, M,
                                          want_xy ? (double *)c->xyp.p : nullptr, xycap, want_xy ? &nb_xy : nullptr);
                if (rc != 0) s.on = false;
            }
            if (rc == 0) {
                LAUNCH_CHECK(c);
                CHK(launch_reduce(c, part, nb, (int)KK, nullptr, 0, red2));
                have = true;
                if (nb_xy > 0) {
                    CHK(launch_reduce(c, (const double *)c->xyp.p, nb_xy, K * M, nullptr, 0, xy_red));
                    *xy_done = true;
                }
            }
        } else {
            c->err.clear();  // no room for the partial blocks: the column-block path needs far less
        }
        if (!have) {  // unaligned layouts: column blocks of X^T X through the column-reduction kernel, into the same slices
            const i64 prow = max_partial_rows(c, N, K);
            CHK(ensure(c, c->part, (size_t)prow * (size_t)K * CB * 8));
            double *part = (double *)c->part.p;
            for (int c0 = 0; c0 < K; c0 += CB) {
                const int cb = std::min(CB, K - c0);
                int nb = 0;
                CHK(launch_xty<T>(c, X, ldx, X + (i64)c0 * ldx, ldx, N, K, cb, part, &nb));
                CHK(launch_reduce(c, part, nb, K * cb, nullptr, 0, red2 + (i64)c0 * K, KK));
            }
        }
    } else {
        HIPCHK(c, hipMemsetAsync(red2, 0, (size_t)plsk::RED_SLICES * KK * 8, c->stream));
    }
    return PLS_HIP_OK;
}

int compute_xx_finish(pls_hip_context *c, int K, double *XX) {
    const i64 KK = (i64)K * K;
    double *red2 = (double *)c->red2.p;
    CHK(do_allreduce(c, red2, (i64)plsk::RED_SLICES * KK));
    hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                       c->stream, (const double *)red2, (int)KK, XX);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

template <typename T>
int compute_xx(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, double *XX) {
    CHK(compute_xx_local<T>(c, X, ldx, N, K));
    return compute_xx_finish(c, K, XX);
}

// Sharded fits: every rank must have derived the same bits (small_kernels.hpp, "replica guard").  Two small launches and one
// 512-byte all-reduce per fit; the verdict lands in a host-mapped flag that pls_hip_synchronize (and the host-memory entry)
// turn into PLS_HIP_ERR_REDUCER.
// the handle's host-mapped status words: [0] the replicas of a sharded fit diverged, [1] a wait of the resident fit ran out
bool host_flags(pls_hip_context *c) {
    if (c->diverged) return true;
    if (hipHostMalloc((void **)&c->diverged, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->diverged_dev, c->diverged, 0) != hipSuccess) {
        (void)hipGetLastError();
        c->diverged = nullptr;
        return false;
    }
    c->diverged[0] = c->diverged[1] = 0;
    return true;
}

int replica_guard(pls_hip_context *c, const double *W, const double *P, const double *Q, const double *R, const double *B,
                  int K, int M, int A) {
    if (!c->reducer || c->nranks < 2 || c->nranks > 1024) return PLS_HIP_OK;
    if (!c->env.replica_guard) return PLS_HIP_OK;
    if (!host_flags(c)) return PLS_HIP_OK;  // (no mapped host memory: the guard is an extra, not a precondition)
    CHK(ensure(c, c->guard, (size_t)plsk::RED_SLICES * 8 * 8));
    double *g = (double *)c->guard.p;
    hipLaunchKernelGGL(plsk::replica_checksum_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, W, P, R, Q, B, (i64)K * A,
                       (i64)M * A, (i64)K * M, g);
    LAUNCH_CHECK(c);
    CHK(do_allreduce(c, g, (i64)plsk::RED_SLICES * 8));
    hipLaunchKernelGGL(plsk::replica_verify_kernel, dim3(1), dim3(plsk::WAVE), 0, c->stream, (const double *)g, c->nranks,
                       c->diverged_dev);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

int check_xchg(pls_hip_context *c);
void xchg_release(pls_hip_context *c);

int check_diverged(pls_hip_context *c) {  // (the stream has been synchronised)
    CHK(check_xchg(c));
    if (c->diverged && c->diverged[1]) {
        c->diverged[1] = 0;
        return fail(c, PLS_HIP_ERR_DEVICE, "the one-launch resident fit waited longer than its time limit for its workgroups to become "
                                           "resident (another process or stream holds the GPU's CUs): results are invalid; "
                                           "PLS_HIP_RESIDENT=0 takes the general plan");
    }
    if (c->diverged && *c->diverged) {
        *c->diverged = 0;
        return fail(c, PLS_HIP_ERR_REDUCER, "the ranks of the sharded fit derived different W / P / Q / R / B: the reducer did not "
                                            "leave identical sums on every rank");
    }
    return PLS_HIP_OK;
}

}  // namespace
