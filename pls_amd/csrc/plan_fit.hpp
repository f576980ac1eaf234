// plan_fit.hpp -- fit_device: Model::plsr (src/pls.cpp:390-437) enqueued on one stream -- the KERNEL, NIPALS and GRAM / KERNEL_TYPE2 plans and the single-launch fits.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

// The resident fits of one process take turns on a device: a launch waits for the previous one's completion event, whatever
// stream that ran on, and records its own (two such kernels together could each hold CUs the other waits for).
struct ResidentTurn {
    std::mutex mu;
    hipEvent_t ev[64] = {};
    bool have[64] = {};
    hipStream_t last[64] = {};  // the stream of the last resident launch (the same stream again: its own order is enough)
    bool any[64] = {};
};
inline ResidentTurn &resident_turns() {
    static ResidentTurn t;
    return t;
}
inline void resident_turn(pls_hip_context *c) {
    ResidentTurn &t = resident_turns();
    std::lock_guard<std::mutex> lock(t.mu);
    const int d = c->device & 63;
    if (t.have[d] && t.any[d] && t.last[d] != c->stream) (void)hipStreamWaitEvent(c->stream, t.ev[d], 0);
}
inline void resident_done(pls_hip_context *c) {
    ResidentTurn &t = resident_turns();
    std::lock_guard<std::mutex> lock(t.mu);
    const int d = c->device & 63;
    if (!t.have[d]) {
        if (hipEventCreateWithFlags(&t.ev[d], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        t.have[d] = true;
    }
    (void)hipEventRecord(t.ev[d], c->stream);
    t.last[d] = c->stream;
    t.any[d] = true;
}

// ---- the fit on device pointers -----------------------------------------------------------
template <typename T>
int fit_device(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
               int A, int method, double *W, double *P, double *Q, double *R, T *Tm, i64 ldt, double *B) {
    // GRAM plan for a KERNEL_TYPE1 request: the K-sized loop runs on XX = X^T X exactly as KERNEL_TYPE2
    // does (no pass over X per component), then the scores are formed in one pass, T = X R.
    // AUTO: pick between the read-only pass plan and the Gram plan from a bandwidth / matrix-core cost
    // model (measured rates on MI355X: ~6 TB/s streaming reads, ~59 TFLOP/s executed in the fp64 SYRK of
    // which the symmetric half is computed).  GRAM pays off for A >~ K/60.
    // A single-response problem small enough for one workgroup's registers: the whole fit in ONE launch (tiny_kernels.hpp) instead of
    // three launches per component, whose dispatch latency would be the entire cost.  The reference's sequence, so the
    // KERNEL plan and AUTO (also when X^T X came with the upload: one launch beats the K x K loop's sixty);
    // an explicit NIPALS or GRAM request keeps its own kernels.
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && c->env.tiny && plsk::tiny_fit_covers(N, K, M, A, ldx, sizeof(T)) &&
        !plsk::micro_fit_covers(N, K, M, A, ldx, sizeof(T))) {  // (the smallest data: one wave is faster than 1024 threads' barriers, below)
        const size_t lds = (size_t)2 * K * A * 8;
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_kernel<T>, (int)plsk::TINY_LDS_MAX)  /* raised once per device: to the most any fit asks for */)
            return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");
        Range r_fit("pls_hip_fit (single launch)");
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
        hipLaunchKernelGGL((plsk::tiny_fit_kernel<T>), dim3(1), dim3(plsk::UPD_THREADS), lds, c->stream, X, ldx, Y, (int)N, K, A,
                           W, P, Q, R, Tm, ldt, B, (const i64 *)nullptr, 0, (i64)0, (double *)nullptr);
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // Mid-size data (1-8 responses) with at most 128 columns under AUTO: ONE launch with three grid-wide hand-offs whatever A -- X^T X and X^T Y on
    // the matrix cores, the component loop on XX in ONE workgroup's LDS, the scores at the end (resident_gram.hpp); the resident
    // fits below exchange once per component.  An explicit KERNEL request keeps the reference's TYPE1 arithmetic (below).
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_AUTO || c->env.resident_gram == 2) && c->opt_fuse && !c->reducer &&
        c->env.tiny && c->env.resident && c->env.resident_gram && !c->opt_graph && plsk::elem_aligned<T>(X) && plsk::elem_aligned<T>(Y) && Tm &&
        !plsk::tiny_fit_covers(N, K, M, A, ldx, sizeof(T)) && !plsk::tiny_fit_m_covers(N, K, M, A, ldx, sizeof(T)) &&
        !plsk::micro_fit_covers(N, K, M, A, ldx, sizeof(T))) {
        int G = plsk::resident_gram_grid(N, K, M, A, ldx, sizeof(T), c->num_cu);
        if (G > 0 && host_flags(c)) {
            plsk::ResidentGram rg;
            // the block form of phase 1 where it pays (PLS_HIP_RESIDENT_GRAM=4: the row form everywhere)
            rg.rs = c->env.resident_gram == 4 ? 0 : plsk::resident_gram_splits(N, K, M, c->num_cu, vec_ok<T>(X, ldx, 16 / (int)sizeof(T)) && vec_ok<T>(Y, ldy, 16 / (int)sizeof(T)));
            if (rg.rs > 0) {
                const int nb = (K + 15) / 16;
                rg.brows = (int)(((N + rg.rs - 1) / rg.rs + 15) & ~(i64)15);
                rg.rs = (int)((N + rg.brows - 1) / rg.brows);
                G = (nb * (nb + 1) / 2 + nb) * rg.rs;
                rg.rows_per = (int)(((N + G - 1) / G + 3) & ~(i64)3);  // (phase 4: a workgroup beyond N has no rows)
            } else {
                rg.rows_per = (int)(((N + G - 1) / G + 3) & ~(i64)3);
                G = (int)((N + rg.rows_per - 1) / rg.rows_per);
            }
            rg.LP = ((i64)K * K + (i64)K * M + 7) & ~(i64)7;
            rg.big = plsk::resident_gram_big(K);
            {  // many rows per workgroup and room in LDS: twice the rows staged at a time
                const int kp = (K + 15) / 16 * 16 + 16;
                if ((i64)rg.rows_per * kp > rg.big && rg.big < 16384 && 16384 + plsk::RG_SMALL + plsk::resident_gram_extra(K, M, A) <= plsk::RG_LDS_DOUBLES - (M > 1 ? 256 : 0))
                    rg.big = 16384;
            }
            const size_t need = 256 + ((size_t)(G + 1) * rg.LP + (size_t)K * A) * 8;
            if (c->resident.bytes < need) {
                CHK(ensure(c, c->resident, need));
                HIPCHK(c, hipMemsetAsync(c->resident.p, 0, 256, c->stream));  // both counters start from zero
                c->resident_launches = 0;
            }
            const size_t lds = ((size_t)rg.big + plsk::RG_SMALL + (size_t)plsk::resident_gram_extra(K, M, A)) * 8;
            const void *fn = rg.rs > 0 ? (M <= 1 ? (const void *)plsk::resident_gram_fit_kernel<T, 1, true> : (const void *)plsk::resident_gram_fit_kernel<T, 2, true>)
                             : M <= 1 ? (const void *)plsk::resident_gram_fit_kernel<T, 1>
                             : M <= 2 ? (const void *)plsk::resident_gram_fit_kernel<T, 2>
                             : M <= 4 ? (const void *)plsk::resident_gram_fit_kernel<T, 4>
                                      : (const void *)plsk::resident_gram_fit_kernel<T, 8>;
            if (plsk::raise_dynamic_lds(fn, (int)lds)) {  // (refused: the per-component resident kernels below take the fit)
            if (!c->rgflags.p) {
                CHK(ensure(c, c->rgflags, (size_t)plsk::RESIDENT_MAX_WG * 4));
                HIPCHK(c, hipMemsetAsync(c->rgflags.p, 0, (size_t)plsk::RESIDENT_MAX_WG * 4, c->stream));
                c->rg_epoch = 0;
            }
            rg.flags = (unsigned *)c->rgflags.p;
            rg.epoch = c->rg_epoch;
            c->rg_epoch += 4;  // (three hand-offs at most; the words wrap with it: the kernel compares signed distances)
            rg.part = (double *)((char *)c->resident.p + 256);
            rg.gred = rg.part + (size_t)G * rg.LP;
            rg.rshare = rg.gred + rg.LP;
            rg.sy.status = c->diverged_dev + 1;
            rg.sy.limit = (long long)(0.05 * 1e8);  // 50 ms of the 100 MHz wall clock
#ifdef PLS_HIP_TESTING
            if (const char *e = getenv("PLS_HIP_TEST_RESIDENT_LIMIT_TICKS")) rg.sy.limit = atoll(e);
#endif
            Range r_fit("pls_hip_fit (single launch, resident, X^T X)");
            Scope s(c, PLS_HIP_FAM_SMALL, (2 * (i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
            resident_turn(c);
            {
                int pit = (int)c->opt_power_iters;
                void *args[] = {(void *)&X, (void *)&ldx, (void *)&Y, (void *)&ldy, (void *)&N, (void *)&K, (void *)&M, (void *)&A, (void *)&pit, (void *)&W,
                                (void *)&P, (void *)&Q, (void *)&R, (void *)&Tm, (void *)&ldt, (void *)&B, (void *)&rg};
                if (hipLaunchKernel(fn, dim3(G), dim3(plsk::UPD_THREADS), args, lds, c->stream) != hipSuccess) {
                    (void)hipGetLastError();
                    return fail(c, PLS_HIP_ERR_DEVICE, "kernel launch: resident_gram_fit");
                }
            }
            resident_done(c);
            return PLS_HIP_OK;
            }
        }
    }
    // Mid-size single-response data (beyond one workgroup's 1024 rows, up to ~50 MB): the same single launch on up to 256
    // workgroups with one grid-wide exchange per component (resident_kernels.hpp)
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && c->env.tiny && c->env.resident && !c->opt_graph /* (a replayed graph would re-use one launch's arrival counter) */ &&
        plsk::elem_aligned<T>(X) && plsk::elem_aligned<T>(Y) && Tm) {
        // (the score columns go out through one buffer descriptor per workgroup: A ld s below 2^31)
        const int wps = (i64)A * ldt * (i64)sizeof(T) < (1ll << 31) ? plsk::resident_wps(N, K, M, A, ldx, sizeof(T), c->num_cu) : 0;
        if (wps > 0 && host_flags(c)) {
            const int G = (int)((N + (i64)plsk::WAVE * wps - 1) / ((i64)plsk::WAVE * wps));
            const int LP = (K + 1 + 7) & ~7;
            const size_t need = 256 + (size_t)2 * G * LP * 8;
            if (c->resident.bytes < need) {
                CHK(ensure(c, c->resident, need));
                HIPCHK(c, hipMemsetAsync(c->resident.p, 0, 256, c->stream));  // both counters start from zero
                c->resident_launches = 0;
            }
            if (!plsk::raise_dynamic_lds((const void *)plsk::resident_fit_kernel<T>, (int)plsk::TINY_LDS_MAX))
                return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the resident fit could not be raised");
            plsk::ResidentSync sy;
            unsigned *ctr = (unsigned *)c->resident.p;
            sy.bar = ctr + 16 * (c->resident_launches & 1);        // (64 bytes apart)
            sy.bar_next = ctr + 16 * ((c->resident_launches + 1) & 1);
            ++c->resident_launches;
            sy.part = (double *)((char *)c->resident.p + 256);
            sy.status = c->diverged_dev + 1;
            sy.limit = (long long)(0.05 * 1e8);  // 50 ms of the 100 MHz wall clock
#ifdef PLS_HIP_TESTING
            if (const char *e = getenv("PLS_HIP_TEST_RESIDENT_LIMIT_TICKS")) sy.limit = atoll(e);  // (the time-out path, tests only)
#endif
            sy.LP = LP;
            Range r_fit("pls_hip_fit (single launch, resident)");
            Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + 1) * A * 8);
            // all G workgroups must be resident together: the resident fits of ONE process take turns (an event chain per
            // device across its streams); a foreign kernel holding CUs ends in the bounded wait's error, not in a hang
            resident_turn(c);
            hipLaunchKernelGGL((plsk::resident_fit_kernel<T>), dim3(G), dim3(plsk::UPD_THREADS), (size_t)2 * K * A * 8, c->stream, X, ldx, Y,
                               N, K, A, W, P, Q, R, Tm, ldt, B, wps, sy);
            LAUNCH_CHECK(c);
            resident_done(c);
            return PLS_HIP_OK;
        }
    }
    // ... and the same for 2..8 responses (resident_fit_m_kernel)
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && c->env.tiny && c->env.resident && !c->opt_graph /* (a replayed graph would re-use one launch's arrival counter) */ &&
        plsk::elem_aligned<T>(X) && plsk::elem_aligned<T>(Y) && Tm) {
        const int wps = plsk::resident_m_wps(N, K, M, A, ldx, sizeof(T), c->num_cu);
        if (wps > 0 && host_flags(c)) {
            const int G = (int)((N + (i64)plsk::WAVE * wps - 1) / ((i64)plsk::WAVE * wps));
            const int LP = (std::max(K + 1, std::min(K * M, (int)plsk::UPD_THREADS)) + 7) & ~7;
            const size_t need = 256 + (size_t)2 * G * LP * 8;
            if (c->resident.bytes < need) {
                CHK(ensure(c, c->resident, need));
                HIPCHK(c, hipMemsetAsync(c->resident.p, 0, 256, c->stream));
                c->resident_launches = 0;
            }
            plsk::ResidentSync sy;
            unsigned *ctr = (unsigned *)c->resident.p;
            sy.bar = ctr + 16 * (c->resident_launches & 1);
            sy.bar_next = ctr + 16 * ((c->resident_launches + 1) & 1);
            ++c->resident_launches;
            sy.part = (double *)((char *)c->resident.p + 256);
            sy.status = c->diverged_dev + 1;
            sy.limit = (long long)(0.05 * 1e8);
#ifdef PLS_HIP_TESTING
            if (const char *e = getenv("PLS_HIP_TEST_RESIDENT_LIMIT_TICKS")) sy.limit = atoll(e);
#endif
            sy.LP = LP;
            const size_t lds = (size_t)(2 * K + M) * A * 8;
            Range r_fit("pls_hip_fit (single launch, resident)");
            Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
#define RES_M(MM_)                                                                                                              \
    do {                                                                                                                        \
        if (!plsk::raise_dynamic_lds((const void *)plsk::resident_fit_m_kernel<T, MM_>, (int)plsk::TINY_LDS_MAX))                 \
            return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the resident fit could not be raised");                      \
        resident_turn(c);                                                                                                       \
        hipLaunchKernelGGL((plsk::resident_fit_m_kernel<T, MM_>), dim3(G), dim3(plsk::UPD_THREADS), lds, c->stream, X, ldx, Y, ldy,  \
                           N, K, M, A, (int)c->opt_power_iters, W, P, Q, R, Tm, ldt, B, wps, sy);                               \
    } while (0)
            if (M <= 2) RES_M(2); else if (M <= 4) RES_M(4); else RES_M(8);
#undef RES_M
            LAUNCH_CHECK(c);
            resident_done(c);
            return PLS_HIP_OK;
        }
    }
    // The smallest problems (N <= 64, K <= 32, 1..8 responses: the reference's README example) as ONE WAVE (micro_fit_kernel)
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && c->env.tiny && plsk::micro_fit_covers(N, K, M, A, ldx, sizeof(T))) {
        Range r_fit("pls_hip_fit (single launch, one wave)");
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
#define MICRO(MM_)                                                                                                           \
    hipLaunchKernelGGL((plsk::micro_fit_kernel<T, MM_>), dim3(1), dim3(plsk::WAVE), 0, c->stream, X, ldx, Y, ldy, (int)N, K, M, A, \
                       (int)c->opt_power_iters, W, P, Q, R, Tm, ldt, B, (const i64 *)nullptr, 0, (i64)0, (double *)nullptr)
        if (M <= 2) MICRO(2); else if (M <= 4) MICRO(4); else MICRO(8);
#undef MICRO
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // ... and the same for 2..8 responses (tiny_fit_m_kernel: the reference's own example, README.md:23, is such a fit)
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && c->env.tiny && plsk::tiny_fit_m_covers(N, K, M, A, ldx, sizeof(T))) {
        const size_t lds = (size_t)(2 * K + M) * A * 8;
        Range r_fit("pls_hip_fit (single launch)");
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
#define TINY_M(MM_)                                                                                                          \
    do {                                                                                                                     \
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_m_kernel<T, MM_>, (int)plsk::TINY_LDS_MAX))               \
            return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");            \
        hipLaunchKernelGGL((plsk::tiny_fit_m_kernel<T, MM_>), dim3(1), dim3(plsk::UPD_THREADS), lds, c->stream, X, ldx, Y, ldy,  \
                           (int)N, K, M, A, (int)c->opt_power_iters, W, P, Q, R, Tm, ldt, B, (const i64 *)nullptr, 0, (i64)0,  \
                           (double *)nullptr);                                                                               \
    } while (0)
        if (M <= 2) TINY_M(2); else if (M <= 4) TINY_M(4); else TINY_M(8);
#undef TINY_M
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    i64 algo = c->opt_algo;
    const bool have_pre = c->pre_xx && c->pre_xy && K <= 32768;
    if (algo == PLS_HIP_ALGO_AUTO && have_pre) {
        algo = PLS_HIP_ALGO_GRAM;  // X^T X is already there: the component loop needs no pass over X at all
    } else if (algo == PLS_HIP_ALGO_AUTO) {
        // Both plans priced with what they launch (profiles/r4/auto_scan.txt): a pass over X per component + ~14 us of launches for
        // KERNEL; for GRAM the SYRK on the blocks it EXECUTES (whole 128-column blocks: a 32-column matrix costs what a 128-column
        // one does), the pass that forms T = X R, and per component the symv + update on K x K data, ~9 us + 18 ns per column.
        const double pass_s = (double)N * K * sizeof(T) / 6.0e12;
        const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
        const double kp = (double)nbk * plsk::SYRK_TB;
        // tiles of 16 x 16 the SYRK executes: the blocks above the diagonal in full, 36 of 64 in a diagonal block (syrk_kernels.hpp)
        const double tile_frac = (32.0 * nbk * (nbk - 1) + 36.0 * nbk) / (64.0 * nbk * nbk);
        const double syrk_s = 2.0 * N * kp * kp * tile_frac / 60.0e12;
        const double kernel_s = (1 + A) * pass_s + A * 14e-6;
        const double gram_s = syrk_s + 1.3 * pass_s + A * (9e-6 + K * 1.8e-8) + 40e-6;
        // ranks of a sharded fit see different N: they must not disagree on the plan -> KERNEL there
        const bool gram_ok = K <= 2048 && N >= 4096 && !c->reducer;
        algo = (gram_ok && kernel_s > gram_s) ? PLS_HIP_ALGO_GRAM : PLS_HIP_ALGO_KERNEL;
    }
    const bool gram = (method == PLS_HIP_KERNEL_TYPE1) && (algo == PLS_HIP_ALGO_GRAM);
    const bool type2 = (method == PLS_HIP_KERNEL_TYPE2) || gram;
    const bool nipals = !type2 && (algo == PLS_HIP_ALGO_NIPALS);
    const int nip = nipals ? 1 : 0;
    const i64 L0 = (i64)K * M;
    const i64 redn = (i64)plsk::RED_SLICES * std::max<i64>(L0, K + 1);
    const i64 prow = max_partial_rows(c, N, K);
    CHK(ensure(c, c->part, (size_t)prow * (size_t)std::max<i64>(L0, K) * 8));
    // t^T t partials: one per workgroup of whichever kernel forms the scores (narrow X*v: N/256; tile kernels: their grid)
    // (the split score kernel of short, wide matrices leaves one partial per 64 rows)
    const i64 ssmax = std::max<i64>(std::max<i64>(K >= 1024 ? (N + 63) / 64 : (N + plsk::WG - 1) / plsk::WG, 1), max_partial_rows(c, N, K));
    CHK(ensure(c, c->sspart, (size_t)ssmax * 8));
    CHK(ensure(c, c->xy, (size_t)L0 * 8));
    CHK(ensure(c, c->v, (size_t)K * 8));
    CHK(ensure(c, c->cs, (size_t)A * 8));
    double *red;
    if (c->user_red) {
        if (c->user_red_count < redn) return fail(c, PLS_HIP_ERR_INVALID, "reduce buffer too small");
        red = c->user_red;
    } else {
        CHK(ensure(c, c->red, (size_t)redn * 8));
        red = (double *)c->red.p;
    }
    // NIPALS keeps the deflated matrix in a library-owned buffer.  When the tile-resident pass covers the fit it
    // is stored row-tile-major (every R x K tile one contiguous block: fused_kernels.hpp), otherwise column-major
    // with ld = N for the one-product kernels.
    // the tile that reads the caller's matrix: 32 column groups x 16 row lanes, taller with fewer groups for narrow matrices
    const int tall_cg = plsk::tall_groups(K);
    const i64 TR = (512 / tall_cg) * (i64)(16 / sizeof(T));
    // Any row count (the last N % V rows go to a tail kernel), any alignment of the columns and leading dimensions up to
    // 2^31 / 4 bytes (mode 2: the EDGE instantiations) -- the same one-sweep traffic for every matrix the reference
    // accepts (src/pls.cpp:419-421)
    const int fused_mode = (c->opt_fuse && N > 0) ? plsk::fused_pass_mode<T>(X, ldx, N, K, Tm, ldt) : 0;
    const bool fused_fit = fused_mode != 0;
    const int wide_mode = (c->opt_fuse && !fused_fit && N > 0) ? plsk::deflate_score_mode<T>(X, ldx, N, K, Tm, ldt) : 0;
    // wide matrices (no resident tile): deflation + score in one sweep, loading in a second read
    const bool semi_fit = nipals && wide_mode != 0;
    // Beyond the semi-fused sweep's reach (its w and p_prev need 16 K bytes of LDS: K <= 4608) and up to 8192 columns the
    // copy's tiles are ONE row pack high (512 column groups x 16 columns per lane): the copy is made in the X^T Y sweep
    // (retile_xty) and every component runs fused on it -- instead of 4 N K s (NIPALS) / 2 N K s (KERNEL) per component
    // through the one-product kernels.
    // (read-only passes take 32 columns per lane there: the KERNEL plan up to 16384 columns)
    // (every precondition of launch_retile_xty for this shape is decided HERE, so that a fit which counts on the copy --
    // wide_only -- cannot find it refused later: the source layout (wide_source_ok), element-aligned responses, M <= 8
    // (wide_only below); the launcher's span limits hold for every row-pack tile -- 16 V K s + 512 bytes < 2^31 at K <= 16384
    // -- and the partial-row capacity is never below 2)
    // (beyond 8192 columns -- 32 columns per lane, a partial row of K doubles per workgroup -- only from 4096 rows on: on a shorter
    // matrix the partial rows weigh as much as the matrix and the one-product kernels win, 500 x 10,000: 0.47 against 0.83 ms per
    // fit, 1,500 x 10,000: 0.70 / 0.89, 3,000 x 12,000: 1.22 / 1.30; 6,000 x 10,000: 1.91 / 1.59 -- profiles/r4/short_wide_plan_choice.txt)
    const bool wide_src = c->opt_fuse && !fused_fit && N > 0 && K <= 512 * (nipals ? 16 : 32) && (K <= 512 * 16 || N >= 4096) &&
                          plsk::wide_source_ok<T>(X, ldx, N, Tm) && plsk::elem_aligned<T>(Y);
    // (more than 8 responses: X^T Y does not ride in the copy's sweep -- a plain copy before the first component and the X^T Y pass
    // of its own, as the KERNEL plan does; 25,000 x 8,192 with 12 responses: 12.3 -> 7.5 ms per 10-component fit)
    const bool wide_only = nipals && wide_src && K > 256 * 16 && K <= 512 * 16 && A >= 3 && c->opt_work_layout != 0;
    const bool tiled_work = nipals && (fused_fit || semi_fit || wide_only) && c->opt_work_layout != 0;
    // Row-tile-major tiles are contiguous whatever their height, so for 1024 < K <= 4096 the working copy uses
    // SHORTER tiles (8-32 rows) that do fit the registers of a CU: from the third component on the fully fused
    // pass runs again (2 N K s per component instead of the semi-fused 3 N K s).  Only the first deflation has
    // to read the caller's column-major X in 256-byte pieces (deflate_score, writing the short tiles).
    // KERNEL plan on such a matrix: X is copied ONCE into short tiles (read + write), after which every component
    // is one fused read-only pass instead of two one-product passes -- pays from the third component on.
    // ... and on a matrix whose columns are not 16-byte aligned (ld odd, a base pointer at 8 mod 16): the one-sweep pass can
    // read it (EDGE level 2) but every 256-byte segment then shares a line with its neighbours and the pass runs at 0.52
    // instead of 0.73 of peak; with the copy (formed in the same sweep as X^T Y: retile_xty_kernel) every component reads
    // aligned tiles.  Costs one write of X; pays from the fourth component on (copy_min below).
    constexpr int FVX = 16 / (int)sizeof(T);
    // The same copy pays for ALIGNED matrices once there are enough components: a read-only pass over the tiled copy, one
    // contiguous 128 KB block per tile, runs at 0.85 of peak (0.63 ms at config 3) against 0.72 (0.74 ms) over the caller's
    // column-major matrix in 256-byte segments; the copy costs 0.86 ms more than the X^T Y pass it replaces
    // (copy_min_al below: 10 components).
    // Beyond 512 columns the direct pass (32 columns per lane, one workgroup per CU) already reads at 0.81 of peak and the
    // copy only pays from ~30 components on (one shard of config 5: 2.65 -> 2.52 ms per pass against 4.1 ms for the copy).
    constexpr int copy_min = 4, copy_min_al = 10;
    const bool copy_fit = fused_fit && M <= 8 &&
                          A >= (vec_ok<T>(X, ldx, FVX) ? (K <= 32 * 16 ? copy_min_al : 3 * copy_min_al + 2) : copy_min);
    bool retile_fit = !nipals && !type2 && A >= 3 && c->opt_work_layout != 0 &&
                      ((wide_mode != 0 && K <= 128 * 32) || (wide_src && K > 128 * 32) || copy_fit);
    // column groups of the short tiles of a wide matrix (1024 < K <= 4096): 16 columns per lane in 128 / 256 groups
    // (8-row fp32 / 4-row fp64 tiles at K <= 4096) -- the register shape of the headline kernel, two workgroups per CU
    // on read-only passes.  Config 4: read+write pass 0.766 -> 0.710 ms (0.70 -> 0.76 of peak), read-only pass
    // 0.364 -> 0.324 ms (0.74 -> 0.83) against 32 columns per lane in 64 / 128 groups (the round-1 shape, deleted in round 4).
    const int wide_groups = fused_fit ? (K <= 32 * 16 ? tall_cg : 64)  // (the copy of a matrix the resident tile covers)
                            : (K <= 128 * 16 ? 128 : (K <= 256 * 16 ? 256 : ((wide_src && (!nipals || wide_only)) ? 512 : 0)));
    if (retile_fit) {  // the copy is optional: without room for it the one-product kernels do the job
        const i64 wr = (512 / wide_groups) * (i64)(16 / sizeof(T));
        if (ensure(c, c->work, (size_t)((N + wr - 1) / wr) * wr * K * sizeof(T)) != PLS_HIP_OK) {
            retile_fit = false;
            c->err.clear();
        }
    }
    const int wide_cg = ((semi_fit && tiled_work) || retile_fit || wide_only) ? wide_groups : 0;
    // 512 < K <= 1024: the resident tile of the caller's layout needs 32 columns per lane (one 8-wave workgroup per
    // CU, 5.7-5.85 TB/s); on the tiled copy the same K fits half-height tiles at 16 columns per lane (6.0 TB/s).
    // Component 0 reads X with the tall tile, the first deflation reads X tall and writes the short tiles (rdst),
    // every later pass runs on the short tiles.
    const int mid_cg = (nipals && fused_fit && tiled_work && K > 32 * 16 && A > 2) ? 64 : 0;
    // opt-in deferred write-back (defer_kernels.hpp): up to `defer` rank-1 updates pending per stored matrix
    const int defer = (nipals && fused_mode == 1 && tall_cg == 32 && N % FVX == 0 && plsk::cols_aligned<T>(Tm, ldt) && tiled_work && K <= 32 * 16)
                          ? (int)c->opt_defer : 1;
    const int work_cg = wide_cg ? wide_cg : (mid_cg ? mid_cg : tall_cg);
    const i64 WR = (512 / work_cg) * (i64)(16 / sizeof(T));  // rows per tile of the working copy
    if ((nipals && A > 1 && N > 0) || retile_fit)
        CHK(ensure(c, c->work, (tiled_work || retile_fit) ? (size_t)((N + WR - 1) / WR) * WR * K * sizeof(T)
                                                          : (size_t)((N + 3) & ~(i64)3) * K * sizeof(T)));
    double *part = (double *)c->part.p, *sspart = (double *)c->sspart.p;
    double *XY = (double *)c->xy.p, *v = (double *)c->v.p;
    // (round 5 moved this copy around inside its allocation -- 20 offsets from 256 bytes to 32 MB, profiles/r5/retile_offsets.txt:
    // retile_xty takes the same 1.60-1.63 ms wherever it lies)
    T *work = (T *)c->work.p;
    // The partial rows of a fused pass are summed in the tail of the pass itself (slice_tail: no reduce launch behind it),
    // and with the device-side exchange attached the push of a sharded component rides there too, the gather in front of
    // the update: pass -> update.  PLS_HIP_TAIL=0: the launches of round 3 (A/B measurements).
    plsk::SliceTail tail;
    if (c->env.tail && c->opt_fuse && N > 0) {
        if (!c->tailcnt.p) CHK(ensure(c, c->tailcnt, 256));
        HIPCHK(c, hipMemsetAsync(c->tailcnt.p, 0, 256, c->stream));  // (whatever an earlier, failed fit left behind)
        tail.cnt = (unsigned *)c->tailcnt.p;
        tail.red = red;
    }
    const bool xep_ok = c->xep.on && c->reducer && (i64)K + 1 <= plsk::XCHG_CAP;

    // prologue: XY = X^T Y (src/pls.cpp:396), summed over ranks
    Range r_fit("pls_hip_fit");
    std::unique_ptr<Range> r_phase(new Range("X^T Y"));
    const bool use_pre = have_pre && (gram || method == PLS_HIP_KERNEL_TYPE2);
    bool xy_from_syrk = false, xx_local_done = false, retiled = false;
    // NIPALS on a wide matrix takes the same first sweep: its working copy is then complete before the first component,
    // which runs fused on it, as every later one does in place (instead of two one-product passes over X for component 0
    // and the semi-fused sweep + a loading pass for component 1: config 4 36.6 -> 36.0 ms per fit)
    const bool nip_copy = nipals && (semi_fit || wide_only) && tiled_work && wide_cg != 0 && A >= 3;
    if (!use_pre && N > 0 && (retile_fit || nip_copy) && M <= 8) {
        // the copy into tiles and X^T Y in ONE sweep over the caller's matrix (instead of retile_kernel + the X^T Y pass)
        int nb = 0, rc = 1;
        {
            Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T) + (i64)N * M * sizeof(T) + L0 * 8);
            // (the source tile is as tall as the copy's for narrow matrices, the 256-byte-segment tile otherwise)
#define RX_CALL(CG_) plsk::launch_retile_xty<T, CG_>(c->stream, c->num_cu, X, ldx, Y, ldy, work, WR, WR * (i64)K, (int)WR, N, K, M, part, (int)prow, &nb)
            rc = (fused_fit && tall_cg == 8) ? RX_CALL(8) : ((fused_fit && tall_cg == 16) ? RX_CALL(16) : RX_CALL(32));
#undef RX_CALL
            if (rc != 0) s.on = false;
        }
        if (rc == 0) {
            LAUNCH_CHECK(c);
            CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
            retiled = true;
        }
    }
    if (wide_only && !retiled && N > 0 && M <= 8) return fail(c, PLS_HIP_ERR_DEVICE, "copy into row-pack tiles failed");
    if (retiled) {
    } else if (use_pre) {
        hipLaunchKernelGGL(plsk::fill_slices_kernel, dim3((unsigned)((L0 + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, c->pre_xy, (int)L0, red);
        LAUNCH_CHECK(c);
    } else if (N > 0) {
        // KERNEL_TYPE2 / GRAM: the SYRK's diagonal workgroups form X^T Y on the way (no pass over X of its own); its
        // partial blocks take c->part, so it runs first and `part` is read again afterwards
        if (type2) {
            CHK(compute_xx_local<T>(c, X, ldx, N, K, Y, ldy, M, red, &xy_from_syrk));
            xx_local_done = true;
            part = (double *)c->part.p;
        }
        if (!xy_from_syrk) {
            int nb = 0;
            CHK(launch_xty<T>(c, X, ldx, Y, ldy, N, K, M, part, &nb));
            CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
        }
    } else {
        HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * L0 * 8, c->stream));
    }
    CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * L0));
    CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, -1, nip));
    r_phase.reset();

    if (type2) {
        // KERNEL_TYPE2 (src/pls.cpp:398, :422-425): XX = X^T X once, then the A-loop never touches X:
        // tt = r^T XX r, p = XX r / tt; T is not computed.  XX is formed in 32-column blocks with the
        // same column-reduction kernel as X^T Y (functional; an MFMA SYRK is the planned fast form).
        CHK(ensure(c, c->xx, (size_t)K * K * 8));
        CHK(ensure(c, c->praw, (size_t)K * 8));
        double *XX = (double *)c->xx.p, *praw = (double *)c->praw.p;
        r_phase.reset(new Range("X^T X (SYRK)"));
        if (use_pre) {  // this member's X^T X came with the upload: present it as slice 0, sum over the members
            const i64 KK = (i64)K * K;
            CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8));
            hipLaunchKernelGGL(plsk::fill_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                               c->stream, c->pre_xx, (int)KK, (double *)c->red2.p);
            LAUNCH_CHECK(c);
            CHK(do_allreduce(c, (double *)c->red2.p, (i64)plsk::RED_SLICES * KK));
            hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                               c->stream, (const double *)c->red2.p, (int)KK, XX);
            LAUNCH_CHECK(c);
        } else {
            if (!xx_local_done) CHK(compute_xx_local<T>(c, X, ldx, N, K));  // (an empty shard: zero slices)
            CHK(compute_xx_finish(c, K, XX));
        }
        r_phase.reset();
        for (int a = 0; a < A; ++a) {
            Range r_comp("component", a);
            {
                Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * K + 2 * K) * 8);
                hipLaunchKernelGGL(plsk::symv_kernel, dim3((K + 3) / 4), dim3(plsk::WG), 0, c->stream,
                                   (const double *)XX, (const double *)v, K, praw);  // XX symmetric: XX r
                LAUNCH_CHECK(c);
            }
            if (update_is_single(K, M, A, a)) {  // tt = r^T praw and the packing: the prologue of the one-workgroup update
                Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + 2 * K) * 8);
                hipLaunchKernelGGL(plsk::component_update_type2_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                                   (size_t)std::min(A, 4096) * sizeof(double), c->stream, (const double *)praw, (const double *)v, red,
                                   XY, W, P, Q, R, v, K, M, A, a, (int)c->opt_power_iters, 0);
                LAUNCH_CHECK(c);
                continue;
            }
            hipLaunchKernelGGL(plsk::type2_pack_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream,
                               (const double *)praw, (const double *)v, K, red);
            LAUNCH_CHECK(c);
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, 0));
        }
        if (B) {
            const int nblk = (int)((L0 + plsk::WG - 1) / plsk::WG);
            hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, c->stream, R, Q, K, M, A, B);
            LAUNCH_CHECK(c);
        }
        if (gram && N > 0) {  // T = X R (src/pls.cpp:439-442 applied to the training data)
            Range r_t("T = X R");
            int nss = 0;
            CHK(launch_xb<T>(c, X, ldx, N, K, R, K, A, Tm, ldt, nullptr, &nss));
        }
        return replica_guard(c, W, P, Q, R, B, K, M, A);
    }

    const T *Xc = X;
    i64 ldc = ldx, tsc = TR;  // column stride and tile stride of the current matrix
    // (a column-major working copy keeps 16-byte columns whatever N is)
    constexpr i64 PV = 16 / (i64)sizeof(T);
    const i64 ldw = (tiled_work || retile_fit) ? WR : (N + PV - 1) / PV * PV, tsw = (tiled_work || retile_fit) ? WR * (i64)K : TR;
    bool cur_tiled = false;  // Xc is the row-tile-major working copy
    int defer_b = 0;         // deferred write-back: index of the stored matrix X_b
    for (int a = 0; a < A; ++a) {
        Range r_comp("component", a);
        bool tail_used = false, upd_done = false;
        // sharded over the device-side exchange, the one-workgroup update behind the pass: the pass pushes (if its tail
        // runs), the update gathers.  The collective number is drawn only when the pass did push.
        const bool want_push = tail.cnt && xep_ok && update_is_single(K, M, A, a);
        // one response: the update is the last act of the pass's tail (update_m1.hpp) -- ONE launch per component; sharded
        // over the device-side exchange the tail pushes, waits for the peers' pushes and updates (other reducers: a call
        // between the pass and the update, so the update stays a launch of its own)
        const bool defl_pass = nipals && a > 0;
        const bool want_upd = tail.cnt && (c->env.tail_update == 2 || (c->env.tail_update == 1 && !defl_pass)) && M == 1 &&
                              K <= plsk::UPD1_KMAX && update_is_single(K, M, A, a) && (!c->reducer || want_push);
        tail.npush = 0;
        tail.upd = plsk::TailUpdate();
        tail.gx = plsk::XchgGather();
        if (want_push) {
            const unsigned long long seq = *c->xep.seq + 1;
            const int par = (int)(seq & 1);
            tail.npush = c->xep.n;
            tail.seq = seq;
            for (int j = 0; j < c->xep.n; ++j) {
                tail.peers.slot[j] = c->xep.inbox[j] + ((i64)par * c->xep.n + c->xep.rank) * plsk::XCHG_CAP;
                tail.peers.flag[j] = c->xep.flags[j] + par * c->xep.n + c->xep.rank;
            }
            if (want_upd) {
                plsk::XchgGather &gx = tail.gx;
                gx.inbox = c->xep.inbox[c->xep.rank] + (i64)par * c->xep.n * plsk::XCHG_CAP;
                gx.flags = c->xep.flags[c->xep.rank] + par * c->xep.n;
                gx.n = c->xep.n; gx.cap = plsk::XCHG_CAP; gx.seq = seq;
                gx.status = c->xep.status; gx.host_status = c->xep.host_status; gx.limit = *c->xep.limit;
            }
        }
        if (want_upd) {
            tail.upd.XY = XY; tail.upd.W = W; tail.upd.P = P; tail.upd.Q = Q; tail.upd.R = R; tail.upd.vnext = v;
            tail.upd.A = A; tail.upd.a = a; tail.upd.nipals = nip;
        }
        if (N > 0) {
            bool done = false;
            if (fused_fit && defer > 1 && a > 0) {
                // deferred write-back: the stored matrix is X_b (the caller's X for b = 0, the working copy after the
                // first store); apply the a - b pending updates in registers, store X_a when `defer` are pending
                int nb = 0, nss = 0, rc;
                const int np = a - defer_b;
                const bool store = (np == defer) && (a + 1 < A);
                plsk::PendingUpdates<T> pend{};
                for (int n = 0; n < np; ++n) {
                    pend.t[n] = Tm + (i64)(defer_b + n) * ldt;
                    pend.p[n] = P + (i64)(defer_b + n) * K;
                }
                {
                    const i64 bytes = (store ? 2 : 1) * (i64)N * K * sizeof(T) + (np + 1) * (i64)N * sizeof(T) +
                                      (np + 2) * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
                    rc = plsk::launch_fused_defer<T>(c->stream, c->num_cu, Xc, ldc, tsc, work, ldw, tsw, N, K, v, np, pend,
                                                     store, Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss);
                    if (rc != 0) s.on = false;
                }
                if (rc != 0) return fail(c, PLS_HIP_ERR_DEVICE, "deferred fused pass launch failed");
                LAUNCH_CHECK(c);
                done = true;
                if (store) { Xc = work; ldc = ldw; tsc = tsw; cur_tiled = true; defer_b = a; }
                CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            } else if (fused_fit && !retile_fit) {
                // tile-resident pass: [deflate with (t_{a-1}, p_{a-1}) +] t_a = X v, X^T t_a partials
                int nb = 0, nss = 0;
                const T *tprev = (nipals && a > 0) ? Tm + (i64)(a - 1) * ldt : nullptr;
                const double *pprev = (nipals && a > 0) ? P + (i64)(a - 1) * K : nullptr;
                int rc;
                {
                    const i64 bytes = (tprev ? 2 : 1) * (i64)N * K * sizeof(T) +
                                      (tprev ? 2 : 1) * (i64)N * sizeof(T) + (tprev ? 3 : 2) * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
                    if (mid_cg && a >= 2)  // half-height tiles of the working copy, in place
                        rc = plsk::launch_fused_pass<T, 64>(c->stream, c->num_cu, work, ldw, tsw, work, ldw, tsw, N, K, v,
                                                            tprev, pprev, Tm + (i64)a * ldt, part, (int)prow, sspart,
                                                            &nb, &nss, (int)c->opt_fused_grid, 0, true, &tail, &tail_used, &upd_done);
                    else {  // (a == 1 with mid_cg: X in 256-byte segments -> half-height tiles)
#define TALL_PASS(CG_) plsk::launch_fused_pass<T, CG_>(c->stream, c->num_cu, Xc, ldc, tsc, tprev ? work : nullptr, ldw, tsw, N, K, v, tprev, \
                                                       pprev, Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss,                        \
                                                       (int)c->opt_fused_grid, (mid_cg && tprev) ? (int)WR : 0, Xc == work, &tail, &tail_used, &upd_done)
                        rc = tall_cg == 8 ? TALL_PASS(8) : (tall_cg == 16 ? TALL_PASS(16) : TALL_PASS(32));
#undef TALL_PASS
                    }
                    if (rc != 0) s.on = false;  // nothing was launched: drop the event pair
                }
                if (rc == 0) {
                    LAUNCH_CHECK(c);
                    done = true;
                    if (tprev) { Xc = work; ldc = ldw; tsc = tsw; cur_tiled = tiled_work; }
                    if (!tail_used) CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
                } else {
                    return fail(c, PLS_HIP_ERR_DEVICE, "fused pass launch failed");
                }
            } else if (wide_cg && (nipals ? (a >= 2 || retiled || wide_only) : true)) {
                // short-tile fused pass on the working copy: NIPALS deflates it in place, KERNEL only reads it
                int nb = 0, nss = 0, rc;
                const T *tprev = (nipals && a > 0) ? Tm + (i64)(a - 1) * ldt : nullptr;
                const double *pprev = (nipals && a > 0) ? P + (i64)(a - 1) * K : nullptr;
                if ((!nipals || wide_only) && a == 0 && !retiled) {  // the one-time copy into short tiles (before the first component: it runs fused, too)
                    Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T));
                    if (plsk::launch_retile<T>(c->stream, c->num_cu, X, ldx, work, ldw, tsw, (int)WR, N, K) != 0) {
                        s.on = false;
                        return fail(c, PLS_HIP_ERR_DEVICE, "retile launch failed");
                    }
                    LAUNCH_CHECK(c);
                }
                {
                    const i64 bytes = (tprev ? 2 : 1) * ((i64)N * K * sizeof(T) + (i64)N * sizeof(T)) + 3 * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
#define WIDE_PASS(CG_) plsk::launch_fused_pass<T, CG_>(c->stream, c->num_cu, work, ldw, tsw, work, ldw, tsw, N, K, v, tprev, pprev, \
                                                        Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss, (int)c->opt_fused_grid, 0, true, \
                                                        &tail, &tail_used, &upd_done)
                    rc = wide_cg == 8 ? WIDE_PASS(8) : (wide_cg == 16 ? WIDE_PASS(16) : (wide_cg == 32 ? WIDE_PASS(32)
                         : (wide_cg == 64 ? WIDE_PASS(64) : (wide_cg == 128 ? WIDE_PASS(128) : (wide_cg == 256 ? WIDE_PASS(256) : WIDE_PASS(512))))));
#undef WIDE_PASS
                    if (rc != 0) s.on = false;
                }
                if (rc != 0) return fail(c, PLS_HIP_ERR_DEVICE, "short-tile fused pass launch failed");
                LAUNCH_CHECK(c);
                done = true;
                if (!tail_used) CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            }
            if (!done) {
                int nss = 0, nb = 0;
                bool have_t = false;
                if (nipals && a > 0) {  // X_a = X_{a-1} - t_{a-1} p_{a-1}^T (first one out of place)
                    const T *tprev = Tm + (i64)(a - 1) * ldt;
                    const double *pprev = P + (i64)(a - 1) * K;
                    if (semi_fit) {  // wide matrices: deflation and score in one sweep (3NK instead of 4NK)
                        const i64 bytes = 2 * (i64)N * K * sizeof(T) + 2 * (i64)N * sizeof(T) + 2 * (i64)K * 8;
                        Scope s(c, PLS_HIP_FAM_DEFLATE, bytes);
                        const int rc = plsk::launch_deflate_score<T>(c->stream, c->num_cu, Xc, ldc, tsc, work, ldw, tsw,
                                                                     (int)(tiled_work ? WR : TR), N, K, tprev, pprev, v,
                                                                     Tm + (i64)a * ldt, sspart, (int)ssmax, &nss, Xc == work);
                        if (rc != 0) {
                            s.on = false;
                            return fail(c, PLS_HIP_ERR_DEVICE, "deflate+score launch failed");
                        }
                        LAUNCH_CHECK(c);
                        have_t = true;
                    }
                    if (!have_t) CHK(launch_deflate<T>(c, Xc, ldc, work, ldw, N, K, tprev, pprev));
                    Xc = work;
                    ldc = ldw;
                    tsc = tsw;
                    cur_tiled = tiled_work;
                }
                if (!have_t)
                    CHK(launch_xb<T>(c, Xc, ldc, N, K, v, K, 1, Tm + (i64)a * ldt, ldt, sspart, &nss));  // :419-420
                if (cur_tiled) {  // row-tile-major work buffer: the loading in tile addressing (:421)
                    Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + (i64)N * sizeof(T) + (i64)K * 8);
                    const T *ta = Tm + (i64)a * ldt;
                    const int xrc =
                        wide_cg == 64    ? plsk::launch_xty_tiled<T, 64>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                        : wide_cg == 128 ? plsk::launch_xty_tiled<T, 128>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                        : wide_cg == 256 ? plsk::launch_xty_tiled<T, 256>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                                         : plsk::launch_xty_tiled<T, 32>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb);
                    if (xrc != 0) {
                        s.on = false;
                        return fail(c, PLS_HIP_ERR_DEVICE, "tiled loading launch failed");
                    }
                    LAUNCH_CHECK(c);
                } else {
                    CHK(launch_xty<T>(c, Xc, ldc, Tm + (i64)a * ldt, ldt, N, K, 1, part, &nb));  // :421
                }
                CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            }
        } else {
            HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * (K + 1) * 8, c->stream));
        }
        if (upd_done) {  // the tail of the pass was the update as well (sharded: its push and gather too)
            if (want_push) ++*c->xep.seq;
        } else if (tail_used && want_push) {  // pushed from the tail of the pass: gather in the prologue of the update
            const unsigned long long seq = ++*c->xep.seq;
            const int par = (int)(seq & 1);
            plsk::XchgGather gx;
            gx.inbox = c->xep.inbox[c->xep.rank] + (i64)par * c->xep.n * plsk::XCHG_CAP;
            gx.flags = c->xep.flags[c->xep.rank] + par * c->xep.n;
            gx.n = c->xep.n; gx.cap = plsk::XCHG_CAP; gx.seq = seq;
            gx.status = c->xep.status; gx.host_status = c->xep.host_status; gx.limit = *c->xep.limit;
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, nip, &gx));
        } else {
            CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * (K + 1)));
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, nip));  // :427-433 and :403-416 of a+1
        }
    }
    if (B) {
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * A + (i64)M * A + (i64)K * M) * 8);
        const int nblk = (int)((L0 + plsk::WG - 1) / plsk::WG);
        hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, c->stream, R, Q,
                           K, M, A, B);
        LAUNCH_CHECK(c);
    }
    return replica_guard(c, W, P, Q, R, B, K, M, A);
}

}  // namespace
