// launch_products.hpp -- typed launchers of the streaming products over X: X B / X v, X^T Y / X^T t, the stand-alone deflation, the fixed-order reduction of partial rows.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

namespace {

template <typename T>
bool vec_ok(const void *p, i64 ld, int vec) {
    return ((uintptr_t)p % (sizeof(T) * vec) == 0) && (ld % vec == 0);
}

// ---- geometry -------------------------------------------------------------------------
constexpr int XTY_KCMT = 32;  // accumulators per lane in xty_kernel
constexpr int DEFL_KC = 32;

struct XtyGeom {
    int G;    // row groups = number of partial rows (same for every m-tile of one product)
    int nkg;  // column groups of this m-tile
};
// All m-tiles of one X^T Y write the same number of partial rows G.  It is derived from the column groups of the
// product's FIRST tile (kc_first columns each; the first tile is the widest in m, i.e. the one with the most column
// groups): with G from the 32-column shape instead, the 8-response tile of config 4 ran 16,384 workgroups of 8 row
// chunks each and spent half its time in their 32 butterfly sums (1.0 ms = 2.1 TB/s, fp32 and fp64 alike).
XtyGeom xty_geom(i64 N, int K, int KC, int vec, int target_wgs, int kc_first) {
    XtyGeom g;
    g.nkg = (K + KC - 1) / KC;
    const int nkg32 = (K + kc_first - 1) / kc_first;
    const i64 nch = (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec);
    i64 G = std::max<i64>(1, target_wgs / nkg32);
    G = std::min<i64>(G, std::max<i64>(nch, 1));
    g.G = (int)G;
    return g;
}
// upper bound of partial rows any product of this fit can write
i64 max_partial_rows(pls_hip_context *c, i64 N, int K) {
    const int nkg32 = (K + XTY_KCMT - 1) / XTY_KCMT;
    const i64 nch = (N + plsk::WG - 1) / plsk::WG;  // vec = 1 is the worst case
    const i64 G = std::min<i64>(std::max<i64>(1, (8 * c->num_cu) / nkg32), std::max<i64>(nch, 1));
    return std::max<i64>(G, std::max<i64>(8 * (i64)c->num_cu, c->opt_fused_grid));
}

// ---- typed launchers --------------------------------------------------------------------
template <typename T, int VEC, int MT>
void launch_xb_t(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const double *Bm, i64 ldb,
                 int ncols, T *out, i64 ldo, double *sspart, int *nss) {
    const i64 per = (i64)plsk::WG * VEC;
    const int nblk = (int)((N + per - 1) / per);
    if (sspart && MT == 1) {
        hipLaunchKernelGGL((plsk::xb_kernel<T, VEC, 1, true>), dim3(nblk), dim3(plsk::WG), 0,
                           c->stream, X, ldx, N, K, Bm, ldb, ncols, out, ldo, sspart);
        *nss = nblk;
    } else {
        hipLaunchKernelGGL((plsk::xb_kernel<T, VEC, MT, false>), dim3(nblk), dim3(plsk::WG), 0,
                           c->stream, X, ldx, N, K, Bm, ldb, ncols, out, ldo, (double *)nullptr);
    }
}

// out(N x C) = X * Bm ; optionally sum of squares partials of column 0 (C must be 1 then)
template <typename T>
int launch_xb(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const double *Bm, i64 ldb,
              int C, T *out, i64 ldo, double *sspart, int *nss) {
    constexpr int FV = 16 / sizeof(T);
    bool wide = vec_ok<T>(X, ldx, FV) && vec_ok<T>(out, ldo, FV);
    // keep >= ~4 workgroups per CU in flight: narrow the per-lane access on short matrices
    if (wide && N / ((i64)FV * plsk::WG) < 4 * (i64)c->num_cu) wide = false;
    // One score column of a short, wide matrix: the rows alone give fewer workgroups than there are CUs -- split the
    // columns as well (xb_split_kernel); ~3 workgroups per CU, at least 128 columns each.
    // (not when the matrix-core kernel below takes the columns in one pass -- more than 8 with fp32 storage, more than 32 with
    // fp64 -- and has a workgroup for at least half the CUs: 32 columns on 131,072 x 4,096 fp32 were eight sweeps of this path,
    // 2.6 instead of 0.75 ms -- profiles/r4/products_scan.txt)
    const i64 rows_per_wg_many = (i64)(plsk::WG / plsk::WAVE) * 16 * FV;
    const bool many = C > (sizeof(T) == 4 ? 8 : 32) && vec_ok<T>(X, ldx, FV) &&
                      (N + rows_per_wg_many - 1) / rows_per_wg_many >= c->num_cu / 2;
    // 5..32 columns of a large matrix on the 4 x 4 x 4 MFMA kernels (xb_mfma4.hpp, xb_mfma4w.hpp): one sweep of X where the column-split
    // path below makes one per 4 columns (20,000 x 2,000, 20 columns: five sweeps, 0.13 of peak).  Not for a handful of columns
    // of a matrix with fewer than 128: its output is a third of the traffic and the VALU kernels are as fast.
    // Nor for a matrix with fewer 16 FV-row tiles than CUs (2,000 x 20,000: the column-split path below spreads it).
    const i64 xb4_tiles = (N + 16 * FV - 1) / (16 * FV);
    const bool xb4_ok = c->env.xb4 && C > 4 && vec_ok<T>(X, ldx, FV) && vec_ok<T>(out, ldo, FV) &&
                        36 * std::max(ldx, ldo) * (i64)sizeof(T) < ((i64)1 << 31) && (i64)N * K * (i64)sizeof(T) >= ((i64)32 << 20) &&
                        (K >= 128 || C > 8) && xb4_tiles >= (i64)c->num_cu;
    // the resident form (all of Bm in LDS, a wave per tile): at least two rounds of 16 tiles per workgroup
    auto xb4_resident = [&](int use) {
        const int ncg = (use + 3) / 4;
        return c->env.xb4 != 3 &&  // (PLS_HIP_XB4=3: the windowed form everywhere, for measurements)
               (size_t)plsk::xb4_kp(K, plsk::xb4_u(FV, ncg)) * plsk::xb4_stride(ncg) * 8 <= 152 * 1024 &&
               N / (16 * FV) >= (i64)2 * (plsk::XB4_WG / plsk::WAVE) * c->num_cu;
    };
    // the windowed form: not for fp64 storage with 8 columns or fewer (the VALU kernel holds them in one sweep at 0.72-0.74 of
    // peak, this one 0.68), not for fp32 beyond 20 (its 24-column form spills)
    // ... unless those 8 would go down the column-split path (two sweeps: 20,000 x 2,000, 8 columns 0.126 against 0.055 ms)
    const bool split_shape = K >= 1024 && (N + plsk::WG - 1) / plsk::WG <= (i64)(sizeof(T) == 4 ? 2 : 1) * c->num_cu;
    auto xb4_windowed = [&](int cols) { return sizeof(T) == 4 ? cols <= 20 : (cols > 8 || split_shape); };  // cols: ALL that remain
    const bool xb4_first = xb4_ok && (xb4_resident(std::min(C, sizeof(T) == 8 ? 32 : 24)) || xb4_windowed(C));
    // 1..4 columns of a TALL matrix (fitted values of a few responses; no sum of squares asked for): the resident form with one
    // column group -- three quarters of its MFMAs are padding and free; what counts is the tile walk (config 3, one column:
    // 0.71 -> 0.67 ms; fp32 0.41 -> 0.34)
    const bool xb4_few = c->env.xb4 && C <= 4 && !sspart && vec_ok<T>(X, ldx, FV) && vec_ok<T>(out, ldo, FV) && K >= 128 &&
                         36 * std::max(ldx, ldo) * (i64)sizeof(T) < ((i64)1 << 31) && (i64)N * K * (i64)sizeof(T) >= ((i64)32 << 20) &&
                         xb4_resident(4);
    // VERY short and wide (fewer 16 FV-row tiles than CUs: 2,000 x 20,000, a usual shape of the method), 5 columns or more: the
    // windowed MFMA kernel with the columns split over blockIdx.y as well, fp64 partial sums, xb_split_finish_kernel behind it --
    // one sweep of X for up to 32 columns where the split path below makes one per 4
    if (c->env.xb4 && C > 4 && N > 0 && K >= 1024 && xb4_tiles < (i64)c->num_cu && vec_ok<T>(X, ldx, FV) &&
        36 * ldx * (i64)sizeof(T) < ((i64)1 << 31) && (i64)N * K * (i64)sizeof(T) >= ((i64)32 << 20)) {
        const i64 ldp = (N + 63) / 64 * 64;
        const int fb = (int)((N + 63) / 64);
        bool ok = true;
        for (int c0 = 0; c0 < C && ok;) {
            const int use = std::min(C - c0, sizeof(T) == 8 ? 32 : 20);
            const int ncg = std::max(2, (use + 3) / 4), nc = 4 * ncg;
            const int kc = 1 << plsk::xb4w_kcl2(FV, ncg);
            int sw = 16;  // one tile per workgroup: its 16 waves share the columns of every window
            int kspl = (int)std::min<i64>((2 * (i64)c->num_cu + xb4_tiles - 1) / xb4_tiles, std::max(1, K / (2 * kc)));
            const int kper = ((K + kspl - 1) / kspl + kc - 1) / kc * kc;
            kspl = (K + kper - 1) / kper;
            const size_t lds = std::max((size_t)2 * kc * plsk::xb4_stride(ncg) * 8, (size_t)16 * FV * 64 * 8);
            const void *fn = nullptr;
#define XB4W_CASE(G_) case G_: fn = (const void *)plsk::xb_mfma4w_kernel<T, FV, G_>; break;
            switch (ncg) {
                XB4W_CASE(2) XB4W_CASE(3) XB4W_CASE(4) XB4W_CASE(5) XB4W_CASE(6)
                default:
                    if constexpr (sizeof(T) == 8) {
                        switch (ncg) { XB4W_CASE(7) XB4W_CASE(8) default: break; }
                    }
                    break;
            }
#undef XB4W_CASE
            if (!fn || kspl > 65535 || !plsk::raise_dynamic_lds(fn, (int)lds) ||
                ensure(c, c->xbpart, (size_t)kspl * nc * ldp * 8) != PLS_HIP_OK) {
                c->err.clear();
                ok = false;
                break;
            }
            double *xp = (double *)c->xbpart.p;
            const double *b = Bm + (i64)c0 * ldb;
            T *o = out + (i64)c0 * ldo;
            {
                Scope s(c, PLS_HIP_FAM_XB, (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8);
                int ncols = use, kp = kper;
                i64 lp = ldp;
                void *args[] = {(void *)&X, (void *)&ldx, (void *)&N, (void *)&K, (void *)&b, (void *)&ldb, (void *)&ncols, (void *)&o, (void *)&ldo,
                                (void *)&sw, (void *)&kp, (void *)&xp, (void *)&lp};
                if (hipLaunchKernel(fn, dim3((unsigned)xb4_tiles, (unsigned)kspl), dim3(plsk::XB4_WG), args, lds, c->stream) != hipSuccess) {
                    c->err = "kernel launch: xb_mfma4w (split)";
                    (void)hipGetLastError();
                    return PLS_HIP_ERR_DEVICE;
                }
                hipLaunchKernelGGL((plsk::xb_split_finish_kernel<T>), dim3(fb, use), dim3(plsk::WG), 0, c->stream, (const double *)xp, ldp, kspl, nc, N, o,
                                   ldo, (double *)nullptr);
                LAUNCH_CHECK(c);
            }
            c0 += use;
        }
        if (ok) return PLS_HIP_OK;
    }
    if (N > 0 && K >= 1024 && !many && !xb4_first && !xb4_few) {
        const bool v2 = vec_ok<T>(X, ldx, FV) && (N + (i64)FV * plsk::WG - 1) / ((i64)FV * plsk::WG) >= 8;
        const i64 per = (i64)plsk::WG * (v2 ? FV : 1);
        const i64 rg = (N + per - 1) / per;
        // taken while one row per lane cannot give every CU a workgroup (fp32: two -- its 4-byte accesses stream worse);
        // measured per shape, tools/xb_split_sweep.py: beyond that the row-parallel kernel is as fast or faster
        const i64 rg1 = (N + plsk::WG - 1) / plsk::WG;
        if (rg1 <= (i64)(sizeof(T) == 4 ? 2 : 1) * c->num_cu) {
            int KS = (int)std::min<i64>(K / 128, (3 * (i64)c->num_cu + rg - 1) / rg);
            const int kper = (K + KS - 1) / KS;
            KS = (K + kper - 1) / kper;
            const i64 ldp = (N + 63) / 64 * 64;
            const int mt = C > 2 ? 4 : (C > 1 ? 2 : 1);  // columns per sweep of X
            if (KS >= 2 && KS <= 65535 && ensure(c, c->xbpart, (size_t)KS * mt * ldp * 8) == PLS_HIP_OK) {
                double *xp = (double *)c->xbpart.p;
                const int fb = (int)((N + 63) / 64);
                for (int c0 = 0; c0 < C; c0 += mt) {
                    const int use = std::min(mt, C - c0);
                    const double *b = Bm + (i64)c0 * ldb;
                    Scope s(c, PLS_HIP_FAM_XB, (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8);
                    const dim3 g((unsigned)rg, (unsigned)KS), blk(plsk::WG);
#define XS_CASE(V_, M_) hipLaunchKernelGGL((plsk::xb_split_kernel<T, V_, M_>), g, blk, 0, c->stream, X, ldx, N, K, kper, b, ldb, use, xp, ldp)
                    if (v2) { if (mt == 4) XS_CASE(FV, 4); else if (mt == 2) XS_CASE(FV, 2); else XS_CASE(FV, 1); }
                    else { if (mt == 4) XS_CASE(1, 4); else if (mt == 2) XS_CASE(1, 2); else XS_CASE(1, 1); }
#undef XS_CASE
                    LAUNCH_CHECK(c);
                    hipLaunchKernelGGL((plsk::xb_split_finish_kernel<T>), dim3(fb, use), blk, 0, c->stream, (const double *)xp, ldp, KS, mt,
                                       N, out + (i64)c0 * ldo, ldo, C == 1 ? sspart : (double *)nullptr);
                    LAUNCH_CHECK(c);
                }
                if (C == 1 && sspart && nss) *nss = fb;
                return PLS_HIP_OK;
            }
            c->err.clear();
        }
    }
    int c0 = 0;
    while (c0 < C) {
        const int rem = C - c0;
        const double *b = Bm + (i64)c0 * ldb;
        T *o = out + (i64)c0 * ldo;
        const int cap = wide ? (FV == 2 ? 32 : 8) : 32;  // fp32 x 4 rows per lane: 8 columns = 32 fp64 accumulators
        if (sizeof(T) == 8 && rem > 32 && vec_ok<T>(X, ldx, FV)) {
            // fp64 storage beyond the 32 columns a pass of the LDS-staged VALU kernel holds: up to 64 per pass on the matrix
            // cores with Bm in LDS (1,048,576 x 512, 64 columns: one pass instead of two of 1.27 ms)
            if constexpr (sizeof(T) == 8) {
                const int use = std::min(rem, 64);
                const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
                Scope s(c, PLS_HIP_FAM_XB, bytes);
                const i64 per = (i64)(plsk::WG / plsk::WAVE) * 16 * FV;  // rows per workgroup
                const dim3 grid((unsigned)((N + per - 1) / per)), blk(plsk::WG);
                if (use > 48)
                    hipLaunchKernelGGL((plsk::xb_mfma_lds_kernel<T, FV, 4>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
                else
                    hipLaunchKernelGGL((plsk::xb_mfma_lds_kernel<T, FV, 3>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
                LAUNCH_CHECK(c);
                c0 += use;
                continue;
            }
        }
        if ((rem > 4 && xb4_ok) || xb4_few) {
            // 5..32 columns (fp32 storage: ..24) with all of Bm in LDS: the 4 x 4 x 4 MFMA form, columns padded to 4 (xb_mfma4.hpp)
            const int use = std::min(rem, sizeof(T) == 8 ? 32 : 24);
            const int ncg = (use + 3) / 4;
            const i64 ntiles = (N + 16 * FV - 1) / (16 * FV);
            const size_t lds = (size_t)plsk::xb4_kp(K, plsk::xb4_u(FV, ncg)) * plsk::xb4_stride(ncg) * 8;
            const int waves = plsk::XB4_WG / plsk::WAVE;
            if (xb4_resident(use)) {
                const void *fn = nullptr;
#define XB4_CASE(G_) case G_: fn = (const void *)plsk::xb_mfma4_kernel<T, FV, G_>; break;
                switch (ncg) {
                    XB4_CASE(1) XB4_CASE(2) XB4_CASE(3) XB4_CASE(4) XB4_CASE(5) XB4_CASE(6)
                    default:
                        if constexpr (sizeof(T) == 8) {
                            switch (ncg) { XB4_CASE(7) XB4_CASE(8) default: break; }
                        }
                        break;
                }
#undef XB4_CASE
                if (fn && plsk::raise_dynamic_lds(fn, (int)lds)) {
                    const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
                    Scope s(c, PLS_HIP_FAM_XB, bytes);
                    const unsigned grid = (unsigned)std::min<i64>(c->num_cu, (ntiles + waves - 1) / waves);
                    int ncols = use;
                    void *args[] = {(void *)&X, (void *)&ldx, (void *)&N, (void *)&K, (void *)&b, (void *)&ldb, (void *)&ncols, (void *)&o, (void *)&ldo};
                    if (hipLaunchKernel(fn, dim3(grid), dim3(plsk::XB4_WG), args, lds, c->stream) != hipSuccess) {
                        c->err = "kernel launch: xb_mfma4";
                        (void)hipGetLastError();
                        return PLS_HIP_ERR_DEVICE;
                    }
                    c0 += use;
                    continue;
                }
            }
        }
        if (rem > 4 && xb4_ok && xb4_windowed(rem)) {
            // the same product where Bm does not fit in LDS or the matrix has too few row tiles for a wave each: Bm in windows,
            // the waves of a workgroup = tile slots x sub-windows (xb_mfma4w.hpp)
            const int use = std::min(rem, sizeof(T) == 8 ? 32 : 20);
            const int ncg = (use + 3) / 4;
            const i64 ntiles = (N + 16 * FV - 1) / (16 * FV);
            const unsigned grid = (unsigned)std::min<i64>(c->num_cu, ntiles);
            const i64 tpw = (ntiles + grid - 1) / grid;
            int tw = 16;
            for (int cand : {8, 4, 2})
                if ((tpw + cand - 1) / cand * cand < (tpw + tw - 1) / tw * tw) tw = cand;
            int sw = 16 / tw;
            const int kc = 1 << plsk::xb4w_kcl2(FV, ncg);
            const size_t lds = std::max((size_t)2 * kc * plsk::xb4_stride(ncg) * 8, (size_t)16 * FV * 64 * 8);
            const void *fn = nullptr;
#define XB4W_CASE(G_) case G_: fn = (const void *)plsk::xb_mfma4w_kernel<T, FV, G_>; break;
            switch (ncg) {
                XB4W_CASE(2) XB4W_CASE(3) XB4W_CASE(4) XB4W_CASE(5) XB4W_CASE(6)
                default:
                    if constexpr (sizeof(T) == 8) {
                        switch (ncg) { XB4W_CASE(7) XB4W_CASE(8) default: break; }
                    }
                    break;
            }
#undef XB4W_CASE
            if (fn && plsk::raise_dynamic_lds(fn, (int)lds)) {
                const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
                Scope s(c, PLS_HIP_FAM_XB, bytes);
                int ncols = use;
                int kp0 = 0;
                double *nopart = nullptr;
                i64 lp0 = 0;
                void *args[] = {(void *)&X, (void *)&ldx, (void *)&N, (void *)&K, (void *)&b, (void *)&ldb, (void *)&ncols, (void *)&o, (void *)&ldo, (void *)&sw,
                                (void *)&kp0, (void *)&nopart, (void *)&lp0};
                if (hipLaunchKernel(fn, dim3(grid), dim3(plsk::XB4_WG), args, lds, c->stream) != hipSuccess) {
                    c->err = "kernel launch: xb_mfma4w";
                    (void)hipGetLastError();
                    return PLS_HIP_ERR_DEVICE;
                }
                c0 += use;
                continue;
            }
        }
        if (sizeof(T) == 4 && rem > 8 && vec_ok<T>(X, ldx, FV)) {
            // fp32 storage, many columns: up to 32 per pass on the matrix cores (xb_mfma_kernel) -- the LDS-staged
            // VALU kernel below holds only 8 columns of fp64 accumulators per pass at 4 rows per lane.  (For fp64
            // storage, where it takes 32 columns per pass, it is the faster one: 0.86 vs 1.04 ms at 20 columns.)
            const int use = std::min(rem, 32);
            const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
            Scope s(c, PLS_HIP_FAM_XB, bytes);
            const i64 per = (i64)(plsk::WG / plsk::WAVE) * 16 * FV;  // rows per workgroup
            const dim3 grid((unsigned)((N + per - 1) / per)), blk(plsk::WG);
            if (use > 16)
                hipLaunchKernelGGL((plsk::xb_mfma_lds_kernel<T, FV, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
            else
                hipLaunchKernelGGL((plsk::xb_mfma_lds_kernel<T, FV, 1>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
            LAUNCH_CHECK(c);
            c0 += use;
            continue;
        }
        if (rem > 4) {
            // many columns: Bm through LDS, up to `cap` columns per pass over X; the tile is the column
            // count rounded up to a multiple of 4 (every extra column costs VEC fp64 FMAs per element)
            const int use = std::min(rem, cap);
            const int mtc = (use + 3) & ~3;
            const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
            Scope s(c, PLS_HIP_FAM_XB, bytes);
            // fp64, 13..20 columns on a large matrix: two row packs per lane (one LDS read of a B value feeds 4 FMAs)
            const bool two = wide && FV == 2 && mtc >= 16 && mtc <= 20 && N >= (i64)c->num_cu * 4 * plsk::WG * FV * 2;
            const i64 per = (i64)plsk::WG * (wide ? FV : 1) * (two ? 2 : 1);
            const dim3 grid((unsigned)((N + per - 1) / per)), blk(plsk::WG);
            if (two) {
                if constexpr (FV == 2) {
                    switch (mtc) {
                        case 16: hipLaunchKernelGGL((plsk::xb_wide_kernel<T, 2, 16, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo); break;
                        default: hipLaunchKernelGGL((plsk::xb_wide_kernel<T, 2, 20, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo); break;
                    }
                }
                LAUNCH_CHECK(c);
                c0 += use;
                continue;
            }
#define XW_CASE(V, M_) hipLaunchKernelGGL((plsk::xb_wide_kernel<T, V, M_>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo)
#define XW_SWITCH(V)                                   \
    switch (mtc) {                                     \
        case 8: XW_CASE(V, 8); break;                  \
        case 12: XW_CASE(V, 12); break;                \
        case 16: XW_CASE(V, 16); break;                \
        case 20: XW_CASE(V, 20); break;                \
        case 24: XW_CASE(V, 24); break;                \
        case 28: XW_CASE(V, 28); break;                \
        default: XW_CASE(V, 32); break;                \
    }
            if (wide) {
                if constexpr (FV == 2) {
                    XW_SWITCH(FV)
                } else {
                    XW_CASE(FV, 8);
                }
            } else {
                XW_SWITCH(1)
            }
#undef XW_SWITCH
#undef XW_CASE
            LAUNCH_CHECK(c);
            c0 += use;
            continue;
        }
        const int mt = rem > 2 ? 4 : rem > 1 ? 2 : 1;
        const int use = std::min(mt, rem);
        const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
        Scope s(c, PLS_HIP_FAM_XB, bytes);
#define XB_CASE(V, M_) launch_xb_t<T, V, M_>(c, X, ldx, N, K, b, ldb, use, o, ldo, sspart, nss)
        if (wide) {
            if (mt == 4) XB_CASE(FV, 4); else if (mt == 2) XB_CASE(FV, 2); else XB_CASE(FV, 1);
        } else {
            if (mt == 4) XB_CASE(1, 4); else if (mt == 2) XB_CASE(1, 2); else XB_CASE(1, 1);
        }
#undef XB_CASE
        LAUNCH_CHECK(c);
        c0 += use;
    }
    return PLS_HIP_OK;
}

template <typename T, int VEC, int KC, int MT>
void launch_xty_t(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
                  int m0, double *part, const XtyGeom &g) {
    hipLaunchKernelGGL((plsk::xty_kernel<T, VEC, KC, MT>), dim3(g.G, g.nkg), dim3(plsk::WG), 0,
                       c->stream, X, ldx, Y, ldy, N, K, M, m0, part);
}

// part[G][K*M] = per-row-group partials of X^T Y; returns G through *nb
template <typename T>
int launch_xty(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
               double *part, int *nb) {
    constexpr int FV = 16 / sizeof(T);
    const bool wide = vec_ok<T>(X, ldx, FV) && vec_ok<T>(Y, ldy, FV);
    // workgroups per CU aimed at: 8; 4 with 8-response tiles, whose 32 butterfly sums per workgroup want longer walks
    // (config 4, fp32: 0.55 ms at 4, 0.59 at 8, 0.77 at 32 -- tools/xty_m8.py)
    const int target = (M >= 8 ? 4 : 8) * c->num_cu;
    int m0 = 0;
    const int kc_first = XTY_KCMT / ((M >= 8) ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1)));
    while (m0 < M) {
        const int mt = (M - m0 >= 8) ? 8 : (M - m0 >= 4 ? 4 : (M - m0 >= 2 ? 2 : 1));
        // 8 responses: 8 columns per workgroup (64 accumulators per lane) -- the Y packs of a row chunk are loaded once
        // per column group, so 4 columns meant twice as many bytes of Y as of X through L2 (1.0 ms = 2.1 TB/s at config 4)
        const int kc = XTY_KCMT / mt;
        const XtyGeom g = xty_geom(N, K, kc, wide ? FV : 1, target, kc_first);
        *nb = g.G;
        const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * mt * sizeof(T) + (i64)K * mt * 8;
        Scope s(c, PLS_HIP_FAM_XTY, bytes);
#define XTY_CASE(V, KC_, M_) launch_xty_t<T, V, KC_, M_>(c, X, ldx, Y, ldy, N, K, M, m0, part, g)
        // 8 responses in one go: the copy-and-product sweep WITHOUT its copy (retile_xty_kernel, dst == nullptr) -- the Y block of
        // a tile goes through LDS once instead of 8 packs per lane and 4 columns (xty8_kernel: 0.49 / 0.60 of peak in fp32 / fp64)
        int rx_nb = 0;
        // ... and 1, 2 or 4 responses up to 2,048 columns: the tile walk streams X at 0.85 of peak where xty_kernel's row chunks
        // reach 0.76 (config 3, one response: 0.708 -> 0.631 ms; 131,072 x 4,096 is faster on the chunks: tools/probe/xty1.py)
        if (wide && ((mt == 8 && M == 8) || (mt == M && K <= 2048)) && m0 == 0 && K >= 256 && plsk::cols_aligned<T>(X, ldx) &&
            plsk::launch_retile_xty<T, 32>(c->stream, c->num_cu, X, ldx, Y, ldy, (T *)nullptr, 0, 0, 0, N, K, M, part,
                                           (int)max_partial_rows(c, N, K), &rx_nb) == 0) {
            *nb = rx_nb;
        } else if (wide && mt == 8 && K % 4 == 0) {
            hipLaunchKernelGGL((plsk::xty8_kernel<T, FV>), dim3(g.G, g.nkg), dim3(plsk::WG), 0, c->stream, X, ldx, Y, ldy, N, K,
                               M, m0, part);
        } else if (wide) {
            if (mt == 8) XTY_CASE(FV, 4, 8); else if (mt == 4) XTY_CASE(FV, 8, 4);
            else if (mt == 2) XTY_CASE(FV, 16, 2); else XTY_CASE(FV, 32, 1);
        } else {
            if (mt == 8) XTY_CASE(1, 4, 8); else if (mt == 4) XTY_CASE(1, 8, 4);
            else if (mt == 2) XTY_CASE(1, 16, 2); else XTY_CASE(1, 32, 1);
        }
#undef XTY_CASE
        LAUNCH_CHECK(c);
        m0 += mt;
    }
    return PLS_HIP_OK;
}

template <typename T>
int launch_deflate(pls_hip_context *c, const T *src, i64 lds, T *dst, i64 ldd, i64 N, int K,
                   const T *t, const double *p) {
    constexpr int FV = 16 / sizeof(T);
    const bool wide = vec_ok<T>(src, lds, FV) && vec_ok<T>(dst, ldd, FV) && vec_ok<T>(t, FV, FV);
    const int nkg = (K + DEFL_KC - 1) / DEFL_KC;
    const int vec = wide ? FV : 1;
    const i64 nch = (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec);
    const i64 G = std::min<i64>(std::max<i64>(nch, 1), std::max<i64>(1, (16 * c->num_cu) / nkg));
    const i64 bytes = 2 * (i64)N * K * sizeof(T) + (i64)N * sizeof(T) + (i64)K * 8;
    Scope s(c, PLS_HIP_FAM_DEFLATE, bytes);
    const i64 nrb = (N + (i64)plsk::WG * FV - 1) / ((i64)plsk::WG * FV);
    if (wide && K <= 65535 && nrb >= 1 && nrb < (1ll << 31)) {  // one 4 KB column piece per workgroup
        hipLaunchKernelGGL((plsk::deflate_piece_kernel<T, FV>), dim3((unsigned)nrb, (unsigned)K), dim3(plsk::WG), 0,
                           c->stream, src, lds, dst, ldd, N, t, p);
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    if (wide)
        hipLaunchKernelGGL((plsk::deflate_kernel<T, FV, DEFL_KC>), dim3((unsigned)G, nkg),
                           dim3(plsk::WG), 0, c->stream, src, lds, dst, ldd, N, K, t, p);
    else
        hipLaunchKernelGGL((plsk::deflate_kernel<T, 1, DEFL_KC>), dim3((unsigned)G, nkg),
                           dim3(plsk::WG), 0, c->stream, src, lds, dst, ldd, N, K, t, p);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

int launch_reduce(pls_hip_context *c, const double *part, int nb, int L, const double *sspart,
                  int nss, double *red, i64 out_stride = 0) {
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)nb * L + nss + (i64)plsk::RED_SLICES * (L + 1)) * 8);
    hipLaunchKernelGGL(plsk::reduce_partials_kernel, dim3((L + 63) / 64, plsk::RED_SLICES),
                       dim3(plsk::WG), 0, c->stream, part, nb, L, sspart, nss, red, out_stride);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

}  // namespace
