// The whole fit of a SMALL single-response problem in one launch (BASELINE config 2: 60 x 401, one response).  The
// regular plan spends three launches per component; below a few hundred kilobytes of X their dispatch latency is the
// entire cost (config 2: 185 us for ten components, one CPU core does the fit in 103 us).  Here ONE workgroup of 1024
// threads keeps X in REGISTERS for the life of the fit and everything K-sized (XY, w, and the P and R columns the
// r recurrence walks) in LDS, and runs the reference's loop (src/pls.cpp:396-434) without leaving the kernel.
// M = 1 only: the direction is w = XY / |XY| (:404), no eigenproblem.  (A first form that called component_update_body
// for any M <= 32 spent 20 us per component in the scratch spills of that body next to the resident X.)
//
// Layout: lane = row inside a 64-row block, a wave = (row block rb, column slice s); the thread of row i and slice s
// holds X[i, s + j*S], j < TINY_RC.  With wps = ceil(N/64) waves per slice there are S = 16 / wps slices, so the
// kernel takes N <= 1024 and K <= S * TINY_RC (N <= 64: K <= 448; N <= 128: K <= 224; ...).
//   t_i   = sum over the S slices of the per-thread partial sum_j x[j] r[s + j*S]           (LDS, fixed order)
//   p_raw = column sums of x[j] * t_i over the rows: wave_multi_sum inside the wave, then the wps waves of a slice in order
// Every sum has a fixed order: equal inputs give equal bits.
#pragma once
#include "fused_kernels.hpp"  // raw buffer descriptors
#include "small_kernels.hpp"

namespace plsk {

constexpr int TINY_RC = 26;       // X values a thread keeps (52 VGPRs of the 128 a 1024-thread workgroup leaves per lane)
constexpr int TINY_HALF = TINY_RC / 2;
constexpr int TINY_KMAX = UPD_WAVES * TINY_RC;

struct TinyShape {
    int wps, S;  // waves per column slice, column slices
    __host__ __device__ explicit TinyShape(int N) : wps((N + WAVE - 1) / WAVE), S(wps > 0 && wps <= UPD_WAVES ? UPD_WAVES / wps : 0) {}
};

// one X value through the buffer descriptor: a lane offset that never changes plus a wave-uniform column offset, so the
// TINY_RC loads in flight cost no address registers; out of range (rows beyond N, columns beyond K) reads as 0
template <typename T>
__device__ __forceinline__ double tiny_ld(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (sizeof(T) == 8) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
        double d;
        __builtin_memcpy(&d, &raw, 8);
        return d;
    } else {
        const unsigned raw = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
        float f;
        __builtin_memcpy(&f, &raw, 4);
        return (double)f;
    }
}

// The K x A outputs (W, P, R) and the score columns are STORED through buffer descriptors as well: one lane offset (k or i) in a
// vector register, the column (a K, a ld) in the instruction's scalar offset.  With plain pointers the compiler keeps a 64-bit
// per-lane address for every output across the component loop -- and, at the 128 registers a 1024-thread workgroup leaves,
// spills them: 17 scratch reloads per component in tiny_fit_kernel (round 5).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(const void *p, i64 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, BUF_WORD3);
}
__device__ __forceinline__ void st_out(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double v) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 raw;
    __builtin_memcpy(&raw, &v, 8);
    __builtin_amdgcn_raw_buffer_store_b64(raw, r, voff, soff, 0);
}
template <typename T>
__device__ __forceinline__ void st_score(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double v) {
    if constexpr (sizeof(T) == 8) {
        st_out(r, voff, soff, v);
    } else {
        const float f = (float)v;
        unsigned raw;
        __builtin_memcpy(&raw, &f, 4);
        __builtin_amdgcn_raw_buffer_store_b32(raw, r, voff, soff, 0);
    }
}

constexpr size_t TINY_LDS_MAX = 96 * 1024;  // dynamic LDS: the P and R columns, 2 * K * A doubles
inline bool tiny_fit_covers(i64 N, int K, int M, int A, i64 ldx, size_t es) {
    if (M != 1 || N < 1 || N > UPD_THREADS || A > K || (i64)TINY_KMAX * ldx * (i64)es >= (1 << 30)) return false;  // 32-bit byte offsets
    const TinyShape sh((int)N);
    return sh.S >= 1 && K <= sh.S * TINY_RC && (size_t)2 * K * A * 8 <= TINY_LDS_MAX;
}

// block_sum (common.hpp) on lds_barrier
__device__ __forceinline__ double tiny_block_sum(double v, double *smem) {
    v = wave_sum(v);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    lds_barrier();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < UPD_WAVES; ++w) t += smem[w];
    return t;
}

// out[k] = sum over the rows of x[.] * f for the column k of every (slice, j); all threads of the workgroup call it; returns
// out[threadIdx.x] (0 beyond K): the caller that only needs its own column reads no LDS and needs no barrier behind the call
// LEAD = false: the caller has passed a barrier of its own since colp (and out) were last read
template <bool LEAD = true>
__device__ __forceinline__ double tiny_column_sums(const double (&x)[TINY_RC], double f, double (*colp)[TINY_RC], int K,
                                                   const TinyShape &shp, double *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (LEAD) lds_barrier();  // the previous use of colp has been read
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double vals[TINY_HALF];
#pragma unroll
        for (int j = 0; j < TINY_HALF; ++j) vals[j] = x[c * TINY_HALF + j] * f;
        bool valid = true;
        const int idx = wave_multi_sum<TINY_HALF, 32>(vals, lane, valid);
        if (valid) colp[wv][c * TINY_HALF + idx] = vals[0];
    }
    lds_barrier();
    double own = 0.0;  // (K <= TINY_KMAX < UPD_THREADS: thread k forms the sum of column k -- and is the thread that owns it afterwards)
    for (int k = threadIdx.x; k < K; k += UPD_THREADS) {
        const int s = k % shp.S, j = k / shp.S;
        double t = 0.0;
        for (int w = 0; w < shp.wps; ++w) t += colp[s * shp.wps + w][j];
        out[k] = t;
        own = t;
    }
    return own;
}

// X: N x K (ld ldx), Y: N x 1; W, P, R: K x A; Q: 1 x A; Tm: N x A (ld ldt); B: K x 1 or null.
// Dynamic LDS: 2 * K * A doubles.  grid = 1.
// FOLD MODE (fold_idx != null; grid = number of cross-validation folds, Model::cv_LOO / cv_LSO, src/pls.cpp:469-549):
// workgroup f fits the rows that are NOT in fold_idx[f*ts .. +ts) -- a held-out row takes part with y = 0 and t = 0,
// which is the fit without it -- and, because X stays whole in the registers, the score pass hands over x_i . r_a of the
// held-out rows for free: their residuals y_i - sum_{c<=a} (x_i . r_c) q_c go to E[(f*ts + j) + a*nobs] (the layout of
// pls_hip_cv_folds, M = 1).  W, P, Q, R, Tm, B are not written.
template <typename T>
__global__ __launch_bounds__(UPD_THREADS) void tiny_fit_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, int N,
                                                               int K, int A, double *__restrict__ W, double *__restrict__ P,
                                                               double *__restrict__ Q, double *__restrict__ R,
                                                               T *__restrict__ Tm, i64 ldt, double *__restrict__ B,
                                                               const i64 *__restrict__ fold_idx, int ts, i64 nobs,
                                                               double *__restrict__ E) {
    extern __shared__ double dyn[];
    double *Pl = dyn, *Rl = dyn + (i64)K * A;  // P[:, j], R[:, j] as they are produced
    __shared__ double tp[UPD_THREADS], colp[UPD_WAVES][TINY_RC], praw[TINY_KMAX], xy[TINY_KMAX], wl[TINY_KMAX], vsl[TINY_KMAX];
    __shared__ double cs[TINY_KMAX], ql[TINY_KMAX], sred[2 * UPD_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const TinyShape shp(N);
    const int s = wv / shp.wps, rb = wv % shp.wps, i = rb * WAVE + lane;
    const bool act = s < shp.S && i < N;
    const int k = tid;  // the column this thread owns in everything K-sized (K <= TINY_KMAX = 416)
    const bool kok = k < K;
    // r in slice-major order, vsl[s*TINY_RC + j] = r[s + j*S]: one LDS base address per thread and immediate
    // offsets (indexed as r[s + j*S] the compiler keeps 28 addresses per lane -- and spills them)
    const int slot = (k % shp.S) * TINY_RC + k / shp.S;

    const uint32_t nrec = (uint32_t)(((i64)(K - 1) * ldx + N) * (i64)sizeof(T));  // < 2^30 (tiny_fit_covers)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), (short)0, (int)nrec, BUF_WORD3);
    const uint32_t voff = act ? (uint32_t)((i + (i64)s * ldx) * (i64)sizeof(T)) : 0x80000000u;
    const uint32_t cstep = (uint32_t)((i64)shp.S * ldx * (i64)sizeof(T));
    double x[TINY_RC];
#pragma unroll
    for (int j = 0; j < TINY_RC; ++j) x[j] = tiny_ld<T>(rs, voff, (uint32_t)j * cstep);
    for (int c = tid; c < TINY_KMAX; c += UPD_THREADS) vsl[c] = 0.0;  // read (times x = 0) beyond K

    const bool fold = fold_idx != nullptr;
    // output descriptors (fold mode writes none of them: zero records, every store dropped)
    const i64 ka8 = fold ? 0 : (i64)K * A * 8;
    const __amdgpu_buffer_rsrc_t rW = out_rsrc(W, ka8), rP = out_rsrc(P, ka8), rR = out_rsrc(R, ka8);
    const __amdgpu_buffer_rsrc_t rT = out_rsrc(Tm, fold ? 0 : ((i64)(A - 1) * ldt + N) * (i64)sizeof(T));
    const uint32_t kof = kok ? (uint32_t)k * 8u : 0x80000000u;                                   // lane offset of entry k
    const uint32_t tof = (act && s == 0) ? (uint32_t)i * (uint32_t)sizeof(T) : 0x80000000u;      // ... of row i
    int hpos = -1;  // position of this thread's row in the fold's held-out list
    if (fold && act)
        for (int j = 0; j < ts; ++j)
            if (fold_idx[(i64)blockIdx.x * ts + j] == i) hpos = j;
    const bool held = hpos >= 0;
    const double yv = act ? (double)Y[i] : 0.0;
    double yhat = 0.0;
    tiny_column_sums(x, held ? 0.0 : yv, colp, K, shp, xy);  // XY = X^T Y (:396)
    lds_barrier();
    double xyk = kok ? xy[k] : 0.0;
    {  // w_0 = XY / |XY| (:404, :411), r_0 = w_0
        const double w = xyk / sqrt(tiny_block_sum(xyk * xyk, sred));
        st_out(rW, kof, 0, w);
        st_out(rR, kof, 0, w);
        if (kok) {
            Rl[k] = w;
            vsl[slot] = w;
        }
        const double c = wave_sum(kok ? w * xyk : 0.0);  // r_0^T XY: see the end of the loop
        if (lane == 0) sred[UPD_WAVES + wv] = c;
    }
    const double *vs = vsl + s * TINY_RC;
    for (int a = 0; a < A; ++a) {
        lds_barrier();  // r_a complete
        double acc = 0.0;  // t = X r (:419)
#pragma unroll
        for (int j = 0; j < TINY_RC; ++j) {
            acc = fma(x[j], vs[j], acc);
            if (j % 8 == 7) asm volatile("" ::: "memory");  // at most 8 values of r in registers next to the 26 of X
        }
        tp[tid] = acc;
        lds_barrier();
        double ti = 0.0;
        if (act)
            for (int q = 0; q < shp.S; ++q) ti += tp[(q * shp.wps + rb) * WAVE + lane];
        const double ui = ti;     // x_i . r_a, also for a held-out row
        if (held) ti = 0.0;       // ... which has no score in its fold's fit
        st_score<T>(rT, tof, (uint32_t)((i64)a * ldt * (i64)sizeof(T)), ti);
        {  // t^T t (:420): summed by waves here, added up behind the barriers of the column sums
            const double c = wave_sum((act && s == 0) ? ti * ti : 0.0);
            if (lane == 0) sred[wv] = c;
        }
        const double pk = tiny_column_sums<false>(x, ti, colp, K, shp, praw);           // X^T t (:427); colp, praw: last read two barriers ago
        double tt = 0.0;  // (the wave sums were stored before the barrier inside the column sums)
#pragma unroll
        for (int w2 = 0; w2 < UPD_WAVES; ++w2) tt += sred[w2];
        const double p = kok ? pk / tt : 0.0;                                                             // (:427)
        double rxy = 0.0;  // r^T XY: its wave sums were left in sred[UPD_WAVES ..] when r_a was formed
#pragma unroll
        for (int w2 = 0; w2 < UPD_WAVES; ++w2) rxy += sred[UPD_WAVES + w2];
        const double q = rxy / tt;                                                                        // q = r^T XY / tt (:428)
        st_out(rP, kof, (uint32_t)a * (uint32_t)K * 8u, p);
        if (kok) Pl[k + (i64)a * K] = p;
        if (tid == 0) {
            if (!fold) Q[a] = q;
            ql[a] = q;
        }
        if (held && s == 0) {  // residual of a held-out row with a+1 components
            yhat = fma(ui, q, yhat);
            E[((i64)blockIdx.x * ts + hpos) + (i64)a * nobs] = yv - yhat;
        }
        xyk -= (p * q) * tt;  // XY -= (p q^T) tt (:429)
        const int n = a + 1;
        if (n >= A) break;
        // |XY|^2 is one more wave's sum beside the p_j^T XY (no reduction of its own: two barriers fewer per component); the norm
        // then divides both w = XY / |XY| (:404, :411) and the c_j = p_j^T w (:415)
        if (kok) wl[k] = xyk;
        lds_barrier();
        for (int j = wv; j <= n; j += UPD_WAVES) {
            double c = 0.0;
            if (j < n) {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
            } else {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
            }
            c = wave_sum(c);
            if (lane == 0) (j < n ? cs[j] : sred[0]) = c;
        }
        lds_barrier();
        const double inv = 1.0 / sqrt(sred[0]);
        const double w = xyk * inv;
        st_out(rW, kof, (uint32_t)n * (uint32_t)K * 8u, w);
        double r = w;
        for (int j = 0; j < n; ++j) r -= (cs[j] * inv) * Rl[(kok ? k : 0) + (i64)j * K];  // the reference's order (:412-416)
        st_out(rR, kof, (uint32_t)n * (uint32_t)K * 8u, r);
        if (kok) {
            Rl[k + (i64)n * K] = r;
            vsl[slot] = r;
        }
        {  // r_n^T XY, the numerator of the next q (:428): summed by waves now, added up behind the barriers of the next t^T t
            const double c = wave_sum(kok ? r * xyk : 0.0);
            if (lane == 0) sred[UPD_WAVES + wv] = c;
        }
    }
    lds_barrier();
    if (B && kok && !fold) {  // B = R Q^T (:444-451)
        double b = 0.0;
        for (int a = 0; a < A; ++a) b = fma(Rl[k + (i64)a * K], ql[a], b);
        B[k] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same single launch for 2 <= M <= 8 responses (BASELINE config 1: toyX / toyY, 10 x 15, two responses, A = 2 --
// README.md:23 of the reference).  X stays in registers as above; XY (K x M) lives in LDS, column-major; the direction of a
// component is the dominant eigenvector of XY^T XY (src/pls.cpp:405-408): its M (M + 1) / 2 entries one wave per pair
// (K <= 448: seven products per lane), the eigenvector by the one-wave solver of the component update
// (dominant_eigvec_wave: repeated squaring, two polishing steps, largest entry positive), w = XY q / |XY q|.  Everything
// else is the single-response loop with M values of q per component.  A kernel of its own (template MM = 2, 4, 8 =
// M rounded up): the single-response instantiation keeps its registers (a shared kernel cost it 11 %).
// Fold mode as above, residuals per response: E[m*(nobs*A) + (f*ts + j) + a*nobs].
// Dynamic LDS: (2 K + M) A doubles (P, R, Q as they are produced).
// ---------------------------------------------------------------------------------------------------------------------
inline bool tiny_fit_m_covers(i64 N, int K, int M, int A, i64 ldx, size_t es) {
    if (M < 2 || M > 8 || N < 1 || N > UPD_THREADS || A > K || (i64)TINY_KMAX * ldx * (i64)es >= (1 << 30)) return false;
    const TinyShape sh((int)N);
    return sh.S >= 1 && K <= sh.S * TINY_RC && (size_t)(2 * K + M) * A * 8 <= TINY_LDS_MAX;
}

template <typename T, int MM>
__global__ __launch_bounds__(UPD_THREADS) void tiny_fit_m_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 ldy,
                                                                 int N, int K, int M, int A, int power_iters,
                                                                 double *__restrict__ W, double *__restrict__ P,
                                                                 double *__restrict__ Q, double *__restrict__ R,
                                                                 T *__restrict__ Tm, i64 ldt, double *__restrict__ B,
                                                                 const i64 *__restrict__ fold_idx, int ts, i64 nobs,
                                                                 double *__restrict__ E) {
    static_assert(MM * MM <= WAVE, "one wave solves the eigenproblem");
    extern __shared__ double dyn[];
    double *Pl = dyn, *Rl = dyn + (i64)K * A, *Ql = dyn + 2 * (i64)K * A;  // P[:, j], R[:, j], Q[:, j] as they are produced
    __shared__ double tp[UPD_THREADS], colp[UPD_WAVES][TINY_RC], praw[TINY_KMAX], xy[MM][TINY_KMAX], wl[TINY_KMAX], vsl[TINY_KMAX];
    __shared__ double cs[TINY_KMAX], sred[UPD_WAVES], Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM], qa[MM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const TinyShape shp(N);
    const int s = wv / shp.wps, rb = wv % shp.wps, i = rb * WAVE + lane;
    const bool act = s < shp.S && i < N;
    const int k = tid;
    const bool kok = k < K;
    const int slot = (k % shp.S) * TINY_RC + k / shp.S;

    const uint32_t nrec = (uint32_t)(((i64)(K - 1) * ldx + N) * (i64)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), (short)0, (int)nrec, BUF_WORD3);
    const uint32_t voff = act ? (uint32_t)((i + (i64)s * ldx) * (i64)sizeof(T)) : 0x80000000u;
    const uint32_t cstep = (uint32_t)((i64)shp.S * ldx * (i64)sizeof(T));
    double x[TINY_RC];
#pragma unroll
    for (int j = 0; j < TINY_RC; ++j) x[j] = tiny_ld<T>(rs, voff, (uint32_t)j * cstep);
    for (int c = tid; c < TINY_KMAX; c += UPD_THREADS) vsl[c] = 0.0;

    const bool fold = fold_idx != nullptr;
    int hpos = -1;
    if (fold && act)
        for (int j = 0; j < ts; ++j)
            if (fold_idx[(i64)blockIdx.x * ts + j] == i) hpos = j;
    const bool held = hpos >= 0;
    double yv[MM], yhat[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        yv[m] = (act && m < M) ? (double)Y[i + (i64)m * ldy] : 0.0;
        yhat[m] = 0.0;
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {  // XY = X^T Y (:396), a held-out row with y = 0; the columns beyond M: zeros
        if (m < M) {
            tiny_column_sums(x, held ? 0.0 : yv[m], colp, K, shp, xy[m]);
        } else if (kok) {
            xy[m][k] = 0.0;
        }
    }
    const double *vs = vsl + s * TINY_RC;
    for (int a = 0; a < A; ++a) {
        // ---- direction (:403-411): G = XY^T XY, one wave per pair (i <= j)
        lds_barrier();  // XY complete
        for (int pr = wv; pr < MM * (MM + 1) / 2; pr += UPD_WAVES) {
            int gi = 0, rem = pr;
            while (rem >= MM - gi) { rem -= MM - gi; ++gi; }
            const int gj = gi + rem;
            double g = 0.0;
            for (int kk = lane; kk < K; kk += WAVE) g = fma(xy[gi][kk], xy[gj][kk], g);
            g = wave_sum(g);
            if (lane == 0) { Gs[gi + gj * MM] = g; Gs[gj + gi * MM] = g; }
        }
        lds_barrier();
        if (wv == 0) dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
        lds_barrier();
        double wk = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) wk = fma(kok ? xy[m][k] : 0.0, qs[m], wk);  // w = XY q (:408)
        // |XY q|^2 is one more wave's sum beside the p_j^T (XY q) (no reduction of its own: two barriers fewer per component); the
        // norm then divides both w (:411) and the c_j = p_j^T w (:415)
        if (kok) wl[k] = wk;
        lds_barrier();
        for (int j = wv; j <= a; j += UPD_WAVES) {
            double c = 0.0;
            if (j < a) {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
            } else {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
            }
            c = wave_sum(c);
            if (lane == 0) (j < a ? cs[j] : sred[0]) = c;
        }
        lds_barrier();
        const double inv = 1.0 / sqrt(sred[0]);
        wk *= inv;
        if (kok && !fold) W[k + (i64)a * K] = wk;
        double r = wk;
        for (int j = 0; j < a; ++j) r -= (cs[j] * inv) * Rl[(kok ? k : 0) + (i64)j * K];  // the reference's order (:412-416)
        if (kok) {
            if (!fold) R[k + (i64)a * K] = r;
            Rl[k + (i64)a * K] = r;
            vsl[slot] = r;
        }
        lds_barrier();  // r_a complete
        // ---- score, loading (:419-427)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < TINY_RC; ++j) {
            acc = fma(x[j], vs[j], acc);
            if (j % 8 == 7) asm volatile("" ::: "memory");
        }
        tp[tid] = acc;
        lds_barrier();
        double ti = 0.0;
        if (act)
            for (int q = 0; q < shp.S; ++q) ti += tp[(q * shp.wps + rb) * WAVE + lane];
        const double ui = ti;
        if (held) ti = 0.0;
        if (act && s == 0 && !fold) Tm[i + (i64)a * ldt] = (T)ti;
        {  // t^T t (:420): summed by waves here, added up behind the barrier inside the column sums
            const double c = wave_sum((act && s == 0) ? ti * ti : 0.0);
            if (lane == 0) sred[wv] = c;
        }
        const double pk = tiny_column_sums<false>(x, ti, colp, K, shp, praw);     // X^T t (:421); colp, praw: last read barriers ago
        double tt = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < UPD_WAVES; ++w2) tt += sred[w2];
        const double p = kok ? pk / tt : 0.0;  // (:427) -- the thread's own column sum: no barrier behind the column sums
        if (kok) {
            if (!fold) P[k + (i64)a * K] = p;
            Pl[k + (i64)a * K] = p;
        }
        // ---- q = XY^T r / tt (:428): one wave per response
        if (wv < MM) {
            double c = 0.0;
            for (int kk = lane; kk < K; kk += WAVE) c = fma(Rl[kk + (i64)a * K], xy[wv][kk], c);
            c = wave_sum(c) / tt;
            if (lane == 0) {
                qa[wv] = c;
                if (wv < M) {
                    Ql[wv + (i64)a * M] = c;
                    if (!fold) Q[wv + (i64)a * M] = c;
                }
            }
        }
        lds_barrier();
        if (held && s == 0) {  // residuals of a held-out row with a + 1 components
#pragma unroll
            for (int m = 0; m < MM; ++m)
                if (m < M) {
                    yhat[m] = fma(ui, qa[m], yhat[m]);
                    E[(i64)m * nobs * A + ((i64)blockIdx.x * ts + hpos) + (i64)a * nobs] = yv[m] - yhat[m];
                }
        }
        if (kok) {
#pragma unroll
            for (int m = 0; m < MM; ++m) xy[m][k] -= (p * qa[m]) * tt;  // XY -= (p q^T) tt (:429)
        }
    }
    lds_barrier();
    if (B && kok && !fold)  // B = R Q^T (:444-447)
        for (int m = 0; m < M; ++m) {
            double b = 0.0;
            for (int a = 0; a < A; ++a) b = fma(Rl[k + (i64)a * K], Ql[m + (i64)a * M], b);
            B[k + (i64)m * K] = b;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// The smallest problems -- N <= 64 rows, K <= 32 columns, 1..8 responses: BASELINE config 1, the reference's README example
// (toyX / toyY, 10 x 15, two responses, two components) and every cross-validation fold of it -- as ONE WAVE.  A workgroup
// of 1024 threads spends such a fit in its barriers (31 us for config 1 in tiny_fit_m_kernel; one CPU core: 10.8 us); a
// single wave has no workgroup barrier at all: lane i holds row i of X in registers, lane k owns entry k of everything
// K-sized (in LDS, handed between the lanes behind a wave-level fence), every sum over rows or columns is a wave sum on
// the DPP / permlane path (wave_sum, wave_multi_sum).  Same operation sequence as the kernels above (src/pls.cpp:396-434,
// the direction by the one-wave eigen solver of the component update); fold mode = one wave per fold.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MICRO_K = 32;
inline bool micro_fit_covers(i64 N, int K, int M, int A, i64 ldx, size_t es) {
    return N >= 1 && N <= WAVE && K >= 1 && K <= MICRO_K && M >= 1 && M <= 8 && A >= 1 && A <= K && (i64)MICRO_K * ldx * (i64)es < (1ll << 31);
}

template <typename T, int MM>
__global__ __launch_bounds__(WAVE) void micro_fit_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 ldy, int N,
                                                         int K, int M, int A, int power_iters, double *__restrict__ W,
                                                         double *__restrict__ P, double *__restrict__ Q, double *__restrict__ R,
                                                         T *__restrict__ Tm, i64 ldt, double *__restrict__ B,
                                                         const i64 *__restrict__ fold_idx, int ts, i64 nobs, double *__restrict__ E) {
    static_assert(MM * MM <= WAVE && MM * (MM + 1) / 2 <= 36, "one wave solves the eigenproblem");
    constexpr int NP = MM * (MM + 1) / 2;
    __shared__ double xy[MM][MICRO_K], rl[MICRO_K], pl[MICRO_K], Pm[MICRO_K][MICRO_K], Rm[MICRO_K][MICRO_K], Ql[MICRO_K][MM];
    __shared__ double Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM];
    const int lane = threadIdx.x;  // row `lane` of X, and entry `lane` of everything K-sized
    const bool act = lane < N, kok = lane < K;
    double x[MICRO_K];
#pragma unroll
    for (int k = 0; k < MICRO_K; ++k) x[k] = (act && k < K) ? (double)X[lane + (i64)k * ldx] : 0.0;
    const bool fold = fold_idx != nullptr;
    int hpos = -1;
    if (fold && act)
        for (int j = 0; j < ts; ++j)
            if (fold_idx[(i64)blockIdx.x * ts + j] == lane) hpos = j;
    const bool held = hpos >= 0;
    double yv[MM], yhat[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        yv[m] = (act && m < M) ? (double)Y[lane + (i64)m * ldy] : 0.0;
        yhat[m] = 0.0;
    }
    rl[lane & (MICRO_K - 1)] = 0.0;  // (read, times x = 0, beyond K)
#pragma unroll
    for (int m = 0; m < MM; ++m) {  // XY = X^T Y (:396); a held-out row takes part with y = 0
        double vals[MICRO_K];
        const double f = held ? 0.0 : yv[m];
#pragma unroll
        for (int k = 0; k < MICRO_K; ++k) vals[k] = x[k] * f;
        bool valid = true;
        const int idx = wave_multi_sum<MICRO_K, 32>(vals, lane, valid);
        if (valid) xy[m][idx] = vals[0];
    }
    wave_lds_sync();
    double xyk[MM];  // this lane's entry of every column of XY
#pragma unroll
    for (int m = 0; m < MM; ++m) xyk[m] = kok ? xy[m][lane] : 0.0;
    for (int a = 0; a < A; ++a) {
        {  // ---- direction (:403-411): G = XY^T XY, its dominant eigenvector, w = XY q / |XY q|
            double g[NP];
#pragma unroll
            for (int i = 0, pr = 0; i < MM; ++i)
#pragma unroll
                for (int j = i; j < MM; ++j, ++pr) g[pr] = xyk[i] * xyk[j];
            bool valid = true;
            const int idx = wave_multi_sum<NP, 32>(g, lane, valid);
            if (valid) {
                int gi = 0, rem = idx;
                while (rem >= MM - gi) { rem -= MM - gi; ++gi; }
                const int gj = gi + rem;
                Gs[gi + gj * MM] = g[0];
                Gs[gj + gi * MM] = g[0];
            }
            wave_lds_sync();
            dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
        }
        double wk = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) wk = fma(xyk[m], qs[m], wk);
        wk = wk / sqrt(wave_sum(wk * wk));
        double rk = wk;
        for (int j = 0; j < a; ++j) {  // r = w - sum_j (p_j^T w) r_j, against the ORIGINAL w, in the reference's order (:412-416)
            const double cj = wave_sum(kok ? Pm[j][lane] * wk : 0.0);
            rk -= cj * (kok ? Rm[j][lane] : 0.0);
        }
        wave_lds_sync();  // (rl of the previous component has been read)
        if (kok) {
            rl[lane] = rk;
            Rm[a][lane] = rk;
            if (!fold) {
                W[lane + (i64)a * K] = wk;
                R[lane + (i64)a * K] = rk;
            }
        }
        wave_lds_sync();
        double ti = 0.0;  // t = X r (:419)
#pragma unroll
        for (int k = 0; k < MICRO_K; ++k) ti = fma(x[k], rl[k], ti);
        const double ui = ti;
        if (held) ti = 0.0;
        if (act && !fold) Tm[lane + (i64)a * ldt] = (T)ti;
        const double tt = wave_sum(ti * ti);  // (:420)
        {  // p = X^T t / tt (:421, :427)
            double vals[MICRO_K];
#pragma unroll
            for (int k = 0; k < MICRO_K; ++k) vals[k] = x[k] * ti;
            bool valid = true;
            const int idx = wave_multi_sum<MICRO_K, 32>(vals, lane, valid);
            if (valid) pl[idx] = vals[0];
        }
        wave_lds_sync();
        const double pk = kok ? pl[lane] / tt : 0.0;
        if (kok) {
            Pm[a][lane] = pk;
            if (!fold) P[lane + (i64)a * K] = pk;
        }
        double qv[MM];  // q = XY^T r / tt (:428)
#pragma unroll
        for (int m = 0; m < MM; ++m) qv[m] = wave_sum(rk * xyk[m]) / tt;
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (lane == m && m < M) {
                Ql[a][m] = qv[m];
                if (!fold) Q[m + (i64)a * M] = qv[m];
            }
        if (held) {  // residuals of a held-out row with a + 1 components
#pragma unroll
            for (int m = 0; m < MM; ++m)
                if (m < M) {
                    yhat[m] = fma(ui, qv[m], yhat[m]);
                    E[(i64)m * nobs * A + ((i64)blockIdx.x * ts + hpos) + (i64)a * nobs] = yv[m] - yhat[m];
                }
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) xyk[m] -= (pk * qv[m]) * tt;  // XY -= (p q^T) tt (:429)
    }
    wave_lds_sync();
    if (B && kok && !fold)  // B = R Q^T (:444-447)
        for (int m = 0; m < M; ++m) {
            double b = 0.0;
            for (int a = 0; a < A; ++a) b = fma(Rm[a][lane], Ql[a][m], b);
            B[lane + (i64)m * K] = b;
        }
}

}  // namespace plsk
