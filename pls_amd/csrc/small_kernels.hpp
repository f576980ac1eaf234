// The K-sized bookkeeping of one component, kept on the device so the A-loop never returns
// to the host (src/pls.cpp:403-416 and :427-433), plus coefficients (:444-447).
// One workgroup of 1024 threads; every rank of a sharded fit runs it on bit-identical
// inputs (the all-reduced partials) and therefore derives bit-identical w, r, p, q.
#pragma once
#include "common.hpp"
#include "exchange_kernels.hpp"  // XchgGather
#include "fused_kernels.hpp"     // raw buffer loads with cache-policy bits
#include "update_m1.hpp"
#include "stream_kernels.hpp"

namespace plsk {

// fixed-order sum of the RED_SLICES slices of reduced value j (slice stride LP)
__device__ __forceinline__ double red_sum(const double *red, int LP, int j) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < RED_SLICES; ++i) s += red[(i64)i * LP + j];
    return s;
}
// the same for a runtime slice count (1 for the per-fold cross-validation loop)
__device__ __forceinline__ double red_sum_n(const double *red, int nsl, int LP, int j) {
    if (nsl == RED_SLICES) return red_sum(red, LP, j);
    double s = 0.0;
    for (int i = 0; i < nsl; ++i) s += red[(i64)i * LP + j];
    return s;
}

// LDS scratch of component_update_body (one workgroup of UPD_THREADS)
struct UpdShared {
    double sred[1024 / 64];
    double qs[32];
    double Gs[32 * 32], Bs[32 * 32], Cs[32 * 32];
};

constexpr int UPD_THREADS = 1024;
constexpr int UPD_WAVES = UPD_THREADS / WAVE;
constexpr int MMAX = 32;  // responses supported by the on-device direction solve (m > 1)

// Dominant eigenvector of G = S^T S (M x M, symmetric PSD) by power iteration carried out as
// repeated squaring: B_0 = G/tr G, B_{j+1} = B_j^2 / tr(B_j^2) -> v1 v1^T with the error
// contracting as (lambda2/lambda1)^(2^j), stopped at the fixed point (at most `iters`
// squarings); then two plain power steps q <- G q / |G q| on the
// original G polish the vector to working precision.  This replaces the reference's general
// EigenSolver + find_dominant_ev (src/pls.cpp:406-408, :113-141).  Sign (left open by the
// reference): largest-|.| entry positive, lowest index on ties.
// Called by all threads of the workgroup; G, Bm, Cm: M*M doubles of LDS; qv: M doubles.
__device__ inline void dominant_eigvec_lds(double *G, double *Bm, double *Cm, double *qv, int M,
                                           int iters) {
    const int tid = threadIdx.x;
    const int a = tid % MMAX, b = tid / MMAX;  // (row, col) when a < M && b < M
    const bool act = (a < M) && (b < M);
    double tr = 0.0;
    for (int c = 0; c < M; ++c) tr += G[c + c * M];
    if (act) Bm[a + b * M] = G[a + b * M] / tr;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (act) {
            double s = 0.0;
            for (int c = 0; c < M; ++c) s = fma(Bm[a + c * M], Bm[c + b * M], s);
            Cm[a + b * M] = s;
        }
        __syncthreads();
        double t2 = 0.0;
        for (int c = 0; c < M; ++c) t2 += Cm[c + c * M];
        int same = 1;
        if (act) {
            const double nb = Cm[a + b * M] / t2, ob = Bm[a + b * M];
            // fixed point: B has become v1 v1^T.  B is trace-normalised (entries <= 1), so entries that are
            // still shrinking towards zero count as converged once they are below 1e-18 in absolute terms
            same = fabs(nb - ob) <= 4.0e-16 * fabs(nb) + 1.0e-18;
            Bm[a + b * M] = nb;
        }
        if (__syncthreads_and(same)) break;  // uniform exit; also the barrier that publishes Bm
    }
    // column with the largest diagonal entry spans the dominant direction
    int best = 0;
    double bd = Bm[0];
    for (int c = 1; c < M; ++c)
        if (Bm[c + c * M] > bd) { bd = Bm[c + c * M]; best = c; }
    if (tid < M) qv[tid] = Bm[tid + best * M];
    __syncthreads();
    for (int pol = 0; pol < 2; ++pol) {
        double s = 0.0;
        if (tid < M)
            for (int c = 0; c < M; ++c) s = fma(G[tid + c * M], qv[c], s);
        __syncthreads();
        if (tid < M) qv[tid] = s;
        __syncthreads();
        double n2 = 0.0;
        for (int c = 0; c < M; ++c) n2 = fma(qv[c], qv[c], n2);
        const double inv = 1.0 / sqrt(n2);
        __syncthreads();
        if (tid < M) qv[tid] = s * inv;
        __syncthreads();
    }
    int big = 0;
    for (int c = 1; c < M; ++c)
        if (fabs(qv[c]) > fabs(qv[big])) big = c;
    const double sgn = (qv[big] < 0.0) ? -1.0 : 1.0;
    __syncthreads();
    if (tid < M) qv[tid] *= sgn;
    __syncthreads();
}

// ---- few responses (2 <= M <= 8): thread-per-k form of the component update ----------------------
// A wave that walks a K-vector on its own is latency-bound (K/64 dependent round trips to L2: 31 us
// for the 36 pair products of M = 8, K = 4096).  Here every thread owns the entries k = tid + i*1024
// of all M columns of XY instead, so the M loads of an entry are independent, the deflation of XY
// (:429) and the Gram matrix XY^T XY (:405) come out of the same registers, and the sums over k are
// one butterfly per value plus a fixed-order sum over the 16 waves.

__device__ __forceinline__ void wave_lds_sync() {  // LDS hand-over between the lanes of ONE wave
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// dominant_eigvec_lds for MM*MM <= 64, run by ONE wave without workgroup barriers.  G, Bm, Cm: MM x MM
// (stride MM, zero-padded beyond the M used rows/columns -- the zeros are exact under every step).
template <int MM>
__device__ __forceinline__ void dominant_eigvec_wave(const double *G, double *Bm, double *Cm, double *qv, int iters) {
    static_assert(MM * MM <= WAVE, "one matrix entry per lane");
    const int lane = threadIdx.x & 63;
    const int a = lane % MM, b = lane / MM;
    const bool act = lane < MM * MM;
    double tr = 0.0;
#pragma unroll
    for (int c = 0; c < MM; ++c) tr += G[c + c * MM];
    double cur = act ? G[a + b * MM] / tr : 0.0;
    if (act) Bm[a + b * MM] = cur;
    wave_lds_sync();
    for (int it = 0; it < iters; ++it) {
        double s = 0.0;
        if (act) {
#pragma unroll
            for (int c = 0; c < MM; ++c) s = fma(Bm[a + c * MM], Bm[c + b * MM], s);
            Cm[a + b * MM] = s;
        }
        wave_lds_sync();
        double t2 = 0.0;
#pragma unroll
        for (int c = 0; c < MM; ++c) t2 += Cm[c + c * MM];
        const double nb = s / t2;
        const bool moved = act && !(fabs(nb - cur) <= 4.0e-16 * fabs(nb) + 1.0e-18);  // see dominant_eigvec_lds
        cur = nb;
        if (act) Bm[a + b * MM] = nb;
        wave_lds_sync();
        if (__builtin_amdgcn_ballot_w64(moved) == 0) break;
    }
    int best = 0;
    double bd = Bm[0];
#pragma unroll
    for (int c = 1; c < MM; ++c)
        if (Bm[c + c * MM] > bd) { bd = Bm[c + c * MM]; best = c; }
    if (lane < MM) qv[lane] = Bm[lane + best * MM];
    wave_lds_sync();
    for (int pol = 0; pol < 2; ++pol) {
        double s = 0.0;
        if (lane < MM) {
#pragma unroll
            for (int c = 0; c < MM; ++c) s = fma(G[lane + c * MM], qv[c], s);
        }
        wave_lds_sync();
        if (lane < MM) qv[lane] = s;
        wave_lds_sync();
        double n2 = 0.0;
#pragma unroll
        for (int c = 0; c < MM; ++c) n2 = fma(qv[c], qv[c], n2);
        const double inv = 1.0 / sqrt(n2);
        wave_lds_sync();
        if (lane < MM) qv[lane] = s * inv;
        wave_lds_sync();
    }
    int big = 0;
#pragma unroll
    for (int c = 1; c < MM; ++c)
        if (fabs(qv[c]) > fabs(qv[big])) big = c;
    const double sgn = (qv[big] < 0.0) ? -1.0 : 1.0;
    wave_lds_sync();
    if (lane < MM) qv[lane] *= sgn;
    wave_lds_sync();
}

// Sums of N per-lane values over the 64 lanes of a wave in N-1 + (levels left) exchanges instead of 6 N:
// at every level the two partner lanes split the remaining values between them, each keeping the sum of
// its half.  Returns the index of the value whose total ends in g[0] of this lane (valid == false: none).
// Every total is a fixed butterfly tree over the lanes, so equal inputs give equal bits.
// All exchanges are VALU operations (common.hpp; no LDS crossbar).  Levels: lanes 32 apart (v_permlane32_swap), 16 apart
// (v_permlane16_swap), then row_mirror, row_half_mirror, lane ^ 2, lane ^ 1 -- the two mirrors pair lanes that differ in
// SEVERAL low bits, which is only legal while those bits have not been used to split the values yet, hence this order.
// The two swap levels need no select at all: the swap itself hands the kept half and the received half to the adder.
template <int N, int LEVEL>
__device__ __forceinline__ int wave_multi_sum_levels(double *g, int lane, bool &valid) {
    if constexpr (LEVEL == 6) {
        return 0;
    } else {
        constexpr int H = (N + 1) / 2;
        constexpr int BIT = LEVEL == 0 ? 32 : LEVEL == 1 ? 16 : LEVEL == 2 ? 8 : LEVEL == 3 ? 4 : LEVEL == 4 ? 2 : 1;
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            double lo = g[i], hi = (N > 1) ? ((i + H < N) ? g[i + H] : 0.0) : g[i];
            if constexpr (LEVEL <= 1) {
                permlane_swap_f64<LEVEL == 1>(lo, hi);
                g[i] = lo + hi;
            } else {
                const double keep = (N > 1 && up) ? hi : lo, send = (N > 1 && up) ? lo : hi;
                constexpr int CTRL = LEVEL == 2 ? 0x140 : LEVEL == 3 ? 0x141 : LEVEL == 4 ? 0x4E : 0xB1;
                g[i] = keep + dpp_mov_f64<CTRL>(send);
            }
        }
        const int idx = wave_multi_sum_levels<H, LEVEL + 1>(g, lane, valid) + ((N > 1 && up) ? H : 0);
        valid = valid && (idx < N);
        return idx;
    }
}

// MASK: kept from the first form of this function (the lane distance of the first level); always 32
template <int N, int MASK>
__device__ __forceinline__ int wave_multi_sum(double *g, int lane, bool &valid) {
    static_assert(MASK == 32, "whole-wave sums");
    return wave_multi_sum_levels<N, 0>(g, lane, valid);
}

// g[pair(i,j)] += x[i] x[j] for i <= j, pairs numbered row by row (compile-time indices: g stays in registers)
template <int MM>
__device__ __forceinline__ void gram_add(const double (&x)[MM], double (&g)[MM * (MM + 1) / 2]) {
#pragma unroll
    for (int i = 0; i < MM; ++i)
#pragma unroll
        for (int j = i; j < MM; ++j) {
            const int idx = i * MM - (i * (i - 1)) / 2 + (j - i);
            g[idx] = fma(x[i], x[j], g[idx]);
        }
}

// Everything of component_update_body up to and including w_n for 2 <= M <= MM.  Returns after the
// workgroup barrier that publishes w_n.
template <int MM>
__device__ __forceinline__ void narrow_update(const double *__restrict__ red, int nsl, double *__restrict__ XY,
                                     double *__restrict__ P, double *__restrict__ Q,
                                     const double *__restrict__ R, double *__restrict__ W, int K, int M, int A,
                                     int a, int power_iters, UpdShared &sh) {
    constexpr int NP = MM * (MM + 1) / 2;
    static_assert(UPD_WAVES * NP <= 32 * 32, "wave partials live in UpdShared::Bs");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool last = (a + 1 >= A);
    // columns m >= M alias column M-1 (loaded unconditionally, value replaced by 0): a conditional load
    // would put a wait on the join of every branch and serialise the M loads of an entry
    i64 co[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) co[m] = (i64)(m < M ? m : M - 1) * K;
    double g[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) g[i] = 0.0;

    if (a >= 0) {
        const double tt = red_sum_n(red, nsl, K + 1, K);
        const double *ra = R + (i64)a * K;
        double qa[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) qa[m] = 0.0;
        double *pa = P + (i64)a * K;
#pragma unroll 2
        for (int k = tid; k < K; k += UPD_THREADS) {  // p = X^T t / tt (:427);  q = XY^T r / tt (:428)
            pa[k] = red_sum_n(red, nsl, K + 1, k) / tt;
            const double rk = ra[k];
            double x[MM];
#pragma unroll
            for (int m = 0; m < MM; ++m) x[m] = XY[k + co[m]];
#pragma unroll
            for (int m = 0; m < MM; ++m) qa[m] = fma(rk, (m < M) ? x[m] : 0.0, qa[m]);
        }
        {
            bool valid = true;
            const int idx = wave_multi_sum<MM, 32>(qa, lane, valid);
            if (valid) sh.Cs[wv * MM + idx] = qa[0];
        }
        __syncthreads();
        if (tid < MM) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < UPD_WAVES; ++w) s += sh.Cs[w * MM + tid];
            s /= tt;
            sh.qs[tid] = s;
            if (tid < M) Q[tid + (i64)a * M] = s;
        }
        __syncthreads();
        double qv[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) qv[m] = sh.qs[m];
        // p is re-read from P (this thread's own store) rather than kept: with the 36 Gram accumulators live the
        // loop has no registers to spare, and short of them the compiler issues its loads one at a time
        for (int k = tid; k < K; k += UPD_THREADS) {  // XY -= (p q^T) tt (:429), Gram of the new XY
            const double p = pa[k];
            double x[MM];
#pragma unroll
            for (int m = 0; m < MM; ++m) x[m] = XY[k + co[m]];
#pragma unroll
            for (int m = 0; m < MM; ++m) x[m] = (m < M) ? x[m] - (p * qv[m]) * tt : 0.0;
#pragma unroll
            for (int m = 0; m < MM; ++m)
                if (m < M) XY[k + (i64)m * K] = x[m];
            if (!last) gram_add<MM>(x, g);
        }
    } else {
        for (int k = tid; k < K; k += UPD_THREADS) {  // prologue: XY = reduced X^T Y (or already in place)
            double x[MM];
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                x[m] = 0.0;
                if (m < M) {
                    if (red) {
                        x[m] = red_sum_n(red, nsl, K * M, k + m * K);
                        XY[k + (i64)m * K] = x[m];
                    } else {
                        x[m] = XY[k + (i64)m * K];
                    }
                }
            }
            gram_add<MM>(x, g);
        }
    }
    if (last) return;

    double *part = sh.Bs;  // [wave][pair]
    {
        bool valid = true;
        const int idx = wave_multi_sum<NP, 32>(g, lane, valid);
        if (valid) part[wv * NP + idx] = g[0];
    }
    __syncthreads();
    if (tid < NP) {
        int i = 0, rem = tid;
        while (rem >= MM - i) { rem -= MM - i; ++i; }
        const int j = i + rem;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < UPD_WAVES; ++w) s += part[w * NP + tid];
        sh.Gs[i + j * MM] = s;
        sh.Gs[j + i * MM] = s;
    }
    __syncthreads();
    if (wv == 0) dominant_eigvec_wave<MM>(sh.Gs, sh.Bs, sh.Cs, sh.qs, power_iters);
    __syncthreads();
    double qv[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) qv[m] = sh.qs[m];
    double *wn = W + (i64)(a + 1) * K;
    // this thread's first 4 entries of w stay in registers between the product and the normalisation
    // (all of them when K <= 4096); their loads are unconditional on a clamped index so that all 4 x M
    // are in flight together
    double ss = 0.0;
    double wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + i * UPD_THREADS, kc = k < K ? k : K - 1;
        double x[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) x[m] = XY[kc + co[m]];
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) s = fma((m < M) ? x[m] : 0.0, qv[m], s);
        wl[i] = (k < K) ? s : 0.0;
        ss = fma(wl[i], wl[i], ss);
    }
    for (int k = tid + 4 * UPD_THREADS; k < K; k += UPD_THREADS) {  // w = XY q (:408)
        double x[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) x[m] = XY[k + co[m]];
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) s = fma((m < M) ? x[m] : 0.0, qv[m], s);
        wn[k] = s;
        ss = fma(s, s, ss);
    }
    ss = block_sum<UPD_WAVES>(ss, sh.sred);
    const double nrm = sqrt(ss);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + i * UPD_THREADS;
        if (k < K) wn[k] = wl[i] / nrm;
    }
    for (int k = tid + 4 * UPD_THREADS; k < K; k += UPD_THREADS) wn[k] = wn[k] / nrm;  // own element: no hazard
}

// a = index of the component whose pass just finished (-1: prologue, red holds X^T Y).
// red is the RED_SLICES-sliced output of reduce_partials_kernel (summed over ranks when sharded).
//   a >= 0 : red = [X^T t (K), t^T t];  p = red/tt (:427) -> P[:,a];  q = XY^T r_a / tt (:428)
//            -> Q[:,a];  XY -= (p q^T) tt (:429).
//   then, when a+1 < A: w from XY (:403-411) -> W[:,a+1];  r (:412-416) -> R[:,a+1];
//   vnext = r (KERNEL algo: next pass is X r) or w (NIPALS algo: next pass is X_a w).
//   (r stays inside the loop for the NIPALS algo too although its passes never read it: q must be formed with r_a.
//   Forming it with w_a instead -- equal in exact arithmetic, the deflated XY being orthogonal to every earlier r_j --
//   and computing R after the loop was tried in round 2: once A exceeds the numerical rank B = R Q^T stops reproducing
//   least squares (A = K = 200: relative error 26 instead of 1e-9), so the reference's form is kept.)
// Dynamic LDS: A doubles (the p_j^T w inner products).
// Body shared by component_update_kernel (one fit) and cv_folds_kernel (one fold per workgroup).
// red == nullptr with a < 0: XY already holds the covariance.  nsl: slices in red.
__device__ __forceinline__ void component_update_body(const double *__restrict__ red, int nsl,
                                             double *__restrict__ XY, double *__restrict__ W,
                                             double *__restrict__ P, double *__restrict__ Q,
                                             double *__restrict__ R, double *__restrict__ vnext, int K,
                                             int M, int A, int a, int nipals, int power_iters,
                                             int split_rotate, double *cs, UpdShared &sh) {
    double *sred = sh.sred, *qs = sh.qs, *Gs = sh.Gs, *Bs = sh.Bs, *Cs = sh.Cs;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    // Single response (the headline configurations) with K <= 4096: every thread keeps its <= 4 entries
    // of XY in registers from the Y-loading update through to the next weight vector; the q dot product
    // is one workgroup reduction instead of one wave walking K, and XY is not re-read.
    const bool fast = (M == 1 && K <= 4 * UPD_THREADS && a >= 0);
    const bool narrow = (M >= 2 && M <= 8);
    const double *Rq = R;
    if (narrow) {
        if (M <= 2)
            narrow_update<2>(red, nsl, XY, P, Q, Rq, W, K, M, A, a, power_iters, sh);
        else if (M <= 4)
            narrow_update<4>(red, nsl, XY, P, Q, Rq, W, K, M, A, a, power_iters, sh);
        else
            narrow_update<8>(red, nsl, XY, P, Q, Rq, W, K, M, A, a, power_iters, sh);
    } else if (fast) {
        // one response: the arithmetic shared with the tail of the fused pass (update_m1.hpp) -- q, p, the XY deflation, w, r
        if (K <= UPD1_VTHREADS)
            update_m1<UPD_THREADS, 1>([&](int j) { return red_sum_n(red, nsl, K + 1, j); }, XY, W, P, Q, R, vnext, K, A, a, nipals,
                                      split_rotate, cs, sred, (double *)nullptr);
        else
            update_m1<UPD_THREADS, 4>([&](int j) { return red_sum_n(red, nsl, K + 1, j); }, XY, W, P, Q, R, vnext, K, A, a, nipals,
                                      split_rotate, cs, sred, (double *)nullptr);
        return;
    } else if (a < 0) {
        if (red)
            for (int j = tid; j < K * M; j += UPD_THREADS) XY[j] = red_sum_n(red, nsl, K * M, j);
    } else {
        const double tt = red_sum_n(red, nsl, K + 1, K);
        const double *ra = Rq + (i64)a * K;
        for (int m = wv; m < M; m += UPD_WAVES) {  // q_m = (r^T XY[:,m]) / tt
            const double *xm = XY + (i64)m * K;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // 4 independent chains: 8 loads in flight per lane
            int k = lane;
            for (; k + 3 * WAVE < K; k += 4 * WAVE) {
                s0 = fma(ra[k], xm[k], s0);
                s1 = fma(ra[k + WAVE], xm[k + WAVE], s1);
                s2 = fma(ra[k + 2 * WAVE], xm[k + 2 * WAVE], s2);
                s3 = fma(ra[k + 3 * WAVE], xm[k + 3 * WAVE], s3);
            }
            for (; k < K; k += WAVE) s0 = fma(ra[k], xm[k], s0);
            double s = wave_sum((s0 + s1) + (s2 + s3));
            if (lane == 0) {
                qs[m] = s / tt;
                Q[m + (i64)a * M] = s / tt;
            }
        }
        __syncthreads();
        for (int k = tid; k < K; k += UPD_THREADS) {
            const double p = red_sum_n(red, nsl, K + 1, k) / tt;
            P[k + (i64)a * K] = p;
            for (int m = 0; m < M; ++m) XY[k + (i64)m * K] -= (p * qs[m]) * tt;
        }
    }
    const int n = a + 1;
    if (n >= A) return;
    __syncthreads();  // XY complete (same workgroup: its own global stores are visible)

    double *wn = W + (i64)n * K;
    if (narrow) {
        // w_n already written above
    } else if (M == 1) {
        double ss = 0.0;
        for (int k = tid; k < K; k += UPD_THREADS) ss = fma(XY[k], XY[k], ss);
        ss = block_sum<UPD_WAVES>(ss, sred);
        const double nrm = sqrt(ss);
        for (int k = tid; k < K; k += UPD_THREADS) wn[k] = XY[k] / nrm;
    } else {
        const int npairs = M * (M + 1) / 2;
        for (int pr = wv; pr < npairs; pr += UPD_WAVES) {  // G = XY^T XY, one wave per (i<=j)
            int i = 0, rem = pr;
            while (rem >= M - i) { rem -= M - i; ++i; }
            const int j = i + rem;
            const double *xi = XY + (i64)i * K, *xj = XY + (i64)j * K;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = lane;
            for (; k + 3 * WAVE < K; k += 4 * WAVE) {
                s0 = fma(xi[k], xj[k], s0);
                s1 = fma(xi[k + WAVE], xj[k + WAVE], s1);
                s2 = fma(xi[k + 2 * WAVE], xj[k + 2 * WAVE], s2);
                s3 = fma(xi[k + 3 * WAVE], xj[k + 3 * WAVE], s3);
            }
            for (; k < K; k += WAVE) s0 = fma(xi[k], xj[k], s0);
            const double s = wave_sum((s0 + s1) + (s2 + s3));
            if (lane == 0) { Gs[i + j * M] = s; Gs[j + i * M] = s; }
        }
        __syncthreads();
        dominant_eigvec_lds(Gs, Bs, Cs, qs, M, power_iters);
        double ss = 0.0;
        for (int k = tid; k < K; k += UPD_THREADS) {  // w = XY q (:408)
            double s = 0.0;
            for (int m = 0; m < M; ++m) s = fma(XY[k + (i64)m * K], qs[m], s);
            wn[k] = s;
            ss = fma(s, s, ss);
        }
        ss = block_sum<UPD_WAVES>(ss, sred);
        const double nrm = sqrt(ss);
        for (int k = tid; k < K; k += UPD_THREADS) wn[k] = wn[k] / nrm;  // own element: no hazard
    }
    if (split_rotate) return;  // r is finished by rotate_dots_kernel + rotate_apply_kernel
    __syncthreads();
    // The two loops below touch O(n*K) values from one workgroup: latency-bound unless several
    // independent loads are in flight per lane, hence the 4-way unrolling.
    for (int j = wv; j < n; j += UPD_WAVES) {  // c_j = P[:,j]^T w  (against the ORIGINAL w, :415)
        const double *pj = P + (i64)j * K;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = lane;
        for (; k + 3 * WAVE < K; k += 4 * WAVE) {
            const double a0 = pj[k], a1 = pj[k + WAVE], a2 = pj[k + 2 * WAVE], a3 = pj[k + 3 * WAVE];
            s0 = fma(a0, wn[k], s0);
            s1 = fma(a1, wn[k + WAVE], s1);
            s2 = fma(a2, wn[k + 2 * WAVE], s2);
            s3 = fma(a3, wn[k + 3 * WAVE], s3);
        }
        for (; k < K; k += WAVE) s0 = fma(pj[k], wn[k], s0);
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) cs[j] = s;
    }
    __syncthreads();
    double *rn = R + (i64)n * K;
    for (int k = tid; k < K; k += UPD_THREADS) {
        const double w = wn[k];
        double r = w;
        int j = 0;
        for (; j + 4 <= n; j += 4) {  // same subtraction order as the reference, loads issued together
            const double r0 = R[k + (i64)j * K], r1 = R[k + (i64)(j + 1) * K];
            const double r2 = R[k + (i64)(j + 2) * K], r3 = R[k + (i64)(j + 3) * K];
            r -= cs[j] * r0;
            r -= cs[j + 1] * r1;
            r -= cs[j + 2] * r2;
            r -= cs[j + 3] * r3;
        }
        for (; j < n; ++j) r -= cs[j] * R[k + (i64)j * K];
        rn[k] = r;
        vnext[k] = nipals ? w : r;
    }
}

// Out-of-line copy for callers that need the body more than once (cv_folds_kernel): inlining it twice next
// to their own loops overflows the 128 registers a 1024-thread workgroup leaves per lane.
__device__ __noinline__ void component_update_call(const double *red, int nsl, double *XY, double *W, double *P,
                                                   double *Q, double *R, double *vnext, int K, int M, int A,
                                                   int a, int nipals, int power_iters, int split_rotate,
                                                   double *cs, UpdShared &sh) {
    component_update_body(red, nsl, XY, W, P, Q, R, vnext, K, M, A, a, nipals, power_iters, split_rotate, cs, sh);
}

__global__ __launch_bounds__(UPD_THREADS) void component_update_kernel(
    const double *__restrict__ red, double *__restrict__ XY, double *__restrict__ W,
    double *__restrict__ P, double *__restrict__ Q, double *__restrict__ R,
    double *__restrict__ vnext, int K, int M, int A, int a, int nipals, int power_iters,
    int split_rotate) {
    extern __shared__ double cs[];  // [A]
    __shared__ UpdShared sh;
    component_update_body(red, RED_SLICES, XY, W, P, Q, R, vnext, K, M, A, a, nipals, power_iters, split_rotate,
                          cs, sh);
}

// The same update with the GATHER of a sharded fit's collective as its prologue (the push rode in the tail of the pass:
// fused_kernels.hpp, slice_tail): wait until every member's flag shows this collective, add the members' vectors in rank
// order -- every member the same bits -- into slice 0 of `red`, and run the body on that one slice.  A wait beyond the time
// limit raises the status words and poisons the sums with NaN, as xchg_gather_kernel does.
constexpr int AUX_SYS = 17;  // sc0 sc1: system scope (the peers' writes into fine-grained memory)
__global__ __launch_bounds__(UPD_THREADS) void component_update_gather_kernel(
    const XchgGather gx, double *red, double *__restrict__ XY, double *__restrict__ W, double *__restrict__ P,
    double *__restrict__ Q, double *__restrict__ R, double *__restrict__ vnext, int K, int M, int A, int a, int nipals,
    int power_iters, int split_rotate) {
    extern __shared__ double cs[];  // [A]
    __shared__ UpdShared sh;
    __shared__ int ok;
    const int tid = threadIdx.x;
    if (tid == 0) ok = (*gx.status == 0);  // an earlier wait of this member timed out: do not wait again
    __syncthreads();
    if (tid < gx.n && ok) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(gx.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < gx.seq) {
            if (wall_clock64() - t0 > gx.limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the vectors behind the flags
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (!ok && tid == 0) {
        *gx.status = 1;
        __hip_atomic_store(gx.host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    {
        const int L = K + 1;
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        for (int j = tid; j < L; j += UPD_THREADS) {
            double x[XCHG_MAX];
#pragma unroll
            for (int m = 0; m < XCHG_MAX; ++m) {  // all members' values in flight together
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(gx.inbox + (i64)(m < gx.n ? m : 0) * gx.cap), (short)0, L * 8, BUF_WORD3);
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)j * 8u, 0, AUX_SYS);
                __builtin_memcpy(&x[m], &raw, 8);
            }
            double sum = x[0];
#pragma unroll
            for (int m = 1; m < XCHG_MAX; ++m)
                if (m < gx.n) sum += x[m];  // rank order, as xchg_gather_kernel
            red[j] = ok ? sum : __builtin_nan("");
        }
    }
    __syncthreads();  // (the workgroup's own stores: visible to all its waves behind the barrier)
    component_update_body(red, 1, XY, W, P, Q, R, vnext, K, M, A, a, nipals, power_iters, split_rotate, cs, sh);
}

// KERNEL_TYPE2 / GRAM (src/pls.cpp:422-425): the update with type2_pack_kernel as its prologue -- given praw = XX r it forms
// tt = r^T praw (the sum of type2_pack_kernel: 1024 strided sums, wave sums, waves in order) and presents [praw, tt] as the
// one slice the body reads.  Two launches per component (symv, this) instead of three.
__global__ __launch_bounds__(UPD_THREADS) void component_update_type2_kernel(
    const double *__restrict__ praw, const double *r, double *red, double *__restrict__ XY, double *__restrict__ W,
    double *__restrict__ P, double *__restrict__ Q, double *__restrict__ R, double *vnext, int K, int M, int A, int a,
    int power_iters, int split_rotate) {
    extern __shared__ double cs[];  // [A]
    __shared__ UpdShared sh;
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += UPD_THREADS) {
        const double p = praw[k];
        s = fma(r[k], p, s);
        red[k] = p;
    }
    s = block_sum<UPD_WAVES>(s, sh.sred);
    if (threadIdx.x == 0) red[K] = s;
    __syncthreads();  // (the workgroup's own stores: visible to all its waves behind the barrier)
    component_update_body(red, 1, XY, W, P, Q, R, vnext, K, M, A, a, 0, power_iters, split_rotate, cs, sh);
}

// Multi-workgroup form of the r update (src/pls.cpp:412-416) for large n*K, where one workgroup
// would be latency-bound on the 2*n*K values of P and R:
//   rotate_dots_kernel   grid n      : cs[j] = P[:,j]^T w_n
//   rotate_apply_kernel  grid K/256  : r_n = w_n - sum_{j<n} cs[j] R[:,j]   (j ascending, as the reference)
__global__ __launch_bounds__(WG) void rotate_dots_kernel(const double *__restrict__ P,
                                                         const double *__restrict__ W, int K, int n,
                                                         double *__restrict__ cs) {
    __shared__ double sm[WG / WAVE];
    const int j = blockIdx.x;
    const double *pj = P + (i64)j * K, *wn = W + (i64)n * K;
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += WG) s = fma(pj[k], wn[k], s);
    s = block_sum<WG / WAVE>(s, sm);
    if (threadIdx.x == 0) cs[j] = s;
}

__global__ __launch_bounds__(WG) void rotate_apply_kernel(const double *__restrict__ W,
                                                          double *__restrict__ R,
                                                          const double *__restrict__ cs,
                                                          double *__restrict__ vnext, int K, int n,
                                                          int nipals) {
    const int k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    const double w = W[k + (i64)n * K];
    double r = w;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        const double r0 = R[k + (i64)j * K], r1 = R[k + (i64)(j + 1) * K];
        const double r2 = R[k + (i64)(j + 2) * K], r3 = R[k + (i64)(j + 3) * K];
        r -= cs[j] * r0;
        r -= cs[j + 1] * r1;
        r -= cs[j + 2] * r2;
        r -= cs[j + 3] * r3;
    }
    for (; j < n; ++j) r -= cs[j] * R[k + (i64)j * K];
    R[k + (i64)n * K] = r;
    vnext[k] = nipals ? w : r;
}

// praw = XX r for the symmetric K x K matrix XX (src/pls.cpp:424): one wave per output, reading
// COLUMN k of XX (= row k by symmetry) contiguously.  grid = ceil(K/4) workgroups of 4 waves.
__global__ __launch_bounds__(WG) void symv_kernel(const double *__restrict__ XX, const double *__restrict__ r,
                                                  int K, double *__restrict__ praw) {
    const int k = blockIdx.x * (WG / WAVE) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= K) return;
    const double *col = XX + (i64)k * K;
    double s0 = 0.0, s1 = 0.0;
    int j = lane;
    for (; j + WAVE < K; j += 2 * WAVE) {
        s0 = fma(col[j], r[j], s0);
        s1 = fma(col[j + WAVE], r[j + WAVE], s1);
    }
    if (j < K) s0 = fma(col[j], r[j], s0);
    const double s = wave_sum(s0 + s1);
    if (lane == 0) praw[k] = s;
}

// KERNEL_TYPE2 (src/pls.cpp:422-425): given praw = XX r, emit what component_update_kernel expects
// from a pass -- red slice 0 = [praw (K), tt = r^T XX r], the other slices zero.
__global__ __launch_bounds__(UPD_THREADS) void type2_pack_kernel(const double *__restrict__ praw,
                                                                const double *__restrict__ r, int K,
                                                                double *__restrict__ red) {
    __shared__ double sm[UPD_WAVES];
    double s = 0.0;
    for (int k = threadIdx.x; k < K; k += UPD_THREADS) {
        const double p = praw[k];
        s = fma(r[k], p, s);
        red[k] = p;
        for (int i = 1; i < RED_SLICES; ++i) red[(i64)i * (K + 1) + k] = 0.0;
    }
    s = block_sum<UPD_WAVES>(s, sm);
    if (threadIdx.x == 0) {
        red[K] = s;
        for (int i = 1; i < RED_SLICES; ++i) red[(i64)i * (K + 1) + K] = 0.0;
    }
}

// mode 0: out = mean = (sum of slices)/n ;  mode 1: out = sd = sqrt(SST/(n-1)), SST = 0 when n < 2
// (src/pls.cpp:69-83)
__global__ __launch_bounds__(WG) void colstat_finish_kernel(const double *__restrict__ red, int K,
                                                            double n, int mode, double *__restrict__ out) {
    const int k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    const double s = red_sum(red, K, k);
    out[k] = (mode == 0) ? s / n : sqrt((n < 2.0 ? 0.0 : s) / (n - 1.0));
}

// out[j] = sum of the RED_SLICES slices (stand-alone X^T Y entry point)
__global__ __launch_bounds__(WG) void sum_slices_kernel(const double *__restrict__ red, int L,
                                                        double *__restrict__ out) {
    const int j = blockIdx.x * WG + threadIdx.x;
    if (j < L) out[j] = red_sum(red, L, j);
}

// out[j] += sum of the RED_SLICES slices (X^T X / X^T Y accumulated row block by row block while X streams in)
__global__ __launch_bounds__(WG) void accumulate_slices_kernel(const double *__restrict__ red, int L,
                                                               double *__restrict__ out) {
    const int j = blockIdx.x * WG + threadIdx.x;
    if (j < L) out[j] += red_sum(red, L, j);
}

// red slice 0 = src, the other slices zero: a locally finished sum presented in the sliced layout the reducer and
// the K-sized kernels expect
__global__ __launch_bounds__(WG) void fill_slices_kernel(const double *__restrict__ src, int L, double *__restrict__ red) {
    const int j = blockIdx.x * WG + threadIdx.x;
    if (j >= L) return;
    red[j] = src[j];
#pragma unroll
    for (int i = 1; i < RED_SLICES; ++i) red[(i64)i * L + j] = 0.0;
}

// B[k + m*K] = sum_{j<c} R[k + j*K] * Q[m + j*M]     Model::coefficients, src/pls.cpp:444-447
__global__ __launch_bounds__(WG) void coefficients_kernel(const double *__restrict__ R,
                                                          const double *__restrict__ Q, int K,
                                                          int M, int c, double *__restrict__ B) {
    const i64 idx = (i64)blockIdx.x * WG + threadIdx.x;
    if (idx >= (i64)K * M) return;
    const int k = (int)(idx % K), m = (int)(idx / K);
    double s = 0.0;
    for (int j = 0; j < c; ++j) s = fma(R[k + (i64)j * K], Q[m + (i64)j * M], s);
    B[idx] = s;
}

// ------------------------------------------------------------------------------------
// Replica guard of a row-sharded fit.  After the reduction every rank runs the same K-sized arithmetic on the same
// reduced partials, so W, P, Q, R, B must agree bit for bit across the ranks -- PROVIDED the reducer really left
// identical bits everywhere (ring / tree all-reduces do; a reducer that does not would silently fork the replicas).
// replica_checksum_kernel: an order-independent 64-bit checksum of the rank's results (sum of mixed bit patterns),
// cut into four 16-bit pieces p_j; out (RED_SLICES x 8 doubles, slice 0 used) = [p_0..p_3, p_0^2..p_3^2].
// After the SUM all-reduce of `out` over n ranks, replica_verify_kernel checks n * sum(p_j^2) == (sum p_j)^2 for every
// piece -- by Cauchy-Schwarz that holds exactly when all ranks sent the same p_j (all values are integers below 2^53
// for n <= 1024) -- and raises the handle's host-mapped flag otherwise; pls_hip_synchronize reports it.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long guard_mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(UPD_THREADS) void replica_checksum_kernel(const double *__restrict__ W, const double *__restrict__ P,
                                                                       const double *__restrict__ Rm, const double *__restrict__ Q,
                                                                       const double *__restrict__ B, i64 nKA, i64 nMA, i64 nKM,
                                                                       double *__restrict__ out) {
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) acc = 0ull;
    __syncthreads();
    unsigned long long h = 0ull;
    const double *arr[5] = {W, P, Rm, Q, B};
    const i64 len[5] = {nKA, nKA, nKA, nMA, B ? nKM : 0};
    for (int a = 0; a < 5; ++a)
        for (i64 i = threadIdx.x; i < len[a]; i += UPD_THREADS)
            h += guard_mix((unsigned long long)__double_as_longlong(arr[a][i]) + 0x9E3779B97F4A7C15ull * (unsigned long long)(i * 5 + a + 1));
    atomicAdd(&acc, h);
    __syncthreads();
    if (threadIdx.x < RED_SLICES * 8) {
        double v = 0.0;
        if (threadIdx.x < 8) {
            const double p = (double)((acc >> (16 * (threadIdx.x & 3))) & 0xFFFFull);
            v = threadIdx.x < 4 ? p : p * p;
        }
        out[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(WAVE) void replica_verify_kernel(const double *__restrict__ red, int nranks, int *host_flag) {
    if (threadIdx.x < 4) {
        const double s1 = red_sum(red, 8, threadIdx.x), s2 = red_sum(red, 8, 4 + threadIdx.x);
        if ((double)nranks * s2 != s1 * s1) __hip_atomic_store(host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace plsk
