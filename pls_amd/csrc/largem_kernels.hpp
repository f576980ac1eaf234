// Component update for MANY responses (M > 32): the generality path.  The reference solves any M x M eigenproblem
// (Eigen::EigenSolver on XY^T XY, src/pls.cpp:405-408); the fast kernels of small_kernels.hpp / coop_update.hpp keep
// the M x M matrix in one workgroup's LDS and stop at M = 32.  Here everything M-sized lives in global memory and every
// step is a plain multi-workgroup kernel -- a few hundred small launches per component (about a millisecond), which is
// what "more slowly" costs; nothing on this path is tuned.  Same operation sequence, same sign convention, same
// repeated-squaring power iteration as dominant_eigvec_lds.
#pragma once
#include "small_kernels.hpp"

namespace plsk {

constexpr int LM_MAX = 1024;  // responses supported by the large-M path (M x M fp64 matrices: 3 x 8 MB at the limit)

// p = X^T t / tt -> P[:,a] (:427);  q_m = (r_a^T XY[:,m]) / tt -> Q[m,a], qv[m] (:428).   grid = M + ceil(K/256)
__global__ __launch_bounds__(WG) void lm_pq_kernel(const double *__restrict__ red, const double *__restrict__ XY,
                                                   const double *__restrict__ R, double *__restrict__ P,
                                                   double *__restrict__ Q, double *__restrict__ qv, int K, int M, int a) {
    __shared__ double sm[WG / WAVE];
    const double tt = red_sum(red, K + 1, K);
    const int b = blockIdx.x;
    if (b < M) {
        const double *xm = XY + (i64)b * K, *ra = R + (i64)a * K;
        double s = 0.0;
        for (int k = threadIdx.x; k < K; k += WG) s = fma(ra[k], xm[k], s);
        s = block_sum<WG / WAVE>(s, sm);
        if (threadIdx.x == 0) {
            Q[b + (i64)a * M] = s / tt;
            qv[b] = s / tt;
        }
    } else {
        const int k = (b - M) * WG + threadIdx.x;
        if (k < K) P[k + (i64)a * K] = red_sum(red, K + 1, k) / tt;
    }
}

// XY -= (p q^T) tt (:429)
__global__ __launch_bounds__(WG) void lm_deflate_kernel(const double *__restrict__ red, double *__restrict__ XY,
                                                        const double *__restrict__ P, const double *__restrict__ qv,
                                                        int K, int M, int a) {
    const i64 idx = (i64)blockIdx.x * WG + threadIdx.x;
    if (idx >= (i64)K * M) return;
    const double tt = red_sum(red, K + 1, K);
    const int k = (int)(idx % K), m = (int)(idx / K);
    XY[idx] -= (P[k + (i64)a * K] * qv[m]) * tt;
}

// (16 x 16 tiles through LDS; grid = (ceil(M/16), ceil(M/16)), block = (16, 16))
// C = (A / tr A)^2 in ONE launch: every workgroup forms the trace itself (fixed order: 256 strided sums, wave sums, waves in
// order -- the same bits in every workgroup), then its 16 x 16 tile of (A / tr) * (A / tr).  One launch per squaring of the
// direction solve instead of three (square, trace, scale): beyond LM_LDS_MAX responses the solve is launch-bound.
__global__ __launch_bounds__(256) void lm_square_normalised_kernel(const double *__restrict__ Am, int M, double *__restrict__ Cm) {
    __shared__ double ta[16][17], tb[16][17], sm[4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 16 + tx;
    double t = 0.0;
    for (int i = tid; i < M; i += 256) t += Am[i + (i64)i * M];
    t = wave_sum(t);
    if ((tid & 63) == 0) sm[tid >> 6] = t;
    __syncthreads();
    const double tr = ((sm[0] + sm[1]) + sm[2]) + sm[3];
    const double sc = 1.0 / tr;  // the OPERANDS are scaled as they are staged: tr^2 or the entries of A * A of an unscaled Gram matrix may leave the fp64 range
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    double s = 0.0;
    for (int k0 = 0; k0 < M; k0 += 16) {
        ta[ty][tx] = (row < M && k0 + tx < M) ? Am[row + (i64)(k0 + tx) * M] * sc : 0.0;
        tb[ty][tx] = (k0 + ty < M && col < M) ? Am[(k0 + ty) + (i64)col * M] * sc : 0.0;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) s = fma(ta[ty][kk], tb[kk][tx], s);
        __syncthreads();
    }
    if (row < M && col < M) Cm[row + (i64)col * M] = s;
}

// The dominant eigenvector of G (M x M, M <= MMAX = 32) in ONE launch, in LDS: the solver of component_update_body
// (repeated squaring to the fixed point, two polishing steps, sign convention).  For 9 <= M <= 32 responses on matrices
// with many columns, where the rest of the update runs as the multi-workgroup kernels of this file.
__global__ __launch_bounds__(UPD_THREADS) void lm_eig_lds_kernel(const double *__restrict__ G, int M, int iters,
                                                                 double *__restrict__ qe) {
    __shared__ double Gs[MMAX * MMAX], Bs[MMAX * MMAX], Cs[MMAX * MMAX], qs[MMAX];
    for (int i = threadIdx.x; i < M * M; i += UPD_THREADS) Gs[i] = G[i];
    __syncthreads();
    dominant_eigvec_lds(Gs, Bs, Cs, qs, M, iters);
    if ((int)threadIdx.x < M) qe[threadIdx.x] = qs[threadIdx.x];
}

// 33 .. LM_LDS_MAX (48) responses: the same solve in ONE launch with B and C in (dynamic) LDS and G read from global memory -- every
// thread owns the entries e = tid, tid + 1024, ... of the M x M matrices; repeated squaring to the fixed point (uniform early
// exit), two polishing steps, sign convention: dominant_eigvec_lds with loops.  Replaces 2 + 3 x power_iters + 1 launches
// (~150 at the default 48 squarings, none of which can stop early: the host does not see the fixed point) -- 40 responses:
// 0.46 -> 0.1 ms per component (profiles/r4/many_responses_scan.txt).
constexpr int LM_LDS_MAX = 48;  // (one CU squares M x M matrices: 33 -> 1.4, 40 -> 1.5, 64 -> 3.5-5.0, 88 -> 11 ms per 8-component fit against ~4.5-6 for the launches)
__global__ __launch_bounds__(UPD_THREADS) void lm_eig_lds_big_kernel(const double *__restrict__ G, int M, int iters,
                                                                     double *__restrict__ qe) {
    extern __shared__ double lm_dyn[];  // Bm [M*M], Cm [M*M], qv [M]
    double *Bm = lm_dyn, *Cm = lm_dyn + M * M, *qv = Cm + M * M;
    const int tid = threadIdx.x, MM = M * M;
    double tr = 0.0;
    for (int c = 0; c < M; ++c) tr += G[c + c * M];
    for (int e = tid; e < MM; e += UPD_THREADS) Bm[e] = G[e] / tr;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int e = tid; e < MM; e += UPD_THREADS) {
            const int a = e % M, b = e / M;
            double s = 0.0;
            for (int c = 0; c < M; ++c) s = fma(Bm[a + c * M], Bm[c + b * M], s);
            Cm[e] = s;
        }
        __syncthreads();
        double t2 = 0.0;
        for (int c = 0; c < M; ++c) t2 += Cm[c + c * M];
        int same = 1;
        for (int e = tid; e < MM; e += UPD_THREADS) {
            const double nb = Cm[e] / t2, ob = Bm[e];
            same &= fabs(nb - ob) <= 4.0e-16 * fabs(nb) + 1.0e-18;  // (see dominant_eigvec_lds)
            Bm[e] = nb;
        }
        if (__syncthreads_and(same)) break;  // uniform exit; also the barrier that publishes Bm
    }
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
    if (tid < M) qe[tid] = qv[tid] * sgn;
}

// From Bm ~ v1 v1^T: the column with the largest diagonal entry, two power steps on G, unit norm, largest-|.| entry
// positive (lowest index on ties) -> qe.   One workgroup of UPD_THREADS; M <= LM_MAX (qe staged in LDS).
__global__ __launch_bounds__(UPD_THREADS) void lm_eig_finish_kernel(const double *__restrict__ G, const double *__restrict__ Bm,
                                                                    int M, double *__restrict__ qe) {
    __shared__ double q[LM_MAX], tmp[LM_MAX], sm[UPD_WAVES];
    const int tid = threadIdx.x;
    int best = 0;
    double bd = Bm[0];
    for (int c = 1; c < M; ++c) {  // every thread walks the diagonal: the same answer everywhere, no exchange
        const double d = Bm[c + (i64)c * M];
        if (d > bd) { bd = d; best = c; }
    }
    for (int i = tid; i < M; i += UPD_THREADS) q[i] = Bm[i + (i64)best * M];
    __syncthreads();
    for (int pol = 0; pol < 2; ++pol) {
        for (int i = tid; i < M; i += UPD_THREADS) {
            double s = 0.0;
            for (int c = 0; c < M; ++c) s = fma(G[i + (i64)c * M], q[c], s);
            tmp[i] = s;
        }
        __syncthreads();
        double n2 = 0.0;
        for (int i = tid; i < M; i += UPD_THREADS) n2 = fma(tmp[i], tmp[i], n2);
        n2 = block_sum<UPD_WAVES>(n2, sm);
        const double inv = 1.0 / sqrt(n2);
        for (int i = tid; i < M; i += UPD_THREADS) q[i] = tmp[i] * inv;
        __syncthreads();
    }
    int big = 0;
    for (int c = 1; c < M; ++c)
        if (fabs(q[c]) > fabs(q[big])) big = c;
    const double sgn = (q[big] < 0.0) ? -1.0 : 1.0;
    for (int i = tid; i < M; i += UPD_THREADS) qe[i] = q[i] * sgn;
}

// wraw = XY qe (:408) with per-workgroup partials of |wraw|^2.   grid = ceil(K/256)
__global__ __launch_bounds__(WG) void lm_w_kernel(const double *__restrict__ XY, const double *__restrict__ qe, int K, int M,
                                                  double *__restrict__ wraw, double *__restrict__ sspart) {
    __shared__ double sm[WG / WAVE];
    const int k = blockIdx.x * WG + threadIdx.x;
    double s = 0.0;
    if (k < K)
        for (int m = 0; m < M; ++m) s = fma(XY[k + (i64)m * K], qe[m], s);
    if (k < K) wraw[k] = s;
    const double ss = block_sum<WG / WAVE>(k < K ? s * s : 0.0, sm);
    if (threadIdx.x == 0) sspart[blockIdx.x] = ss;
}

// w = wraw / sqrt(sum of the partials) (:411); n == 0 (first component): r_0 = w_0 and the next direction as well
__global__ __launch_bounds__(WG) void lm_normalize_kernel(const double *__restrict__ wraw, const double *__restrict__ sspart,
                                                          int nparts, int K, double *__restrict__ wout, double *R0,
                                                          double *vnext) {
    double ss = 0.0;
    for (int i = 0; i < nparts; ++i) ss += sspart[i];  // index order: the same bits in every workgroup
    const int k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    const double w = wraw[k] / sqrt(ss);
    wout[k] = w;
    if (R0) R0[k] = w;
    if (vnext) vnext[k] = w;
}

}  // namespace plsk
