// update_m1.hpp -- the K-sized bookkeeping of one component for ONE response (src/pls.cpp:403-404, :411-416, :427-433), written
// once for every place that runs it:
//   * component_update_kernel / component_update_gather_kernel / component_update_type2_kernel (small_kernels.hpp): one workgroup
//     of 1024 threads behind the pass;
//   * the TAIL of the fused pass itself (fused_kernels.hpp, slice_tail): the 512 threads of the workgroup that summed the last
//     slice run the update as the last act of the launch -- a component is then ONE launch, also on a shard.
// The ranks of a row-sharded fit may take different routes (a shard with fewer than TAIL_MIN_WG tiles has no tail), and the
// replica guard compares bits: so the arithmetic is the same whatever the thread count.  A workgroup of NT threads plays the
// 1024 VIRTUAL threads of the original kernel, VT = 1024 / NT each: virtual thread v owns the columns v, v + 1024, v + 2048,
// v + 3072 (K <= 4096), its wave is v / 64, block sums are wave sums (DPP order) added over the 16 virtual waves in index
// order; the p_j^T w products are one (real) wave per j with the lane-strided four-chain sum; the r recurrence is per column.
#pragma once
#include "common.hpp"

namespace plsk {

constexpr int UPD1_VTHREADS = 1024;
constexpr int UPD1_VWAVES = UPD1_VTHREADS / WAVE;
constexpr int UPD1_KMAX = 4 * UPD1_VTHREADS;

// sum over the 1024 virtual threads; x[u] = the value of virtual thread tid + u NT.  sm: >= 16 doubles of LDS.
template <int NT>
__device__ __forceinline__ double vblock_sum(const double (&x)[UPD1_VTHREADS / NT], double *sm) {
    constexpr int VT = UPD1_VTHREADS / NT, NW = NT / WAVE;
    double xs[VT];
#pragma unroll
    for (int u = 0; u < VT; ++u) xs[u] = wave_sum(x[u]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int u = 0; u < VT; ++u) sm[w + u * NW] = xs[u];
    }
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < UPD1_VWAVES; ++i) s += sm[i];
    return s;
}

// S(j): reduced value j of [X^T t (K values), t^T t] -- the same bits on every rank and route.
// a >= 0: the component whose pass just finished.  p = S/tt (:427) -> P[:,a]; q = XY^T r_a / tt (:428) -> Q[a];
// XY -= (p q) tt (:429); when a + 1 < A: w = XY / |XY| (:404, :411) -> W[:,a+1]; r (:412-416) -> R[:,a+1];
// vnext = w (NIPALS: the next pass is X_a w) or r (KERNEL: X r).  split_rotate: r is left to rotate_dots / rotate_apply.
// cs: >= a + 1 doubles of LDS; sm: >= 16 doubles of LDS; wl: K doubles of LDS for w (nullptr: w is read back from W, the
// workgroup's own stores -- the values are the same).
// KI: columns per virtual thread, 1 (K <= 1024) or 4 (K <= 4096) -- every route takes the same KI for the same K.
template <int NT, int KI, typename SumF>
__device__ __forceinline__ void update_m1(SumF S, double *__restrict__ XY, double *__restrict__ W, double *__restrict__ P,
                                          double *__restrict__ Q, double *__restrict__ R, double *__restrict__ vnext, int K, int A,
                                          int a, int nipals, int split_rotate, double *cs, double *sm, double *wl) {
    constexpr int VT = UPD1_VTHREADS / NT, NW = NT / WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double tt = S(K);
    const double *ra = R + (i64)a * K;
    double xv[VT][KI], pv[VT][KI], part[VT];
#pragma unroll
    for (int u = 0; u < VT; ++u) {
        part[u] = 0.0;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int k = tid + u * NT + i * UPD1_VTHREADS;
            xv[u][i] = (k < K) ? XY[k] : 0.0;
            pv[u][i] = (k < K) ? S(k) / tt : 0.0;
            part[u] = fma((k < K) ? ra[k] : 0.0, xv[u][i], part[u]);
        }
    }
    const double q = vblock_sum<NT>(part, sm) / tt;  // q = (r^T XY) / tt  (:428)
    if (tid == 0) Q[(i64)a] = q;
    double ssn[VT];
#pragma unroll
    for (int u = 0; u < VT; ++u) {
        ssn[u] = 0.0;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int k = tid + u * NT + i * UPD1_VTHREADS;
            if (k < K) {
                P[k + (i64)a * K] = pv[u][i];
                xv[u][i] -= (pv[u][i] * q) * tt;  // XY -= (p q^T) tt  (:429)
                XY[k] = xv[u][i];
                ssn[u] = fma(xv[u][i], xv[u][i], ssn[u]);
            }
        }
    }
    const int n = a + 1;
    if (n >= A) return;
    const double nrm = sqrt(vblock_sum<NT>(ssn, sm));  // w = XY / |XY|  (:404, :411)
    double *wn = W + (i64)n * K;
#pragma unroll
    for (int u = 0; u < VT; ++u)
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int k = tid + u * NT + i * UPD1_VTHREADS;
            if (k < K) {
                const double w = xv[u][i] / nrm;
                wn[k] = w;
                if (wl) wl[k] = w;
            }
        }
    if (split_rotate) return;
    __syncthreads();  // w complete (LDS, or the workgroup's own global stores)
    const double *ws = wl ? wl : wn;
    for (int j = wv; j < n; j += NW) {  // c_j = P[:,j]^T w  (against the ORIGINAL w, :415)
        const double *pj = P + (i64)j * K;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = lane;
        for (; k + 3 * WAVE < K; k += 4 * WAVE) {
            const double a0 = pj[k], a1 = pj[k + WAVE], a2 = pj[k + 2 * WAVE], a3 = pj[k + 3 * WAVE];
            s0 = fma(a0, ws[k], s0);
            s1 = fma(a1, ws[k + WAVE], s1);
            s2 = fma(a2, ws[k + 2 * WAVE], s2);
            s3 = fma(a3, ws[k + 3 * WAVE], s3);
        }
        for (; k < K; k += WAVE) s0 = fma(pj[k], ws[k], s0);
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) cs[j] = s;
    }
    __syncthreads();
    double *rn = R + (i64)n * K;
    for (int k = tid; k < K; k += NT) {
        const double w = ws[k];
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

}  // namespace plsk
