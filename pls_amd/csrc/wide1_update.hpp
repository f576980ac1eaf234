// Component update of a SINGLE-response fit on very many columns (src/pls.cpp:403-416, :427-429 with M = 1), spread over
// the chip.  component_update_kernel walks the K-sized vectors through ONE workgroup and the r recurrence needs its n
// dot products p_j^T w on top (rotate_dots_kernel: one workgroup per j): 143 + 57 us per component at K = 50,000 -- more
// than the two passes over a 512 x 50,000 matrix (35 + 48 us).  With one response the eigenproblem of :405-408 is 1 x 1,
// w = XY / |XY|, and the update is element-wise work plus three kinds of K-long sums, which fall on the two launch
// boundaries of a two-kernel form (no in-launch exchange, no co-residency requirement):
//
//   wide1_a_kernel   q_a = (sum of the q partials the PREVIOUS launch left) / tt                       (:428)
//                    p_a = red / tt (:427),  XY -= (p q) tt (:429)
//                    partials of |XY|^2 and of p_j^T XY, j <= a      -> part[(1 + j) * G + workgroup]
//   wide1_b_kernel   totals (every workgroup, same order, same bits),  w = XY / |XY| (:404, :411),
//                    c_j = (p_j^T XY) / |XY| = p_j^T w (:415),  r = w - sum_j c_j r_j in the reference's order (:412-416),
//                    q partial r^T XY for the next component        -> qpart[workgroup]
//
// A workgroup owns a contiguous slice of 256 E columns (G <= 128 workgroups); the slice of XY stays in LDS between the
// element-wise step and the n dot products, which the four waves share out by j.  Every total is an index-ordered sum of
// workgroup partials: bit-reproducible, identical on every rank of a sharded fit.
#pragma once
#include "coop_update.hpp"  // wg_totals, gram_add, the one-wave eigenvector
#include "small_kernels.hpp"

namespace plsk {

constexpr int W1_WG = 256, W1_MAXG = 128;

// out[t] (LDS) = sum over the G workgroups of the value-major partials part[t * stride_t + g * stride_g], t < nv: a wave per
// value, its lanes over the workgroups (G <= 128: two loads per lane), then the fixed-order wave sum -- every workgroup
// forms the same bits.  (One thread per value walking G partials one after the other: 15 us of a 20 us kernel.)
__device__ __forceinline__ void w1_totals(const double *__restrict__ part, int nv, int G, i64 stride_t, i64 stride_g, double *out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = wv; t < nv; t += W1_WG / WAVE) {
        double s = 0.0;
        for (int g = lane; g < G; g += WAVE) s += part[(i64)t * stride_t + (i64)g * stride_g];
        s = wave_sum(s);
        if (lane == 0) out[t] = s;
    }
}

// a, red, nipals: as component_update_kernel (a = -1: prologue, XY = reduced X^T Y).  E: columns per thread.
__global__ __launch_bounds__(W1_WG) void wide1_a_kernel(const double *__restrict__ red, double *__restrict__ XY, double *P,
                                                        double *__restrict__ Q, int K, int A, int a, int E,
                                                        const double *__restrict__ qpart, double *__restrict__ part) {
    extern __shared__ double xs[];  // [W1_WG * E] this workgroup's slice of the new XY
    __shared__ double sred[W1_WG / WAVE];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k0 = wg * W1_WG * E, kn = min(W1_WG * E, K - k0);  // the slice [k0, k0 + kn)
    const int n = a + 1;
    double ss = 0.0;
    if (a >= 0) {
        const double tt = red_sum(red, K + 1, K);
        const double qs = block_sum<W1_WG / WAVE>(tid < G ? qpart[tid] : 0.0, sred);
        const double q = qs / tt;
        if (wg == 0 && tid == 0) Q[a] = q;
        for (int i = tid; i < kn; i += W1_WG) {
            const int k = k0 + i;
            const double p = red_sum(red, K + 1, k) / tt;
            P[k + (i64)a * K] = p;
            const double x = XY[k] - (p * q) * tt;
            XY[k] = x;
            xs[i] = x;
            ss = fma(x, x, ss);
        }
    } else {
        for (int i = tid; i < kn; i += W1_WG) {
            const double x = red_sum(red, K, k0 + i);
            XY[k0 + i] = x;
            xs[i] = x;
            ss = fma(x, x, ss);
        }
    }
    if (n >= A) return;
    ss = block_sum<W1_WG / WAVE>(ss, sred);  // (its barriers also publish xs and this launch's column of P to the workgroup)
    if (tid == 0) part[wg] = ss;
    for (int j = wv; j < n; j += W1_WG / WAVE) {
        const double *pj = P + (i64)j * K + k0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = lane;
        for (; i + 3 * WAVE < kn; i += 4 * WAVE) {
            s0 = fma(pj[i], xs[i], s0);
            s1 = fma(pj[i + WAVE], xs[i + WAVE], s1);
            s2 = fma(pj[i + 2 * WAVE], xs[i + 2 * WAVE], s2);
            s3 = fma(pj[i + 3 * WAVE], xs[i + 3 * WAVE], s3);
        }
        for (; i < kn; i += WAVE) s0 = fma(pj[i], xs[i], s0);
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) part[(i64)(1 + j) * G + wg] = s;
    }
}

// More than 4096 components (hence K > 4096): the totals no longer fit a workgroup's LDS (and n x G loads per workgroup
// would be repeated G times) -- one launch forms them once: tot[0] = |XY|, tot[1 + j] = c_j.  grid = ceil((n + 1) / 4).
__global__ __launch_bounds__(W1_WG) void wide1_totals_kernel(const double *__restrict__ part, int n, int G, double *__restrict__ tot) {
    __shared__ double v[1 + W1_WG / WAVE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    w1_totals(part, 1, G, G, 1, v);  // |XY|^2 (wave 0)
    {
        const int t = 1 + blockIdx.x * (W1_WG / WAVE) + wv;  // this wave's value
        double s = 0.0;
        if (t <= n)
            for (int g = lane; g < G; g += WAVE) s += part[(i64)t * G + g];
        s = wave_sum(s);
        if (lane == 0) v[1 + wv] = s;
    }
    __syncthreads();
    const double nrm = sqrt(v[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) tot[0] = nrm;
    const int t = 1 + blockIdx.x * (W1_WG / WAVE) + (int)threadIdx.x;
    if (threadIdx.x < W1_WG / WAVE && t <= n) tot[t] = v[1 + threadIdx.x] / nrm;
}

// n = a + 1 >= 0 is the component whose w and r are formed (n < A).  cs: dynamic LDS, n + 1 doubles -- or, GCS, the
// totals wide1_totals_kernel left in global memory (wave-uniform reads: the scalar cache).
template <bool GCS>
__global__ __launch_bounds__(W1_WG) void wide1_b_kernel(const double *__restrict__ XY, double *__restrict__ W, double *R,
                                                        double *__restrict__ vnext, int K, int n, int E, int nipals,
                                                        const double *__restrict__ part, double *__restrict__ qpart,
                                                        const double *__restrict__ tot) {
    extern __shared__ double cs_lds[];  // [n + 1]: |XY|^2, then p_j^T XY
    __shared__ double sred[W1_WG / WAVE];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const int k0 = wg * W1_WG * E, kn = min(W1_WG * E, K - k0);
    const double *cs = GCS ? tot : cs_lds;
    double nrm;
    if constexpr (GCS) {
        nrm = tot[0];
    } else {
        w1_totals(part, n + 1, G, G, 1, cs_lds);
        __syncthreads();
        nrm = sqrt(cs_lds[0]);
        __syncthreads();
        for (int t = tid; t < n; t += W1_WG) cs_lds[1 + t] /= nrm;  // c_j = p_j^T w
        __syncthreads();
    }
    double qa = 0.0;
    double *wn = W + (i64)n * K, *rn = R + (i64)n * K;
    for (int i = tid; i < kn; i += W1_WG) {
        const int k = k0 + i;
        const double x = XY[k];
        const double w = x / nrm;
        wn[k] = w;
        double r = w;
        int j = 0;
        for (; j + 4 <= n; j += 4) {  // same subtraction order as the reference, loads issued together
            const double r0 = R[k + (i64)j * K], r1 = R[k + (i64)(j + 1) * K];
            const double r2 = R[k + (i64)(j + 2) * K], r3 = R[k + (i64)(j + 3) * K];
            r -= cs[1 + j] * r0;
            r -= cs[2 + j] * r1;
            r -= cs[3 + j] * r2;
            r -= cs[4 + j] * r3;
        }
        for (; j < n; ++j) r -= cs[1 + j] * R[k + (i64)j * K];
        rn[k] = r;
        vnext[k] = nipals ? w : r;
        qa = fma(r, x, qa);
    }
    qa = block_sum<W1_WG / WAVE>(qa, sred);
    if (tid == 0) qpart[wg] = qa;
}

// geometry: E columns per thread so that at most W1_MAXG workgroups cover K; false = not this path (LDS slice too large)
inline bool wide1_geometry(int K, int *G, int *E) {
    const int e = (K + W1_WG * W1_MAXG - 1) / (W1_WG * W1_MAXG);
    if ((size_t)W1_WG * e * 8 > 48 * 1024) return false;
    *E = e;
    *G = (K + W1_WG * e - 1) / (W1_WG * e);
    return true;
}

// ---- 2 <= M <= 8 responses beyond the cooperative kernel's 16,384 columns --------------------------------------------
// coop_update_kernel (coop_update.hpp) needs all its workgroups co-resident and one thread per column; past that the
// update fell back to ONE workgroup walking K x M values.  The same arithmetic with the kernel cut at its two grid-wide
// exchanges -- three launches, a slice of 256 E columns per workgroup, G <= 128:
//   widem_a_kernel   q_a from the previous launch's partials (:428), p_a (:427), XY -= (p q^T) tt (:429), Gram partials (:405)
//   widem_b_kernel   G = XY^T XY, its dominant eigenvector qe (every workgroup, same bits; :405-408),
//                    w = XY qe / sqrt(qe^T G qe) (:408-411), partials of c_j = p_j^T w (:415)
//   widem_c_kernel   r = w - sum_j c_j r_j (:412-416), partials of r^T XY for the next component's q
constexpr int WM_GSTRIDE = 36, WM_QSTRIDE = 8;

template <int MM>
__global__ __launch_bounds__(W1_WG) void widem_a_kernel(const double *__restrict__ red, double *__restrict__ XY,
                                                        double *__restrict__ P, double *__restrict__ Q, int K, int M, int A,
                                                        int a, int E, const double *__restrict__ qpart, double *__restrict__ gpart) {
    constexpr int NP = MM * (MM + 1) / 2;
    __shared__ double sp[4][COOP_SP];
    __shared__ double qs[MM];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k0 = wg * W1_WG * E, kn = min(W1_WG * E, K - k0);
    double g[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) g[i] = 0.0;
    double tt = 1.0;
    if (a >= 0) {
        tt = red_sum(red, K + 1, K);
        if (tid < MM) qs[tid] = 0.0;
        __syncthreads();
        w1_totals(qpart, M, G, 1, WM_QSTRIDE, qs);
        __syncthreads();
        if (tid < M) {
            const double q = qs[tid] / tt;
            qs[tid] = q;
            if (wg == 0) Q[tid + (i64)a * M] = q;
        }
        __syncthreads();
    }
    for (int i = tid; i < kn; i += W1_WG) {
        const int k = k0 + i;
        double x[MM];
        if (a >= 0) {
            const double p = red_sum(red, K + 1, k) / tt;
            P[k + (i64)a * K] = p;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                x[m] = 0.0;
                if (m < M) {
                    x[m] = XY[k + (i64)m * K] - (p * qs[m]) * tt;
                    XY[k + (i64)m * K] = x[m];
                }
            }
        } else {
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                x[m] = 0.0;
                if (m < M) {
                    x[m] = red_sum(red, K * M, k + m * K);
                    XY[k + (i64)m * K] = x[m];
                }
            }
        }
        gram_add<MM>(x, g);
    }
    if (a + 1 >= A) return;
    const double tot = wg_totals<NP>(g, sp, lane, wv);
    if (tid < NP) gpart[(i64)tid * G + wg] = tot;
}

template <int MM>
__global__ __launch_bounds__(W1_WG) void widem_b_kernel(const double *__restrict__ XY, double *__restrict__ W,
                                                        const double *__restrict__ P, int K, int M, int n, int E, int power_iters,
                                                        const double *__restrict__ gpart, double *__restrict__ cpart) {
    constexpr int NP = MM * (MM + 1) / 2;
    extern __shared__ double ws[];  // [W1_WG * E] this workgroup's slice of w
    __shared__ double Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM], lam_s;
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k0 = wg * W1_WG * E, kn = min(W1_WG * E, K - k0);
    w1_totals(gpart, NP, G, G, 1, Cs);  // (Cs: free until the eigenvector iteration)
    __syncthreads();
    if (tid < NP) {
        const double s = Cs[tid];
        int i = 0, rem = tid;
        while (rem >= MM - i) { rem -= MM - i; ++i; }
        const int j = i + rem;
        Gs[i + j * MM] = s;
        Gs[j + i * MM] = s;
    }
    __syncthreads();
    if (wv == 0) {
        dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
        const int ea = lane % MM, eb = lane / MM;
        const double term = (lane < MM * MM) ? qs[ea] * Gs[ea + eb * MM] * qs[eb] : 0.0;
        const double lam = wave_sum(term);  // |XY qe|^2 = qe^T G qe
        if (lane == 0) lam_s = lam;
    }
    __syncthreads();
    const double nrm = sqrt(lam_s);
    double *wn = W + (i64)n * K;
    for (int i = tid; i < kn; i += W1_WG) {
        const int k = k0 + i;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (m < M) s = fma(XY[k + (i64)m * K], qs[m], s);
        const double w = s / nrm;
        wn[k] = w;
        ws[i] = w;
    }
    __syncthreads();
    for (int j = wv; j < n; j += W1_WG / WAVE) {
        const double *pj = P + (i64)j * K + k0;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = lane;
        for (; i + 3 * WAVE < kn; i += 4 * WAVE) {
            s0 = fma(pj[i], ws[i], s0);
            s1 = fma(pj[i + WAVE], ws[i + WAVE], s1);
            s2 = fma(pj[i + 2 * WAVE], ws[i + 2 * WAVE], s2);
            s3 = fma(pj[i + 3 * WAVE], ws[i + 3 * WAVE], s3);
        }
        for (; i < kn; i += WAVE) s0 = fma(pj[i], ws[i], s0);
        const double s = wave_sum((s0 + s1) + (s2 + s3));
        if (lane == 0) cpart[(i64)j * G + wg] = s;
    }
}

template <int MM>
__global__ __launch_bounds__(W1_WG) void widem_c_kernel(const double *__restrict__ XY, const double *__restrict__ W, double *R,
                                                        double *__restrict__ vnext, int K, int M, int n, int E, int nipals,
                                                        const double *__restrict__ cpart, double *__restrict__ qpart) {
    extern __shared__ double cs[];  // [n]
    __shared__ double sp[4][COOP_SP];
    const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k0 = wg * W1_WG * E, kn = min(W1_WG * E, K - k0);
    w1_totals(cpart, n, G, G, 1, cs);
    __syncthreads();
    double qa[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) qa[m] = 0.0;
    const double *wn = W + (i64)n * K;
    double *rn = R + (i64)n * K;
    for (int i = tid; i < kn; i += W1_WG) {
        const int k = k0 + i;
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
#pragma unroll
        for (int m = 0; m < MM; ++m)
            if (m < M) qa[m] = fma(r, XY[k + (i64)m * K], qa[m]);
    }
    const double tot = wg_totals<MM>(qa, sp, lane, wv);
    if (tid < MM) qpart[(i64)wg * WM_QSTRIDE + tid] = tot;
}

}  // namespace plsk
