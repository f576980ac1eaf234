// Component update over SEVERAL workgroups for K x M matrices too large for one (2 <= M <= 8, K*M >= 16K values:
// BASELINE config 4 has K = 4096, M = 8).  component_update_kernel walks the 1.5 MB of K-sized data of such a
// component through ONE workgroup (30 us) and needs two more launches for the r recurrence (11 us); here a thread
// owns one row k of XY, W, P, R, the grid is K/256 workgroups, and the four K-long reductions of the update
// (q = XY^T r, G = XY^T XY, |w|, c_j = p_j^T w) are arranged so that only TWO grid-wide exchanges remain:
//
//   launch a      q_a = (sum of the q_raw partials the PREVIOUS launch left) / tt        no exchange: launch boundary
//                 p_a = red/tt, XY -= p q^T tt (src/pls.cpp:427-429), Gram partial of the new XY
//   -- exchange 1: G = XY^T XY (src/pls.cpp:405) --
//                 every workgroup: dominant eigenvector qe of G (one wave), w = XY qe / |XY qe| with
//                 |XY qe|^2 = qe^T G qe (an identity for ANY qe: no separate reduction for the norm, :408-411),
//                 c_j partials = p_j[k] w[k], j < n
//   -- exchange 2: c_j = p_j^T w (:415) --
//                 r = w - sum c_j r_j (:412-416), q_raw partial = r[k] XY[k,:] for the NEXT launch's q (:428)
//
// An exchange = per-workgroup partials (written by wave 0 with agent-scope stores), one agent-scope atomic add per
// workgroup, a relaxed agent-scope poll by one lane, sc1 loads of the partials -- the hand-off form of
// MI355X_MICROARCH.md "Valid forms" (no L2 write-back fence).  Every total is the index-ordered sum of the workgroup
// partials, so the result is bit-reproducible and identical on every rank of a sharded fit.
// All workgroups are co-resident (at most 64 x 256 threads on an otherwise idle stream); the poll is bounded so that
// a lost workgroup could never hang the device.
#pragma once
#include "fused_kernels.hpp"  // raw buffer loads with cache-policy bits
#include "small_kernels.hpp"

namespace plsk {

constexpr int COOP_WG = 256, COOP_MAXG = 64, COOP_CH = 32, COOP_SP = 40;
constexpr int COOP_QSTRIDE = 8, COOP_GSTRIDE = 36;

// Every partial this workgroup publishes has been stored by wave 0 (agent scope) before the call.
__device__ __forceinline__ void grid_exchange(unsigned *cnt, unsigned G) {
    if (threadIdx.x < WAVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G && ++spins < (1u << 22))
            __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

// Totals over the G workgroups of nv published values.  The partials are stored TRANSPOSED, part[t*G + w] = value t
// of workgroup w, so that the readers' loads are contiguous across the threads (an sc1 load is served by the fabric,
// not by L2: lane-scattered rows cost one fabric read per lane and instruction -- 10 us for 36 values x 16
// workgroups -- and relaxed ATOMIC loads are issued one at a time).  All threads of the workgroup call it; the total
// of value t is returned in thread t (t < nv <= COOP_SP) as the index-ordered sum over the workgroups.
constexpr int COOP_LBUF = COOP_SP * COOP_MAXG;  // doubles of LDS staging
__device__ __forceinline__ double gather_totals(const double *part, int nv, unsigned G, double *lbuf) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(part), (short)0, 0x7fffffff, BUF_WORD3);
    const int total = nv * (int)G;
    __syncthreads();  // the previous use of lbuf has been read
    for (int idx = threadIdx.x; idx < total; idx += COOP_WG) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)idx * 8u, 0, AUX_SC1);
        double d;
        __builtin_memcpy(&d, &raw, 8);
        lbuf[idx] = d;
    }
    __syncthreads();
    double s = 0.0;
    if ((int)threadIdx.x < nv)
        for (unsigned w = 0; w < G; ++w) s += lbuf[threadIdx.x * G + w];
    return s;
}

// totals of NV per-thread values over the workgroup: the total of value t ends in thread t (t < NV) as the return value
template <int NV>
__device__ __forceinline__ double wg_totals(double (&v)[NV], double (*sp)[COOP_SP], int lane, int wv) {
    static_assert(NV <= COOP_SP, "wave partial row");
    bool valid = true;
    const int idx = wave_multi_sum<NV, 32>(v, lane, valid);
    __syncthreads();  // the previous use of sp has been read
    if (valid) sp[wv][idx] = v[0];
    __syncthreads();
    const int t = threadIdx.x;
    return (t < NV) ? (sp[0][t] + sp[1][t]) + (sp[2][t] + sp[3][t]) : 0.0;
}

// a, red, nipals, power_iters: as component_update_kernel.  cnt: two exchange counters (zero between fits);
// qraw [COOP_MAXG][8] (workgroup-major); gpart [36][G], cpart [A][G] (value-major, see gather_totals).
template <int MM>
__global__ __launch_bounds__(COOP_WG) void coop_update_kernel(const double *__restrict__ red, double *__restrict__ XY,
                                                              double *__restrict__ W, double *__restrict__ P,
                                                              double *__restrict__ Q, double *R, double *__restrict__ vnext,
                                                              int K, int M, int A, int a, int nipals, int power_iters,
                                                              unsigned *cnt, double *qraw, double *gpart, double *cpart) {
    constexpr int NP = MM * (MM + 1) / 2;
    static_assert(NP <= COOP_GSTRIDE && MM <= COOP_QSTRIDE && MM * MM <= WAVE, "scratch strides");
    __shared__ double sp[4][COOP_SP];
    __shared__ double lbuf[COOP_LBUF];
    __shared__ double Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM], lam_s;
    extern __shared__ double cs[];  // [A]
    const unsigned G = gridDim.x;
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k = wg * COOP_WG + tid;
    const bool kok = k < K;
    const int kc = kok ? k : K - 1;
    const int n = a + 1;
    const bool last = (n >= A);
    unsigned *cntA = cnt, *cntB = cnt + 32;  // on cache lines of their own
    // cntB is zeroed before this workgroup arrives at exchange 1, i.e. before any workgroup can reach exchange 2
    if (wg == 0 && tid == 0 && !last) __hip_atomic_exchange(cntB, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double pcur = 0.0;  // this launch's p[k]

    // Every load below that does not depend on an exchange of THIS launch is issued before the exchange it can hide
    // behind: the K-sized data was written by other CUs (earlier launches), so each dependent round trip is 1-2 us.
    double x[MM];
    if (a >= 0) {
        double ts[RED_SLICES], ps[RED_SLICES];
#pragma unroll
        for (int i = 0; i < RED_SLICES; ++i) {
            ts[i] = red[(i64)i * (K + 1) + K];
            ps[i] = red[(i64)i * (K + 1) + kc];
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) x[m] = XY[kc + (i64)(m < M ? m : M - 1) * K];
        // q = (r^T XY)/tt (:428): the partials of r_a^T XY_a are the previous launch's last step; every thread sums the
        // column m = tid % MM (no divergent branch in front of the loads), the first MM threads publish it
        double qsum = 0.0;
        for (unsigned w = 0; w < G; ++w) qsum += qraw[w * COOP_QSTRIDE + (tid & (MM - 1))];
        double tt = 0.0, pr = 0.0;
#pragma unroll
        for (int i = 0; i < RED_SLICES; ++i) {
            tt += ts[i];
            pr += ps[i];
        }
        const double p = pr / tt;  // p = X^T t / tt (:427)
        if (tid < MM) {
            qs[tid] = qsum / tt;
            if (wg == 0 && tid < M) Q[tid + (i64)a * M] = qsum / tt;
        }
        if (kok) P[k + (i64)a * K] = p;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            x[m] = (m < M && kok) ? x[m] - (p * qs[m]) * tt : 0.0;  // XY -= (p q^T) tt (:429)
            if (m < M && kok) XY[k + (i64)m * K] = x[m];
        }
        pcur = p;
    } else {
#pragma unroll
        for (int m = 0; m < MM; ++m) {  // prologue: XY = reduced X^T Y (:396)
            x[m] = (m < M && kok) ? red_sum(red, K * M, kc + m * K) : 0.0;
            if (m < M && kok) XY[k + (i64)m * K] = x[m];
        }
    }
    if (last) return;

    // P[k, j], j < min(n, 64), for the c_j partials after exchange 1 (column a is this launch's own p)
    constexpr int PF = 2 * COOP_CH;
    double pv[PF];
    // (unconditional loads on a clamped column index: a conditional load would put a wait on every join)
#pragma unroll
    for (int i = 0; i < PF; ++i) pv[i] = P[kc + (i64)(i < a ? i : 0) * K];  // consumed after exchange 1

    {  // G = XY^T XY (:405)
        double g[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) g[i] = 0.0;
        gram_add<MM>(x, g);
        const double tot = wg_totals<NP>(g, sp, lane, wv);
        if (tid < NP) st_agent(gpart + (i64)tid * G + wg, tot);
    }
    grid_exchange(cntA, G);
    {
        const double gt = gather_totals(gpart, NP, G, lbuf);
        if (tid < NP) {
            int i = 0, rem = tid;
            while (rem >= MM - i) { rem -= MM - i; ++i; }
            const int j = i + rem;
            Gs[i + j * MM] = gt;
            Gs[j + i * MM] = gt;
        }
    }
    __syncthreads();
    if (wv == 0) {  // every workgroup solves the same tiny problem on the same bits
        dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
        const int ea = lane % MM, eb = lane / MM;
        const double term = (lane < MM * MM) ? qs[ea] * Gs[ea + eb * MM] * qs[eb] : 0.0;
        const double lam = wave_sum(term);  // |XY qe|^2 = qe^T G qe
        if (lane == 0) lam_s = lam;
    }
    __syncthreads();
    double wk = 0.0;
#pragma unroll
    for (int m = 0; m < MM; ++m) wk = fma(x[m], qs[m], wk);  // w = XY q (:408)
    wk = kok ? wk / sqrt(lam_s) : 0.0;                      // w /= sqrt(w^T w) (:411)
    if (kok) W[k + (i64)n * K] = wk;

    // R[k, j], j < min(n, 64), for the r recurrence after exchange 2: issued now, hidden behind the exchange
    double rv[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) rv[i] = R[kc + (i64)(i < n ? i : 0) * K];
#pragma unroll
    for (int i = 0; i < PF; ++i) rv[i] = (i < n) ? rv[i] : 0.0;

    for (int j0 = 0; j0 < n; j0 += COOP_CH) {  // c_j = p_j^T w, j < n (against the ORIGINAL w, :415)
        double acc[COOP_CH];
        if (j0 < PF) {
#pragma unroll
            for (int i = 0; i < COOP_CH; ++i) {
                const int j = j0 + i;
                const double pj = (j0 == 0 ? pv[i] : pv[COOP_CH + i]);
                acc[i] = ((j < a) ? pj : ((j == a) ? pcur : 0.0)) * wk;  // zero beyond n
            }
        } else {
#pragma unroll
            for (int i = 0; i < COOP_CH; ++i) {
                const int j = j0 + i;
                acc[i] = (j < n) ? P[kc + (i64)j * K] * wk : 0.0;
            }
        }
        const double tot = wg_totals<COOP_CH>(acc, sp, lane, wv);
        if (tid < COOP_CH && j0 + tid < n) st_agent(cpart + (i64)(j0 + tid) * G + wg, tot);
    }
    grid_exchange(cntB, G);
    // every workgroup has passed exchange 1: its counter can be zeroed for the next launch
    if (wg == 0 && tid == 0) __hip_atomic_exchange(cntA, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j0 = 0; j0 < n; j0 += COOP_CH) {
        const int nv = min(COOP_CH, n - j0);
        const double ct = gather_totals(cpart + (i64)j0 * G, nv, G, lbuf);
        if (tid < nv) cs[j0 + tid] = ct;
    }
    __syncthreads();
    double r = wk;
#pragma unroll
    for (int i = 0; i < PF; ++i) r -= cs[i < n ? i : 0] * rv[i];  // same subtraction order as the reference; rv = 0 beyond n
    for (int j = PF; j < n; ++j) r -= cs[j] * R[kc + (i64)j * K];
    if (kok) {
        R[k + (i64)n * K] = r;
        vnext[k] = nipals ? wk : r;
    }
    {  // q_raw = r_n^T XY_n for the next launch (:428)
        double qa[MM];
#pragma unroll
        for (int m = 0; m < MM; ++m) qa[m] = kok ? r * x[m] : 0.0;
        const double tot = wg_totals<MM>(qa, sp, lane, wv);
        if (tid < MM) qraw[(i64)wg * COOP_QSTRIDE + tid] = tot;
    }
}

// is the shape one the cooperative kernel takes?
inline bool coop_update_covers(int K, int M) {
    return M >= 2 && M <= 8 && (i64)K * M >= 16384 && K <= COOP_MAXG * COOP_WG;
}
// bytes of scratch for A components
inline size_t coop_scratch_bytes(int A) {
    return 256 + ((size_t)COOP_MAXG * (COOP_QSTRIDE + COOP_GSTRIDE) + (size_t)COOP_MAXG * A) * 8;
}

}  // namespace plsk
