// Deferred write-back of the rank-1 deflation (opt-in: PLS_HIP_OPT_DEFER = D > 1; NIPALS plan, K <= 512).
//
// The explicit NIPALS plan materialises X_a = X_{a-1} - t_{a-1} p_{a-1}^T in HBM for every component (one read +
// one write of X per component).  Here up to D rank-1 updates stay PENDING: a pass loads the last stored matrix
// X_b, applies the pending updates (t_b, p_b) ... (t_{a-1}, p_{a-1}) to the tile in registers -- the same fused
// multiply-adds, rounded to the storage type after each one, so the tile holds exactly the bits the explicit plan
// would have stored -- computes t_a and X_a^T t_a from it, and writes X_a back only when D updates are pending.
// HBM traffic per component drops from 2 to (D + 1) / D sweeps; scores, loadings and coefficients agree with the
// explicit plan to the rounding of the partial sums (the workgroup count of a read-only pass differs).
// The explicit plan (D = 1) remains the default and the benchmark headline.
#pragma once
#include "fused_kernels.hpp"

namespace plsk {

constexpr int DEFER_MAX = 4;

template <typename T>
struct PendingUpdates {
    const T *t[DEFER_MAX];       // score columns t_b .. t_{a-1}
    const double *p[DEFER_MAX];  // loading vectors p_b .. p_{a-1}
};

// NP: pending updates applied to every tile (1..DEFER_MAX); ST: write the updated tile to dst.
template <typename T, int V, int R, int NT, int CPT, int NP, bool ST>
__global__ __launch_bounds__(NT, (NT / 256) * (ST ? 1 : 2)) void fused_defer_kernel(
    const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd, i64 N, int K, const double *__restrict__ v,
    PendingUpdates<T> pend, T *__restrict__ tout, double *__restrict__ part, double *__restrict__ sspart) {
    constexpr int RP = R / V, CG = NT / RP, NW = NT / WAVE;
    static_assert(NP >= 1 && NP <= DEFER_MAX && CPT <= 16, "shape");
    __shared__ __attribute__((aligned(16))) double ps[NP][CG * CPT];
    __shared__ __attribute__((aligned(16))) double vs[CG * CPT];
    __shared__ __attribute__((aligned(16))) double tred[2][NW][R];
    __shared__ double sred[NW];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = tid % RP, cg = tid / RP;
    for (int k = tid; k < CG * CPT; k += NT) {
        vs[k] = (k < K) ? v[k] : 0.0;
#pragma unroll
        for (int n = 0; n < NP; ++n) ps[n][k] = (k < K) ? pend.p[n][k] : 0.0;
    }
    __syncthreads();

    double pacc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) pacc[j] = 0.0;
    double ss = 0.0;
    int buf = 0;
    const uint32_t xoff = (uint32_t)(((i64)rp * V + (i64)cg * ldx) * (i64)sizeof(T));
    const uint32_t doff = ST ? (uint32_t)(((i64)rp * V + (i64)cg * ldd) * (i64)sizeof(T)) : 0u;
    constexpr uint32_t OOR = 0x80000000u;

    for (i64 tile = blockIdx.x; tile * R < N; tile += gridDim.x, buf ^= 1) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);
        const uint32_t xo = rowok ? xoff : OOR, dof = rowok ? doff : OOR;
        int cgz = cg;
        asm volatile("" : "+v"(cgz));  // LDS operands are re-read every tile (see fused_pass_kernel)
        Pack<T, V> x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cols = min(CG, K - CG * j);
            const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldx * (i64)sizeof(T)) : 0u;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<T *>(X + tile * tsx + (i64)j * CG * ldx), (short)0, (int)nrec, BUF_WORD3);
            x[j] = buf_ld<T, V, AUX_NT>(rs, xo);
            __builtin_amdgcn_sched_barrier(0);
        }
        double tp[NP][V];
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            if (rowok) {
                const Pack<T, V> tpk = ld_pack<T, V>(pend.t[n] + i0);
#pragma unroll
                for (int e = 0; e < V; ++e) tp[n][e] = -(double)tpk.v[e];
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) tp[n][e] = 0.0;
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
#pragma unroll
            for (int n = 0; n < NP; ++n) {  // oldest update first, rounded to storage after each: the explicit plan's bits
                const double pk = ps[n][cgz + CG * j];
#pragma unroll
                for (int e = 0; e < V; ++e) x[j].v[e] = (T)fma(tp[n][e], pk, (double)x[j].v[e]);
            }
            if constexpr (ST) {
                const int cols = min(CG, K - CG * j);
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldd * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                    dst + tile * tsd + (i64)j * CG * ldd, (short)0, (int)nrec, BUF_WORD3);
                buf_st<T, V, AUX_NT>(rd, dof, x[j]);
            }
        }
        double tp2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) tp2[e] = 0.0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const double vk = vs[cgz + CG * j];
#pragma unroll
            for (int e = 0; e < V; ++e) tp2[e] = fma((double)x[j].v[e], vk, tp2[e]);
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
            tp2[e] = xor_range_sum<RP, WAVE>(tp2[e]);
        if (lane < RP)
#pragma unroll
            for (int e = 0; e < V; ++e) tred[buf][wv][rp * V + e] = tp2[e];
        __syncthreads();
        double t[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += tred[buf][w][rp * V + e];
            t[e] = (double)(T)s;
        }
        if (cg == 0 && rowok) {
            Pack<T, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = (T)t[e];
            st_pack<T, V>(tout + i0, o);
#pragma unroll
            for (int e = 0; e < V; ++e) ss = fma(t[e], t[e], ss);
        }
        if (sizeof(T) < sizeof(double)) {
#pragma unroll
            for (int j = 0; j < CPT; ++j)
#pragma unroll
                for (int e = 0; e < V; ++e) asm volatile("" : "+v"(x[j].v[e]));
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) pacc[j] = fma((double)x[j].v[e], t[e], pacc[j]);
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const double s = xor_range_sum<1, RP>(pacc[j]);
        const int k = cg + CG * j;
        if (rp == 0 && k < K) part[(i64)blockIdx.x * K + k] = s;
    }
    ss = block_sum<NW>(ss, sred);
    if (tid == 0) sspart[blockIdx.x] = ss;
}

// rc as launch_fused_pass.  np pending updates (1..DEFER_MAX); store: write the updated matrix to (dst, ldd, tsd).
template <typename T>
int launch_fused_defer(hipStream_t stream, int num_cu, const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd, i64 N,
                       int K, const double *v, int np, const PendingUpdates<T> &pend, bool store, T *tout, double *part,
                       int max_rows, double *sspart, int *nb, int *nss) {
    constexpr int V = 16 / sizeof(T);
    constexpr int R = tile_rows<T, 32>(), NT = 512, CG = 32;
    auto al = [](const void *p, i64 ld) { return ((uintptr_t)p % 16 == 0) && (ld % V == 0); };
    if (np < 1 || np > DEFER_MAX || K > CG * 16 || N < 1 || N % V != 0) return 1;
    if (!al(X, ldx) || !al(tout, V) || tsx % V != 0 || (store && (!al(dst, ldd) || tsd % V != 0))) return 1;
    for (int n = 0; n < np; ++n)
        if (!al(pend.t[n], V)) return 1;
    if ((i64)CG * ldx * (i64)sizeof(T) >= (1ll << 31) || (store && (i64)CG * ldd * (i64)sizeof(T) >= (1ll << 31))) return 1;
    const i64 ntiles = (N + R - 1) / R;
    const i64 grid = std::min<i64>(std::min<i64>((store ? 1 : 2) * (i64)num_cu, ntiles), max_rows);
    if (grid < 1) return 1;
    const dim3 g((unsigned)grid), b(NT);
#define DEFER_LAUNCH(CPT_, NP_, ST_)                                                                                  \
    hipLaunchKernelGGL((fused_defer_kernel<T, V, R, NT, CPT_, NP_, ST_>), g, b, 0, stream, X, ldx, tsx, dst, ldd, tsd, N, K, \
                       v, pend, tout, part, sspart)
#define DEFER_NP(CPT_)                                                                                                \
    do {                                                                                                              \
        switch (np * 2 + (store ? 1 : 0)) {                                                                           \
            case 2: DEFER_LAUNCH(CPT_, 1, false); break;                                                              \
            case 3: DEFER_LAUNCH(CPT_, 1, true); break;                                                               \
            case 4: DEFER_LAUNCH(CPT_, 2, false); break;                                                              \
            case 5: DEFER_LAUNCH(CPT_, 2, true); break;                                                               \
            case 6: DEFER_LAUNCH(CPT_, 3, false); break;                                                              \
            case 7: DEFER_LAUNCH(CPT_, 3, true); break;                                                               \
            case 8: DEFER_LAUNCH(CPT_, 4, false); break;                                                              \
            default: DEFER_LAUNCH(CPT_, 4, true); break;                                                              \
        }                                                                                                             \
    } while (0)
    if (K <= CG * 4) DEFER_NP(4);
    else if (K <= CG * 8) DEFER_NP(8);
    else DEFER_NP(16);
#undef DEFER_NP
#undef DEFER_LAUNCH
    *nb = (int)grid;
    *nss = (int)grid;
    return 0;
}

}  // namespace plsk
