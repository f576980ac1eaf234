// One-product-per-launch streaming kernels over the column-major N x K matrix X:
//   xb_kernel      out = X * Bm        (t = X r, src/pls.cpp:419; X B :449-451; X R :439-442)
//   xty_kernel     part = X^T Y        (XY = X^T Y, src/pls.cpp:396; p = X^T t, :421)
//   deflate_kernel dst = src - t p^T   (the north-star rank-1 deflation; no reference line)
//   reduce_partials_kernel             fixed-order sum of per-workgroup partials
// Every one is HBM-bound (<= 4 flop/B).  Layout rule: element (i,k) is at
// X[i + k*ld], so lanes map to consecutive ROWS -- a wave's load of one column is one
// contiguous 1 KiB segment (16 B per lane), the operand that varies with k (r_k, p_k, B[k,m])
// is wave-uniform and comes through the scalar cache, and the contraction over rows is a
// per-lane accumulation finished by a wave butterfly + LDS step once per workgroup.
#pragma once
#include "common.hpp"

namespace plsk {

// ------------------------------------------------------------------------------------
// out[i, m] = sum_k X[i,k] * Bm[k + m*ldb],  m < MT.   One thread owns VEC consecutive rows.
// SS: also emit sum_i out[i,0]^2 per workgroup (t^T t partial, src/pls.cpp:420).
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int MT, bool SS>
__global__ __launch_bounds__(WG) void xb_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                const double *__restrict__ Bm, i64 ldb,
                                                T *__restrict__ out, i64 ldo,
                                                double *__restrict__ sspart) {
    __shared__ double red[WG / WAVE];
    const i64 i0 = ((i64)blockIdx.x * WG + threadIdx.x) * VEC;
    double acc[VEC][MT];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[v][m] = 0.0;

    if (i0 + VEC <= N) {
        const T *xp = X + i0;
        constexpr int U = 8;
        int k = 0;
        for (; k + U <= K; k += U) {
            Pack<T, VEC> x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld_pack_nt<T, VEC>(xp + (i64)(k + u) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const double b = Bm[(k + u) + m * ldb];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v][m] = fma((double)x[u].v[v], b, acc[v][m]);
                }
        }
        for (; k < K; ++k) {
            Pack<T, VEC> x = ld_pack_nt<T, VEC>(xp + (i64)k * ldx);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const double b = Bm[k + m * ldb];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v][m] = fma((double)x.v[v], b, acc[v][m]);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            Pack<T, VEC> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.v[v] = (T)acc[v][m];
            st_pack<T, VEC>(out + i0 + m * ldo, o);
        }
    } else if (i0 < N) {  // ragged tail: element-wise
        const int nv = (int)(N - i0);
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (v < nv) {
                    const double x = (double)X[i0 + v + (i64)k * ldx];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[v][m] = fma(x, Bm[k + m * ldb], acc[v][m]);
                }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (v < nv)
#pragma unroll
                for (int m = 0; m < MT; ++m) out[i0 + v + m * ldo] = (T)acc[v][m];
    }
    if (SS) {
        // the sum of squares uses the value as STORED (rounded to T), so that t^T t matches
        // what a later pass over the stored scores would see
        double ss = 0.0;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (i0 + v < N) {
                const double tv = (double)(T)acc[v][0];
                ss = fma(tv, tv, ss);
            }
        ss = block_sum<WG / WAVE>(ss, red);
        if (threadIdx.x == 0) sspart[blockIdx.x] = ss;
    }
}

// ------------------------------------------------------------------------------------
// part[blockIdx.x][(k0+kc) + (m0+m)*K] = sum over this workgroup's rows of X[i,k0+kc]*Y[i,m0+m]
// grid = (row groups G, column groups ceil(K/KC)); a workgroup walks row chunks
// blockIdx.x, blockIdx.x+G, ... of WG*VEC rows and keeps KC*MT fp64 accumulators per lane.
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int KC, int MT>
__global__ __launch_bounds__(WG) void xty_kernel(const T *__restrict__ X, i64 ldx,
                                                 const T *__restrict__ Y, i64 ldy, i64 N, int K,
                                                 int M, int m0, double *__restrict__ part) {
    __shared__ double red[WG / WAVE][KC * MT];
    const int k0 = blockIdx.y * KC;
    const int kn = min(KC, K - k0);
    const int mn = min(MT, M - m0);
    double acc[KC][MT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[kc][m] = 0.0;

    constexpr i64 CH = (i64)WG * VEC;
    for (i64 c = blockIdx.x; c * CH < N; c += gridDim.x) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
        if (i0 + VEC <= N) {
            Pack<T, VEC> y[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < mn) y[m] = ld_pack<T, VEC>(Y + i0 + (i64)(m0 + m) * ldy);
            if (kn == KC) {
                constexpr int U = (KC < 8) ? KC : 8;  // loads in flight per lane per batch
#pragma unroll
                for (int kb = 0; kb < KC; kb += U) {
                    Pack<T, VEC> x[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) x[u] = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + kb + u) * ldx);
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            if (m < mn)
#pragma unroll
                                for (int v = 0; v < VEC; ++v)
                                    acc[kb + u][m] = fma((double)x[u].v[v], (double)y[m].v[v], acc[kb + u][m]);
                }
            } else {
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
                    if (kc < kn) {
                        Pack<T, VEC> x = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + kc) * ldx);
#pragma unroll
                        for (int m = 0; m < MT; ++m)
                            if (m < mn)
#pragma unroll
                                for (int v = 0; v < VEC; ++v)
                                    acc[kc][m] = fma((double)x.v[v], (double)y[m].v[v], acc[kc][m]);
                    }
            }
        } else if (i0 < N) {
            const int nv = (int)(N - i0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                if (kc < kn)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (v < nv) {
                            const double x = (double)X[i0 + v + (i64)(k0 + kc) * ldx];
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                if (m < mn)
                                    acc[kc][m] = fma(x, (double)Y[i0 + v + (i64)(m0 + m) * ldy], acc[kc][m]);
                        }
        }
    }
    // workgroup reduction: butterfly per value, then waves in order
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const double s = wave_sum(acc[kc][m]);
            if (lane == 0) red[w][kc * MT + m] = s;
        }
    __syncthreads();
    if (threadIdx.x < KC * MT) {
        const int kc = threadIdx.x / MT, m = threadIdx.x % MT;
        if (kc < kn && m < mn) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < WG / WAVE; ++i) s += red[i][threadIdx.x];
            part[(i64)blockIdx.x * ((i64)K * M) + (k0 + kc) + (i64)(m0 + m) * K] = s;
        }
    }
}

// ------------------------------------------------------------------------------------
// dst[i,k] = src[i,k] - t[i] * p[k].   grid = (row groups, column groups of KC).
// Algorithmic bytes: 2*N*K*s + N*s + K*8.
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int KC>
__global__ __launch_bounds__(WG) void deflate_kernel(const T *__restrict__ src, i64 lds,
                                                     T *__restrict__ dst, i64 ldd, i64 N, int K,
                                                     const T *__restrict__ t,
                                                     const double *__restrict__ p) {
    const int k0 = blockIdx.y * KC;
    const int kn = min(KC, K - k0);
    constexpr i64 CH = (i64)WG * VEC;
    constexpr int U = 8;
    static_assert(KC % U == 0, "KC must be a multiple of the unroll");
    for (i64 c = blockIdx.x; c * CH < N; c += gridDim.x) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
        if (i0 + VEC <= N) {
            const Pack<T, VEC> tv = ld_pack<T, VEC>(t + i0);
            double td[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) td[v] = -(double)tv.v[v];
            if (kn == KC) {
#pragma unroll
                for (int kb = 0; kb < KC; kb += U) {
                    Pack<T, VEC> x[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) x[u] = ld_pack_nt<T, VEC>(src + i0 + (i64)(k0 + kb + u) * lds);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double pk = p[k0 + kb + u];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) x[u].v[v] = (T)fma(td[v], pk, (double)x[u].v[v]);
                        st_pack_nt<T, VEC>(dst + i0 + (i64)(k0 + kb + u) * ldd, x[u]);
                    }
                }
            } else {
                for (int kc = 0; kc < kn; ++kc) {
                    Pack<T, VEC> x = ld_pack_nt<T, VEC>(src + i0 + (i64)(k0 + kc) * lds);
                    const double pk = p[k0 + kc];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) x.v[v] = (T)fma(td[v], pk, (double)x.v[v]);
                    st_pack_nt<T, VEC>(dst + i0 + (i64)(k0 + kc) * ldd, x);
                }
            }
        } else if (i0 < N) {
            const int nv = (int)(N - i0);
            for (int kc = 0; kc < kn; ++kc) {
                const double pk = p[k0 + kc];
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (v < nv)
                        dst[i0 + v + (i64)(k0 + kc) * ldd] =
                            (T)fma(-(double)t[i0 + v], pk, (double)src[i0 + v + (i64)(k0 + kc) * lds]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Fixed-order sum of per-workgroup partials, in RED_SLICES independent slices so that enough
// workgroups take part:   red[s*LP + j] = sum_{b in slice s} part[b*L + j]   (j < L),
// and, when nss > 0,      red[s*LP + L] = sum_{b in slice s of nss} sspart[b]     (LP = L+1).
// With nss == 0, LP = L.  The consumer (component_update_kernel, after the all-reduce of the
// whole RED_SLICES*LP buffer in a sharded fit) adds the slices in index order.
// grid = (ceil(L/64), RED_SLICES); 256 threads = 64 columns x 4 interleaved sub-slices.
// ------------------------------------------------------------------------------------
constexpr int RED_SLICES = 8;

__global__ __launch_bounds__(WG) void reduce_partials_kernel(const double *__restrict__ part,
                                                             int nb, int L,
                                                             const double *__restrict__ sspart,
                                                             int nss, double *__restrict__ red) {
    __shared__ double sm[4][64];
    __shared__ double sm1[WG / WAVE];
    const int LP = L + (nss > 0 ? 1 : 0);
    const int sl = blockIdx.y;
    const int jl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + jl;
    const int lo = (int)((i64)nb * sl / RED_SLICES), hi = (int)((i64)nb * (sl + 1) / RED_SLICES);
    double a = 0.0;
    if (j < L) {
        int b = lo + q;
        for (; b + 12 < hi; b += 16) {  // 4 independent loads in flight per lane
            const double x0 = part[(i64)b * L + j], x1 = part[(i64)(b + 4) * L + j];
            const double x2 = part[(i64)(b + 8) * L + j], x3 = part[(i64)(b + 12) * L + j];
            a += x0; a += x1; a += x2; a += x3;
        }
        for (; b < hi; b += 4) a += part[(i64)b * L + j];
    }
    sm[q][jl] = a;
    __syncthreads();
    if (q == 0 && j < L) red[(i64)sl * LP + j] = (sm[0][jl] + sm[1][jl]) + (sm[2][jl] + sm[3][jl]);
    if (nss > 0 && blockIdx.x == 0) {
        const int slo = (int)((i64)nss * sl / RED_SLICES), shi = (int)((i64)nss * (sl + 1) / RED_SLICES);
        double s = 0.0;
        for (int b = slo + threadIdx.x; b < shi; b += WG) s += sspart[b];
        s = block_sum<WG / WAVE>(s, sm1);
        if (threadIdx.x == 0) red[(i64)sl * LP + L] = s;
    }
}

}  // namespace plsk
