// One-product-per-launch streaming kernels over the column-major N x K matrix X:
//   xb_kernel      out = X * Bm        (t = X r, src/pls.cpp:419; X B :449-451; X R :439-442)
//   xty_kernel     part = X^T Y        (XY = X^T Y, src/pls.cpp:396; p = X^T t, :421)
//   deflate_piece_kernel / deflate_kernel   dst = src - t p^T   (the north-star rank-1 deflation; no reference
//                  line): one contiguous 4 KB column piece per workgroup / the unaligned fallback
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
// out[i, m] = sum_k X[i,k] * Bm[k + m*ldb],  m < ncols <= MT.   One thread owns VEC consecutive rows.
// SS: also emit sum_i out[i,0]^2 per workgroup (t^T t partial, src/pls.cpp:420).
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int MT, bool SS>
__global__ __launch_bounds__(WG) void xb_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                const double *__restrict__ Bm, i64 ldb, int ncols,
                                                T *__restrict__ out, i64 ldo,
                                                double *__restrict__ sspart) {
    __shared__ double red[WG / WAVE];
    const i64 i0 = ((i64)blockIdx.x * WG + threadIdx.x) * VEC;
    double acc[VEC][MT];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[v][m] = 0.0;

    // columns m >= ncols of a partially used tile read a valid column again and are never stored:
    // no conditional loads (the masked form compiled to a branch per scalar load)
    i64 boff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) boff[m] = (i64)min(m, ncols - 1) * ldb;

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
                    const double b = Bm[(k + u) + boff[m]];  // wave-uniform: scalar load
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v][m] = fma((double)x[u].v[v], b, acc[v][m]);
                }
        }
        for (; k < K; ++k) {
            Pack<T, VEC> x = ld_pack_nt<T, VEC>(xp + (i64)k * ldx);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const double b = Bm[k + boff[m]];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v][m] = fma((double)x.v[v], b, acc[v][m]);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < ncols) {
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
                    for (int m = 0; m < MT; ++m)
                        if (m < ncols) acc[v][m] = fma(x, Bm[k + m * ldb], acc[v][m]);
                }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            if (v < nv)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (m < ncols) out[i0 + v + m * ldo] = (T)acc[v][m];
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
// t = X v for SHORT, WIDE matrices (few rows, very many columns: N = 512, K = 50,000 is a usual shape of the method).
// xb_kernel parallelises over rows only -- N / 256 workgroups, TWO of them for 512 rows, each walking all K columns: 3.5 ms
// for 200 MB.  Here the columns are split as well: grid = (row groups, KS column ranges), a thread owns VEC rows over the
// columns [k_lo, k_hi) of its range and leaves a partial score, part[ks * ldp + i]; xb_split_finish_kernel sums the KS
// partials of a row in range order (bit-reproducible), rounds to the storage type and forms the t^T t partials.
// ------------------------------------------------------------------------------------
// MT columns of Bm (ldb) at a time: part[(ks * MT + m) * ldp + i].
template <typename T, int VEC, int MT>
__global__ __launch_bounds__(WG) void xb_split_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K, int kper,
                                                      const double *__restrict__ v, i64 ldb, int ncols, double *__restrict__ part,
                                                      i64 ldp) {
    const i64 i0 = ((i64)blockIdx.x * WG + threadIdx.x) * VEC;
    const int k_lo = blockIdx.y * kper, k_hi = min(K, k_lo + kper);
    double acc[VEC][MT];
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[e][m] = 0.0;
    i64 boff[MT];  // columns m >= ncols read a valid column again and are never stored
#pragma unroll
    for (int m = 0; m < MT; ++m) boff[m] = (i64)min(m, ncols - 1) * ldb;
    if (i0 + VEC <= N) {
        const T *xp = X + i0;
        constexpr int U = 8;
        int k = k_lo;
        for (; k + U <= k_hi; k += U) {
            Pack<T, VEC> x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld_pack_nt<T, VEC>(xp + (i64)(k + u) * ldx);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const double b = v[k + u + boff[m]];  // wave-uniform: scalar load
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e][m] = fma((double)x[u].v[e], b, acc[e][m]);
                }
        }
        for (; k < k_hi; ++k) {
            const Pack<T, VEC> x = ld_pack_nt<T, VEC>(xp + (i64)k * ldx);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const double b = v[k + boff[m]];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e][m] = fma((double)x.v[e], b, acc[e][m]);
            }
        }
    } else if (i0 < N) {
        for (int k = k_lo; k < k_hi; ++k)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const double b = v[k + boff[m]];
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (i0 + e < N) acc[e][m] = fma((double)X[i0 + e + (i64)k * ldx], b, acc[e][m]);
            }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
        if (m < ncols)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (i0 + e < N) part[((i64)blockIdx.y * MT + m) * ldp + i0 + e] = acc[e][m];
}

// t[i] = sum over the KS ranges, in order; workgroup = 64 rows x 4 interleaved sets of ranges (LDS for the last step);
// blockIdx.y = column m of the MT the split kernel formed.
// sspart[blockIdx.x] = sum of the workgroup's t^2 (as stored), when asked for (one column only).
template <typename T>
__global__ __launch_bounds__(WG) void xb_split_finish_kernel(const double *__restrict__ part, i64 ldp, int KS, int MT, i64 N,
                                                             T *__restrict__ out, i64 ldo, double *__restrict__ sspart) {
    __shared__ double sh[4][64];
    __shared__ double red[WG / WAVE];
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6, m = blockIdx.y;
    const i64 i = (i64)blockIdx.x * 64 + r;
    // set q sums the ranges [q * per, (q + 1) * per): contiguous, so that the order of the whole sum is the range order
    const int per = (KS + 3) / 4, lo = q * per, hi = min(KS, lo + per);
    double s = 0.0;
    if (i < N) {
        int ks = lo;
        for (; ks + 8 <= hi; ks += 8) {  // 8 loads in flight, added in range order
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = part[((i64)(ks + u) * MT + m) * ldp + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += pv[u];
        }
        for (; ks < hi; ++ks) s += part[((i64)ks * MT + m) * ldp + i];
    }
    sh[q][r] = s;
    __syncthreads();
    double ss = 0.0;
    if (q == 0 && i < N) {
        const T t = (T)(((sh[0][r] + sh[1][r]) + sh[2][r]) + sh[3][r]);
        out[i + (i64)m * ldo] = t;
        ss = (double)t * (double)t;
    }
    if (sspart) {
        ss = block_sum<WG / WAVE>(ss, red);
        if (threadIdx.x == 0) sspart[blockIdx.x] = ss;
    }
}

// ------------------------------------------------------------------------------------
// out(N x ncols) = X * Bm for 4 < ncols <= 16 NB columns on the matrix cores (scores T = X R,
// src/pls.cpp:439-442; fitted values X B :449-451).  X*Bm -- unlike X^T Y -- is MFMA-shaped as it lies in
// memory: the M dimension of v_mfma_f64_16x16x4_f64 runs along the ROWS of X, the contiguous direction, so
// the A operand comes straight from global memory in 16-byte accesses (lane = (row group li, column lq of the
// 4-column step): a wave-load is 4 column segments of 256 bytes, the tile pattern of the fused pass) and the
// V rows of a lane's pack feed V separate MFMAs (row set e = rows row0 + V i + e).  At 4 flop per byte the
// kernel stays HBM-bound (the MFMA pipe could take 19 TB/s).  Accumulation is fp64 for both storage types.
// Used for fp32 storage beyond 8 columns (where the VALU kernel below holds only 8 per pass) and for fp64 storage beyond 32;
// with fp64 storage and 20 columns the VALU kernel is the faster one.
// A wave owns 16 V rows and walks all K; 4 waves per workgroup, one-shot workgroups.
// ------------------------------------------------------------------------------------
typedef double xb_f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------
// out(N x ncols) = X * Bm on the matrix cores with Bm staged through LDS: up to 16 NB columns per pass over X (NB <= 4 for
// fp64 storage: 64 columns; NB <= 2 for fp32: 32), a wave per 16 V rows.  Its predecessor (rounds 2-3) fetched every B operand from
// global memory (one 8-byte load per lane and MFMA: 2.6 ms per pass over 131,072 x 4,096 fp32, 0.8 TB/s, neither bandwidth
// nor matrix pipe); here a chunk of KB rows of Bm sits in LDS -- [k][ST] doubles with a row stride of 16 mod 32 doubles, which
// puts the two k-rows a half-wave reads on disjoint bank halves -- and one ds_read_b64 feeds V MFMAs.
template <typename T, int V, int NB>
__global__ __launch_bounds__(WG) void xb_mfma_lds_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                         const double *__restrict__ Bm, i64 ldb, int ncols,
                                                         T *__restrict__ out, i64 ldo) {
    constexpr int NC = 16 * NB, KB = 32, ST = (NB % 2) ? NC : NC + 16, U = KB / 4;  // (row stride = 16 mod 32 doubles)
    __shared__ __attribute__((aligned(16))) double bs[2][KB * ST];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    constexpr int RW = 16 * V;  // rows per wave
    const i64 row0 = ((i64)blockIdx.x * (WG / WAVE) + wv) * RW;
    const bool wave_on = row0 < N;  // (wave-uniform; idle waves still stage B and keep the barriers)
    const i64 r = row0 + (i64)V * li;
    const bool rfull = (r + V <= N);
    xb_f64x4 acc[V][NB];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[e][b] = xb_f64x4{0.0, 0.0, 0.0, 0.0};
    auto stage = [&](int buf, int k0) {  // Bm[k0 .. k0+KB) x ncols -> bs[buf][k][col], zero padded (consecutive threads: consecutive k)
        for (int j = tid; j < KB * NC; j += WG) {
            const int kk = j % KB, m = j / KB;
            bs[buf][kk * ST + m] = (k0 + kk < K && m < ncols) ? Bm[(k0 + kk) + (i64)m * ldb] : 0.0;
        }
    };
    auto load_x = [&](Pack<T, V> (&x)[U], int k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 4 * u + lq;
            if (k < K && rfull) {
                x[u] = ld_pack_nt<T, V>(X + r + (i64)k * ldx);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) x[u].v[e] = (k < K && r + e < N) ? X[r + e + (i64)k * ldx] : (T)0;
            }
        }
    };
    stage(0, 0);
    Pack<T, V> xc[U], xn[U];
    if (wave_on) load_x(xc, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KB, buf ^= 1) {
        // the next chunk -- B into the other LDS buffer, X into the second register set -- goes out ahead of this chunk's MFMAs
        if (k0 + KB < K) {
            stage(buf ^ 1, k0 + KB);
            if (wave_on) load_x(xn, k0 + KB);
        }
        if (wave_on) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double *brow = &bs[buf][(4 * u + lq) * ST + li];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const double bv = brow[16 * b];
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        acc[e][b] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)xc[u].v[e], bv, acc[e][b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xc[u] = xn[u];
        }
        __syncthreads();
    }
    if (!wave_on) return;
    // D layout: lane holds D[i = lq + 4 q][j = li]; row of X = row0 + V i + e, column of out = 16 b + j
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = 16 * b + li;
        if (col >= ncols) continue;
#pragma unroll
        for (int e = 0; e < V; ++e)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i64 row = row0 + (i64)V * (lq + 4 * q) + e;
                if (row < N) out[row + (i64)col * ldo] = (T)acc[e][b][q];
            }
    }
}

// ------------------------------------------------------------------------------------
// out(N x ncols) = X * Bm for 4 < ncols <= MT columns in ONE pass over X (scores T = X R with A columns,
// src/pls.cpp:439-442; fitted values X B :449-451).  With many columns the per-column scalar loads of the
// narrow kernel above expose their latency (SGPRs cannot hold K x ncols operands), so Bm is staged
// through LDS in chunks of KB rows, stored [k][MT] so that one k's columns are contiguous and every
// lane reads the same address (LDS broadcast, conflict-free), two columns per ds_read_b128.
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int MT, int NP = 1>
__global__ __launch_bounds__(WG) void xb_wide_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                     const double *__restrict__ Bm, i64 ldb, int ncols,
                                                     T *__restrict__ out, i64 ldo) {
    // NP row packs per lane (WG*VEC rows apart: every load stays a coalesced 4 KB piece): one LDS read of a B value
    // feeds NP*VEC FMAs.  The broadcast reads of B are the bound at 20 fp64 columns (10 ds_read_b128 per X pack).
    constexpr int KB = 64;
    __shared__ __attribute__((aligned(16))) double bs[2][KB * MT];
    const i64 i0 = (i64)blockIdx.x * (WG * VEC * NP) + (i64)threadIdx.x * VEC;
    constexpr i64 PS = (i64)WG * VEC;  // rows between the packs of a lane
    bool full = true;
#pragma unroll
    for (int p = 0; p < NP; ++p) full = full && (i0 + p * PS + VEC <= N);
    double acc[NP][VEC][MT];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[p][v][m] = 0.0;

    auto stage = [&](int buf, int k0) {  // Bm[k0 .. k0+KB) x ncols -> bs[buf][k][m], zero padded
        for (int j = threadIdx.x; j < KB * MT; j += WG) {
            const int kk = j % KB, m = j / KB;  // consecutive threads: consecutive k of one column (coalesced)
            bs[buf][kk * MT + m] = (k0 + kk < K && m < ncols) ? Bm[(k0 + kk) + (i64)m * ldb] : 0.0;
        }
    };
    stage(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KB, buf ^= 1) {
        if (k0 + KB < K) stage(buf ^ 1, k0 + KB);
        const int kn = min(KB, K - k0);
        constexpr int U = 8 / NP;
        for (int kb = 0; kb < kn; kb += U) {
            Pack<T, VEC> x[U][NP];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const i64 ip = i0 + p * PS;
                    if (full && kb + u < kn) x[u][p] = ld_pack_nt<T, VEC>(X + ip + (i64)(k0 + kb + u) * ldx);
                    else
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
                            x[u][p].v[v] = (kb + u < kn && ip + v < N) ? X[ip + v + (i64)(k0 + kb + u) * ldx] : (T)0;
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double *brow = &bs[buf][(kb + u) * MT];  // rows beyond kn hold zeros or stale data x 0
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const double b = brow[m];
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[p][v][m] = fma((double)x[u][p].v[v], b, acc[p][v][m]);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const i64 ip = i0 + p * PS;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (m < ncols) {
                if (ip + VEC <= N) {
                    Pack<T, VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = (T)acc[p][v][m];
                    st_pack<T, VEC>(out + ip + (i64)m * ldo, o);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (ip + v < N) out[ip + v + (i64)m * ldo] = (T)acc[p][v][m];
                }
            }
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
// The 8-response tile of X^T Y with the loads of the NEXT row chunk issued before the FMAs of the current one
// (register double buffer): xty_kernel<T, V, 4, 8> alternates "12 loads, wait, 128 FMAs" with two waves per SIMD to
// hide a 1-2 us load latency behind 0.25 us of arithmetic, and runs at 2.1 TB/s whatever the storage type.
// Same grid, same partial layout, same summation order as xty_kernel<T, VEC, 4, 8> (full 16-byte row packs only).
// ------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(WG) void xty8_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 ldy,
                                                  i64 N, int K, int M, int m0, double *__restrict__ part) {
    constexpr int KC = 4, MT = 8;
    __shared__ double red[WG / WAVE][KC * MT];
    const int k0 = blockIdx.y * KC;
    double acc[KC][MT];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[kc][m] = 0.0;
    constexpr i64 CH = (i64)WG * VEC;
    const i64 nfull = N / CH;  // chunks in which every lane has a full pack
    Pack<T, VEC> xa[KC], ya[MT], xb[KC], yb[MT];
    auto load = [&](i64 c, Pack<T, VEC> (&x)[KC], Pack<T, VEC> (&y)[MT]) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
#pragma unroll
        for (int m = 0; m < MT; ++m) y[m] = ld_pack<T, VEC>(Y + i0 + (i64)(m0 + m) * ldy);
#pragma unroll
        for (int u = 0; u < KC; ++u) x[u] = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + u) * ldx);
    };
    auto fmas = [&](const Pack<T, VEC> (&x)[KC], const Pack<T, VEC> (&y)[MT]) {
#pragma unroll
        for (int u = 0; u < KC; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[u][m] = fma((double)x[u].v[v], (double)y[m].v[v], acc[u][m]);
    };
    i64 c = blockIdx.x;
    if (c < nfull) load(c, xa, ya);
    while (c < nfull) {
        const i64 c1 = c + gridDim.x;
        if (c1 < nfull) load(c1, xb, yb);
        fmas(xa, ya);
        c = c1;
        if (c >= nfull) break;
        const i64 c2 = c + gridDim.x;
        if (c2 < nfull) load(c2, xa, ya);
        fmas(xb, yb);
        c = c2;
    }
    // the ragged last chunk (N % (256 VEC) rows), element-wise, by the workgroup whose turn it is
    if (nfull * CH < N && (nfull % gridDim.x) == blockIdx.x) {
        for (i64 i = nfull * CH + threadIdx.x; i < N; i += WG)
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                const double x = (double)X[i + (i64)(k0 + u) * ldx];
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[u][m] = fma(x, (double)Y[i + (i64)(m0 + m) * ldy], acc[u][m]);
            }
    }
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
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < WG / WAVE; ++i) s += red[i][threadIdx.x];
        part[(i64)blockIdx.x * ((i64)K * M) + (k0 + kc) + (i64)(m0 + m) * K] = s;
    }
}

// ------------------------------------------------------------------------------------
// dst[i,k] = src[i,k] - t[i] * p[k] on column-major matrices, one-shot workgroups: a workgroup is ONE
// contiguous 4 KB piece of one column (256 lanes x 16 bytes), one load and one store per lane.
// grid = (row blocks, K), row blocks fastest.  This is the fastest read+write form on MI355X
// (pls_amd/csrc/tune/rw_probe.hip, profiles/r1/rw_probe.txt: 6.15 TB/s out of place, 5.96 in place;
// two columns per workgroup 5.7, the 256-byte tile pattern 5.0): the t piece is re-read once per
// column, but from L2 / Infinity Cache, not from HBM.
// ------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(WG) void deflate_piece_kernel(const T *__restrict__ src, i64 lds, T *dst, i64 ldd,
                                                           i64 N, const T *__restrict__ t,
                                                           const double *__restrict__ p) {
    const i64 i0 = ((i64)blockIdx.x * WG + threadIdx.x) * VEC;
    const int k = blockIdx.y;
    const double pk = p[k];
    const T *s = src + (i64)k * lds;
    T *d = dst + (i64)k * ldd;
    if (i0 + VEC <= N) {
        const Pack<T, VEC> tv = ld_pack<T, VEC>(t + i0);
        Pack<T, VEC> x = ld_pack_nt<T, VEC>(s + i0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) x.v[v] = (T)fma(-(double)tv.v[v], pk, (double)x.v[v]);
        st_pack_nt<T, VEC>(d + i0, x);
    } else {
        for (i64 i = i0; i < N; ++i) d[i] = (T)fma(-(double)t[i], pk, (double)s[i]);
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
// Column statistics for the pre-processing in front of the fit (src/pls.cpp:69-111, src/main.cpp:24-25):
//   MODE 0: part[g][k] = sum_i X[i,k]                 (-> column mean)
//   MODE 1: part[g][k] = sum_i (X[i,k] - mean[k])^2   (-> SST, :69-73; sd = sqrt(SST/(N-1)), :79-83)
// Two passes like the reference (mean first, then squared deviations about it), same grid and
// reduction structure as xty_kernel.
// ------------------------------------------------------------------------------------
template <typename T, int VEC, int KC, int MODE>
__global__ __launch_bounds__(WG) void colstat_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                     const double *__restrict__ mean,
                                                     double *__restrict__ part) {
    __shared__ double red[WG / WAVE][KC];
    const int k0 = blockIdx.y * KC;
    const int kn = min(KC, K - k0);
    double acc[KC], mu[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        acc[kc] = 0.0;
        mu[kc] = (MODE == 1 && kc < kn) ? mean[k0 + kc] : 0.0;
    }
    constexpr i64 CH = (i64)WG * VEC;
    for (i64 c = blockIdx.x; c * CH < N; c += gridDim.x) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            if (kc < kn) {
                if (i0 + VEC <= N) {
                    const Pack<T, VEC> x = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + kc) * ldx);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double d = (double)x.v[v] - mu[kc];
                        acc[kc] = (MODE == 0) ? acc[kc] + d : fma(d, d, acc[kc]);
                    }
                } else {
                    for (int v = 0; v < VEC; ++v)
                        if (i0 + v < N) {
                            const double d = (double)X[i0 + v + (i64)(k0 + kc) * ldx] - mu[kc];
                            acc[kc] = (MODE == 0) ? acc[kc] + d : fma(d, d, acc[kc]);
                        }
                }
            }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const double s = wave_sum(acc[kc]);
        if (lane == 0) red[w][kc] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < kn) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < WG / WAVE; ++i) s += red[i][threadIdx.x];
        part[(i64)blockIdx.x * K + k0 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------
// The same statistics in ONE sweep (single rank).  Every WAVE keeps, per column, the sums of (x - s) and (x - s)^2 about
// the first element s IT reads of that column (a scalar register pair), and turns them into (count, mean, M2 = sum of
// squared deviations about that mean) of its rows: the cancellation in S2 - S1^2/cnt is eps * ((s - mean)/sd)^2, a few
// eps for any s drawn from the column -- and at most eps * cnt (a wave's few thousand rows) if s is an outlier, because
// the outlier's own deviation is then part of M2.  No column offset enters.  The triples are merged pairwise with the
// update of Chan, Golub & LeVeque (mean += d*nb/n, M2 += M2b + d^2*na*nb/n): waves -> workgroup here, workgroups in
// colmoments_finish_kernel.  The merge is FIRST order in the error of the two means (the two-pass form is second order
// in the error of its one mean), so the means travel as unevaluated sums hi + lo (two_sum): with a column offset of 1e8
// sd a plain fp64 mean would be rounded at 1e-8 sd and every d = mean_b - mean_a with it; hi_b - hi_a is exact there
// (Sterbenz) and lo carries what the rounding dropped.
//   part[g][0..2][k] = mean hi, mean lo, M2 of workgroup g's rows;  cnt[g] = its row count.
// A constant column gives M2 = 0 exactly (every difference is 0), like the two-pass form.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
    s = a + b;
    const double bb = s - a;
    e = (a - (s - bb)) + (b - bb);
}
// (na, ah + al, qa) <- merged with (nb, bh + bl, qb); wb = nb / (na + nb), wc = na * wb
__device__ __forceinline__ void moments_merge(double &ah, double &al, double &qa, double bh, double bl, double qb, double wb,
                                              double wc) {
    const double d = (bh - ah) + (bl - al);
    qa = qa + qb + d * d * wc;
    double e;
    two_sum(ah, d * wb, ah, e);
    al += e;
}
__device__ __forceinline__ double wave_first(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

template <typename T, int VEC, int KC>
__global__ __launch_bounds__(WG) void colmoments_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K,
                                                        double *__restrict__ part, double *__restrict__ cnt) {
    __shared__ double red[WG / WAVE][3][KC];
    __shared__ double redn[WG / WAVE];
    const int k0 = blockIdx.y * KC;
    const int kn = min(KC, K - k0);
    constexpr i64 CH = (i64)WG * VEC;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 wfirst = (i64)blockIdx.x * CH + (i64)w * WAVE * VEC;  // the first row this wave reads (wave-uniform)
    double s1[KC], s2[KC], mu[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        s1[kc] = 0.0; s2[kc] = 0.0;
        mu[kc] = wave_first((kc < kn && wfirst < N) ? (double)X[wfirst + (i64)(k0 + kc) * ldx] : 0.0);
    }
    double n = 0.0;
    for (i64 c = blockIdx.x; c * CH < N; c += gridDim.x) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
        if (i0 + VEC <= N) {
            n += (double)VEC;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                if (kc < kn) {
                    const Pack<T, VEC> x = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + kc) * ldx);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const double d = (double)x.v[v] - mu[kc];
                        s1[kc] += d;
                        s2[kc] = fma(d, d, s2[kc]);
                    }
                }
        } else if (i0 < N) {
            n += (double)(N - i0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                if (kc < kn)
                    for (int v = 0; v < VEC; ++v)
                        if (i0 + v < N) {
                            const double d = (double)X[i0 + v + (i64)(k0 + kc) * ldx] - mu[kc];
                            s1[kc] += d;
                            s2[kc] = fma(d, d, s2[kc]);
                        }
        }
    }
    // the wave's (count, mean hi + lo, M2)
    n = wave_sum(n);
    const double rn = n > 0.0 ? 1.0 / n : 0.0;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const double a = wave_sum(s1[kc]), b = wave_sum(s2[kc]);
        if (lane == 0) {
            const double m1 = a * rn;
            double hi, lo;
            two_sum(mu[kc], m1, hi, lo);
            red[w][0][kc] = hi; red[w][1][kc] = lo; red[w][2][kc] = fmax(fma(-a, m1, b), 0.0);
        }
    }
    if (lane == 0) redn[w] = n;
    __syncthreads();
    if ((int)threadIdx.x < KC) {
        double na = redn[0], ah = red[0][0][threadIdx.x], al = red[0][1][threadIdx.x], qa = red[0][2][threadIdx.x];
#pragma unroll
        for (int i = 1; i < WG / WAVE; ++i) {
            const double nb = redn[i], nt = na + nb;
            const double wb = nt > 0.0 ? nb / nt : 0.0;
            moments_merge(ah, al, qa, red[i][0][threadIdx.x], red[i][1][threadIdx.x], red[i][2][threadIdx.x], wb, na * wb);
            na = nt;
        }
        if ((int)threadIdx.x < kn) {
            part[((i64)blockIdx.x * 3 + 0) * K + k0 + threadIdx.x] = ah;
            part[((i64)blockIdx.x * 3 + 1) * K + k0 + threadIdx.x] = al;
            part[((i64)blockIdx.x * 3 + 2) * K + k0 + threadIdx.x] = qa;
        }
        if (threadIdx.x == 0 && blockIdx.y == 0) cnt[blockIdx.x] = na;
    }
}

// merge of the G workgroup triples of a column, in order (G <= a few thousand): mean[k], sd[k] = sqrt(M2 / (n - 1)).
// With tri given (a row-sharded matrix): the shard's own triple instead, tri[0..2][k] = mean hi, mean lo, M2 and
// tri[3 K] = its row count, for the two exchanges below.
__global__ __launch_bounds__(WG) void colmoments_finish_kernel(const double *__restrict__ part, const double *__restrict__ cnt,
                                                               int G, int K, double *__restrict__ mean, double *__restrict__ sd,
                                                               double *__restrict__ tri) {
    const int k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    double na = cnt[0], ah = part[k], al = part[(i64)K + k], qa = part[2 * (i64)K + k];
    for (int g = 1; g < G; ++g) {
        const double nb = cnt[g], nt = na + nb;
        const double wb = nt > 0.0 ? nb / nt : 0.0;
        moments_merge(ah, al, qa, part[((i64)g * 3 + 0) * K + k], part[((i64)g * 3 + 1) * K + k], part[((i64)g * 3 + 2) * K + k], wb,
                      na * wb);
        na = nt;
    }
    if (tri) {
        tri[k] = ah; tri[(i64)K + k] = al; tri[2 * (i64)K + k] = qa;
        if (k == 0) tri[3 * (i64)K] = na;
        return;
    }
    mean[k] = ah + al;
    sd[k] = sqrt((na < 2.0 ? 0.0 : qa) / (na - 1.0));
}

// Row-sharded statistics from the shards' triples with TWO all-reduces of K sums (the two-pass form needs two as well,
// and a second sweep of X):
//   step 0: buf[k] = n_r * mean_r[k]                         -> sum over ranks / n = the global mean g
//   step 1: mean[k] = g;  buf[k] = M2_r + n_r (mean_r - g)^2  -> sum over ranks = SST about g
//           (every term is >= 0, and an error d in g adds n d^2: second order, like the two-pass form)
//   step 2: sd[k] = sqrt(SST / (n - 1))
__global__ __launch_bounds__(WG) void colmoments_shard_kernel(const double *__restrict__ tri, int K, double n_total, int step,
                                                              double *__restrict__ buf, double *__restrict__ mean,
                                                              double *__restrict__ sd) {
    const int k = blockIdx.x * WG + threadIdx.x;
    if (k >= K) return;
    const double nr = tri[3 * (i64)K];
    if (step == 0) {
        buf[k] = nr * (tri[k] + tri[(i64)K + k]);
    } else if (step == 1) {
        const double g = buf[k] / n_total;
        mean[k] = g;
        const double d = (tri[k] - g) + tri[(i64)K + k];
        buf[k] = nr > 0.0 ? fma(nr * d, d, tri[2 * (i64)K + k]) : 0.0;
    } else {
        sd[k] = sqrt((n_total < 2.0 ? 0.0 : buf[k]) / (n_total - 1.0));
    }
}

// Z[i,k] = (X[i,k] - mean[k]) / sd[k]   (src/pls.cpp:93-105: the division is by the UNGUARDED stdev,
// so a constant column becomes NaN exactly as in the reference).  grid = (row groups, column groups).
template <typename T, int VEC, int KC>
__global__ __launch_bounds__(WG) void zscale_kernel(const T *X, i64 ldx, T *Z, i64 ldz, i64 N, int K,
                                                    const double *__restrict__ mean,
                                                    const double *__restrict__ sd) {
    const int k0 = blockIdx.y * KC;
    const int kn = min(KC, K - k0);
    constexpr i64 CH = (i64)WG * VEC;
    // The quotient y / sd is formed as q = y*r, q += fma(-q, sd, y) * r with r = 1/sd correctly rounded: the correctly
    // rounded quotient (Markstein) in 3 full-rate instructions instead of the ~15 of the division sequence with its
    // quarter-rate v_rcp_f64.  sd = 0, NaN or so large / small that r or the residual could leave the normal range: the
    // plain division (src/pls.cpp:103 divides by the unguarded sd -- a constant column is NaN / inf there and here).
    // (One column at a time, eight waves per SIMD: measured faster than all KC packs in flight at four.)
    for (i64 c = blockIdx.x; c * CH < N; c += gridDim.x) {
        const i64 i0 = c * CH + (i64)threadIdx.x * VEC;
        for (int kc = 0; kc < kn; ++kc) {
            const double m = mean[k0 + kc], sdev = sd[k0 + kc], rinv = 1.0 / sdev;
            const bool fast = fabs(sdev) > 1e-150 && fabs(sdev) < 1e150;
            auto quot = [&](double y) -> double {
                if (!fast) return y / sdev;
                const double q = y * rinv;
                const double q1 = fma(fma(-q, sdev, y), rinv, q);
                return (fabs(q) > 1e-290 && fabs(q) < 1e290) ? q1 : y / sdev;  // (near the subnormals the residual is not exact)
            };
            if (i0 + VEC <= N) {
                Pack<T, VEC> x = ld_pack_nt<T, VEC>(X + i0 + (i64)(k0 + kc) * ldx);
#pragma unroll
                for (int v = 0; v < VEC; ++v) x.v[v] = (T)quot((double)x.v[v] - m);
                st_pack_nt<T, VEC>(Z + i0 + (i64)(k0 + kc) * ldz, x);
            } else {
                for (int v = 0; v < VEC; ++v)
                    if (i0 + v < N) Z[i0 + v + (i64)(k0 + kc) * ldz] = (T)quot((double)X[i0 + v + (i64)(k0 + kc) * ldx] - m);
            }
        }
    }
}

// The same scale pass as one-shot workgroups on ONE contiguous column piece each (grid = (row pieces, K), the shape of
// deflate_piece_kernel): used for 16-byte aligned columns.
template <typename T, int VEC>
__global__ __launch_bounds__(WG) void zscale_piece_kernel(const T *X, i64 ldx, T *Z, i64 ldz, i64 N,
                                                          const double *__restrict__ mean, const double *__restrict__ sd) {
    const i64 i0 = ((i64)blockIdx.x * WG + threadIdx.x) * VEC;
    const int k = blockIdx.y;
    const double m = mean[k], s = sd[k], r = 1.0 / s;
    const bool fast = fabs(s) > 1e-150 && fabs(s) < 1e150;
    auto quot = [&](double y) -> double {
        if (!fast) return y / s;
        const double q = y * r;
        const double q1 = fma(fma(-q, s, y), r, q);
        return (fabs(q) > 1e-290 && fabs(q) < 1e290) ? q1 : y / s;
    };
    const T *xs = X + (i64)k * ldx;
    T *zs = Z + (i64)k * ldz;
    if (i0 + VEC <= N) {
        Pack<T, VEC> x = ld_pack_nt<T, VEC>(xs + i0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) x.v[v] = (T)quot((double)x.v[v] - m);
        st_pack_nt<T, VEC>(zs + i0, x);
    } else {
        for (i64 i = i0; i < N; ++i) zs[i] = (T)quot((double)xs[i] - m);
    }
}

// SSE of the model with 1..A components in ONE sweep over the scores (src/pls.cpp:453-459, and the
// per-ncomp loop of print_explained_variance :551-562, which upstream costs A full X*B passes):
// with S = X R (N x A) the fitted values are Yhat_c = S[:, :c] Q[:, :c]^T, so a row's residuals for
// c = 1..A follow from one running sum.  part[g][m + c*M] = sum over the group's rows of
// (Y[i,m] - Yhat_c[i,m])^2.  One thread per row; Q (M x A) read through the scalar cache.
template <typename T>
__global__ __launch_bounds__(WG) void sse_components_kernel(const T *__restrict__ S, i64 lds_,
                                                            const T *__restrict__ Y, i64 ldy, i64 N,
                                                            int c_lo, int c_hi, int M, const double *__restrict__ Q,
                                                            double *__restrict__ part) {
    // component counts c_lo+1 .. c_hi of the model (the fitted values of the first c_lo components are rebuilt
    // without being recorded, so that a long component list can be covered in ranges of (c_hi - c_lo)*M <= 1024 sums)
    extern __shared__ double acc[];  // [WG/WAVE][(c_hi - c_lo)*M] per-wave running sums (no barrier in the sweep)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int AM = (c_hi - c_lo) * M;
    double *mine = acc + (i64)wv * AM;
    for (int j = lane; j < AM; j += WAVE) mine[j] = 0.0;
    for (i64 i0 = (i64)blockIdx.x * WG; i0 < N; i0 += (i64)gridDim.x * WG) {
        const i64 i = i0 + threadIdx.x;
        for (int m = 0; m < M; ++m) {
            const double y = (i < N) ? (double)Y[i + (i64)m * ldy] : 0.0;
            double yhat = 0.0;
            for (int c = 0; c < c_hi; ++c) {
                const double s = (i < N) ? (double)S[i + (i64)c * lds_] : 0.0;
                yhat = fma(s, Q[m + (i64)c * M], yhat);
                if (c < c_lo) continue;
                const double e = y - yhat;
                const double tot = wave_sum(e * e);
                if (lane == 0) mine[m + (c - c_lo) * M] += tot;
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < AM; j += WG) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WG / WAVE; ++w) t += acc[(i64)w * AM + j];
        part[(i64)blockIdx.x * AM + j] = t;
    }
}

// ------------------------------------------------------------------------------------
// Fixed-order sum of per-workgroup partials, in RED_SLICES independent slices so that enough
// workgroups take part:   red[s*LP + j] = sum_{b in slice s} part[b*L + j]   (j < L),
// and, when nss > 0,      red[s*LP + L] = sum_{b in slice s of nss} sspart[b]     (LP = L+1).
// With nss == 0, LP = L.  The consumer (component_update_kernel, after the all-reduce of the
// whole RED_SLICES*LP buffer in a sharded fit) adds the slices in index order.
// out_stride > 0 replaces LP as the distance between the slices of `red` (a column block of a larger sliced matrix).
// grid = (ceil(L/64), RED_SLICES); 256 threads = 64 columns x 4 interleaved sub-slices.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void reduce_partials_kernel(const double *__restrict__ part,
                                                             int nb, int L,
                                                             const double *__restrict__ sspart,
                                                             int nss, double *__restrict__ red,
                                                             i64 out_stride) {
    __shared__ double sm[4][64];
    __shared__ double sm1[WG / WAVE];
    const i64 LP = out_stride > 0 ? out_stride : (i64)(L + (nss > 0 ? 1 : 0));
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
