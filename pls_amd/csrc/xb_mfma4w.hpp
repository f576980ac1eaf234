// xb_mfma4w.hpp -- out(N x ncols) = X * Bm, 4 < ncols <= 4 NCG, on v_mfma_f64_4x4x4_4b_f64 where xb_mfma4_kernel (xb_mfma4.hpp:
// lane layouts, addressing, the barrier ahead of the stores) does not apply:
//   * Bm does not fit in LDS (K x ncols x 8 bytes beyond ~150 KB: 1,024 x 20, config 4's 4,096 columns) -- here it passes
//     through LDS in WINDOWS of KC rows, two buffers: the next window's values are loaded into registers at the start of
//     a window and written to the other buffer at its end, one barrier per window;
//   * the matrix is SHORT -- fewer than 16 row tiles per workgroup, so 16 waves that each own a tile would leave waves (or
//     CUs) idle: the 16 waves of a workgroup are TW tile slots x SW <= 8 sub-windows (TW SW = 16); wave (tw, sw) walks the
//     sw-th part of every window for tile slot tw, and the SW partial sums of a tile meet in LDS at the end of the round
//     in a fixed order.
// A round of a workgroup = TW tiles x all K; the accumulators live across the windows of a round.  Every wave of a
// workgroup runs the same number of rounds, windows and batches (a wave without a tile loads out-of-range offsets and
// stores nothing), so the barriers are uniform.  The last tile may be partial: lanes whose rows lie beyond N load
// out-of-range offsets, a pack that straddles N reads the padding of the column (ldx is a multiple of the pack: checked
// by the launcher) and the stores of that tile are guarded per element.
// Scores T = X R and fitted values (src/pls.cpp:439-442, :449-451) of wide or short matrices: 131,072 x 4,096 fp32 with
// 8 / 20 columns 0.68 / 0.72 ms -> see profiles/r5/xb_cols.txt.
#pragma once
#include "xb_mfma4.hpp"

namespace plsk {

__host__ __device__ constexpr int xb4w_u(int v, int ncg) { return v * ncg > 10 ? 2 : 4; }  // column steps per batch

// rows of Bm per window (a power of two: the staging's index arithmetic is shifts): as many as two buffers of [KC][ST] doubles
// fit in ~150 KB of LDS and 8 doubles per thread carry -- 1,024 for 8 columns, 512 up to 16, 256 beyond
// (fp32 storage, V = 4 rows per lane: twice the accumulators -- 128 rows from 16 columns on, so that nothing spills in the loop)
__host__ __device__ constexpr int xb4w_kcl2(int v, int ncg) { return ncg <= 2 ? 10 : ncg <= 3 ? 9 : v > 2 ? 7 : ncg <= 4 ? 9 : 8; }

template <typename T, int V, int NCG>
__global__ __launch_bounds__(XB4_WG) void xb_mfma4w_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K, const double *__restrict__ Bm,
                                                           i64 ldb, int ncols, T *__restrict__ out, i64 ldo, int SW, int kper, double *__restrict__ part, i64 ldp) {
    constexpr int NC = 4 * NCG, ST = xb4_stride(NCG), U = xb4w_u(V, NCG), RW = 16 * V, NWV = XB4_WG / WAVE;
    constexpr uint32_t OOR = 0xFFFFFFF0u;
    constexpr int KL = xb4w_kcl2(V, NCG), KC = 1 << KL;
    constexpr int BR = (KC * NC + XB4_WG - 1) / XB4_WG;  // doubles of the next window a thread carries through a window (<= 8)
    extern __shared__ __attribute__((aligned(16))) double xb4w_bs[];  // [2][KC][ST]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4, lj = lane & 3;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int TW = NWV / SW, tw = wvu % TW, sw = wvu / TW;
    const i64 ntiles = (N + RW - 1) / RW;
    const int G = gridDim.x;
    const i64 mytiles = (ntiles - blockIdx.x + G - 1) / G;       // tiles b, b + G, b + 2 G ... of this workgroup
    const int rounds = (int)((mytiles + TW - 1) / TW);
    // part != nullptr: the columns are split over blockIdx.y as well (VERY short matrices: fewer tiles than CUs) -- this
    // workgroup walks [k_lo, k_hi) and leaves fp64 partial sums, part[(blockIdx.y * 4 NCG + column) * ldp + row], which
    // xb_split_finish_kernel (stream_kernels.hpp) adds in range order
    const int k_lo = part ? (int)blockIdx.y * kper : 0, k_hi = part ? min(K, k_lo + kper) : K;
    const int NQ = (k_hi - k_lo + KC - 1) / KC;                  // windows
    const int KS = KC / SW, nbw = KS / (4 * U);                  // a wave's part of a window, in column steps and batches
    const uint32_t voff = (uint32_t)((V * li + (i64)lq * ldx) * (i64)sizeof(T));
    const int cstep = (int)(4 * ldx * (i64)sizeof(T));
    const uint32_t soff = (uint32_t)((V * (4 * ((lane >> 2) & 3) + lq) + (i64)lj * ldo) * (i64)sizeof(T));
    const int ostep = (int)(4 * ldo * (i64)sizeof(T));

    // window q of Bm -> registers (consecutive threads: consecutive k of one column), registers -> buffer
    double breg[BR];
    auto fetch_b = [&](int q) {
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int j = tid + i * XB4_WG, kk = j & (KC - 1), m = j >> KL, k = k_lo + q * KC + kk;
            breg[i] = (j < KC * NC && k < k_hi && m < ncols) ? Bm[k + (i64)m * ldb] : 0.0;
        }
    };
    auto put_b = [&](int buf) {
        double *bs = xb4w_bs + (size_t)buf * KC * ST;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int j = tid + i * XB4_WG, kk = j & (KC - 1), m = j >> KL;
            if (j < KC * NC) bs[kk * ST + m] = breg[i];
        }
    };
    // tile of (round r, slot tw) -- or -1
    auto tile_of = [&](int r) -> i64 {
        const i64 t = ((i64)r * TW + tw) * G + blockIdx.x;
        return t < ntiles ? t : -1;
    };
    auto load_x = [&](Pack<T, V> (&x)[U], i64 t, int k0) {  // columns k0 + 4 u + lq of tile t (t < 0: nothing)
        const i64 tt = t < 0 ? 0 : t;
        // (the descriptor ends with the matrix: a straddling pack of the last column reads nothing beyond the caller's allocation)
        const i64 ext = (tt + 1) * RW <= N ? (i64)0x7fffffff  // (a full tile: no bound to compute)
                                         : ((i64)(K - 1 - k0) * ldx + N - tt * RW) * (i64)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X + tt * RW + (i64)k0 * ldx), (short)0,
                                                                            (int)min(ext > 0 ? ext : (i64)0, (i64)0x7fffffff), BUF_WORD3);
        const bool rowok = t >= 0 && tt * RW + V * li < N;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = buf_ld_so<T, V, 2>(rs, (rowok && k0 + 4 * u + lq < k_hi) ? voff : OOR, u * cstep);
    };

    double acc[V][NCG];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int c = 0; c < NCG; ++c) acc[e][c] = 0.0;

    // position of the batch in flight: (round r, window q, batch j of the wave's part)
    int r = 0, q = 0, j = 0, wseq = 0;  // (wseq: windows done so far -- the buffer of the current one is wseq & 1)
    i64 tile = tile_of(0);
    fetch_b(0);
    put_b(0);
    if (NQ > 1 || rounds > 1) fetch_b(NQ > 1 ? 1 : 0);
    __syncthreads();
    if (rounds == 0) return;  // (uniform: no tile at all for this workgroup)

    auto step = [&](Pack<T, V> (&xa)[U], Pack<T, V> (&xb)[U]) -> bool {
        // the batch after this one
        int jn = j + 1, qn = q, rn = r;
        if (jn == nbw) {
            jn = 0;
            if (++qn == NQ) {
                qn = 0;
                ++rn;
            }
        }
        const bool more = rn < rounds;
        const i64 tn = rn == r ? tile : (more ? tile_of(rn) : -1);
        if (more) load_x(xb, tn, k_lo + qn * KC + sw * KS + jn * 4 * U);
        const double *brow = xb4w_bs + (size_t)(wseq & 1) * KC * ST + (size_t)(sw * KS + j * 4 * U + lq) * ST + lj;
        double bc[NCG], bn[NCG];
#pragma unroll
        for (int c = 0; c < NCG; ++c) bc[c] = brow[4 * c];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + 1 < U) {
#pragma unroll
                for (int c = 0; c < NCG; ++c) bn[c] = brow[(4 * (u + 1)) * ST + 4 * c];
            }
#pragma unroll
            for (int c = 0; c < NCG; ++c)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e][c] = __builtin_amdgcn_mfma_f64_4x4x4f64((double)xa[u].v[e], bc[c], acc[e][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NCG; ++c) bc[c] = bn[c];
        }
        if (jn == 0) {  // the window is done
            const bool last = (qn == 0);  // ... and with it the round
            if (last) {
                // the SW partial sums of a tile meet in LDS, column group by column group, in the order sw = 1, 2, ...
                // (both buffers are free behind the barrier: the next window is still in registers)
                __syncthreads();
                if (SW > 1) {
                    double *ex = xb4w_bs;  // [16 waves][V][64 lanes]
#pragma unroll
                    for (int c = 0; c < NCG; ++c) {
                        if (sw > 0) {
#pragma unroll
                            for (int e = 0; e < V; ++e) ex[((size_t)wvu * V + e) * WAVE + lane] = acc[e][c];
                        }
                        __syncthreads();
                        if (sw == 0) {
                            for (int s = 1; s < SW; ++s)
#pragma unroll
                                for (int e = 0; e < V; ++e) acc[e][c] += ex[((size_t)(s * TW + tw) * V + e) * WAVE + lane];
                        }
                        __syncthreads();
                    }
                }
                if (sw == 0 && tile >= 0) {
                    const i64 r0 = tile * RW + V * (4 * ((lane >> 2) & 3) + lq);
                    if (part) {
#pragma unroll
                        for (int c = 0; c < NCG; ++c)
#pragma unroll
                            for (int e = 0; e < V; ++e)
                                if (4 * c + lj < ncols && r0 + e < N) part[((i64)blockIdx.y * NC + 4 * c + lj) * ldp + r0 + e] = acc[e][c];
                    } else if ((tile + 1) * RW <= N) {
                        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tile * RW, (short)0, 0x7fffffff, BUF_WORD3);
#pragma unroll
                        for (int c = 0; c < NCG; ++c) {
                            Pack<T, V> o;
#pragma unroll
                            for (int e = 0; e < V; ++e) o.v[e] = (T)acc[e][c];
                            buf_st_so<T, V, 2>(ro, (4 * c + lj < ncols) ? soff : OOR, c * ostep, o);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < NCG; ++c)
#pragma unroll
                            for (int e = 0; e < V; ++e)
                                if (4 * c + lj < ncols && r0 + e < N) out[r0 + e + (i64)(4 * c + lj) * ldo] = (T)acc[e][c];
                    }
                }
#pragma unroll
                for (int e = 0; e < V; ++e)
#pragma unroll
                    for (int c = 0; c < NCG; ++c) acc[e][c] = 0.0;
            }
            if (more) {
                // the next window (of this round, or the first of the next): registers -> the other buffer -- free since the
                // barrier at the end of the window before this one -- and the window after it into the registers
                put_b((wseq & 1) ^ 1);
                __syncthreads();
                int q2 = qn + 1, more2 = 1;
                if (q2 == NQ) {
                    q2 = 0;
                    more2 = rn + 1 < rounds;
                }
                if (more2) fetch_b(q2);
            }
        }
        if (jn == 0) ++wseq;
        j = jn;
        q = qn;
        if (rn != r) tile = tn;
        r = rn;
        return more;
    };
    Pack<T, V> x0[U], x1[U];
    load_x(x0, tile, k_lo + sw * KS);
    while (true) {
        if (!step(x0, x1)) break;
        if (!step(x1, x0)) break;
    }
}

}  // namespace plsk
