// xb_mfma4.hpp -- out(N x ncols) = X * Bm for 4 < ncols <= 4 NCG columns on v_mfma_f64_4x4x4_4b_f64 (scores T = X R,
// src/pls.cpp:439-442; fitted values :449-451).
// Four 4 x 4 x 4 blocks per instruction: the column count is padded to 4, not to 16 -- 20 columns are five groups where the
// 16-wide MFMA pays for 32 -- at the rate of the vector FMA (measured, tune/mfma_f64_4x4x4_probe: 73.9 TFLOP/s against
// 69.4), with the operand of Bm in ONE register per lane instead of a broadcast LDS read per FMA.
// Lane layout (measured by the same probe): A lane = 16 k + 4 b + i, B lane = 16 k + 4 b + j, D lane = 16 i + 4 b + j
// (b = block, i = row of the block, j = column, k = step of the 4-deep product).  The four blocks take rows 4 b + i of a
// 16-row set and the SAME 4 k x 4 columns of Bm, so the A operand is a 16 x 4 access (lane = (row li, column lq): 16-byte
// loads, 4 column segments of 256 bytes per wave-load -- the tile pattern of the fused pass) and the V rows of a lane's pack
// feed V MFMAs.  All of Bm sits in LDS ([k][ST] doubles, rows beyond K zero) for the whole launch: no barrier in the loop.
// Persistent workgroups of 16 waves; a wave owns the 16 V rows of a tile and walks all K with the next batch of U loads in
// flight behind the MFMAs of this one (two register sets that swap roles: no copy, also across the tile boundary).
// Addressing: one buffer descriptor per (tile, batch) -- scalar -- plus ONE lane offset for the whole launch and the
// instruction's scalar offset per column step; a column beyond K -- and, in the partial last tile, a row beyond N -- is an
// out-of-range offset (returns 0).
// What the time is made of (tune/xb4_tune.hip, 1,048,576 x 512 fp64, 20 columns; profiles/r5/xb4_tune.txt):
//   * without its stores the kernel streams X at 0.655 ms whatever the column count (the MFMAs are free: 0.27 ms of pipe);
//   * the 168 MB of output -- 4 % of the bytes -- cost 0.06 ms in some processes and 0.19 ms in most (per process, not per
//     placement of the output: offsets into one arena change nothing).  16 waves that store whenever each finishes a tile
//     trickle their 256-byte pieces into the read stream; ONE barrier per round ahead of the stores (BAR) makes a burst
//     of 80 KB per workgroup and takes the 0.19 ms to 0.11 (0.855 -> 0.765 ms; nothing lost where stores were cheap);
//   * MAP: the 16 tiles of a workgroup's round lie a whole grid apart (tile = workgroup + grid x wave), the walk of the
//     fused pass -- workgroup b, on XCD b % 8, only ever touches the 256-byte pieces b mod 256 of every column.  Tiles 1,
//     2 ... 128 apart (a workgroup's round contiguous in each column) are 4-15 % slower;
//   * nt stores (STAUX 2) before plain, sc1 or sc0 sc1 ones by 0-4 %; one burst per TWO rounds (timed with wrong values)
//     another 1.5 %: not built; 512-thread workgroups, two row packs per lane, barriers inside the round, pacing
//     with s_sleep: nothing or worse.
#pragma once
#include "fused_kernels.hpp"  // buf_ld_so, BUF_WORD3

namespace plsk {

constexpr int XB4_WG = 1024;
// column steps per batch: 4 (2 where the accumulators of fp32 storage's 4-row packs leave no room) -- 8 fit the registers up to
// 20 fp64 columns and are 4 % slower with the barrier (0.757 against 0.730 ms; 2: 0.723, 1: 0.791)
__host__ __device__ constexpr int xb4_u(int v, int ncg) { return v * ncg > 20 ? 2 : 4; }
__host__ __device__ constexpr int xb4_kp(int K, int u) { return (K + (4 * u > 32 ? 4 * u : 32) - 1) / (4 * u > 32 ? 4 * u : 32) * (4 * u > 32 ? 4 * u : 32); }  // rows of Bm in LDS
__host__ __device__ constexpr int xb4_stride(int ncg) { return (8 * ncg) % 64 == 0 ? 4 * ncg + 4 : 4 * ncg; }  // (k-rows on disjoint banks)

// MAP, WGT, UU, AUXL, STAUX, BAR: the choices the header's measurements settled (tune/xb4_tune.hip instantiates the others)
template <typename T, int V, int NCG, int MAP = 0, int WGT = XB4_WG, int UU = 0, int AUXL = 2, int STAUX = 2, int BAR = 1>
__global__ __launch_bounds__(WGT) void xb_mfma4_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K, const double *__restrict__ Bm,
                                                          i64 ldb, int ncols, T *__restrict__ out, i64 ldo) {
    constexpr int NC = 4 * NCG, ST = xb4_stride(NCG), U = UU ? UU : xb4_u(V, NCG), RW = 16 * V;
    constexpr uint32_t OOR = 0xFFFFFFF0u;
    extern __shared__ __attribute__((aligned(16))) double xb4_bs[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4, lj = lane & 3;
    constexpr int KR = 4 * U > 32 ? 4 * U : 32;
    const int Kp = (K + KR - 1) / KR * KR;  // (xb4_kp)
    const i64 ntiles = (N + RW - 1) / RW;  // (the last one may be partial)
    for (int j = tid; j < Kp * NC; j += WGT) {  // consecutive threads: consecutive k of one column (coalesced)
        const int kk = j % Kp, m = j / Kp;
        xb4_bs[kk * ST + m] = (kk < K && m < ncols) ? Bm[kk + (i64)m * ldb] : 0.0;
    }
    __syncthreads();
    const i64 wstride = (i64)gridDim.x * (WGT / WAVE);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);  // (wave-uniform: scalar descriptors)
    // the tiles of a round (16 per workgroup): the waves of a workgroup take tiles MAP apart (MAP = 0: a whole grid apart)
    i64 tile = MAP == 0 ? (i64)wvu * gridDim.x + blockIdx.x
                        : (i64)(blockIdx.x / MAP) * ((WGT / WAVE) * MAP) + (i64)wvu * MAP + blockIdx.x % MAP;
    if (tile >= ntiles) return;
    const uint32_t voff = (uint32_t)((V * li + (i64)lq * ldx) * (i64)sizeof(T));
    const int cstep = (int)(4 * ldx * (i64)sizeof(T));  // bytes between the column steps of a batch
    // the stores: D lane = 16 i + 4 b + j holds rows V (4 b + i) + e, e < V, of column 4 c + j -- 16 contiguous bytes per lane
    const uint32_t soff = (uint32_t)((V * (4 * ((lane >> 2) & 3) + lq) + (i64)lj * ldo) * (i64)sizeof(T));
    const int ostep = (int)(4 * ldo * (i64)sizeof(T));  // bytes between the column groups
    auto load_x = [&](Pack<T, V> (&x)[U], i64 t, int k0) {
        // the descriptor ends with the matrix (element (N - 1, K - 1)): a pack of the partial last tile that straddles N reads the
        // padding between the columns (ldx is a multiple of the pack: the launcher's vec_ok) -- and, in the LAST column, nothing
        // beyond the caller's allocation (out of range: zeros).  Lanes whose rows lie beyond N load out-of-range offsets.
        const i64 ext = (t + 1) * RW <= N ? (i64)0x7fffffff  // (a full tile: no bound to compute)
                                         : ((i64)(K - 1 - k0) * ldx + N - t * RW) * (i64)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X + t * RW + (i64)k0 * ldx), (short)0,
                                                                            (int)min(ext > 0 ? ext : (i64)0, (i64)0x7fffffff), BUF_WORD3);
        const bool rowok = t * RW + V * li < N;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = buf_ld_so<T, V, AUXL>(rs, (rowok && k0 + 4 * u + lq < K) ? voff : OOR, u * cstep);
    };
    const double *bl = xb4_bs + lq * ST + lj;
    double acc[V][NCG];
#pragma unroll
    for (int e = 0; e < V; ++e)
#pragma unroll
        for (int c = 0; c < NCG; ++c) acc[e][c] = 0.0;
    int k0 = 0;
    i64 round = 0;
    const i64 rfull = ntiles / wstride;  // rounds in which every wave of the workgroup has a tile
    // one batch: the loads of the NEXT batch (of this tile, or the first of the wave's next tile) go out into xb, then the
    // MFMAs of this one out of xa; the two register sets swap roles from step to step (no copy, no wait for the loads in flight)
    auto step = [&](Pack<T, V> (&xa)[U], Pack<T, V> (&xb)[U]) -> bool {
        int kn = k0 + 4 * U;
        i64 tn = tile;
        if (kn >= K) {
            kn = 0;
            tn = tile + wstride;
        }
        if (tn < ntiles) load_x(xb, tn, kn);
        const double *brow = bl + k0 * ST;
        double bc[NCG], bn[NCG];
#pragma unroll
        for (int c = 0; c < NCG; ++c) bc[c] = brow[4 * c];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // the operands of step u + 1 are read from LDS ahead of the MFMAs of step u
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
        if (kn == 0) {
            if constexpr (BAR != 0) {  // the 16 waves store their tiles together: a burst of 80 KB, not a trickle into the read stream
                if (round < rfull) __syncthreads();
                ++round;
            }
            // D: lane holds (row 4 b + i, column j) with i = lane / 16, b = (lane / 4) % 4: rows row0 + V (4 b + i) + e, e < V,
            // are contiguous -- one 16-byte store per lane and column group
            const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tile * RW, (short)0, 0x7fffffff, BUF_WORD3);
            const bool tfull = (tile + 1) * RW <= N;  // (wave-uniform)
#pragma unroll
            for (int c = 0; c < NCG; ++c) {
                const bool ok = 4 * c + lj < ncols;
                if (tfull) {
                    Pack<T, V> o;
#pragma unroll
                    for (int e = 0; e < V; ++e) o.v[e] = (T)acc[e][c];
                    buf_st_so<T, V, STAUX>(ro, ok ? soff : OOR, c * ostep, o);
                } else {  // the partial last tile, element by element
                    const i64 r0 = tile * RW + V * (4 * ((lane >> 2) & 3) + lq);
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        if (ok && r0 + e < N) out[r0 + e + (i64)(4 * c + lj) * ldo] = (T)acc[e][c];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e][c] = 0.0;
            }
        }
        k0 = kn;
        tile = tn;
        return tn < ntiles;
    };
    Pack<T, V> x0[U], x1[U];
    load_x(x0, tile, 0);
    while (true) {
        if (!step(x0, x1)) break;
        if (!step(x1, x0)) break;
    }
}

}  // namespace plsk
