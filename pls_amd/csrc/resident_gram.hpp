// resident_gram.hpp -- the whole fit of a mid-size problem with K <= 128 columns in ONE launch and THREE (block form: TWO)
// grid-wide hand-offs, whatever the number of components (round 5).
//
// resident_fit_kernel (resident_kernels.hpp) keeps X in registers and exchanges [X^T t, t^T t] once per COMPONENT: 10-13 us
// each, of which the exchange is 5-6 (a store, an arrival, a poll and a gather: four trips over the fabric on a chip that
// idles at low clocks).  With few columns the component loop needs no pass over X at all (src/pls.cpp:398, :422-425:
// t^T t = r^T XX r, X^T t = XX r) -- the GRAM plan's algebra -- and XX = X^T X of 128 columns is 128 KB: it fits in the LDS
// of ONE workgroup.  So:
//   1. every workgroup stages its rows of X (and Y) in LDS and forms its part of XX and XY on v_mfma_f64_16x16x4_f64
//      (a wave per pair of 16-column blocks), stores it sc1;                                          -- hand-off 1 --
//   2. every workgroup sums a SLICE of the K^2 + K values over the G parts, in workgroup order         -- hand-off 2 --
//      (G (K^2 + K) <= 32,768: workgroup 0 sums them itself and the hand-off is saved);
//      BLOCK form of 1 and 2 (48 columns and more, 16-byte-aligned columns; resident_gram_splits): workgroup (block pair, row
//      split) forms ONE 16 x 16 block over its rows from 4-row packs straight out of global memory (no staging, no barrier in
//      the loop), stores 256 values, and workgroup 0 adds the RS splits of the lower block triangle itself with 16-byte loads:
//      TWO hand-offs in all.  5,000 x 128, A = 10: XX in workgroup 0's LDS at 20.5 us instead of 33.5, the fit 63.5 us
//      instead of 76 (profiles/r5/resident_gram_stamps.txt, resident_gram_block_ab.txt);
//   3. workgroup 0 alone, XX and every p_j, r_j in LDS: for every component XX r (a wave per output), r^T XX r, q, p, the
//      XY deflation, w, the r recurrence (the single-launch kernels' update: thread k owns column k); B = R Q^T; R goes
//      out sc1;                                                                                       -- hand-off 3 --
//   4. every workgroup forms the scores of its rows, T = X R (src/pls.cpp:439-442).
// The hand-offs are resident_kernels.hpp's (first row of MI355X_MICROARCH.md's "Valid forms": sc1 stores, vmcnt(0), barrier,
// one lane's agent-scope arrival; sc1 loads behind the poll and a barrier; one workgroup per CU; bounded waits that end in a
// status word, never a hang).  1-8 responses (more than one: the direction by the one-wave eigen solver, still ONE product XX r per
// component), fp64 or fp32 storage.
#pragma once
#include "resident_kernels.hpp"

namespace plsk {

#ifdef PLS_HIP_TESTING
// testing/libpls_hip.so only: wall-clock stamps of the phases (tools/resident_gram_stamps.py)
#define RG_STAMP(slot)                                                                                      \
    do {                                                                                                    \
        if (g_pass_stamps && threadIdx.x == 0) g_pass_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define RG_STAMP(slot) do { } while (0)
#endif

constexpr int RG_KMAX = 128;
constexpr int RG_ACH = 8;        // score columns per pass of phase 4
constexpr int RG_SMALL = 512 + 4 * RG_KMAX + 16;  // partial sums of XX r (or of a slice), r, the p_j^T w, w, q: doubles of dynamic LDS behind the big block
constexpr int RG_LDS_DOUBLES = 20384;          // what one workgroup may ask for (160 KB less the static few and a margin)
constexpr i64 RG_DIRECT = 32768;               // G (K^2 + K) at or below this: workgroup 0 sums the parts itself (no hand-off 2)

struct ResidentGram {
    ResidentSync sy;           // status, limit (the counters, part and LP are the per-component kernels')
    unsigned *flags = nullptr; // [256]: workgroup g's arrival word -- epoch + (hand-offs passed): grows from launch to launch, never reset
    unsigned epoch = 0;        // the value before this launch's first hand-off
    double *part = nullptr;    // [G][LP]: XX (K x K, column-major) then XY (K) of every workgroup
    double *gred = nullptr;    // [LP]: their sums
    double *rshare = nullptr;  // [K A]: R for phase 4
    i64 LP = 0;
    int rows_per = 0;          // rows per workgroup (a multiple of 4)
    int big = 0;               // doubles of the big LDS block: XX, or the staged rows, or the score partials
    int rs = 0;                // block form (resident_gram_splits): row splits per block pair, 0 = the row form
    int brows = 0;             //   rows per split (a multiple of 16)
};

// doubles of the big LDS block: XX and XY (a workgroup's part on its way out); 16 x RG_ACH x 64 score partials; at least 32 staged rows
inline int resident_gram_big(int K) {
    const int kp = (K + 15) / 16 * 16 + 16;
    return std::max(std::max(K * K + K, 16 * RG_ACH * 64), 32 * kp);
}
// doubles of dynamic LDS behind the big block and the small vectors: P and R of every component; more responses: XY and Q as well
inline int resident_gram_extra(int K, int M, int A) {
    const int mm = M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8;
    return 2 * K * A + (M > 1 ? mm * K + M * A : 0);
}
// 0: not covered; else the number of workgroups.  1-8 responses.
inline int resident_gram_grid(i64 N, int K, int M, int A, i64 ldx, size_t es, int num_cu) {
    if (M < 1 || M > 8 || K < 1 || K > RG_KMAX || A < 1 || 2 * A > K || N < 256) return 0;  // (A close to K: the last directions are noise, and XX squares
                                                                                      // the condition number -- P^T R = I to 1e-8 only up to ~K/2; the TYPE1 kernels take those)
    if (resident_gram_big(K) + RG_SMALL + resident_gram_extra(K, M, A) > RG_LDS_DOUBLES - (M > 1 ? 256 : 0)) return 0;   // XX + P and R of every component in LDS
                                                                                                                          // (more responses: 2 KB of static LDS for the eigen solver)
    if ((i64)N * K * (i64)es > ((i64)64 << 20) || (i64)K * ldx * (i64)es >= (1ll << 31)) return 0;
    const i64 L = (i64)K * K + (i64)K * M;
    // phase 1 is matrix-core work on G CUs at the clocks of a nearly idle chip (~1.1 GHz: 5,000 x 128 on 60 workgroups 19 us):
    // as many workgroups as leave 32 rows each (64 below 64 columns), up to ~20 MB of parts
    i64 G = std::min<i64>(std::min<i64>(num_cu, RESIDENT_MAX_WG), N / (K >= 64 ? 32 : 64));  // (few columns: the hand-offs cost more than the product)
    G = std::min<i64>(G, std::max<i64>(8, (i64)2500000 / L));
    return (int)std::max<i64>(G, 2);
}

// The BLOCK form of phase 1 (0: not taken; else the row splits RS): workgroup (pair, split) forms ONE 16 x 16 block of XX (or of XY)
// over its split of the rows, operands straight from global memory in 4-row packs (whole 128-byte lines, no staging, no barrier in
// the loop) and stores 256 values -- not K^2 + K M: nothing to drain before the hand-off, and workgroup 0 adds the RS splits of
// the lower block triangle itself: no slices, no second hand-off.  X is read nb + 1 times, from L2.
inline int resident_gram_splits(i64 N, int K, int M, int num_cu, bool vec) {
    if (!vec || K < 48 || M > 2) return 0;  // (more responses: the eigen solver's loop came out 25 % slower in the kernel that holds both forms, and the
                                            //  block form instantiated for it gained nothing at 8,000 x 100 with 8: 78.7 against 78.5 us)
    const int nb = (K + 15) / 16, nxx = nb * (nb + 1) / 2, npairs = nxx + nb;
    i64 rs = std::min<i64>(std::min<i64>(num_cu, RESIDENT_MAX_WG) / npairs, N / 256);
    rs = std::min<i64>(rs, 49152 / ((i64)nxx * 256));  // what workgroup 0 reads: RS x the lower triangle
    if (rs > 4) rs &= ~(i64)3;                          //   (four splits per round of its loads)
    if ((i64)N * (nb + 1) * K > (i64)32 << 20) return 0;  // (elements through L2: beyond this the row form's single read -- 50,000 x 100, 8 responses: 174 against 169 us)
    return rs >= 2 ? (int)rs : 0;
}

// A grid-wide hand-off without a shared counter: every wave's sc1 stores are complete (vmcnt(0)) before the barrier behind which
// one lane publishes the workgroup's OWN arrival word (an agent-scope store); wave 0 polls all G words in one 16-byte load per
// lane until none is behind.  (One counter for all: 139 agent-scope additions to one address took 10 us -- profiles/r5/
// resident_gram_stamps.txt.)  The words only ever grow, from launch to launch (rg.epoch): a launch that ended in a time-out
// leaves nothing to clean up.  Returns false when the wait ran out (status raised).
__device__ __forceinline__ bool resident_gram_barrier(const ResidentGram &rg, unsigned phase, int *flag, bool wait);
__device__ __forceinline__ double ld_sc1(const double *base, i64 idx) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), (short)0, 0x7fffffff, BUF_WORD3);
    const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)(idx * 8), 0, AUX_SC1);
    double v;
    __builtin_memcpy(&v, &raw, 8);
    return v;
}

__device__ __forceinline__ bool resident_gram_barrier(const ResidentGram &rg, unsigned phase, int *flag, bool wait) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned target = rg.epoch + phase + 1u;
    if (threadIdx.x == 0) __hip_atomic_store(rg.flags + blockIdx.x, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < WAVE) {
        const int G = gridDim.x, lane = threadIdx.x;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(rg.flags, (short)0, RESIDENT_MAX_WG * 4, BUF_WORD3);
        const long long t0 = wall_clock64();
        int ok = wait ? 1 : 0;
        while (wait) {
            const u32x4 f = __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)lane * 16u, 0, AUX_SC1);
            bool mine = true;  // (signed distance: the words wrap with the epoch)
#pragma unroll
            for (int q = 0; q < 4; ++q) mine = mine && (4 * lane + q >= G || (int)(f[q] - target) >= 0);
            if (__all(mine)) break;
            if (wall_clock64() - t0 > rg.sy.limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    if (!ok && threadIdx.x == 0) __hip_atomic_store(rg.sy.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return ok;
}

// X: N x K (ld ldx), Y: N x M (ld ldy), M <= MM; W, P, R: K x A; Q: M x A; Tm: N x A (ld ldt); B: K x M or null.
// grid = G workgroups of 1024 threads; dynamic LDS: rg.big + RG_SMALL + resident_gram_extra(K, M, A) doubles.
template <typename T, int MM, bool BLK = false>  // BLK: the block form of phases 1-2 (rg.rs > 0)
__global__ __launch_bounds__(UPD_THREADS) void resident_gram_fit_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 ldy, i64 N,
                                                                        int K, int M, int A, int power_iters, double *__restrict__ W,
                                                                        double *__restrict__ P, double *__restrict__ Q, double *__restrict__ R,
                                                                        T *__restrict__ Tm, i64 ldt, double *__restrict__ B,
                                                                        const ResidentGram rg) {
    static_assert(MM == 1 || MM * MM <= WAVE, "one wave solves the eigenproblem");
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) double rgd[];
    __shared__ double sred[2 * UPD_WAVES + 8], Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM], qa[MM];
    __shared__ int flag;
    double *big = rgd;                          // [rg.big]
    double *sp = rgd + rg.big;                  // [512]: partial sums (XX r by column group; a slice by workgroup subset)
    double *rl = sp + 512;                      // [K]: r_a
    double *cs = rl + RG_KMAX;                  // [A]: p_j^T w
    double *wl = cs + RG_KMAX;                  // [K]: w
    double *ql = wl + RG_KMAX;                  // [M A]: q (M = 1: room for A <= 136; more responses: behind R)
    double *Pl = ql + RG_KMAX + 8;              // [K A]
    double *Rq = Pl + (i64)K * A;               // [K A]
    double *xyl = Rq + (i64)K * A;              // MM > 1: [MM][K] XY, then [M A] Q
    if (MM > 1) ql = xyl + (i64)MM * K;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int G = gridDim.x, g = blockIdx.x;
    const i64 L = (i64)K * K + (i64)K * M;
    const bool direct = (i64)G * L <= RG_DIRECT;  // (the same in every workgroup)
    bool ok = true;
    unsigned phase = 0;

    RG_STAMP(0);
    // ---- 1. this workgroup's rows: XX and XY parts on the matrix cores, operands from LDS ----
    const i64 r0 = (i64)g * rg.rows_per, r1 = min(N, r0 + (i64)rg.rows_per);
    const int nb = (K + 15) / 16;                 // 16-column blocks of X; block index nb: the responses
    const int KP = nb * 16 + 16;                  // row stride of the staged rows: X blocks, then 16 response slots
    // rows staged at a time: a power of two (>= 32), so that the staging's (row, column) of an element are a mask and a shift
    const int clg = 31 - __builtin_clz((unsigned)(rg.big / KP)), chunk = 1 << clg;
    const int npairs = nb * (nb + 1) / 2 + nb;    // (bi <= bj) and (bi, Y)
    const int li = lane & 15, lk = lane >> 4;
    double *mine = rg.part + (i64)g * rg.LP;
    // few pairs (few columns): the 16 waves are (pair, row split) -- wave = split * npairs + pair walks every RSPL-th 4-row step
    const int RSPL = npairs < UPD_WAVES ? UPD_WAVES / npairs : 1;
    const int task = RSPL > 1 ? wv % npairs : wv, rsp = RSPL > 1 ? wv / npairs : 0;
    const bool tact = rsp < RSPL;  // (RSPL > 1: the waves beyond npairs * RSPL idle)
    // a wave owns the pairs wv, wv + 16, ...: at most 3 for K = 128 (44 pairs)
    constexpr int MAXP = 3;
    f64x4 acc[MAXP];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) acc[p] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto pair_of = [&](int idx, int &bi, int &bj) {  // row-major over bj >= bi, then the (bi, Y) pairs
        int rest = idx;
        for (bi = 0; bi < nb; ++bi) {
            const int len = nb - bi;
            if (rest < len) {
                bj = bi + rest;
                return;
            }
            rest -= len;
        }
        bi = rest;
        bj = nb;
    };
    const int nxx = nb * (nb + 1) / 2;            // the (bi <= bj) pairs
    if constexpr (BLK) {
        // ---- 1 (block form). workgroup (pair, split): ONE block over the split's rows.  Lane (li, lk) of a wave holds the 4-row pack
        //      rows 16 t + 4 lk .. + 3 of column li of both blocks (the k index of the MFMA is only summed over: step e takes row e of
        //      every lane's pack) -- whole 128-byte lines per column, no staging.  The 16 waves' sums meet in LDS in wave order. ----
        constexpr int PV = 16 / (int)sizeof(T), NPK = 4 / PV;  // 16-byte packs per lane and batch
        const int pr = g % npairs, sp2 = g / npairs;
        int bi, bj;
        pair_of(pr, bi, bj);
        const i64 b0 = (i64)sp2 * rg.brows, b1 = min(N, b0 + (i64)rg.brows);
        const T *ca = bi * 16 + li < K ? X + (i64)(bi * 16 + li) * ldx : nullptr;
        const T *cb = bj < nb ? (bj * 16 + li < K ? X + (i64)(bj * 16 + li) * ldx : nullptr) : (li < M ? Y + (i64)li * ldy : nullptr);
        const bool same = bi == bj;
        auto ld4 = [&](const T *col, i64 row, double (&o)[4]) {
            if (col != nullptr && row + 4 <= b1) {
#pragma unroll
                for (int h = 0; h < NPK; ++h) {
                    const Pack<T, PV> pk = ld_pack<T, PV>(col + row + h * PV);
#pragma unroll
                    for (int e = 0; e < PV; ++e) o[h * PV + e] = (double)pk.v[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (col != nullptr && row + e < b1) ? (double)col[row + e] : 0.0;
            }
        };
        f64x4 ac = f64x4{0.0, 0.0, 0.0, 0.0};
        i64 row = b0 + (i64)wv * 16 + 4 * lk;
        if (sp2 < rg.rs) {
            for (; row - 4 * lk + 16 * UPD_WAVES < b1; row += 2 * 16 * UPD_WAVES) {  // two batches in flight
                double a0[4], a1[4], c0[4], c1[4];
                ld4(ca, row, a0);
                ld4(ca, row + 16 * UPD_WAVES, a1);
                if (!same) {
                    ld4(cb, row, c0);
                    ld4(cb, row + 16 * UPD_WAVES, c1);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) ac = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], same ? a0[e] : c0[e], ac, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) ac = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], same ? a1[e] : c1[e], ac, 0, 0, 0);
            }
            if (row - 4 * lk < b1) {
                double a0[4], c0[4];
                ld4(ca, row, a0);
                if (!same) ld4(cb, row, c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) ac = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], same ? a0[e] : c0[e], ac, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) big[(wv * 4 + q) * WAVE + lane] = ac[q];
        __syncthreads();
        if (tid < 256 && sp2 < rg.rs) {
            double t = 0.0;
            for (int w = 0; w < UPD_WAVES; ++w) t += big[(w * 4 + (tid >> 6)) * WAVE + (tid & 63)];
            st_agent(rg.part + ((i64)sp2 * npairs + pr) * 256 + tid, t);  // element q 64 + lane of the block: row lk + 4 q, column li
        }
    } else {
        // (loading the NEXT chunk into registers behind this chunk's MFMAs was tried: sixteen guarded loads per thread whatever the
        // chunk cost 5 us more than the waits they hid)
        for (i64 c0 = r0; c0 < r1; c0 += chunk) {
            const int rc = (int)min((i64)chunk, r1 - c0), rc4 = (rc + 3) & ~3;
            __syncthreads();  // the previous chunk has been read
            for (int base = 0; base < chunk * KP; base += 8 * UPD_THREADS) {  // consecutive threads: consecutive rows of one column; eight loads in flight
                double v[8];
    #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * UPD_THREADS + tid, row = idx & (chunk - 1), col = idx >> clg;
                    v[u] = 0.0;
                    if (col < KP && row < rc) {
                        if (col < K) v[u] = (double)X[c0 + row + (i64)col * ldx];
                        else if (col >= nb * 16 && col - nb * 16 < M) v[u] = (double)Y[c0 + row + (i64)(col - nb * 16) * ldy];
                    }
                }
    #pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * UPD_THREADS + tid, row = idx & (chunk - 1), col = idx >> clg;
                    if (col < KP && row < rc4) big[row * KP + col] = v[u];
                }
            }
            __syncthreads();
            const double *ap[MAXP], *bp[MAXP];
    #pragma unroll
            for (int p = 0; p < MAXP; ++p) {
                int bi = 0, bj = 0;
                if (task + p * UPD_WAVES < npairs) pair_of(task + p * UPD_WAVES, bi, bj);
                ap[p] = big + lk * KP + bi * 16 + li;
                bp[p] = big + lk * KP + bj * 16 + li;
            }
            if (tact)
                for (int r = 4 * rsp; r < rc4; r += 4 * RSPL) {  // the wave's (up to) three accumulation chains side by side
    #pragma unroll
                    for (int p = 0; p < MAXP; ++p)
                        if (task + p * UPD_WAVES < npairs) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[p][r * KP], bp[p][r * KP], acc[p], 0, 0, 0);
                }
        }
        if (RSPL > 1) {  // the row splits of a pair meet in LDS, in split order (the staged rows are done with)
            __syncthreads();
            if (tact && rsp > 0) {
    #pragma unroll
                for (int q = 0; q < 4; ++q) big[(wv * 4 + q) * WAVE + lane] = acc[0][q];
            }
            __syncthreads();
            if (tact && rsp == 0)
                for (int s2 = 1; s2 < RSPL; ++s2)
    #pragma unroll
                    for (int q = 0; q < 4; ++q) acc[0][q] += big[((s2 * npairs + task) * 4 + q) * WAVE + lane];
        }
        // D: lane holds rows (lane >> 4) + 4 q, column lane & 15 of the 16 x 16 block (assembling the part in LDS for consecutive stores
        // was tried: two barriers and a store loop cost more than the scattered stores, 19 -> 25 us at K = 128)
    #pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            const int idx = task + p * UPD_WAVES;
            if (idx < npairs && rsp == 0) {
                int bi, bj;
                pair_of(idx, bi, bj);
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = bi * 16 + lk + 4 * q, j = (bj < nb ? bj * 16 : 0) + li;
                    const double v = acc[p][q];
                    if (bj < nb) {
                        if (i < K && j < K) {
                            st_agent(mine + i + (i64)j * K, v);
                            if (bi != bj) st_agent(mine + j + (i64)i * K, v);
                        }
                    } else if (i < K && li < M) {
                        st_agent(mine + (i64)K * K + i + (i64)li * K, v);
                    }
                }
            }
        }
    }
    RG_STAMP(1);
    ok = resident_gram_barrier(rg, phase++, &flag, ok);
    RG_STAMP(2);

    // ---- 2. a slice of the L values summed over the G parts, in workgroup order (few values: workgroup 0 does it itself) ----
    auto sum_parts = [&](i64 j) -> double {  // sixteen loads in flight, added in workgroup order
        double s = 0.0;
        int h = 0;
        for (; h + 16 <= G; h += 16) {
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = ld_sc1(rg.part + (i64)(h + u) * rg.LP, j);
#pragma unroll
            for (int u = 0; u < 16; ++u) s += x[u];
        }
        for (; h < G; ++h) s += ld_sc1(rg.part + (i64)h * rg.LP, j);
        return s;
    };
    auto sum_blk = [&](int p, int e) -> double {  // block form: element e of pair p over the row splits, in split order
        double t = 0.0;
        for (int h = 0; h < rg.rs; h += 8) {  // eight loads in flight
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = h + u < rg.rs ? ld_sc1(rg.part, ((i64)(h + u) * npairs + p) * 256 + e) : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) t += x[u];
        }
        return t;
    };
    auto xy_blk = [&](int m, int kk) -> double {  // XY(kk, m): row kk % 16 of the block (kk / 16, Y), column m
        return sum_blk(nxx + (kk >> 4), ((kk & 15) >> 2) * 64 + (kk & 3) * 16 + m);
    };
    if (!BLK && !direct) {
        // the slice's S values: NS = 512 / LW workgroup subsets per value (LW = S rounded up to whole waves, at most 512 at a time):
        // thread (e, hs) adds the parts hs, hs + NS, ... in that order, then the NS sums meet in order -- fixed, whatever the run
        const i64 S = (L + G - 1) / G, j0 = (i64)g * S, j1 = min(L, j0 + S);
        for (i64 jb = j0; jb < j1; jb += 512) {
            const int cnt = (int)min((i64)512, j1 - jb), LW = (cnt + WAVE - 1) / WAVE * WAVE, NS = 512 / LW;
            const int e = tid % LW, hs = tid / LW;
            double s = 0.0;
            if (hs < NS && e < cnt) {
                int h = hs;
                for (; h + 7 * NS < G; h += 8 * NS) {  // eight loads in flight, added in order
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = ld_sc1(rg.part + (i64)(h + u * NS) * rg.LP, jb + e);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += x[u];
                }
                for (; h < G; h += NS) s += ld_sc1(rg.part + (i64)h * rg.LP, jb + e);
            }
            __syncthreads();  // (the previous chunk's sums have been read)
            if (hs < NS) sp[hs * LW + e] = s;
            __syncthreads();
            if (tid < cnt) {
                double t = 0.0;
                for (int q = 0; q < NS; ++q) t += sp[q * LW + tid];
                st_agent(rg.gred + jb + tid, ok ? t : __builtin_nan(""));
            }
        }
        ok = resident_gram_barrier(rg, phase++, &flag, ok);
    }
    RG_STAMP(3);

    // ---- 3. workgroup 0: the components, everything K-sized in LDS (thread k owns column k) ----
    if (g == 0) {
        const int k = tid;
        const bool kok = k < K;
        const int KW = K <= WAVE ? WAVE : 2 * WAVE, NJG = 512 / KW, JL = (K + NJG - 1) / NJG, kq = tid % KW, jg = tid / KW;
        if constexpr (BLK) {  // the lower block triangle, mirrored: four element pairs (16-byte loads) x four splits in flight per thread
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(rg.part, (short)0, 0x7fffffff, BUF_WORD3);
            for (int base = 0; base < nxx * 128; base += 4 * UPD_THREADS) {
                double v[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u][0] = v[u][1] = 0.0;
                for (int h = 0; h < rg.rs; h += 4) {
                    u32x4 x[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i2 = base + u * UPD_THREADS + tid;  // pair i2: elements 2 (i2 % 128), + 1 of block pair i2 / 128
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            x[u][q] = (i2 < nxx * 128 && h + q < rg.rs)
                                          ? __builtin_amdgcn_raw_buffer_load_b128(rp, (uint32_t)((((h + q) * npairs + (i2 >> 7)) * 256 + 2 * (i2 & 127)) * 8), 0, AUX_SC1)
                                          : u32x4{0u, 0u, 0u, 0u};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            double d[2];
                            __builtin_memcpy(d, &x[u][q], 16);
                            v[u][0] += d[0];
                            v[u][1] += d[1];
                        }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i2 = base + u * UPD_THREADS + tid;
                    if (i2 < nxx * 128) {
                        int bi, bj;
                        pair_of(i2 >> 7, bi, bj);
#pragma unroll
                        for (int z = 0; z < 2; ++z) {
                            const int e = 2 * (i2 & 127) + z, i = bi * 16 + ((e >> 4) & 3) + 4 * (e >> 6), j = bj * 16 + (e & 15);
                            if (i < K && j < K) {
                                big[i + j * K] = ok ? v[u][z] : __builtin_nan("");
                                if (bi != bj) big[j + i * K] = ok ? v[u][z] : __builtin_nan("");
                            }
                        }
                    }
                }
            }
        } else if (direct) {
            for (i64 j = tid; j < (i64)K * K; j += UPD_THREADS) big[j] = ok ? sum_parts(j) : __builtin_nan("");
        } else {
            for (int j0 = 0; j0 < K * K; j0 += 16 * UPD_THREADS) {  // sixteen loads in flight per lane
                double x[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int j = j0 + u * UPD_THREADS + tid;
                    x[u] = j < K * K ? ld_sc1(rg.gred, j) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int j = j0 + u * UPD_THREADS + tid;
                    if (j < K * K) big[j] = ok ? x[u] : __builtin_nan("");
                }
            }
        }
        if constexpr (MM == 1) {
            double xyk = kok ? (BLK ? xy_blk(0, k) : direct ? sum_parts((i64)K * K + k) : ld_sc1(rg.gred, (i64)K * K + k)) : 0.0;  // XY = X^T Y (:396)
            auto bsum = [&](double v) -> double {  // block sum on lds_barrier (every thread calls it)
                v = wave_sum(v);
                lds_barrier();
                if (lane == 0) sred[wv] = v;
                lds_barrier();
                double t = 0.0;
    #pragma unroll
                for (int w = 0; w < UPD_WAVES; ++w) t += sred[w];
                return t;
            };
            RG_STAMP(4);
            {  // w_0 = XY / |XY| (:404, :411), r_0 = w_0
                const double w = xyk / sqrt(bsum(xyk * xyk));
                if (kok) {
                    W[k] = w;
                    R[k] = w;
                    Rq[k] = w;
                    rl[k] = w;
                }
            }
            // Four barriers per component: what would be a reduction of its own rides on a barrier that is there anyway --
            // r^T XX r and r^T XY are summed by waves next to the partial sums of XX r; |XY| next to the p_j^T XY (the norm then
            // divides both w and the p_j^T w).
            for (int a = 0; a < A; ++a) {
                lds_barrier();  // r_a (rl) complete; XX in LDS
                // XX r (:424): thread (kq, jg) adds XX[kq][j] r[j] over the jg-th group of the columns j (row kq of column j: consecutive
                // lanes, consecutive addresses; r[j] a broadcast); the NJG partial sums of an output meet in order behind the barrier
                double s = 0.0;
                if (jg < NJG && kq < K) {
                    double s0 = 0.0, s1 = 0.0;
                    const int ja = jg * JL, jb2 = min(K, ja + JL);
                    int j = ja;
                    for (; j + 1 < jb2; j += 2) {
                        s0 = fma(big[(i64)j * K + kq], rl[j], s0);
                        s1 = fma(big[(i64)(j + 1) * K + kq], rl[j + 1], s1);
                    }
                    if (j < jb2) s0 = fma(big[(i64)j * K + kq], rl[j], s0);
                    s = s0 + s1;
                    sp[jg * KW + kq] = s;
                }
                {  // tt = r^T XX r (:425) = the sum of r[kq] * (partial sum) over every (kq, jg); r^T XY from the threads that own a column
                    const double v0 = wave_sum((jg < NJG && kq < K) ? rl[kq] * s : 0.0), v1 = wave_sum(kok ? rl[k] * xyk : 0.0);
                    if (lane == 0) {
                        sred[wv] = v0;
                        sred[UPD_WAVES + wv] = v1;
                    }
                }
                lds_barrier();
                double pr = 0.0;
                if (kok)
                    for (int q = 0; q < NJG; ++q) pr += sp[q * KW + k];
                double tt = 0.0, rxy = 0.0;
    #pragma unroll
                for (int w = 0; w < UPD_WAVES; ++w) {
                    tt += sred[w];
                    rxy += sred[UPD_WAVES + w];
                }
                const double p = pr / tt, q = rxy / tt;  // (:427, :428)
                if (kok) {
                    P[k + (i64)a * K] = p;
                    Pl[k + (i64)a * K] = p;
                }
                if (tid == 0) {
                    Q[a] = q;
                    ql[a] = q;
                }
                xyk -= (p * q) * tt;  // XY -= (p q^T) tt (:429)
                const int n = a + 1;
                if (n >= A) break;
                if (kok) wl[k] = xyk;  // the deflated XY, not yet normalised
                lds_barrier();
                for (int j = wv; j <= n; j += UPD_WAVES) {  // p_j^T XY for j < n (a wave each), |XY|^2 (one more wave)
                    double c = 0.0;
                    if (j < n) {
                        for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
                    } else {
                        for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
                    }
                    c = wave_sum(c);
                    if (lane == 0) (j < n ? cs[j] : sred[2 * UPD_WAVES]) = c;
                }
                lds_barrier();
                const double inv = 1.0 / sqrt(sred[2 * UPD_WAVES]);
                const double w = xyk * inv;  // w = XY / |XY| (:404, :411)
                double r = w;
                for (int j = 0; j < n; ++j) r -= (cs[j] * inv) * Rq[(kok ? k : 0) + (i64)j * K];  // c_j = p_j^T w; the reference's order (:412-416)
                if (kok) {
                    W[k + (i64)n * K] = w;
                    R[k + (i64)n * K] = r;
                    Rq[k + (i64)n * K] = r;
                    rl[k] = r;
                }
            }
            lds_barrier();
            if (B && kok) {  // B = R Q^T (:444-451)
                double b = 0.0;
                for (int a = 0; a < A; ++a) b = fma(Rq[k + (i64)a * K], ql[a], b);
                B[k] = b;
            }
        } else {
            // ---- 2..8 responses: the direction by the one-wave eigen solver (src/pls.cpp:403-411), ONE product XX r per component ----
            for (int m = 0; m < MM; ++m)
                if (kok) xyl[m * K + k] = m < M ? (BLK ? xy_blk(m, k) : direct ? sum_parts((i64)K * K + (i64)m * K + k) : ld_sc1(rg.gred, (i64)K * K + (i64)m * K + k)) : 0.0;
            RG_STAMP(4);
            for (int a = 0; a < A; ++a) {
                lds_barrier();  // XY complete (and XX in LDS)
                for (int pr = wv; pr < MM * (MM + 1) / 2; pr += UPD_WAVES) {  // G = XY^T XY, one wave per pair (i <= j)
                    int gi = 0, rem = pr;
                    while (rem >= MM - gi) { rem -= MM - gi; ++gi; }
                    const int gj = gi + rem;
                    double gsum = 0.0;
                    for (int kk = lane; kk < K; kk += WAVE) gsum = fma(xyl[gi * K + kk], xyl[gj * K + kk], gsum);
                    gsum = wave_sum(gsum);
                    if (lane == 0) { Gs[gi + gj * MM] = gsum; Gs[gj + gi * MM] = gsum; }
                }
                lds_barrier();
                if (wv == 0) dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
                lds_barrier();
                double wk = 0.0;
#pragma unroll
                for (int m = 0; m < MM; ++m) wk = fma(kok ? xyl[m * K + k] : 0.0, qs[m], wk);  // w = XY q (:408), not yet normalised
                if (kok) wl[k] = wk;
                lds_barrier();
                for (int j = wv; j <= a; j += UPD_WAVES) {  // p_j^T (XY q) for j < a (a wave each), |XY q|^2 (one more wave)
                    double c = 0.0;
                    if (j < a) {
                        for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
                    } else {
                        for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
                    }
                    c = wave_sum(c);
                    if (lane == 0) (j < a ? cs[j] : sred[2 * UPD_WAVES]) = c;
                }
                lds_barrier();
                const double inv = 1.0 / sqrt(sred[2 * UPD_WAVES]);
                wk *= inv;  // (:411)
                double r = wk;
                for (int j = 0; j < a; ++j) r -= (cs[j] * inv) * Rq[(kok ? k : 0) + (i64)j * K];  // c_j = p_j^T w; the reference's order (:412-416)
                if (kok) {
                    W[k + (i64)a * K] = wk;
                    R[k + (i64)a * K] = r;
                    Rq[k + (i64)a * K] = r;
                    rl[k] = r;
                }
                lds_barrier();  // r_a complete
                double s = 0.0;  // XX r (:424) by column groups, r^T XX r by waves beside it (as for one response)
                if (jg < NJG && kq < K) {
                    double s0 = 0.0, s1 = 0.0;
                    const int ja = jg * JL, jb2 = min(K, ja + JL);
                    int j = ja;
                    for (; j + 1 < jb2; j += 2) {
                        s0 = fma(big[(i64)j * K + kq], rl[j], s0);
                        s1 = fma(big[(i64)(j + 1) * K + kq], rl[j + 1], s1);
                    }
                    if (j < jb2) s0 = fma(big[(i64)j * K + kq], rl[j], s0);
                    s = s0 + s1;
                    sp[jg * KW + kq] = s;
                }
                {
                    const double v0 = wave_sum((jg < NJG && kq < K) ? rl[kq] * s : 0.0);
                    if (lane == 0) sred[wv] = v0;
                }
                if (wv >= UPD_WAVES - MM) {  // q_m = r^T XY[:, m] (:428): the last MM waves, one response each (the first ones carry XX r)
                    const int m = wv - (UPD_WAVES - MM);
                    double c = 0.0;
                    for (int kk = lane; kk < K; kk += WAVE) c = fma(rl[kk], xyl[m * K + kk], c);
                    c = wave_sum(c);
                    if (lane == 0) qa[m] = c;
                }
                lds_barrier();
                double pr = 0.0;
                if (kok)
                    for (int q = 0; q < NJG; ++q) pr += sp[q * KW + k];
                double tt = 0.0;
#pragma unroll
                for (int w = 0; w < UPD_WAVES; ++w) tt += sred[w];
                const double p = pr / tt;  // (:427)
                if (kok) {
                    P[k + (i64)a * K] = p;
                    Pl[k + (i64)a * K] = p;
#pragma unroll
                    for (int m = 0; m < MM; ++m) xyl[m * K + k] -= (p * (qa[m] / tt)) * tt;  // XY -= (p q^T) tt (:429)
                }
                if (tid < M) {
                    Q[tid + (i64)a * M] = qa[tid] / tt;
                    ql[tid + (i64)a * M] = qa[tid] / tt;
                }
            }
            lds_barrier();
            if (B && kok)  // B = R Q^T (:444-447)
                for (int m = 0; m < M; ++m) {
                    double b = 0.0;
                    for (int a = 0; a < A; ++a) b = fma(Rq[k + (i64)a * K], ql[m + (i64)a * M], b);
                    B[k + (i64)m * K] = b;
                }
        }
        for (int j = tid; j < K * A; j += UPD_THREADS) st_agent(rg.rshare + j, Rq[j]);
    }
    RG_STAMP(5);
    ok = resident_gram_barrier(rg, phase++, &flag, ok);
    RG_STAMP(6);

    // ---- 4. the scores of this workgroup's rows: T = X R.  A wave per slice of the columns, a lane per row of a 64-row block,
    //         RG_ACH score columns at a time; the 16 partial sums of a score meet in LDS in wave order ----
    double *Rl = Rq, *tp = big;  // tp: [16 waves][RG_ACH][64]
    if (g != 0 || !ok)
        for (int j = tid; j < K * A; j += UPD_THREADS) Rl[j] = ok ? ld_sc1(rg.rshare, j) : __builtin_nan("");
    if (r1 - r0 >= 6 * WAVE) {
        // many rows: a wave per 64-row block, a lane per row, all K columns -- no partial sums to meet, no barrier; 16 score columns
        // at a time (X is read again for the next 16: from L2)
        __syncthreads();  // Rl complete
        for (i64 b0 = r0 + (i64)wv * WAVE; b0 < r1; b0 += (i64)UPD_WAVES * WAVE) {
            const i64 row = b0 + lane;
            for (int a0 = 0; a0 < A; a0 += 16) {
                double ac[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) ac[c] = 0.0;
                for (int k0 = 0; k0 < K; k0 += 8) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = (row < r1 && k0 + u < K) ? (double)X[row + (i64)(k0 + u) * ldx] : 0.0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (k0 + u < K) {
#pragma unroll
                            for (int c = 0; c < 16; ++c)
                                if (a0 + c < A) ac[c] = fma(x[u], Rl[k0 + u + (i64)(a0 + c) * K], ac[c]);
                        }
                }
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (a0 + c < A && row < r1) Tm[row + (i64)(a0 + c) * ldt] = (T)ac[c];
            }
        }
        RG_STAMP(7);
        return;
    }
    const int ksl = (K + UPD_WAVES - 1) / UPD_WAVES, k_lo = wv * ksl, k_hi = min(K, k_lo + ksl);  // (ksl <= 8)
    double xn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xn[u] = (r0 + lane < r1 && k_lo + u < k_hi) ? (double)X[r0 + lane + (i64)(k_lo + u) * ldx] : 0.0;
    for (i64 b0 = r0; b0 < r1; b0 += WAVE) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = xn[u];
        {  // the next block's values are asked for ahead of this block's arithmetic
            const i64 row = b0 + WAVE + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u) xn[u] = (row < r1 && k_lo + u < k_hi) ? (double)X[row + (i64)(k_lo + u) * ldx] : 0.0;
        }
        for (int a0 = 0; a0 < A; a0 += RG_ACH) {
            __syncthreads();  // Rl complete / the previous tp has been read
            double ac[RG_ACH];
#pragma unroll
            for (int c = 0; c < RG_ACH; ++c) {
                ac[c] = 0.0;
                if (a0 + c < A) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (k_lo + u < k_hi) ac[c] = fma(x[u], Rl[k_lo + u + (i64)(a0 + c) * K], ac[c]);
                }
                tp[(wv * RG_ACH + c) * WAVE + lane] = ac[c];
            }
            __syncthreads();
            if (tid < RG_ACH * WAVE) {
                const int c = tid / WAVE, l = tid % WAVE;
                double s = 0.0;
                for (int w = 0; w < UPD_WAVES; ++w) s += tp[(w * RG_ACH + c) * WAVE + l];
                if (a0 + c < A && b0 + l < r1) Tm[b0 + l + (i64)(a0 + c) * ldt] = (T)s;
            }
        }
    }
    RG_STAMP(7);
}

}  // namespace plsk
