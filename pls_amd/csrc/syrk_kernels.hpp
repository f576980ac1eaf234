// XX = X^T X for KERNEL_TYPE2 (src/pls.cpp:398) -- the one contraction of the library that is
// matrix-core shaped: 2*N*K^2 flops (550 GF at N = 2^20, K = 512) on K^2 outputs.
//
// v_mfma_f64_16x16x4_f64, D(16x16) += A(16x4) B(4x16): A rows = 16 columns a of X, B columns = 16
// columns b of X, the contraction index = 4 rows of X.  Rows are the contiguous direction of the
// column-major X, so operands cannot come straight from global memory (16 columns x 32 bytes per
// wave-load); a workgroup stages a slab of RB = 32 rows x 128 columns per panel through LDS with
// the tile access pattern (256-byte column segments), stored [column][row] with the row count
// padded to 34 doubles so that the MFMA operand reads (lane: column l&15, row l>>4) hit 32 distinct
// bank pairs per half-wave.
//
// Workgroup = 4 waves = one 128 x 128 block of XX (each wave a 64 x 64 quadrant = 4 x 4 MFMA tiles,
// 64 fp64 accumulators per lane); 8 operand reads feed 16 MFMAs per 4-row step.  Only blocks on or
// above the diagonal are computed (the mirror image is written from the same accumulators).  The
// sum over rows is split over gridDim.y workgroups per block; their partial blocks are added by
// reduce_partials_kernel in a fixed order.
#pragma once
#include "common.hpp"

namespace plsk {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int SYRK_TB = 128;  // block tile (columns of X per panel)
// per storage type: V rows per 16-byte access, slab of RB = 16*V rows (256-byte column segments), LDS row
// count padded by 2 elements: conflict-free operand reads for 8-byte (fp64: stride 34 doubles) and 4-byte
// (fp32: stride 66 floats) elements alike.  fp32 panels are converted to fp64 at the operand read, so
// the accumulation policy (fp64) is the same as everywhere else in the library.
template <typename T>
struct SyrkCfg {
    static constexpr int V = 16 / sizeof(T);
    static constexpr int RB = 16 * V;
    static constexpr int LDP = RB + 2;
    static constexpr size_t LDS_BYTES = 2 * (size_t)SYRK_TB * LDP * sizeof(T);
};

// blockIdx.x enumerates the nbk*(nbk+1)/2 blocks (bi <= bj); blockIdx.y = row split.
template <typename T>
__global__ __launch_bounds__(256, 2) void syrk_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K, int nbk,
                                                      double *__restrict__ part) {
    constexpr int V = SyrkCfg<T>::V, SYRK_RB = SyrkCfg<T>::RB, SYRK_LDP = SyrkCfg<T>::LDP;
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_raw[];  // As[TB][LDP], Bs[TB][LDP]
    T *As = reinterpret_cast<T *>(slab_raw), *Bs = As + SYRK_TB * SYRK_LDP;

    int bi = 0, rem = blockIdx.x;
    while (rem >= nbk - bi) { rem -= nbk - bi; ++bi; }
    const int bj = bi + rem;
    const bool diag = (bi == bj);
    if (diag) Bs = As;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // staging map: row group rp (16 of them, V rows each), column group cgi (16 of them); 8 columns per thread
    const int rp = tid & 15, cgi = tid >> 4;
    // compute map
    const int a0 = (wv >> 1) * 64, b0 = (wv & 1) * 64;
    const int li = lane & 15, lq = lane >> 4;

    f64x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};

    const i64 nslabs = (N + SYRK_RB - 1) / SYRK_RB;
    Pack<T, V> ga[8], gb[8];

    auto load_slab = [&](i64 s) {
        const i64 r0 = s * SYRK_RB + V * rp;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ca = bi * SYRK_TB + cgi + 16 * j, cb = bj * SYRK_TB + cgi + 16 * j;
#pragma unroll
            for (int e = 0; e < V; ++e) ga[j].v[e] = gb[j].v[e] = (T)0;
            if (r0 + V <= N) {
                if (ca < K) ga[j] = ld_pack_nt<T, V>(X + r0 + (i64)ca * ldx);
                if (!diag && cb < K) gb[j] = ld_pack_nt<T, V>(X + r0 + (i64)cb * ldx);
            } else if (r0 < N) {  // ragged last rows: element-wise, the missing slots stay zero
                for (int e = 0; e < V; ++e)
                    if (r0 + e < N) {
                        if (ca < K) ga[j].v[e] = X[r0 + e + (i64)ca * ldx];
                        if (!diag && cb < K) gb[j].v[e] = X[r0 + e + (i64)cb * ldx];
                    }
            }
        }
    };
    auto store_slab = [&]() {  // 8-byte pieces: (col*LDP + V*rp)*sizeof(T) is 8-byte aligned for both types
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cgi + 16 * j;
            constexpr int H = 8 / sizeof(T);  // elements per 8-byte piece
#pragma unroll
            for (int e = 0; e < V; e += H) {
                *reinterpret_cast<Pack<T, H> *>(As + c * SYRK_LDP + V * rp + e) = *reinterpret_cast<const Pack<T, H> *>(&ga[j].v[e]);
                if (!diag)
                    *reinterpret_cast<Pack<T, H> *>(Bs + c * SYRK_LDP + V * rp + e) = *reinterpret_cast<const Pack<T, H> *>(&gb[j].v[e]);
            }
        }
    };

    // No software prefetch: the 64 staging registers would not fit beside the 128 accumulator
    // registers without spilling (measured: 6.8 ms with prefetch + spills, 6.5 ms without); the
    // second workgroup on the CU covers the load phase instead.
    // (a contiguous range of slabs per row split, as in the LDS-DMA kernel below: syrk8_slab_range)
    const i64 per_split = (nslabs + gridDim.y - 1) / gridDim.y;
    const i64 s_end = min(nslabs, (i64)(blockIdx.y + 1) * per_split);
    for (i64 s = (i64)blockIdx.y * per_split; s < s_end; ++s) {
        load_slab(s);
        __syncthreads();  // everyone is done reading the previous slab
        store_slab();
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SYRK_RB; kk += 4) {
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = (double)As[(a0 + 16 * m + li) * SYRK_LDP + kk + lq];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = (double)Bs[(b0 + 16 * n + li) * SYRK_LDP + kk + lq];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }

    // f64 C/D layout: lane holds D[row = (lane>>4) + 4*reg][col = lane&15]; row <-> a, col <-> b
    double *out = part + (i64)blockIdx.y * ((i64)K * K);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ga_ = bi * SYRK_TB + a0 + 16 * m + lq + 4 * r;
                const int gb_ = bj * SYRK_TB + b0 + 16 * n + li;
                if (ga_ < K && gb_ < K) {
                    const double v = acc[m][n][r];
                    out[ga_ + (i64)gb_ * K] = v;
                    if (!diag) out[gb_ + (i64)ga_ * K] = v;  // the mirror block
                }
            }
}

// 16 bytes per lane, global -> LDS without a register destination: lane l's bytes land at lds_base + 16 l
// (lds_base wave-uniform).  The builtin exists in the device pass only.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (cache policy of the DMA: default.  nt / sc0 / sc1 measured in round 4: 4.98 / 4.92 / 4.94 ms against 4.92 for the SYRK
    // launch at config 3 -- no effect, profiles/r4/syrk_sixteen_waves_32_row_slabs.txt)
    __builtin_amdgcn_global_load_lds(gsrc, lds_base, 16, 0, 0);
#else
    (void)gsrc;
    (void)lds_base;
#endif
}

// ---- LDS-DMA variant (the kernel that runs whenever N % V == 0; the register-staged one above takes ragged N) ----------
// The register-staged kernel above exposes the global-load latency of every slab (load -> barrier -> LDS
// store -> barrier -> 128 MFMAs; only the second workgroup of the CU covers it): 67-68 % MFMA-busy.  Here
// the panels go global -> LDS directly (global_load_lds_dwordx4: no staging registers),
// into one of two LDS buffers, one slab ahead of the MFMAs, with ONE
// barrier per slab:
//     barrier (slab s landed, slab s-1 consumed) -> issue slab s+1 -> 16 MFMAs x RB/4 steps on slab s.
// An LDS-DMA instruction writes 1 KB lane-linearly, so the LDS image of a panel is [column][8 positions of
// 16 bytes] (V rows per position, RB = 8 V rows per slab, 128 bytes per column) and cannot be padded; bank
// conflicts of the operand reads (16 lanes = 16 columns, stride 128 B) are removed by an XOR swizzle
// instead: the row group q of column c is stored at position q ^ ((c >> 1) & 7) -- applied to the SOURCE
// address of the DMA and to the READ address alike.
// Columns >= K read column K-1 again (their products are never written); the launcher requires N % V == 0
// and sends rows beyond N to a zero block.
// ---- eight waves per workgroup ------------------------------------------------------------------------------------------
// 512 threads = 8 waves, 64 accumulator registers per lane, so that two workgroups per CU are FOUR waves per SIMD: with two
// (the 4-wave form of rounds 1-2, 128 accumulator registers per lane, deleted in round 4: profiles/r3/syrk_eight_vs_four_waves_ab.txt)
// the MFMA pipe idles whenever both waves of a SIMD sit at their slab barriers.  Full blocks 5.61 ms -> 5.13 ms on config 3.
// Diagonal blocks: only the 36 tiles on or above the diagonal (a diagonal workgroup costs 0.74 of a full one per slab with
// X^T Y on board and gets that many more rows: sd row splits instead of so).
//   off-diagonal blocks: waves in a 4 x 2 arrangement, wave tile 32 x 64: 8 MFMAs and 6 operand reads per 4-row step;
//   diagonal blocks: the 36 tiles (16 x 16) on or above the diagonal of the 8 x 8 tile grid dealt out to the 8 waves in
//   enumeration order (5 or 4 each -- 5/8 of a full block's time per slab), every tile with its own two operand reads;
//   X^T Y rides along as before (thread = (column, quarter of the slab's rows)).
// The slabs of row split `split` of `nsplit`: a CONTIGUOUS range [first, end).  With every nsplit-th slab instead, all workgroups
// in flight read within nsplit consecutive slabs of each other -- at config 3 a 10 KB window of each of the 512 columns, the
// columns 8 MiB apart: the same few HBM channels for the whole chip.  Contiguous ranges put the row splits N / nsplit rows
// apart: 3.6 -> 4.06 TB/s of panel reads, 0.734 -> 0.826 of the matrix pipe in the stand-alone form of this kernel
// (pls_amd/csrc/tune/syrk_stream_probe.hip, profiles/r4/syrk_stream_probe.txt) -- the same as padding the columns' stride
// away from a power of two does.
// (Round 5 also weighted the row splits by DISPATCH POSITION -- the workgroups of the first residency wave, ids below the CU
// count, win the arbitration for their CU's matrix pipe and finish at 3.6 ms where the second workgroup of each CU takes 4.4
// (tools/syrk_stamps.py) -- first wave : second = 1.07 ... 1.23 x the pair's weight 1.17 ... 1.6: nothing below the unweighted
// 4.29 ms (profiles/r5/syrk_position_weights.txt); the launch is bound by what the CUs' pipes take in all, as the streaming
// sweeps are by the memory.)
__device__ __forceinline__ void syrk8_slab_range(i64 N, int RB, i64 split, int nsplit, i64 &first, i64 &end) {
    const i64 nall = (N + RB - 1) / RB;
    const i64 per = (nall + nsplit - 1) / nsplit;
    first = min(nall, split * per);
    end = min(nall, first + per);
}

// Arguments of an out-of-line device function arrive in VECTOR registers: the compiler no longer knows that they are the same
// in every lane, keeps the slab counter and every pointer derived from them per lane, and wraps each buffer instruction whose
// descriptor it cannot prove uniform in a read-first-lane loop.  uni() hands the value back as a scalar.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ i64 uni(i64 x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)x >> 32));
    return (i64)(((unsigned long long)hi << 32) | lo);
}
template <typename P>
__device__ __forceinline__ P *uni(P *p) { return reinterpret_cast<P *>(uni((i64)reinterpret_cast<uintptr_t>(p))); }

// The LDS-DMA of one slab through BUFFER descriptors (round 5): buffer_load_dwordx4 ... lds.  A wave's two DMA instructions per
// panel cover the column groups i = wv and wv + 8 (8 columns each); one descriptor per (panel, group) -- base = the group's first
// column, num_records = its columns that exist x ldx -- lives in scalar registers for the whole slab loop, the lane's offset
// (column inside the group, swizzled row position) in ONE vector register per group, and the slab's row offset rides in the
// instruction's scalar offset.  No 64-bit address arithmetic per slab, no select against a zero block: columns >= K are out of
// range (zeros), rows >= N get an out-of-range lane offset.  (With global_load_lds the pointers cost the loop 8-16 vector
// registers -- the difference between the nine-tile diagonal body fitting its 128 registers and reloading spilled addresses
// behind s_waitcnt vmcnt(0), i.e. behind the NEXT slab's DMA, in every 4-row step.)
constexpr int SYRK_BUF_WORD3 = 0x00020000;  // raw buffer, 32-bit data format, stride 0 (fused_kernels.hpp: BUF_WORD3)
template <typename T>
struct SyrkDma {
    __amdgpu_buffer_rsrc_t ra[2], rb[2];  // panels A and B, column groups wv and wv + 8
    uint32_t voff[2];                     // the lane's byte offset inside its group
    uint32_t rowq[2];                     // V x its swizzled row position (for the rows >= N test)
};
template <typename T>
__device__ __forceinline__ SyrkDma<T> syrk8_dma_setup(const T *__restrict__ X, i64 ldx, int K, int bi, int bj, int wv, int scol, int spos) {
    constexpr int V = 16 / sizeof(T);
    SyrkDma<T> d;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wv + 8 * j;
        const int col = 8 * i + scol;
        const int q = spos ^ ((col >> 1) & 7);
        d.voff[j] = (uint32_t)(((i64)scol * ldx + (i64)q * V) * (i64)sizeof(T));
        d.rowq[j] = (uint32_t)(q * V);
        const int ca0 = bi * SYRK_TB + 8 * i, cb0 = bj * SYRK_TB + 8 * i;
        const int na = max(0, min(8, K - ca0)), nb = max(0, min(8, K - cb0));
        // (everything in a descriptor is wave-uniform; said explicitly, or the buffer instruction is wrapped in a read-first-lane loop)
        d.ra[j] = __builtin_amdgcn_make_buffer_rsrc(uni(const_cast<T *>(X + (i64)min(ca0, K - 1) * ldx)), (short)0,
                                                    uni((int)((i64)na * ldx * (i64)sizeof(T))), SYRK_BUF_WORD3);
        d.rb[j] = __builtin_amdgcn_make_buffer_rsrc(uni(const_cast<T *>(X + (i64)min(cb0, K - 1) * ldx)), (short)0,
                                                    uni((int)((i64)nb * ldx * (i64)sizeof(T))), SYRK_BUF_WORD3);
    }
    return d;
}
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t r, void *lds_base, uint32_t voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_base, 16, (int)voff, soff, 0, 0);
#else
    (void)r; (void)lds_base; (void)voff; (void)soff;
#endif
}
template <typename T>
__device__ __forceinline__ void syrk8_dma(T *lds, int buf, const SyrkDma<T> &d, i64 N, bool diag, i64 s, int wv) {
    constexpr int V = 16 / sizeof(T), RB = 8 * V, CS = 8 * V, PANEL = SYRK_TB * CS;
    T *Ab = lds + (size_t)buf * 2 * PANEL, *Bb = Ab + PANEL;
    const i64 row0 = s * RB;
    const int soff = (int)(row0 * (i64)sizeof(T));
    const bool whole = row0 + RB <= N;  // (uniform) every row of the slab exists: all but the last slab of an N % RB != 0 matrix
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = wv + 8 * j;
        const uint32_t vo = (whole || row0 + d.rowq[j] < N) ? d.voff[j] : 0xFFFFFFF0u;  // rows >= N: out of range, zeros
        bufdma16(d.ra[j], Ab + i * (8 * CS), vo, soff);
        if (!diag) bufdma16(d.rb[j], Bb + i * (8 * CS), vo, soff);
    }
}

// the slab's rows of Y: [response m (8 of them)][8 row positions of 16 bytes]; responses >= M and rows >= N arrive as zeros
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t syrk8_y_rsrc(const T *Y, i64 ldy, int M) {
    return __builtin_amdgcn_make_buffer_rsrc(uni(const_cast<T *>(Y)), (short)0, uni((int)((i64)M * ldy * (i64)sizeof(T))), SYRK_BUF_WORD3);
}
template <typename T>
__device__ __forceinline__ void syrk8_dma_y(T *lds_dst, __amdgpu_buffer_rsrc_t ry, i64 ldy, i64 N, i64 s, int lane) {
    constexpr int V = 16 / sizeof(T), RB = 8 * V;
    const int m = lane >> 3, pos = lane & 7;
    const i64 row = s * RB + (i64)pos * V;
    const uint32_t vo = row < N ? (uint32_t)(((i64)m * ldy + (i64)pos * V) * (i64)sizeof(T)) : 0xFFFFFFF0u;
    bufdma16(ry, lds_dst, vo, (int)(s * RB * (i64)sizeof(T)));
}

// (one out-of-line body per wave index W, as for the paired diagonal blocks below: tile offsets are instruction immediates)
template <typename T, int W>
__device__ __noinline__ void syrk8_full_wave(const T *X_, i64 ldx_, i64 N_, int K_, int bi_, int bj_, i64 s0_, i64 s1_, double *out_) {
    const T *__restrict__ X = uni(X_);
    double *__restrict__ out = uni(out_);
    const i64 ldx = uni(ldx_), N = uni(N_), s0 = uni(s0_);
    const int K = uni(K_), bi = uni(bi_), bj = uni(bj_);
    const i64 s1 = uni(s1_);
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_raw[];
    T *lds = reinterpret_cast<T *>(slab_raw);
    constexpr int V = 16 / sizeof(T), RB = 8 * V, CS = 8 * V, PANEL = SYRK_TB * CS;
    constexpr int a0 = (W >> 1) * 32, b0 = (W & 1) * 64;
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int fl = li >> 1;
    const int scol = lane >> 3, spos = lane & 7;
    f64x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
    i64 s = s0;
    const i64 nslabs = s1;  // this workgroup's slabs [s0, s1): syrk_glds8_kernel
    int buf = 0;
    const SyrkDma<T> dma = syrk8_dma_setup<T>(X, ldx, K, bi, bj, W, scol, spos);
    if (s < nslabs) syrk8_dma<T>(lds, 0, dma, N, false, s, W);
    for (; s < nslabs; ++s, buf ^= 1) {
        // (the slab's DMA must have LANDED: for buffer_load ... lds the compiler does not put the wait in front of the barrier
        // by itself -- it left vmcnt(3) there, and repeated fits differed: test_race_screen_repeated_fits)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const T *As = lds + (size_t)buf * 2 * PANEL + li * CS, *Bs = As + PANEL;
        if (s + 1 < nslabs) syrk8_dma<T>(lds, buf ^ 1, dma, N, false, s + 1, W);
#pragma unroll
        for (int kk = 0; kk < RB; kk += 4) {
            const int r = kk + lq;
            const int off = (((r / V) ^ fl) * V) + (r % V);
            double a[2], b[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = (double)As[(a0 + 16 * m) * CS + off];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = (double)Bs[(b0 + 16 * n) * CS + off];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ga_ = bi * SYRK_TB + a0 + 16 * m + lq + 4 * r;
                const int gb_ = bj * SYRK_TB + b0 + 16 * n + li;
                if (ga_ < K && gb_ < K) {
                    const double v = acc[m][n][r];
                    out[ga_ + (i64)gb_ * K] = v;
                    out[gb_ + (i64)ga_ * K] = v;
                }
            }
}

template <typename T>
__device__ __forceinline__ void syrk8_full_body(const T *__restrict__ X, i64 ldx, i64 N, int K, int bi, int bj, i64 s0, i64 s1,
                                             const T *__restrict__ zeros, double *__restrict__ out) {
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {  // (wave-uniform: a scalar branch)
#define FULL_WAVE(W_) case W_: syrk8_full_wave<T, W_>(X, ldx, N, K, bi, bj, s0, s1, out); break;
        FULL_WAVE(0) FULL_WAVE(1) FULL_WAVE(2) FULL_WAVE(3) FULL_WAVE(4) FULL_WAVE(5) FULL_WAVE(6) FULL_WAVE(7)
#undef FULL_WAVE
    }
}

// tile t of the upper triangle of the 8 x 8 tile grid, row by row: (i, j), i <= j
__device__ __forceinline__ void syrk8_tri_tile(int t, int &i, int &j) {
    i = 0;
    while (t >= 8 - i) { t -= 8 - i; ++i; }
    j = i + t;
}

template <typename T, bool WITH_Y>
__device__ __forceinline__ void syrk8_diag_body(const T *__restrict__ X, i64 ldx, i64 N, int K, int bi, i64 s0, i64 s1,
                                             const T *__restrict__ zeros, double *__restrict__ out, const T *__restrict__ Y, i64 ldy,
                                             int M, double *__restrict__ xy_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_raw[];
    T *lds = reinterpret_cast<T *>(slab_raw);
    constexpr int V = 16 / sizeof(T), RB = 8 * V, CS = 8 * V, PANEL = SYRK_TB * CS;
    constexpr int NT5 = 5;  // tiles per wave at most: 36 tiles over 8 waves
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int fl = li >> 1;
    const int scol = lane >> 3, spos = lane & 7;
    // This wave's tiles: up to two ROW SEGMENTS of the upper triangle of the 8 x 8 tile grid -- (ra; columns ca .. ca+na-1)
    // and (rb; cb .. cb+nb-1), na + nb = 5 or 4 -- so that a 4-row step reads 2 row operands + 5 column operands instead of
    // two per tile:   w0-w3: row w, columns w .. w+4;   w4: (0; 5-7) + (7; 7);   w5: (1; 6-7) + (6; 6-7);
    //                 w6: (2; 7) + (5; 5-7);   w7: (4; 4-7).   5 5 5 5 4 4 4 4 = 36 tiles.  (wave-uniform: scalar registers)
    const int wq = wv & 3;
    const bool hi = wv >= 4;
    const int ra = __builtin_amdgcn_readfirstlane(hi ? (wv == 7 ? 4 : wq) : wq);
    const int ca = __builtin_amdgcn_readfirstlane(hi ? (wv == 7 ? 4 : 5 + wq) : wq);
    const int na = __builtin_amdgcn_readfirstlane(hi ? (wv == 7 ? 4 : 3 - wq) : 5);
    const int rb = __builtin_amdgcn_readfirstlane(hi ? 7 - wq : 0);
    const int cb = __builtin_amdgcn_readfirstlane(hi ? 7 - wq : 0);
    const int nb = __builtin_amdgcn_readfirstlane((hi && wv != 7) ? 1 + wq : 0);
    int ti[NT5], tj[NT5];
#pragma unroll
    for (int t = 0; t < NT5; ++t) {
        const bool inA = t < na;
        ti[t] = inA ? ra : rb;
        tj[t] = inA ? ca + t : cb + (t - na);
        if (t >= na + nb) { ti[t] = 0; tj[t] = 0; }
    }
    const bool last = (na + nb == NT5);  // does the fifth tile exist?
    f64x4 acc[NT5];
#pragma unroll
    for (int t = 0; t < NT5; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    double accy[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) accy[m] = 0.0;
    const int yc = tid >> 2, yh = tid & 3;  // X^T Y: column of the panel, quarter of the slab's 8 row positions
    const int ykey = (yc >> 1) & 7;
    i64 s = s0;
    const i64 nslabs = s1;  // this workgroup's slabs [s0, s1): syrk_glds8_kernel
    const SyrkDma<T> dma = syrk8_dma_setup<T>(X, ldx, K, bi, bi, wv, scol, spos);
    const __amdgpu_buffer_rsrc_t ry = syrk8_y_rsrc<T>(Y, ldy, WITH_Y ? M : 0);
    auto issue = [&](i64 s, int buf) {
        syrk8_dma<T>(lds, buf, dma, N, true, s, wv);
        if constexpr (WITH_Y) {
            if (wv == 0) syrk8_dma_y<T>(lds + (size_t)2 * 2 * PANEL + (size_t)buf * (8 * CS), ry, ldy, N, s, lane);
        }
    };
    int buf = 0;
    if (s < nslabs) issue(s, 0);
    for (; s < nslabs; ++s, buf ^= 1) {
        // (the slab's DMA must have LANDED: for buffer_load ... lds the compiler does not put the wait in front of the barrier
        // by itself -- it left vmcnt(3) there, and repeated fits differed: test_race_screen_repeated_fits)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const T *As = lds + (size_t)buf * 2 * PANEL;
        if (s + 1 < nslabs) issue(s + 1, buf ^ 1);
#pragma unroll
        for (int kk = 0; kk < RB; kk += 4) {
            const int r = kk + lq;
            const int off = (((r / V) ^ fl) * V) + (r % V);
            double a[NT5], b[NT5];
            const double opa = (double)As[(16 * ra + li) * CS + off], opb = (double)As[(16 * rb + li) * CS + off];
#pragma unroll
            for (int t = 0; t < NT5; ++t) {
                a[t] = (t < na) ? opa : opb;
                b[t] = (double)As[(16 * tj[t] + li) * CS + off];
            }
#pragma unroll
            for (int t = 0; t < NT5 - 1; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc[t], 0, 0, 0);
            if (last) acc[NT5 - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[NT5 - 1], b[NT5 - 1], acc[NT5 - 1], 0, 0, 0);
        }
        if constexpr (WITH_Y) {  // after the slab's MFMAs: the wait the compiler puts in front of these LDS reads (vmcnt(0): the
                                 // next slab's DMA) is the wait of the coming barrier anyway (measured equal to the placement
                                 // ahead of the DMA issue that the 4-wave form uses: 3,340-3,350 components/s either way)
            const T *Ys = lds + (size_t)2 * 2 * PANEL + (size_t)buf * (8 * CS);
            Pack<T, V> xv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) xv[q] = *reinterpret_cast<const Pack<T, V> *>(As + yc * CS + (((2 * yh + q) ^ ykey) * V));
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (m < M) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const Pack<T, V> yv = *reinterpret_cast<const Pack<T, V> *>(Ys + (m * 8 + 2 * yh + q) * V);
#pragma unroll
                        for (int e = 0; e < V; ++e) accy[m] = fma((double)xv[q].v[e], (double)yv.v[e], accy[m]);
                    }
                }
        }
    }
    if constexpr (WITH_Y) {
        __syncthreads();  // the panels have been read
        double *ysh = reinterpret_cast<double *>(slab_raw);  // [4][128][8]
#pragma unroll
        for (int m = 0; m < 8; ++m) ysh[(yh * SYRK_TB + yc) * 8 + m] = accy[m];
        __syncthreads();
        const int col = bi * SYRK_TB + yc;
        if (yh == 0 && col < K)
            for (int m = 0; m < M; ++m)
                xy_out[col + (i64)m * K] = (ysh[yc * 8 + m] + ysh[(SYRK_TB + yc) * 8 + m]) +
                                           (ysh[(2 * SYRK_TB + yc) * 8 + m] + ysh[(3 * SYRK_TB + yc) * 8 + m]);
    }
#pragma unroll
    for (int t = 0; t < NT5; ++t) {
        if (t == NT5 - 1 && !last) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ga_ = bi * SYRK_TB + 16 * ti[t] + lq + 4 * r;
            const int gb_ = bi * SYRK_TB + 16 * tj[t] + li;
            if (ga_ < K && gb_ < K) {
                const double v = acc[t][r];
                out[ga_ + (i64)gb_ * K] = v;
                out[gb_ + (i64)ga_ * K] = v;
            }
        }
    }
}


// ---- TWO diagonal blocks per workgroup (round 5) ------------------------------------------------------------------------------
// A diagonal block alone gives its eight waves 5 + 4 tiles (36 of 64) with 7 operand reads per 5 MFMAs and X^T Y as VALU work on
// LDS data of its own: 0.63 of the matrix pipe on 27 % of config 3's tiles, and its row splits never balance exactly against
// the full blocks'.  Here a workgroup takes the diagonal blocks ba and bb TOGETHER -- both panels in LDS, exactly the LDS-DMA
// stream of an off-diagonal workgroup -- and wave w computes row w of the upper triangle of ba (tiles (w, w..7): 8 - w of them)
// and row 7 - w of bb's (w + 1 tiles): NINE tiles for every wave, 9 operand reads per step (the diagonal tile's two operands are
// the same register).  X^T Y costs no read of X at all: a wave's two ROW operands are the X values (row kk + lq, column 16 w + li)
// and (.., 16 (7 - w) + li) that X^T Y needs, every row tile of the two panels is the row operand of exactly one wave, so
// X^T Y = one broadcast LDS read of Y and two FMAs per response and 4-row step, summed over the four lq lane groups at the end.
// MT: responses the instantiation carries accumulators for (0: no X^T Y).
// One out-of-line body per WAVE INDEX W: every tile index, LDS offset and operand choice is a compile-time constant (a body
// with the wave index at run time selects operands per tile and spills 50-100 registers at the 128 four waves per SIMD leave;
// round 2 found the same for the single diagonal block's patterns).  The waves of a workgroup run different functions with the
// same slab loop, so they meet at the same barriers.
template <typename T, int MT, int W>
__device__ __noinline__ void syrk8_dd_wave(const T *X_, i64 ldx_, i64 N_, int K_, int ba_, int bb_, i64 s0_, i64 s1_, double *out_,
                                           const T *Y_, i64 ldy_, int M_, double *xy_out_) {
    const T *__restrict__ X = uni(X_), *__restrict__ Y = uni(Y_);
    double *__restrict__ out = uni(out_), *__restrict__ xy_out = uni(xy_out_);
    const i64 ldx = uni(ldx_), N = uni(N_), s0 = uni(s0_), ldy = uni(ldy_);
    const int K = uni(K_), ba = uni(ba_), bb = uni(bb_), M = uni(M_);
    const i64 s1 = uni(s1_);
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_raw[];
    T *lds = reinterpret_cast<T *>(slab_raw);
    constexpr int V = 16 / sizeof(T), RB = 8 * V, CS = 8 * V, PANEL = SYRK_TB * CS;
    constexpr int NA = 8 - W, NB = W + 1, RBW = 7 - W;  // tiles (W, W .. 7) of panel A, (RBW, RBW .. 7) of panel B
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int fl = li >> 1;
    const int scol = lane >> 3, spos = lane & 7;
    f64x4 acca[NA], accb[NB];
#pragma unroll
    for (int t = 0; t < NA; ++t) acca[t] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NB; ++t) accb[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    double accy[2][MT > 0 ? MT : 1];
#pragma unroll
    for (int m = 0; m < (MT > 0 ? MT : 1); ++m) accy[0][m] = accy[1][m] = 0.0;
    i64 s = s0;
    const i64 nslabs = s1;  // this workgroup's slabs [s0, s1): syrk_glds8_kernel
    const SyrkDma<T> dma = syrk8_dma_setup<T>(X, ldx, K, ba, bb, W, scol, spos);
    const __amdgpu_buffer_rsrc_t ry = syrk8_y_rsrc<T>(Y, ldy, MT > 0 ? M : 0);
    auto issue = [&](i64 s, int buf) {
        syrk8_dma<T>(lds, buf, dma, N, false, s, W);
        if constexpr (MT > 0 && W == 0) syrk8_dma_y<T>(lds + (size_t)2 * 2 * PANEL + (size_t)buf * (8 * CS), ry, ldy, N, s, lane);
    };
    int buf = 0;
    if (s < nslabs) issue(s, 0);
    for (; s < nslabs; ++s, buf ^= 1) {
        // (the slab's DMA must have LANDED: for buffer_load ... lds the compiler does not put the wait in front of the barrier
        // by itself -- it left vmcnt(3) there, and repeated fits differed: test_race_screen_repeated_fits)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const T *As = lds + (size_t)buf * 2 * PANEL + li * CS, *Bs = As + PANEL;  // (+ the lane's column inside a tile)
        const T *Ys = lds + (size_t)2 * 2 * PANEL + (size_t)buf * (8 * CS);
        if (s + 1 < nslabs) issue(s + 1, buf ^ 1);
#pragma unroll
        for (int kk = 0; kk < RB; kk += 4) {
            const int r = kk + lq;
            const int off = (((r / V) ^ fl) * V) + (r % V);
            double xa[NA], xb[NB];  // column operands; xa[0] / xb[0] are the row operands too (the diagonal tiles)
#pragma unroll
            for (int t = 0; t < NA; ++t) xa[t] = (double)As[(16 * (W + t)) * CS + off];
#pragma unroll
            for (int t = 0; t < NB; ++t) xb[t] = (double)Bs[(16 * (RBW + t)) * CS + off];
#pragma unroll
            for (int t = 0; t < NA; ++t) acca[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], xa[t], acca[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NB; ++t) accb[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[0], xb[t], accb[t], 0, 0, 0);
            if constexpr (MT > 0) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const double y = (double)Ys[(m * 8 + r / V) * V + r % V];  // (responses >= M: zeros came with the DMA)
                    accy[0][m] = fma(xa[0], y, accy[0][m]);
                    accy[1][m] = fma(xb[0], y, accy[1][m]);
                }
            }
        }
    }
    if constexpr (MT > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const double sum = xor_range_sum<16, 64>(accy[h][m]);  // over the four lq groups
                const int col = (h == 0 ? ba * SYRK_TB + 16 * W : bb * SYRK_TB + 16 * RBW) + li;
                if (lq == 0 && m < M && col < K) xy_out[col + (i64)m * K] = sum;
            }
    }
    auto store = [&](const f64x4 &acc, int blk, int ti, int tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ga_ = blk * SYRK_TB + 16 * ti + lq + 4 * r;
            const int gb_ = blk * SYRK_TB + 16 * tj + li;
            if (ga_ < K && gb_ < K) {
                const double v = acc[r];
                out[ga_ + (i64)gb_ * K] = v;
                out[gb_ + (i64)ga_ * K] = v;
            }
        }
    };
#pragma unroll
    for (int t = 0; t < NA; ++t) store(acca[t], ba, W, W + t);
#pragma unroll
    for (int t = 0; t < NB; ++t) store(accb[t], bb, RBW, RBW + t);
}

template <typename T, int MT>
__device__ __forceinline__ void syrk8_dd_body(const T *__restrict__ X, i64 ldx, i64 N, int K, int ba, int bb, i64 s0, i64 s1,
                                              const T *__restrict__ zeros, double *__restrict__ out, const T *__restrict__ Y, i64 ldy,
                                              int M, double *__restrict__ xy_out) {
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {  // (wave-uniform: a scalar branch)
#define DD_WAVE(W_) case W_: syrk8_dd_wave<T, MT, W_>(X, ldx, N, K, ba, bb, s0, s1, out, Y, ldy, M, xy_out); break;
        DD_WAVE(0) DD_WAVE(1) DD_WAVE(2) DD_WAVE(3) DD_WAVE(4) DD_WAVE(5) DD_WAVE(6) DD_WAVE(7)
#undef DD_WAVE
    }
}

template <typename T>
__global__ __launch_bounds__(512, 4) void syrk_glds8_kernel(const T *__restrict__ X, i64 ldx, i64 N, int K, int nbk,
                                                            const T *__restrict__ zeros, double *__restrict__ part, int so, int sd,
                                                            const T *__restrict__ Y, i64 ldy, int M, double *__restrict__ xypart,
                                                            int pairs) {
    // (Round 4: starting half of the workgroups late -- the second half of the grid, odd ids, or every second group of 8, by
    // 16 ... 96 x 64 cycles -- so that the two workgroups of a CU do not sit at their slab barriers together: no effect,
    // 4.91-4.96 ms in all 18 combinations, profiles/r4/syrk_stagger.txt.)
    // 1-D grid, exactly the workgroups that have rows: first so splits of every off-diagonal block, then sd of every PAIR of
    // diagonal blocks (pairs != 0: blocks 2 p and 2 p + 1 in one workgroup, syrk8_dd_body), then sd of every diagonal block left
    // (workgroups that only exit still take a dispatch slot: a 2-D grid with idle members ran 7.8 ms instead of 6.1).  A
    // diagonal block has fewer partials than the so the reduction sums: its workgroup j also zeroes the slots j + sd, ... < so.
#ifdef PLS_STAMP
    PLS_STAMP(0);
#endif
    const int npair = pairs ? nbk / 2 : 0, nsingle = nbk - 2 * npair, noff = nbk * (nbk + 1) / 2 - nbk;
    int id = blockIdx.x, split, blk, kind;  // kind 0: off-diagonal, 1: a pair of diagonal blocks, 2: one diagonal block
    if (id < noff * so) {
        kind = 0;
        blk = id % noff;
        split = id / noff;
    } else if (id < noff * so + npair * sd) {
        kind = 1;
        id -= noff * so;
        blk = id % npair;
        split = id / npair;
    } else {
        kind = 2;
        id -= noff * so + npair * sd;
        blk = 2 * npair + id % nsingle;
        split = id / nsingle;
    }
    double *out = part + (i64)split * ((i64)K * K);
    // the reduction sums max(so, sd) partial matrices: a block with fewer row splits zeroes the slots it does not write
    const int nslots = max(so, sd);
    auto zero_block = [&](int bzi, int bzj, int mine) {
        for (int z = split + mine; z < nslots; z += mine) {
            double *zo = part + (i64)z * ((i64)K * K);
            for (int e = threadIdx.x; e < SYRK_TB * SYRK_TB; e += 512) {
                const int ga_ = bzi * SYRK_TB + (e & (SYRK_TB - 1)), gb_ = bzj * SYRK_TB + e / SYRK_TB;
                if (ga_ < K && gb_ < K) zo[ga_ + (i64)gb_ * K] = 0.0;
            }
        }
    };
    if (kind == 1) {
        zero_block(2 * blk, 2 * blk, sd);
        zero_block(2 * blk + 1, 2 * blk + 1, sd);
    } else if (kind == 2) {
        zero_block(blk, blk, sd);
    }
    double *xyo = xypart + (i64)split * ((i64)K * M);
    constexpr int RBK = 8 * (16 / (int)sizeof(T));
    i64 s0, s1;  // this workgroup's slabs
    syrk8_slab_range(N, RBK, split, kind == 0 ? so : sd, s0, s1);
    if (kind == 1) {
        if (!Y) syrk8_dd_body<T, 0>(X, ldx, N, K, 2 * blk, 2 * blk + 1, s0, s1, zeros, out, nullptr, 0, 0, nullptr);
        else if (M <= 1) syrk8_dd_body<T, 1>(X, ldx, N, K, 2 * blk, 2 * blk + 1, s0, s1, zeros, out, Y, ldy, M, xyo);
        else if (M <= 2) syrk8_dd_body<T, 2>(X, ldx, N, K, 2 * blk, 2 * blk + 1, s0, s1, zeros, out, Y, ldy, M, xyo);
        else syrk8_dd_body<T, 4>(X, ldx, N, K, 2 * blk, 2 * blk + 1, s0, s1, zeros, out, Y, ldy, M, xyo);  // (M <= 4: the launcher)
    } else if (kind == 2) {
        if (Y)
            syrk8_diag_body<T, true>(X, ldx, N, K, blk, s0, s1, zeros, out, Y, ldy, M, xyo);
        else
            syrk8_diag_body<T, false>(X, ldx, N, K, blk, s0, s1, zeros, out, nullptr, 0, 0, nullptr);
    } else {
        int bi = 0, rem = blk;
        while (rem >= nbk - 1 - bi) { rem -= nbk - 1 - bi; ++bi; }
        if (so < nslots) {
            zero_block(bi, bi + 1 + rem, so);
            zero_block(bi + 1 + rem, bi, so);
        }
        syrk8_full_body<T>(X, ldx, N, K, bi, bi + 1 + rem, s0, s1, zeros, out);
    }
#ifdef PLS_STAMP
    PLS_STAMP(1);
#endif
}

// (Round 4: a 16-wave form with 32-row slabs -- 256-byte DMA pieces, half the barriers, one workgroup per CU -- was built after
// the PMC reading of this kernel and measured EQUAL, 4.93-4.96 vs 4.94-4.97 ms at config 3: profiles/r4/syrk_sixteen_waves_32_row_slabs.txt.
// Nor is it the 17.2 GB per launch that leave L2 (an XCD-grouped grid cut them to 10.5 GB at the same launch time): on the CUs that
// set the launch time the MFMA / ds_read loop alone runs at 0.857 of the pipe, DMA issue + slab barrier cost another 12 % --
// profiles/r4/syrk_where_the_time_goes.txt.  Deleted.)

// rc: 0 launched (part holds *nb partial K x K matrices), 1 shape not covered
// zeros: >= 16 bytes of device zeros (source of the rows beyond N in the LDS-DMA variant); nullptr = register-staged kernel
template <typename T>
int launch_syrk(hipStream_t stream, int num_cu, const T *X, i64 ldx, i64 N, int K, double *part,
                i64 part_capacity_doubles, int *nb, const void *zeros = nullptr, const T *Y = nullptr, i64 ldy = 0, int M = 0,
                double *xypart = nullptr, i64 xypart_capacity_doubles = 0, int *nb_xy = nullptr) {
    if (nb_xy) *nb_xy = 0;
    constexpr int SYRK_RB = SyrkCfg<T>::RB;
    const size_t SYRK_LDS_BYTES = SyrkCfg<T>::LDS_BYTES;
    if (((uintptr_t)X % 16) != 0 || (ldx % SyrkCfg<T>::V) != 0 || N < 1) return 1;
    const int nbk = (K + SYRK_TB - 1) / SYRK_TB;
    const int nblocks = nbk * (nbk + 1) / 2;
    const i64 nslabs = (N + SYRK_RB - 1) / SYRK_RB;
    // Row split: all workgroups resident at once (2 per CU) when the blocks allow it -- a grid that is
    // one workgroup over a residency wave takes twice as long (measured: 520 workgroups 10.0 ms, 510
    // workgroups 6.5 ms) -- otherwise at least ~8 waves so that the tail is small.
    const i64 slots = 2 * (i64)num_cu;
    i64 S = nblocks <= slots ? slots / nblocks : (8 * slots + nblocks - 1) / nblocks;
    S = std::max<i64>(1, std::min<i64>(S, nslabs));
    S = std::min<i64>(S, part_capacity_doubles / ((i64)K * K));
    if (S < 1) return 1;
    constexpr int V = SyrkCfg<T>::V;
    // (the LDS-DMA variant addresses a group of 8 columns through one 32-bit buffer descriptor and a slab through a 32-bit scalar
    // offset: 8 ld s and N s below 2^32 -- 67 M fp64 rows per column; beyond, the register-staged kernel)
    const bool dma_span_ok = (i64)8 * ldx * (i64)sizeof(T) < (1ll << 32) - 4096 && (i64)N * (i64)sizeof(T) < (1ll << 32) - 4096 &&
                             (!Y || (i64)8 * ldy * (i64)sizeof(T) < (1ll << 32) - 4096);
    if (zeros && N % V == 0 && K >= 1 && dma_span_ok) {  // LDS-DMA variant: slabs of 8 V rows, two LDS buffers of two panels
        constexpr int RBG = 8 * V;
        constexpr size_t LDS_G = 2 * 2 * (size_t)SYRK_TB * 128 + 2 * 1024;  // two buffers of two panels + two slabs of Y
        const i64 nslabs_g = (N + RBG - 1) / RBG;
        i64 Sg = (8 * slots + nblocks - 1) / nblocks;  // (several waves of workgroups; overwritten for one residency wave)
        i64 Sd = 0;  // row splits of the diagonal blocks; 0 = as the others (several residency waves balance themselves)
        // X^T Y rides along in the diagonal workgroups (one partial row per row split of a diagonal block)
        const bool fuse_y0 = Y && xypart && nb_xy && M >= 1 && M <= 8 && ((uintptr_t)Y % 16) == 0 && (ldy % V) == 0;
        // one residency wave: a diagonal workgroup costs 36/64 per slab (more with X^T Y on board).  Only while the blocks leave
        // at least sixteen row splits each (up to 7 column blocks, K <= 896): with fewer the slots do not fill and the two kinds of
        // workgroup balance coarsely -- K = 2048: 136 blocks, 3 + 2 splits, 392 of 512 slots, 0.58 of the matrix pipe -- while eight
        // waves of workgroups balance themselves and have few diagonal blocks among many: K = 1024 0.77 -> 0.80, 1280 0.68 -> 0.80,
        // 1536 0.65 -> 0.81, 2048 0.58 -> 0.82 (profiles/r4/syrk_scan.txt)
        // Diagonal blocks in PAIRS (syrk8_dd_body: 72 tiles per workgroup and slab against a full block's 64, nine reads per nine
        // MFMAs against six per eight): a pair weighs dw2 full blocks, a diagonal block left over (odd block count) rides with
        // the pairs' row splits.
        // Measured at config 3 (profiles/r5/syrk_pairs.txt, same box): one diagonal block per workgroup 4.83 ms (0.746 of the fp64
        // matrix pipe); pairs at dw2 = 0.95 / 1.0 / 1.05 / 1.1 / 1.15 / 1.2 / 1.3: 4.61 / 4.41 / 4.38 / 4.32 / 4.296 / 4.306 / 4.38 ms
        // (0.840).  PLS_HIP_SYRK_PAIRS=0 keeps the single blocks (A/B measurements; read once per process).
        static const int exp_pairs = getenv("PLS_HIP_SYRK_PAIRS") ? atoi(getenv("PLS_HIP_SYRK_PAIRS")) : 1;
        constexpr double exp_dw2 = 1.17;
        // (more than 4 responses on board: 2 x M accumulators beside the 72 of the tiles no longer fit the 128 registers of four
        // waves per SIMD -- those fits keep one diagonal block per workgroup)
        const int pairs = (exp_pairs && nbk >= 2 && !(fuse_y0 && M > 4)) ? 1 : 0;
        const int npair = pairs ? nbk / 2 : 0, nsingle = nbk - 2 * npair;
        const i64 ndunits = pairs ? npair + nsingle : nbk;  // workgroup kinds on the diagonal, per row split
        const double dw = pairs ? exp_dw2 : (fuse_y0 ? (M == 1 ? 0.74 : 0.82) : 0.66);
        if (nblocks <= slots && (i64)(slots / ((nblocks - nbk) + dw * ndunits)) >= 16) {
            // measured optima of the row-split weight (profiles/r3/syrk_diagonal_weight_sweep.txt): 0.74 with X^T Y of one
            // response on board, 0.82 for several, 0.66 without
            const double units = (nblocks - nbk) + dw * ndunits;
            Sg = (i64)(slots / units);
            Sd = std::max<i64>(1, (i64)(dw * Sg));
            while ((nblocks - nbk) * Sg + ndunits * Sd > slots && Sg > 1) { --Sg; Sd = std::max<i64>(1, (i64)(dw * Sg)); }
        }
        Sg = std::max<i64>(1, std::min<i64>(Sg, nslabs_g));
        Sg = std::min<i64>(Sg, part_capacity_doubles / ((i64)K * K));
        if (Sg < 1) return 1;
        Sd = Sd ? std::min<i64>(Sd, std::min<i64>(nslabs_g, part_capacity_doubles / ((i64)K * K))) : Sg;
        if (!raise_dynamic_lds(reinterpret_cast<const void *>(&syrk_glds8_kernel<T>), (int)LDS_G)) return 1;
        const i64 nwg8 = (i64)(nblocks - nbk) * Sg + ndunits * Sd;
        const bool fuse_y8 = fuse_y0 && Sd * (i64)K * M <= xypart_capacity_doubles;
        const i64 nslots = std::max(Sg, Sd);
        hipLaunchKernelGGL(syrk_glds8_kernel<T>, dim3((unsigned)nwg8), dim3(512), LDS_G, stream, X, ldx, N, K, nbk,
                           static_cast<const T *>(zeros), part, (int)Sg, (int)Sd, fuse_y8 ? Y : nullptr, ldy, M, xypart, pairs);
        if (fuse_y8) *nb_xy = (int)Sd;
        *nb = (int)nslots;
        return 0;
    }
    if (!raise_dynamic_lds(reinterpret_cast<const void *>(&syrk_kernel<T>), (int)SYRK_LDS_BYTES)) return 1;
    hipLaunchKernelGGL(syrk_kernel<T>, dim3(nblocks, (unsigned)S), dim3(256), SYRK_LDS_BYTES, stream, X, ldx, N, K, nbk,
                       part);
    *nb = (int)S;
    return 0;
}

}  // namespace plsk
