// Row-tile-resident fused pass: ONE sweep over X per component.
//
//   KERNEL algo :  t = X r ;  p_raw = X^T t ;  tt = t^T t                    (src/pls.cpp:419-421)
//                  -> X is read once per component instead of twice.
//   NIPALS algo :  X' = X - t_prev p_prev^T (written once) ;  t = X' w ;  p_raw = X'^T t ; tt
//                  -> the north star's rank-1 deflation fused with the next component's score
//                     and loading products: one read + one write of X per component.
//
// t_i needs the WHOLE row i before p can use it, so a workgroup keeps a full-width tile
// (R rows x all K columns) resident in registers: thread (rp, cg) owns V consecutive rows
// (one 16-byte access) of the columns cg, cg+CG, cg+2CG, ...  A tile is K separate R*s-byte
// segments (column stride ld*s); R*s = 256 B is the shortest segment that still streams at full
// HBM rate on MI355X (measured: profiles/r1/tile_probe.txt -- 5.9 TB/s read-only at R = 32 fp64
// rows, 4.3 TB/s at R = 16), so R = 32 (fp64) keeps the register tile small enough for two
// 512-thread workgroups per CU, which is what overlaps one workgroup's reduction phase with
// the other's loads.
//
// Per tile: all loads issued back to back (CPT x 16 B per lane in flight), [deflate + store],
// per-lane partial t over the lane's columns, butterfly over the lanes that share rows, one
// LDS exchange between the waves (double-buffered: one barrier per tile), then the loading
// accumulation into CPT per-lane accumulators that live across all tiles of the workgroup.
// The column set of a lane belongs to exactly one wave, so the final X^T t partial needs only
// an in-wave butterfly -- no atomics anywhere, results are bit-reproducible.
#pragma once
#include "common.hpp"

namespace plsk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// gfx9-family raw buffer descriptor word 3 (DST_SEL xyzw, 32-bit data format); stride 0:
// offsets are plain byte offsets, range-checked against num_records -- an out-of-range load
// returns 0 and an out-of-range store is dropped, which is how the ragged edges are handled.
constexpr int BUF_WORD3 = 0x00020000;

// AUX: cache-policy bits of the buffer instruction (gfx94x/gfx950: bit0 sc0, bit1 nt, bit4 sc1)
template <typename T, int V, int AUX = 0>
__device__ __forceinline__ Pack<T, V> buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
    static_assert(sizeof(Pack<T, V>) == 16, "16-byte accesses");
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUX);
    Pack<T, V> p;
    __builtin_memcpy(&p, &raw, 16);
    return p;
}
template <typename T, int V, int AUX = 0>
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t voff, const Pack<T, V> &p) {
    u32x4 raw;
    __builtin_memcpy(&raw, &p, 16);
    __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, 0, AUX);
}

// Matrix layouts: element (i, k) of a matrix with column stride ld and tile stride ts lives at
//   (i / R) * ts + (i % R) + k * ld.
// The caller's column-major matrices are (ld, ts = R): a tile is K separate 256-byte segments.  The
// library's own deflated copy of X (the NIPALS work buffer, never seen by the caller) is stored
// ROW-TILE-MAJOR, (ld = R, ts = R*K): the R x K tile is one contiguous R*K*s-byte block, column-major
// inside.  Same kernel, same arithmetic; only the two strides differ.  Why: a read+write sweep that
// touches memory in 256-byte pieces stops at ~5.0 TB/s on MI355X, in >= 4 KB contiguous pieces it
// reaches 5.9-6.3 TB/s (profiles/r1/rw_probe.txt) -- reads alone do not care (5.9 TB/s either way).
//
// Addressing: every access of a tile is (wave-uniform descriptor for the column group) +
// (per-lane 32-bit byte offset that never changes): the descriptor base X + tile*R + j*CG*ld
// lives in SGPRs, the lane offset (rp*V + cg*ld)*s in ONE VGPR, so the CPT loads in flight cost
// no address registers.  Lanes whose rows lie beyond N use an offset past num_records.
// X is streamed exactly once per pass and is far larger than the 256 MiB Infinity Cache: its
// loads and stores carry the nt (streaming) policy so they do not evict the small reused vectors
// (scores, partials).  Measured on config 3: read-only pass 0.80 -> 0.71 ms (6.05 TB/s),
// read+write pass 1.81 -> 1.70 ms (profiles/r1/tune_fused_cache_policy.txt).
constexpr int AUX_NT = 2;

// RDST: the destination is stored in tiles of rdst rows, rdst dividing R (the first deflation of a fit whose working
// copy uses shorter tiles than the R rows read at a time from the caller's column-major X); otherwise rdst is ignored.
template <typename T, int V, int R, int NT, int CPT, bool DEFL, int LDAUX = AUX_NT, int STAUX = AUX_NT, bool RDST = false>
__global__ __launch_bounds__(NT, (NT / 256) * (CPT <= 16 ? 2 : 1)) void fused_pass_kernel(
    const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd, i64 N, int K,  // dst may alias X (in-place deflation)
    const double *__restrict__ v, const T *__restrict__ tprev, const double *__restrict__ pprev,
    T *__restrict__ tout, double *__restrict__ part, double *__restrict__ sspart, int rdst) {
    constexpr int RP = R / V;    // lanes along the rows of a tile
    constexpr int CG = NT / RP;  // column groups
    constexpr int NW = NT / WAVE;
    static_assert(RP <= WAVE && WAVE % RP == 0 && NT % RP == 0, "tile shape");
    // LDS: the operand vectors v [CG*CPT] and, when DEFL, p_prev [CG*CPT], the score exchange, the block-sum
    // scratch.  One static block with a fixed layout (p_prev first): the instruction schedule of the headline
    // kernel turned out to depend on these addresses (1.3 % slower with the vectors after the exchange buffer or
    // in dynamic LDS).  Only the 128-column-group shape, whose vectors exceed a static allocation, takes them
    // from dynamic LDS.
    constexpr bool DYN = (size_t)2 * CG * CPT * sizeof(double) > 48 * 1024;
    struct alignas(16) Lds {
        alignas(16) double ps[(DEFL && !DYN) ? CG * CPT : 2];
        alignas(16) double vs[DYN ? 2 : CG * CPT];
        alignas(16) double tred[2][NW][R];
        alignas(16) double sred[NW];
    };
    __shared__ Lds lds;
    extern __shared__ double fused_dyn[];
    double *vs = DYN ? fused_dyn : lds.vs, *ps = DYN ? fused_dyn + CG * CPT : lds.ps;
    auto &tred = lds.tred;
    double *sred = lds.sred;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = tid % RP, cg = tid / RP;
    for (int k = tid; k < CG * CPT; k += NT) {
        vs[k] = (k < K) ? v[k] : 0.0;
        if (DEFL) ps[k] = (k < K) ? pprev[k] : 0.0;
    }
    __syncthreads();

    double pacc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) pacc[j] = 0.0;
    double ss = 0.0;
    int buf = 0;
    const uint32_t xoff = (uint32_t)(((i64)rp * V + (i64)cg * ldx) * (i64)sizeof(T));
    uint32_t doff = DEFL ? (uint32_t)(((i64)rp * V + (i64)cg * ldd) * (i64)sizeof(T)) : 0u;
    i64 dtile = tsd;  // destination elements per source tile
    if constexpr (RDST && DEFL) {
        // the lane's rows rp*V.. fall into sub-tile (rp*V)/rdst of the R/rdst destination tiles a source tile covers
        doff = (uint32_t)(((i64)((rp * V) / rdst) * tsd + (rp * V) % rdst + (i64)cg * ldd) * (i64)sizeof(T));
        dtile = (R / rdst) * tsd;
    }
    constexpr uint32_t OOR = 0x80000000u;  // beyond every num_records the launcher allows

    for (i64 tile = blockIdx.x; tile * R < N; tile += gridDim.x, buf ^= 1) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);  // N % V == 0 (launcher): a pack is all-valid or all-invalid
        const uint32_t xo = rowok ? xoff : OOR, dof = rowok ? doff : OOR;
        // LDS operands (v, p_prev) are re-read every tile: an index the compiler cannot prove
        // loop-invariant keeps 2*CPT fp64 values out of the register file
        int cgz = cg;
        asm volatile("" : "+v"(cgz));
        Pack<T, V> x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cols = min(CG, K - CG * j);  // columns of this group that exist (may be <= 0)
            const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldx * (i64)sizeof(T)) : 0u;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<T *>(X + tile * tsx + (i64)j * CG * ldx), (short)0, (int)nrec, BUF_WORD3);
            x[j] = buf_ld<T, V, LDAUX>(rs, xo);
            __builtin_amdgcn_sched_barrier(0);  // build one descriptor, issue its load, repeat
        }
        if (DEFL) {
            double tp[V];
            if (rowok) {
                const Pack<T, V> tpk = ld_pack<T, V>(tprev + i0);
#pragma unroll
                for (int e = 0; e < V; ++e) tp[e] = -(double)tpk.v[e];
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) tp[e] = 0.0;
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const double pk = ps[cgz + CG * j];
#pragma unroll
                for (int e = 0; e < V; ++e) x[j].v[e] = (T)fma(tp[e], pk, (double)x[j].v[e]);
                if constexpr (RDST) {  // lane offsets span several destination tiles: columns >= K masked per lane
                    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                        dst + tile * dtile + (i64)j * CG * ldd, (short)0, 0x7fffffff, BUF_WORD3);
                    buf_st<T, V, STAUX>(rd, (cg + CG * j < K) ? dof : OOR, x[j]);
                } else {
                    const int cols = min(CG, K - CG * j);
                    const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldd * (i64)sizeof(T)) : 0u;
                    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                        dst + tile * tsd + (i64)j * CG * ldd, (short)0, (int)nrec, BUF_WORD3);
                    buf_st<T, V, STAUX>(rd, dof, x[j]);
                }
            }
        }
        // score: partial over this lane's columns, then over the lanes / waves sharing the rows
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
            t[e] = (double)(T)s;  // the score as stored
        }
        if (cg == 0 && rowok) {
            Pack<T, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) o.v[e] = (T)t[e];
            st_pack<T, V>(tout + i0, o);
#pragma unroll
            for (int e = 0; e < V; ++e) ss = fma(t[e], t[e], ss);
        }
        // loading: p_raw[k] += sum over the lane's rows of x[i,k] * t[i]   (rows >= N hold x = 0)
        // fp32 storage: the tile is converted to fp64 again here -- the empty asm hides the stored values from
        // common-subexpression elimination, which would otherwise keep the fp64 copies of the whole tile made for
        // the score alive across the barrier (2x the registers of the tile: 135-217 spilled VGPRs, 2.3 TB/s)
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

    // epilogue: sum over the RP lanes that share a column group (low lane bits), then one lane
    // per column group writes this workgroup's partial row
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const double s = xor_range_sum<1, RP>(pacc[j]);
        const int k = cg + CG * j;
        if (rp == 0 && k < K) part[(i64)blockIdx.x * K + k] = s;
    }
    ss = block_sum<NW>(ss, sred);
    if (tid == 0) sspart[blockIdx.x] = ss;
}

// Semi-fused sweep for matrices too wide for the resident tile (K > 32 column groups' worth):
//   X' = X - t_prev p_prev^T (written),  t = X' w,  t^T t partials      -- the loading p = X'^T t then
// takes one more READ of X' (xty_kernel): 3 N K s of traffic per component instead of 4.
// Same tile access pattern as the fused pass; the column groups of a tile are streamed CPT at
// a time (nothing stays resident), the per-lane partial scores are combined once per tile.
// Dynamic LDS: 2*K doubles (w and p_prev).
template <typename T, int V, int R, int NT, int CPT>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void deflate_score_kernel(
    const T *src, i64 lds_, i64 tss, T *dst, i64 ldd, i64 tsd, int rdst, i64 N, int K, const T *__restrict__ tprev,
    const double *__restrict__ pprev, const double *__restrict__ w, T *__restrict__ tout,
    double *__restrict__ sspart) {
    constexpr int RP = R / V, CG = NT / RP, NW = NT / WAVE;
    extern __shared__ double dyn[];  // [2K]: w, p_prev
    __shared__ double tred[2][NW][R];
    __shared__ double sred[NW];
    double *ws = dyn, *ps = dyn + K;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int rp = tid % RP, cg = tid / RP;
    for (int k = tid; k < K; k += NT) {
        ws[k] = w[k];
        ps[k] = pprev[k];
    }
    __syncthreads();
    const uint32_t soff = (uint32_t)(((i64)rp * V + (i64)cg * lds_) * (i64)sizeof(T));
    // destination tiles may be shorter than the R rows read at a time (rdst divides R; rdst == R otherwise): the
    // lane's rows rp*V.. fall into sub-tile (rp*V)/rdst of the R/rdst destination tiles this source tile covers
    const int dsub = (rp * V) / rdst, dwithin = (rp * V) % rdst, dtiles = R / rdst;
    const uint32_t doff = (uint32_t)(((i64)dsub * tsd + dwithin + (i64)cg * ldd) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const int ngroups = (K + CG - 1) / CG;
    double ss = 0.0;
    int buf = 0;
    for (i64 tile = blockIdx.x; tile * R < N; tile += gridDim.x, buf ^= 1) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);
        const uint32_t so = rowok ? soff : OOR, dof = rowok ? doff : OOR;
        double tp[V], tacc[V];
        if (rowok) {
            const Pack<T, V> tpk = ld_pack<T, V>(tprev + i0);
#pragma unroll
            for (int e = 0; e < V; ++e) tp[e] = -(double)tpk.v[e];
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) tp[e] = 0.0;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) tacc[e] = 0.0;
        for (int g0 = 0; g0 < ngroups; g0 += CPT) {
            Pack<T, V> x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cols = min(CG, K - CG * (g0 + j));
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * lds_ * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(src + tile * tss + (i64)(g0 + j) * CG * lds_), (short)0, (int)nrec, BUF_WORD3);
                x[j] = buf_ld<T, V, AUX_NT>(rs, so);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int k = cg + CG * (g0 + j);
                const double pk = (k < K) ? ps[k] : 0.0, wk = (k < K) ? ws[k] : 0.0;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    x[j].v[e] = (T)fma(tp[e], pk, (double)x[j].v[e]);
                    tacc[e] = fma((double)x[j].v[e], wk, tacc[e]);
                }
                // columns >= K are masked per lane (the lane offset may span several destination tiles, so the
                // descriptor's range check cannot do it here)
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                    dst + tile * dtiles * tsd + (i64)(g0 + j) * CG * ldd, (short)0, 0x7fffffff, BUF_WORD3);
                buf_st<T, V, AUX_NT>(rd, (k < K) ? dof : OOR, x[j]);
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
            tacc[e] = xor_range_sum<RP, WAVE>(tacc[e]);
        if (lane < RP)
#pragma unroll
            for (int e = 0; e < V; ++e) tred[buf][wv][rp * V + e] = tacc[e];
        __syncthreads();
        if (cg == 0 && rowok) {
            Pack<T, V> o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < NW; ++q) s += tred[buf][q][rp * V + e];
                o.v[e] = (T)s;
                const double ts = (double)o.v[e];
                ss = fma(ts, ts, ss);
            }
            st_pack<T, V>(tout + i0, o);
        }
    }
    ss = block_sum<NW>(ss, sred);
    if (tid == 0) sspart[blockIdx.x] = ss;
}

// rc as launch_fused_pass; *nss = number of t^T t partials written
template <typename T>
int launch_deflate_score(hipStream_t stream, int num_cu, const T *src, i64 lds_, i64 tss, T *dst, i64 ldd, i64 tsd,
                         int rdst, i64 N, int K, const T *tprev, const double *pprev, const double *w, T *tout,
                         double *sspart, int max_rows, int *nss) {
    constexpr int V = 16 / sizeof(T);
    constexpr int R = 256 / sizeof(T), NT = 512, CPT = 8;
    constexpr int CG = NT / (R / V);
    auto al = [](const void *q, i64 ld) { return ((uintptr_t)q % 16 == 0) && (ld % V == 0); };
    if (!al(src, lds_) || !al(dst, ldd) || !al(tprev, V) || !al(tout, V) || N < 1 || N % V != 0) return 1;
    if (tss % V != 0 || tsd % V != 0 || rdst < V || R % rdst != 0) return 1;
    if (((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if ((i64)CG * lds_ * (i64)sizeof(T) >= (1ll << 31) || (i64)CG * ldd * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const size_t dyn = (size_t)K * 16;  // w and p_prev
    if (dyn > 72 * 1024) return 1;      // two workgroups per CU must fit the 160 KiB LDS
    if (dyn > 48 * 1024 &&
        !raise_dynamic_lds(reinterpret_cast<const void *>(&deflate_score_kernel<T, V, R, NT, CPT>), 72 * 1024))
        return 1;
    const i64 ntiles = (N + R - 1) / R;
    const i64 grid = std::min<i64>(std::min<i64>(ntiles, 2 * (i64)num_cu), max_rows);
    hipLaunchKernelGGL((deflate_score_kernel<T, V, R, NT, CPT>), dim3((unsigned)grid), dim3(NT), (size_t)K * 16,
                       stream, src, lds_, tss, dst, ldd, tsd, rdst, N, K, tprev, pprev, w, tout, sspart);
    *nss = (int)grid;
    return 0;
}

// Copy of a column-major matrix into row-tile-major storage with tiles of rdst rows (rdst divides R): read in the
// 256-byte-segment pattern, written as contiguous tile pieces.  Used once per fit by the KERNEL plan on wide
// matrices (1024 < K <= 4096), whose read-only passes then run fused on the short tiles.
template <typename T, int V, int R, int NT, int CPT>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void retile_kernel(const T *src, i64 lds_, T *dst, i64 ldd, i64 tsd,
                                                                    int rdst, i64 N, int K) {
    constexpr int RP = R / V, CG = NT / RP;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const uint32_t soff = (uint32_t)(((i64)rp * V + (i64)cg * lds_) * (i64)sizeof(T));
    const int dsub = (rp * V) / rdst, dwithin = (rp * V) % rdst, dtiles = R / rdst;
    const uint32_t doff = (uint32_t)(((i64)dsub * tsd + dwithin + (i64)cg * ldd) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const int ngroups = (K + CG - 1) / CG;
    for (i64 tile = blockIdx.x; tile * R < N; tile += gridDim.x) {
        const bool rowok = (tile * R + (i64)rp * V < N);
        const uint32_t so = rowok ? soff : OOR, dof = rowok ? doff : OOR;
        for (int g0 = 0; g0 < ngroups; g0 += CPT) {
            Pack<T, V> x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cols = min(CG, K - CG * (g0 + j));
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * lds_ * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(src + tile * R + (i64)(g0 + j) * CG * lds_), (short)0, (int)nrec, BUF_WORD3);
                x[j] = buf_ld<T, V, AUX_NT>(rs, so);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int k = cg + CG * (g0 + j);
                const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                    dst + tile * dtiles * tsd + (i64)(g0 + j) * CG * ldd, (short)0, 0x7fffffff, BUF_WORD3);
                buf_st<T, V, AUX_NT>(rd, (k < K) ? dof : OOR, x[j]);
            }
        }
    }
}

// rc as launch_fused_pass
template <typename T>
int launch_retile(hipStream_t stream, int num_cu, const T *src, i64 lds_, T *dst, i64 ldd, i64 tsd, int rdst, i64 N,
                  int K) {
    constexpr int V = 16 / sizeof(T);
    constexpr int R = 256 / sizeof(T), NT = 512, CPT = 8;
    constexpr int CG = NT / (R / V);
    auto al = [](const void *q, i64 ld) { return ((uintptr_t)q % 16 == 0) && (ld % V == 0); };
    if (!al(src, lds_) || !al(dst, ldd) || tsd % V != 0 || rdst < V || R % rdst != 0 || N < 1 || N % V != 0) return 1;
    if ((i64)CG * lds_ * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if (((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const i64 ntiles = (N + R - 1) / R;
    const i64 grid = std::min<i64>(ntiles, 2 * (i64)num_cu);
    hipLaunchKernelGGL((retile_kernel<T, V, R, NT, CPT>), dim3((unsigned)grid), dim3(NT), 0, stream, src, lds_, dst, ldd,
                       tsd, rdst, N, K);
    return 0;
}

// Will launch_deflate_score accept every deflating pass of a fit (decided once per fit, like fused_pass_covers)?
template <typename T>
bool deflate_score_covers(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    constexpr int V = 16 / sizeof(T);
    constexpr int CG = 512 / ((256 / (int)sizeof(T)) / V);
    auto al = [](const void *q, i64 ld) { return ((uintptr_t)q % 16 == 0) && (ld % V == 0); };
    return al(X, ldx) && al(Tm, ldt) && N >= 1 && N % V == 0 && (size_t)K * 16 <= 72 * 1024 &&
           (i64)CG * ldx * (i64)sizeof(T) < (1ll << 31);
}

// Loading partials p_raw = X^T t for a matrix in (ld, ts) tile addressing -- the row-tile-major work buffer of
// the semi-fused plan:   part[blockIdx.x*K + k] = sum over the workgroup's tiles and rows of X[i,k] * t[i].
// grid = (row chunks of tpw tiles, column blocks of CG*CPT columns); one-shot workgroups, 16-byte accesses,
// a tile's column block is one contiguous CG*CPT*256-byte piece.
template <typename T, int V, int R, int NT, int CPT>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void xty_tiled_kernel(const T *X, i64 ldx, i64 tsx, i64 N, int K,
                                                                       const T *__restrict__ t,
                                                                       double *__restrict__ part, int tpw) {
    constexpr int RP = R / V, CG = NT / RP;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const int g0 = blockIdx.y * CPT;  // first column group of this block
    const i64 ntiles = (N + R - 1) / R;
    const i64 tile0 = (i64)blockIdx.x * tpw, tile1 = min(ntiles, tile0 + (i64)tpw);
    const uint32_t xoff = (uint32_t)(((i64)rp * V + (i64)cg * ldx) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    double pacc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) pacc[j] = 0.0;
    for (i64 tile = tile0; tile < tile1; ++tile) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);
        const uint32_t xo = rowok ? xoff : OOR;
        Pack<T, V> x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int cols = min(CG, K - CG * (g0 + j));
            const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldx * (i64)sizeof(T)) : 0u;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<T *>(X + tile * tsx + (i64)(g0 + j) * CG * ldx), (short)0, (int)nrec, BUF_WORD3);
            x[j] = buf_ld<T, V, AUX_NT>(rs, xo);
            __builtin_amdgcn_sched_barrier(0);
        }
        double tv[V];
        if (rowok) {
            const Pack<T, V> tpk = ld_pack<T, V>(t + i0);
#pragma unroll
            for (int e = 0; e < V; ++e) tv[e] = (double)tpk.v[e];
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) tv[e] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) pacc[j] = fma((double)x[j].v[e], tv[e], pacc[j]);
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const double s = xor_range_sum<1, RP>(pacc[j]);
        const int k = cg + CG * (g0 + j);
        if (rp == 0 && k < K) part[(i64)blockIdx.x * K + k] = s;
    }
}

// rc as launch_fused_pass; *nb = partial rows written (<= max_rows)
template <typename T, int CGX = 32>
int launch_xty_tiled(hipStream_t stream, int num_cu, const T *X, i64 ldx, i64 tsx, i64 N, int K, const T *t,
                     double *part, int max_rows, int *nb) {
    constexpr int V = 16 / sizeof(T);
    constexpr int R = (512 / CGX) * V, NT = 512, CPT = 16;  // = tile_rows<T, CGX>()
    constexpr int CG = NT / (R / V);
    auto al = [](const void *q, i64 ld) { return ((uintptr_t)q % 16 == 0) && (ld % V == 0); };
    if (!al(X, ldx) || !al(t, V) || tsx % V != 0 || N < 1 || N % V != 0 || max_rows < 1) return 1;
    if ((i64)CG * ldx * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const i64 ntiles = (N + R - 1) / R;
    const int nkb = (K + CG * CPT - 1) / (CG * CPT);
    // ~8 workgroups per CU in total, at most max_rows row chunks
    const i64 want = std::max<i64>(1, std::min<i64>((8 * (i64)num_cu + nkb - 1) / nkb, max_rows));
    const i64 tpw = (ntiles + want - 1) / want;
    const i64 gx = (ntiles + tpw - 1) / tpw;
    hipLaunchKernelGGL((xty_tiled_kernel<T, V, R, NT, CPT>), dim3((unsigned)gx, (unsigned)nkb), dim3(NT), 0, stream, X,
                       ldx, tsx, N, K, t, part, (int)tpw);
    *nb = (int)gx;
    return 0;
}

// Rows per tile of the tile-resident kernels.  CGX = column groups of a 512-thread workgroup:
//   32  -> 16 lanes along the rows: 256-byte column segments (32 fp64 / 64 fp32 rows), K <= 1024 -- the shape
//          that can also read the caller's column-major matrices at full rate;
//   64, 128 -> 8 / 4 lanes along the rows (16 / 8 fp64, 32 / 16 fp32 rows), K <= 2048 / 4096: only for the
//          row-tile-major working copy, where a tile is contiguous whatever its height.
template <typename T, int CGX = 32>
constexpr int tile_rows() { return (512 / CGX) * (16 / (int)sizeof(T)); }

// Will launch_fused_pass accept every pass of a fit on (X, ldx) with score columns Tm + a*ldt?  (Decided once
// per fit: the work buffer's layout depends on it.)
template <typename T>
bool fused_pass_covers(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    constexpr int V = 16 / sizeof(T);
    constexpr int CG = 32;
    auto al = [](const void *p, i64 ld) { return ((uintptr_t)p % 16 == 0) && (ld % V == 0); };
    return al(X, ldx) && al(Tm, ldt) && K <= CG * 32 && N >= 1 && N % V == 0 &&
           (i64)CG * ldx * (i64)sizeof(T) < (1ll << 31);
}

// rc: 0 = launched, 1 = shape/alignment not covered (caller falls back to the one-product
// kernels), <0 = launch error.  grid_hint: 0 = auto.  (ldx, tsx) / (ldd, tsd): column and tile strides.
// rdst > 0 (with CGX = 32 and a deflating pass): the destination uses tiles of rdst rows.
template <typename T, int CGX = 32>
int launch_fused_pass(hipStream_t stream, int num_cu, const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd,
                      i64 N, int K, const double *v, const T *tprev, const double *pprev, T *tout,
                      double *part, int max_rows, double *sspart, int *nb, int *nss, int grid_hint, int rdst = 0) {
    constexpr int V = 16 / sizeof(T);
    constexpr int R = tile_rows<T, CGX>(), NT = 512;
    constexpr int CG = NT / (R / V);
    static_assert(CG == CGX, "tile shape");
    const bool defl = (tprev != nullptr);
    auto al = [](const void *p, i64 ld) { return ((uintptr_t)p % 16 == 0) && (ld % V == 0); };
    if (!al(X, ldx) || !al(tout, V) || (defl && (!al(dst, ldd) || !al(tprev, V)))) return 1;
    if (tsx % V != 0 || (defl && tsd % V != 0)) return 1;
    if (K > CG * (CGX == 256 ? 16 : 32) || N < 1 || N % V != 0) return 1;
    // a column group's byte span (its num_records, and every lane offset) must stay below 2^31
    if ((i64)CG * ldx * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if (defl && (i64)CG * ldd * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if (rdst > 0 && (CGX != 32 || !defl || rdst < V || R % rdst != 0 ||
                     ((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)))
        return 1;
    const i64 ntiles = (N + R - 1) / R;
    // Workgroups per CU.  Read-only passes: two (5 % faster than one on the caller's column-major X).  Read+write
    // passes on the tiled copy: ONE (2.7 % faster than two at 16 columns per lane, 4 % at 4, equal at 8 -- less
    // in flight is better for the read/write mix, tools/fused_grid_sweep.py); 32 columns per lane (256 VGPRs)
    // never fit two.
    const int per_cu = (K <= CG * 16 && !defl) ? 2 : 1;
    i64 grid = grid_hint > 0 ? grid_hint : per_cu * (i64)num_cu;
    grid = std::min<i64>(std::min<i64>(grid, ntiles), max_rows);
    if (grid < 1) return 1;
    const dim3 g((unsigned)grid), b(NT);
#define FUSED_CASE(CPT_)                                                                                      \
    do {                                                                                                      \
        const size_t dyn = ((size_t)2 * CG * CPT_ * sizeof(double) > 48 * 1024) ? (size_t)2 * CG * CPT_ * sizeof(double) : 0; \
        if (dyn > 48 * 1024) {                                                                                \
            const void *fn = defl ? reinterpret_cast<const void *>(&fused_pass_kernel<T, V, R, NT, CPT_, true>) \
                                  : reinterpret_cast<const void *>(&fused_pass_kernel<T, V, R, NT, CPT_, false>); \
            if (!raise_dynamic_lds(fn, (int)dyn)) return 1;                                                   \
        }                                                                                                     \
        if (defl)                                                                                             \
            hipLaunchKernelGGL((fused_pass_kernel<T, V, R, NT, CPT_, true>), g, b, dyn, stream, X, ldx, tsx,  \
                               dst, ldd, tsd, N, K, v, tprev, pprev, tout, part, sspart, 0);                  \
        else                                                                                                  \
            hipLaunchKernelGGL((fused_pass_kernel<T, V, R, NT, CPT_, false>), g, b, dyn, stream, X, ldx, tsx, \
                               dst, ldd, tsd, N, K, v, tprev, pprev, tout, part, sspart, 0);                  \
    } while (0)
    if constexpr (CGX == 32) {
        if (rdst > 0) {  // first deflation into shorter tiles: only the 32-columns-per-lane shape needs it
            if (K <= CG * 16) return 1;
            hipLaunchKernelGGL((fused_pass_kernel<T, V, R, NT, 32, true, AUX_NT, AUX_NT, true>), g, b, 0, stream, X, ldx, tsx,
                               dst, ldd, tsd, N, K, v, tprev, pprev, tout, part, sspart, rdst);
        } else if (K <= CG * 4) FUSED_CASE(4);
        else if (K <= CG * 8) FUSED_CASE(8);
        else if (K <= CG * 16) FUSED_CASE(16);
        else FUSED_CASE(32);
    } else if constexpr (CGX == 64 || CGX == 128) {
        if (K <= CG * 16) FUSED_CASE(16);
        else FUSED_CASE(32);
    } else {
        FUSED_CASE(16);  // 256 column groups x 2 row lanes: K <= 4096 at 16 columns per lane
    }
#undef FUSED_CASE
    *nb = (int)grid;
    *nss = (int)grid;
    return 0;
}

}  // namespace plsk
