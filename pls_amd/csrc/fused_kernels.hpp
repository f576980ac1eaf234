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
#include "exchange_kernels.hpp"  // XchgPeers: the push of a sharded fit's sums rides in the tail of the pass
#include "update_m1.hpp"         // the one-response component update: the last act of the tail

namespace plsk {

#ifdef PLS_HIP_TESTING
// testing/libpls_hip.so only: wall-clock stamps of every workgroup of a fused pass (start, tile loop done, exit) for
// tools/pass_stamps.py -- where a short pass spends its time (dispatch skew, imbalance at the end of the tile loop, the tail)
__device__ unsigned long long *g_pass_stamps = nullptr;
#define PLS_STAMP(slot)                                                                                 \
    do {                                                                                                \
        if (g_pass_stamps && threadIdx.x == 0) {                                                        \
            g_pass_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64();                            \
            if ((slot) == 0) {  /* where the workgroup runs: XCC_ID (hwreg 20), HW_ID (hwreg 4) */       \
                g_pass_stamps[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  \
                g_pass_stamps[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   \
            }                                                                                           \
        }                                                                                               \
    } while (0)
#else
#define PLS_STAMP(slot) do { } while (0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// gfx9-family raw buffer descriptor word 3 (DST_SEL xyzw, 32-bit data format); stride 0:
// offsets are plain byte offsets, range-checked against num_records -- an out-of-range load
// returns 0 and an out-of-range store is dropped, which is how the ragged edges are handled.
constexpr int BUF_WORD3 = 0x00020000;
constexpr int AUX_SC1 = 16;  // cache-policy bit sc1: served by the fabric, not by this XCD's L2 (what another XCD's sc1 stores wrote)

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
// the same with a wave-uniform byte offset beside the lane offset (the instruction's scalar offset operand): one
// descriptor serves every column group of a contiguous tile
template <typename T, int V, int AUX = 0>
__device__ __forceinline__ Pack<T, V> buf_ld_so(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff) {
    static_assert(sizeof(Pack<T, V>) == 16, "16-byte accesses");
    const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    Pack<T, V> p;
    __builtin_memcpy(&p, &raw, 16);
    return p;
}
template <typename T, int V, int AUX = 0>
__device__ __forceinline__ void buf_st_so(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff, const Pack<T, V> &p) {
    u32x4 raw;
    __builtin_memcpy(&raw, &p, 16);
    __builtin_amdgcn_raw_buffer_store_b128(raw, r, voff, soff, AUX);
}

// 16-byte access to a score column (t_prev in, t out): the caller's T may have any leading dimension, so its columns are
// only element-aligned; global memory accesses need dword alignment, nothing more.
template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld_pack_u(const T *p) {
    Pack<T, V> o;
    __builtin_memcpy(&o, p, sizeof(o));
    return o;
}
template <typename T, int V>
__device__ __forceinline__ void st_pack_u(T *p, const Pack<T, V> &x) {
    __builtin_memcpy(p, &x, sizeof(x));
}

// The previous score column (t_prev) is read through a buffer descriptor as well: ONE descriptor, num_records = the VALID
// rows (the launchers keep N s below 2^31), built before the tile loop; the lane's offset is its first row.  Raw buffer
// accesses are range-checked per dword (gfx9 family), so the 16-byte load of the pack that straddles row N reads zeros
// for the rows behind the end -- a sweep over the library's zero-padded tiled copy can therefore run to the next
// multiple of V rows with no code for the partial pack.  (The score STORE keeps its plain 16-byte form with an
// element-wise branch for that one pack: as a buffer store it cost the fp32 instantiations 34 spilled registers.)
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t score_rsrc(const T *col, i64 nvalid) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(col), (short)0, (int)(nvalid * (i64)sizeof(T)), BUF_WORD3);
}

// ---------------------------------------------------------------------------------------------------------------------
// Reduction of the per-workgroup partial rows INSIDE the launch that wrote them (instead of reduce_partials_kernel behind
// it), and -- in a row-sharded fit over the device-side exchange -- the push of the sums to the peers from the same tail:
// a sharded component is then two launches, pass -> update, instead of pass -> reduce -> exchange -> update.
//
// The rows [lo, hi) of slice s are those of reduce_partials_kernel (lo = nrows s / 8); every workgroup stores its row with
// sc1 stores, waits for them, and ONE lane adds 1 to the slice's arrival counter (agent scope).  The workgroup whose add
// came last -- told by the value the add returned, so the order of arrival is free -- acquires (agent scope) and sums the
// slice's rows in exactly the order of reduce_partials_kernel: four interleaved chains per column, row index ascending,
// (c0 + c1) + (c2 + c3); the bits do not depend on which workgroup does it or when.  Rows written BEFORE the launch
// (tail_rows_kernel's row of the last N % V rows: row nrows - 1) are complete by stream order and not counted.
// Sharded (npush > 0): the last of the RED_SLICES slice reducers adds the slices in index order and writes the vector into
// slot [seq & 1][rank] of every member's inbox, then publishes seq in the flags (exchange_kernels.hpp: xchg_push_kernel);
// the gather is the prologue of the component update (small_kernels.hpp).
// Counters: RED_SLICES + 1 words, zero before the first launch of a fit; every counter is reset by its last arriver.
// Every wait is an atomic's return value or a barrier: no spinning, nothing that needs other workgroups to be resident.
// ---------------------------------------------------------------------------------------------------------------------
// The one-response component update (update_m1.hpp) as the LAST act of the tail: the workgroup that summed the last slice
// holds [X^T t, t^T t] and runs p, q, the XY deflation, w and the r recurrence right there -- a component is ONE launch.
struct TailUpdate {
    double *XY = nullptr;  // nullptr: no update in the tail (the component update is a launch of its own)
    double *W = nullptr, *P = nullptr, *Q = nullptr, *R = nullptr, *vnext = nullptr;
    int A = 0, a = 0, nipals = 0;
};

// Unequal shares of the tiles for the workgroups of the fast / slow XCD classes (fused_pass_kernel, "weighted walk")
struct WalkWeights {
    int nfull = -1;     // rounds dealt to all workgroups; -1: plain cyclic walk
    unsigned mask = 0;  // bit c: the workgroups with blockIdx % 8 == c also share the tiles behind those rounds
};

struct SliceTail {
    unsigned *cnt = nullptr;  // nullptr: no tail (reduce_partials_kernel follows the launch)
    double *red = nullptr;    // RED_SLICES slices of K + 1 values
    int nrows = 0;            // partial rows in all: this launch's workgroups (+ 1 written before the launch)
    int npush = 0;            // members of the exchange to push to; 0: single rank or a reducer of the caller's
    unsigned long long seq = 0;
    XchgPeers peers;
    TailUpdate upd;
    XchgGather gx;            // sharded, with upd: the gather of this collective behind the push (n = 0: none)
};

constexpr int TAIL_MIN_WG = 32;  // fewer workgroups: a slice could be left without one (reduce_partials_kernel instead)
constexpr int TAIL_AUX_SYS = 17;  // sc0 sc1: system scope (the peers' writes into fine-grained memory)

// all NT threads of the workgroup call it after their stores of part / sspart have been waited for and a barrier;
// role: one int of LDS; sm: >= 4 doubles of LDS; scratch: >= 16 + A doubles of LDS and wl: >= K doubles of LDS (the update)
// Hand-off: every handed-off byte is stored sc1, waited for by the storing wave, and loaded sc1 behind the returned add
// and a barrier -- MI355X_MICROARCH.md "Valid forms", first row of its table (hipMalloc memory, ONE workgroup per CU, 8-byte
// accesses): the launchers use the tail for launches of one workgroup per CU only.  (With two per CU -- the read-only
// passes at 16 columns per lane -- a slice is 64 rows, its last workgroup reads 256 KB behind an agent-scope acquire, and
// the pass + tail lost 4.7 us per component to pass + reduce_partials_kernel at config 3: profiles/r4/tail_ab.txt.)
// The update itself, out of line (ONE copy for all instantiations of the pass; its arguments travel in registers): the reduced
// value j is 0.0 + src[j] + src[stride + j] + ... over n vectors -- the RED_SLICES slices this launch's slice reducers stored
// (sc1 loads), or, sharded, the members' vectors in this member's inbox in rank order (sys: sc0 sc1 loads; !ok: a wait that
// timed out -- NaN).  The same bits as red_sum over the sliced form the other routes present (0.0 + x0 = x0, and a sum that
// is -0.0 either way ends as +0.0).
__device__ __noinline__ void tail_update_m1(const double *src, i64 stride, int n, int sys, int ok, double *XY, double *W, double *P,
                                            double *Q, double *R, double *vnext, int K, int A, int a, int nipals, double *scratch,
                                            double *wl) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    static_assert(XCHG_MAX == 2 * RED_SLICES, "two batches of RED_SLICES loads");
    const int L = K + 1;
    update_m1<512, 1>(  // (K <= 1024: the launcher)
        [&](int j) -> double {
            double sum = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h * RED_SLICES >= n) break;  // (uniform)
                double x[RED_SLICES];
#pragma unroll
                for (int m = 0; m < RED_SLICES; ++m) {  // the vectors' values in flight together
                    const int mm = h * RED_SLICES + m;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<double *>(src + (i64)(mm < n ? mm : 0) * stride), (short)0, L * 8, BUF_WORD3);
                    const u32x2 raw = sys ? __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)j * 8u, 0, TAIL_AUX_SYS)
                                          : __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)j * 8u, 0, AUX_SC1);
                    __builtin_memcpy(&x[m], &raw, 8);
                }
#pragma unroll
                for (int m = 0; m < RED_SLICES; ++m)
                    if (h * RED_SLICES + m < n) sum += x[m];
            }
            return ok ? sum : __builtin_nan("");
        },
        XY, W, P, Q, R, vnext, K, A, a, nipals, 0, scratch + UPD1_VWAVES, scratch, wl);
}

template <int NT>
__device__ __forceinline__ void slice_tail(const SliceTail &st, const double *part, const double *sspart, int K, int *role,
                                        double *sm, double *scratch, double *wl) {
    static_assert(NT >= WG, "the t^T t sum takes the first four waves");
    const int tid = threadIdx.x, nb = st.nrows, nwg = gridDim.x, b = blockIdx.x;
    // (Round 5 measured the tail on SMALL grids too -- one arrival counter, the last workgroup sums all slices, update behind it:
    // the last workgroup then spends 9 us in the tail of a 3 us pass (every hand-off is a round trip to memory on a chip that
    // idles at a low clock) and mid-size fits LOSE to the three launches, 300 x 400: 132 against 94 us for five components,
    // 3,000 x 100: 129 against 92 -- profiles/r5/small_vs_r4_tail_on_small_grids.txt.  The tail stays with grids of
    // TAIL_MIN_WG workgroups or more.)
    constexpr int NSL = RED_SLICES;
    const int sl = (int)(((i64)(b + 1) * NSL + nb - 1) / nb) - 1;  // lo(sl) <= b < hi(sl)
    if (tid == 0) {
        const int lo_ = (int)((i64)nb * sl / NSL), hi_ = (int)((i64)nb * (sl + 1) / NSL);
        const unsigned mine = (unsigned)(min(hi_, nwg) - lo_);  // workgroups of this launch behind this counter
        const unsigned ticket = __hip_atomic_fetch_add(st.cnt + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (ticket + 1u == mine);
        if (last) {
            __hip_atomic_store(st.cnt + sl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (nobody else comes)
        }
        *role = last;
    }
    __syncthreads();
    if (!*role) return;
    const i64 LP = K + 1;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    {
        const int s8 = sl;
        const int lo = (int)((i64)nb * s8 / RED_SLICES), hi = (int)((i64)nb * (s8 + 1) / RED_SLICES);
        double ssl = 0.0;  // t^T t of the slice, this thread's strided share: loaded ahead of the rows
        {
            const __amdgpu_buffer_rsrc_t rq =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(sspart), (short)0, nb * 8, BUF_WORD3);
            if (tid < WG)
                for (int r = lo + tid; r < hi; r += WG) {
                    const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rq, (uint32_t)r * 8u, 0, AUX_SC1);
                    double d;
                    __builtin_memcpy(&d, &raw, 8);
                    ssl += d;
                }
        }
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(part + (i64)lo * K), (short)0,
                                                                                (int)((i64)(hi - lo) * K * 8), BUF_WORD3);
            auto ld = [&](int row, int j) -> double {  // (rows beyond hi - lo: out of range, zero)
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, (uint32_t)(((i64)row * K + j) * 8), 0, AUX_SC1);
                double d;
                __builtin_memcpy(&d, &raw, 8);
                return d;
            };
            const int nr = hi - lo;
            for (int j = tid; j < K; j += NT) {
                double c[4] = {0.0, 0.0, 0.0, 0.0};
                int r = 0;
                for (; r + 16 <= nr; r += 16) {  // 16 loads in flight per lane; chain q takes the rows lo + q, lo + q + 4, ...
                    double x[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = ld(r + u, j);
#pragma unroll
                    for (int u = 0; u < 16; ++u) c[u & 3] += x[u];
                }
                for (; r < nr; r += 4) {
                    double x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = (r + u < nr) ? ld(r + u, j) : 0.0;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (r + u < nr) c[u] += x[u];
                }
                st_agent(st.red + (i64)s8 * LP + j, (c[0] + c[1]) + (c[2] + c[3]));
            }
        }
        {  // t^T t of the slice: reduce_partials_kernel's sum -- 256 strided sums, wave sums, waves 0..3 in order
            double sw = wave_sum(ssl);
            __syncthreads();
            if ((tid & 63) == 0 && tid < WG) sm[tid >> 6] = sw;
            __syncthreads();
            if (tid == 0) st_agent(st.red + (i64)s8 * LP + K, ((sm[0] + sm[1]) + sm[2]) + sm[3]);
        }
    }
    if (st.npush <= 0 && !st.upd.XY) return;
    // ---- the last of the RED_SLICES slice reducers: the sums of all slices -> every member's inbox (sharded), the component
    // update (one response)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(st.cnt + RED_SLICES, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (ticket + 1u == (unsigned)NSL);
            if (last) {
                __hip_atomic_store(st.cnt + RED_SLICES, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *role = last;
        }
        __syncthreads();
        if (!*role) return;
    }
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(st.red, (short)0, (int)((i64)RED_SLICES * LP * 8), BUF_WORD3);
    if (st.npush > 0) {
        for (int j = tid; j <= K; j += NT) {
            double x[RED_SLICES];
#pragma unroll
            for (int i = 0; i < RED_SLICES; ++i) {
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rr, (uint32_t)(((i64)i * LP + j) * 8), 0, AUX_SC1);
                __builtin_memcpy(&x[i], &raw, 8);
            }
            double sum = x[0];
#pragma unroll
            for (int i = 1; i < RED_SLICES; ++i) sum += x[i];  // (the order of xchg_push_kernel)
            for (int d = 0; d < st.npush; ++d) __hip_atomic_store(st.peers.slot[d] + j, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // the inboxes are fine-grained (uncached) memory: a store that has been acknowledged is visible to its device; every
        // wave waits for its own, the flags go out behind the barrier (release: ordered behind this thread's stores too)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < st.npush) __hip_atomic_store(st.peers.flag[tid], st.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!st.upd.XY) return;
    const TailUpdate &u = st.upd;
    if (st.gx.n > 0) {
        // sharded: the GATHER of the collective just pushed (component_update_gather_kernel's prologue) -- wait until every
        // member's flag shows it, then the members' vectors in rank order.  Only this one workgroup is left of the launch:
        // a peer that shares the GPU (the rehearsal on one device) finds the other CUs free, so the wait needs no
        // co-residency; a wait beyond the time limit raises the status words and poisons the sums, as xchg_gather_kernel does.
        const XchgGather &gx = st.gx;
        if (tid == 0) *role = (*gx.status == 0);  // an earlier wait of this member timed out: do not wait again
        __syncthreads();
        if (tid < gx.n && *role) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(gx.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < gx.seq) {
                if (wall_clock64() - t0 > gx.limit) {
                    *role = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the vectors behind the flags
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int ok = *role;
        if (!ok && tid == 0) {
            *gx.status = 1;
            __hip_atomic_store(gx.host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        tail_update_m1(gx.inbox, gx.cap, gx.n, 1, ok, u.XY, u.W, u.P, u.Q, u.R, u.vnext, K, u.A, u.a, u.nipals, scratch, wl);
        return;
    }
    tail_update_m1(st.red, LP, RED_SLICES, 0, 1, u.XY, u.W, u.P, u.Q, u.R, u.vnext, K, u.A, u.a, u.nipals, scratch, wl);
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

// Which tiles a workgroup visits.  Cyclic: blockIdx.x, blockIdx.x + gridDim.x, ...  XCD-contiguous (EDGE = 2, below):
// workgroup b runs on XCD b % 8 under round-robin dispatch; every XCD takes one contiguous eighth of the tiles and its
// workgroups walk it cyclically (grids that are a multiple of 8; cyclic otherwise).
template <bool XCD, int R>
struct TileWalk {
    __device__ __forceinline__ explicit TileWalk(i64) {}
    __device__ __forceinline__ unsigned first() const { return blockIdx.x; }
    __device__ __forceinline__ unsigned step() const { return gridDim.x; }
    __device__ __forceinline__ i64 nlim(i64 N) const { return N; }
};
template <int R>
struct TileWalk<true, R> {
    i64 first_, step_, nlim_;
    __device__ __forceinline__ explicit TileWalk(i64 N) : first_(blockIdx.x), step_(gridDim.x), nlim_(N) {
        if ((gridDim.x & 7) == 0) {
            const i64 per = ((N + R - 1) / R + 7) / 8;
            first_ = (i64)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
            step_ = gridDim.x >> 3;
            nlim_ = min(N, (i64)((blockIdx.x & 7) + 1) * per * R);
        }
    }
    __device__ __forceinline__ i64 first() const { return first_; }
    __device__ __forceinline__ i64 step() const { return step_; }
    __device__ __forceinline__ i64 nlim(i64) const { return nlim_; }
};

// RDST: the destination is stored in tiles of rdst rows, rdst dividing R (the first deflation of a fit whose working
// copy uses shorter tiles than the R rows read at a time from the caller's column-major X); otherwise rdst is ignored.
// N is a multiple of V here: the last N % V rows of a matrix (fewer than one row pack) are the tail kernel's
// (tail_rows_kernel below), so that no instantiation carries code for partial packs.
// EDGE -- the layouts of the caller's matrix that the plain addressing does not reach (src/pls.cpp:419-421 takes any
// Eigen map):
//   1: leading dimensions beyond 2^31 / (CG s) bytes: the descriptor of a load is built per WAVE (its base includes the
//      wave's first column group), so a lane offset spans the WAVE / RP column groups of one wave, not all CG;
//   2: columns that are not 16-byte aligned (odd ld, an X pointer at 8 mod 16).  Buffer loads only need dword alignment,
//      but the 256-byte segment of a tile then shares a 128-byte line with the tile above and the tile below it, and every
//      shared line is fetched twice: 4.7 instead of 6.7 TB/s however the loads are formed (two 8-byte loads, aligned
//      supersets + lane shuffles, one row per lane: pls_amd/csrc/tune/unaligned_probe.hip).  Two changes bring most of it
//      back (5.9 TB/s in the probe): the tiles are dealt out XCD-contiguously -- workgroup b runs on XCD b % 8 under
//      round-robin dispatch, every XCD takes one contiguous eighth of the tiles and its workgroups walk it cyclically, so
//      the two tiles that share a line are read at about the same time behind the SAME L2 -- and the loads drop the
//      streaming (nt) policy, so that the line is still there.  Per-wave descriptors as in 1.
// TILED: X (and dst, when DEFL) are the library's row-tile-major copy in THIS tile shape -- ld = R, ts = R K: a tile is
// one contiguous R K s-byte block, so ONE descriptor per tile (num_records = K R s: columns >= K are out of range) serves
// all CPT loads and all CPT stores, the column group chosen by the instruction's scalar offset j CG R s.  With a
// descriptor per column group (the caller's column-major matrix needs those: a column group spans CG ld s bytes) the
// compiler carried 2 x 16 descriptors across the tile loop as running pointers -- 141 spilled SGPRs and 11 spilled VGPRs
// in the headline kernel, 0.29 GB of scratch stores per launch at config 3 (profiles/r3/pmc_traffic_C3_nipals_fused.txt).
// ONEWG: a read-only pass at 16 or fewer columns per lane compiled for ONE workgroup per CU (its default is two, 5 % faster
// on a long sweep): the form that may sum its partial rows -- and run the one-response update -- in its own tail, which is
// what a SHORT pass wants (below ~1.5 GB the two launches behind the pass cost more than the 5 %).
template <typename T, int V, int R, int NT, int CPT, bool DEFL, int LDAUX_ = AUX_NT, int STAUX = AUX_NT, bool RDST = false,
          int EDGE = 0, bool TILED = false, bool ONEWG = false>
__global__ __launch_bounds__(NT, (NT / 256) * ((CPT <= 16 && !DEFL && !ONEWG) ? 2 : 1)) void fused_pass_kernel(
    const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd, i64 N, int K,  // dst may alias X (in-place deflation)
    const double *__restrict__ v, const T *__restrict__ tprev, const double *__restrict__ pprev,
    T *__restrict__ tout, double *__restrict__ part, double *__restrict__ sspart, int rdst, i64 NV, const SliceTail st,
    const WalkWeights wk) {
    // N: rows swept (a multiple of V); NV <= N: valid rows of the score columns (the rows between are zero padding of
    // the library's own copy)
    constexpr int RP = R / V;    // lanes along the rows of a tile
    constexpr int CG = NT / RP;  // column groups
    constexpr int NW = NT / WAVE;
    constexpr int LDAUX = (EDGE == 2) ? 0 : LDAUX_;
    static_assert(RP <= WAVE && WAVE % RP == 0 && NT % RP == 0, "tile shape");
    static_assert(!TILED || (EDGE == 0 && !RDST), "the tiled copy is aligned and keeps one tile shape");
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
    PLS_STAMP(0);
    // TILED deflating pass: the operand vectors are fetched here but go to LDS behind the FIRST tile's loads (below) -- those
    // need neither, and a short pass (a shard) should not wait a memory round trip before its first byte of X is asked for
    constexpr bool LATE_OK = TILED && DEFL && CG * CPT <= NT;
    const bool LATE_FILL = LATE_OK && (rdst & 0x10000);
    double v_late = 0.0, p_late = 0.0;
    if (LATE_FILL) {
        if (tid < CG * CPT) {
            v_late = (tid < K) ? v[tid] : 0.0;
            p_late = (tid < K) ? pprev[tid] : 0.0;
        }
    } else {
        for (int k = tid; k < CG * CPT; k += NT) {
            vs[k] = (k < K) ? v[k] : 0.0;
            if (DEFL) ps[k] = (k < K) ? pprev[k] : 0.0;
        }
        __syncthreads();
    }
    bool first_tile = true;

    double pacc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) pacc[j] = 0.0;
    double ss = 0.0;
    int buf = 0;
    // EDGE: one descriptor spans the CGD column groups of a wave (first group cgw), otherwise all CG of the workgroup
    constexpr int CGD = EDGE ? WAVE / RP : CG;
    const int cgw = EDGE ? __builtin_amdgcn_readfirstlane(cg) : 0;
    const uint32_t xoff = (uint32_t)(((i64)rp * V + (i64)(cg - cgw) * ldx) * (i64)sizeof(T));
    uint32_t doff = DEFL ? (uint32_t)(((i64)rp * V + (i64)cg * ldd) * (i64)sizeof(T)) : 0u;
    i64 dtile = tsd;  // destination elements per source tile
    if constexpr (RDST && DEFL) {
        // the lane's rows rp*V.. fall into sub-tile (rp*V)/rdst of the R/rdst destination tiles a source tile covers
        doff = (uint32_t)(((i64)((rp * V) / rdst) * tsd + (rp * V) % rdst + (i64)cg * ldd) * (i64)sizeof(T));
        dtile = (R / rdst) * tsd;
    }
    constexpr uint32_t OOR = 0x80000000u;  // beyond every num_records the launcher allows
    constexpr bool TBUF = DEFL && sizeof(T) == 8 && R == 32 && CPT == 16;
    const __amdgpu_buffer_rsrc_t rs_tin = score_rsrc<T>(DEFL ? tprev : tout, NV);

    // EDGE = 2: XCD-contiguous tiles (grids that are a multiple of 8; cyclic otherwise).  The other instantiations fold
    // the three values below to blockIdx.x, gridDim.x and N at compile time.
    TileWalk<EDGE == 2, R> walk(N);
    const int pace = rdst & 0xffff;
    // Weighted walk (wk.nfull >= 0; grids that are a multiple of 8): the rounds [0, nfull) are dealt out cyclically to all
    // workgroups, the tiles behind them to the workgroups of the FAST classes only -- workgroup b runs on XCD b % 8, and on
    // MI355X the XCDs do not stream alike: in a read+write sweep the odd ones need 5-10 % longer per tile whatever the tile
    // (profiles/r5/pass_stamps_shapes.txt), so with equal shares half the chip idles for the last 8 % of the launch.
    // Static, so the tile set of a workgroup -- and with it every sum -- is the same in every run.
    // (compiled into the deflating instantiations only: the read-only ones sit exactly at their 128 registers, and the loop's
    // extra state cost the KERNEL plan's pass 0.66 -> 0.79 ms at config 3 -- profiles/r5/vs_r4.txt, first run)
    constexpr bool WEIGHTED = DEFL && EDGE != 2;
    const int wfast = (WEIGHTED && wk.nfull >= 0) ? ((wk.mask >> (blockIdx.x & 7)) & 1) : 0;
    const i64 wswitch = (WEIGHTED && wk.nfull >= 0) ? (i64)wk.nfull * gridDim.x : (i64)1 << 62;
    i64 wstep = walk.step();
    bool wshared = true;
    for (i64 tile = walk.first(); tile * R < walk.nlim(N); buf ^= 1) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);  // N % V == 0 (launcher): a pack is all-valid or all-invalid
        const uint32_t xo = rowok ? xoff : OOR, dof = rowok ? doff : OOR;
        // LDS operands (v, p_prev) are re-read every tile: an index the compiler cannot prove
        // loop-invariant keeps 2*CPT fp64 values out of the register file
        int cgz = cg;
        asm volatile("" : "+v"(cgz));
        if constexpr (TILED && DEFL) {
            // pacing: `rdst` x 64 cycles of s_sleep before a tile's loads go out (launcher)
            if (!LATE_FILL || !first_tile)  // (nothing is in flight before the first tile)
                for (int q = 0; q < pace; ++q) __builtin_amdgcn_s_sleep(1);
        }
        Pack<T, V> x[CPT];
        constexpr int GSTEP = CG * R * (int)sizeof(T);  // TILED: bytes between the column groups of a tile
        const int trec = K * R * (int)sizeof(T);        // TILED: bytes of a tile
        if constexpr (TILED) {
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X + tile * tsx), (short)0, trec, BUF_WORD3);
#pragma unroll
            for (int j = 0; j < CPT; ++j) x[j] = buf_ld_so<T, V, LDAUX>(rs, xo, j * GSTEP);
            __builtin_amdgcn_sched_barrier(0);  // all CPT loads in flight before anything consumes the first
            if constexpr (LATE_OK) {
                if (LATE_FILL && first_tile) {  // (uniform)
                    first_tile = false;
                    if (tid < CG * CPT) {
                        vs[tid] = v_late;
                        ps[tid] = p_late;
                    }
                    __syncthreads();
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cols = min(CGD, K - CG * j - cgw);  // columns of this group that exist (may be <= 0)
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * ldx * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(X + tile * tsx + (i64)(j * CG + cgw) * ldx), (short)0, (int)nrec, BUF_WORD3);
                x[j] = buf_ld<T, V, LDAUX>(rs, xo);
                __builtin_amdgcn_sched_barrier(0);  // build one descriptor, issue its load, repeat
            }
        }
        if (DEFL) {
            double tp[V];
            {
                // t_prev of the lane's rows; behind row NV (padded sweeps) zeros.  Two forms, chosen per instantiation by
                // measurement (same box, A/B): through a range-checked buffer descriptor -- the headline shape: 701 vs 693
                // components/s at config 3 -- or as a plain 16-byte load with an element-wise branch for the one straddling
                // pack -- every other shape: +0.3 ... +0.7 % (config 4, the shards of configs 3 and 5).
                Pack<T, V> tpk;
                if constexpr (TBUF) {
                    tpk = buf_ld<T, V>(rs_tin, rowok ? (uint32_t)(i0 * (i64)sizeof(T)) : OOR);
                } else if (rowok && i0 + V <= NV) {
                    tpk = ld_pack_u<T, V>(tprev + i0);
                } else {  // (rows the sweep does not cover must contribute nothing: the tail kernel owns row NV - 1 of an odd matrix)
#pragma unroll
                    for (int e = 0; e < V; ++e) tpk.v[e] = (rowok && i0 + e < NV) ? tprev[i0 + e] : (T)0;
                }
#pragma unroll
                for (int e = 0; e < V; ++e) tp[e] = -(double)tpk.v[e];
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const double pk = ps[cgz + CG * j];
#pragma unroll
                for (int e = 0; e < V; ++e) x[j].v[e] = (T)fma(tp[e], pk, (double)x[j].v[e]);
                if constexpr (TILED) {
                    const __amdgpu_buffer_rsrc_t rd =
                        __builtin_amdgcn_make_buffer_rsrc(dst + tile * tsd, (short)0, trec, BUF_WORD3);
                    // (write-through stores -- sc1 | nt, sc0 | sc1 | nt: nothing dirty in L2 when the launch ends -- measured in
                    // round 5: the pass +2.5 us on a shard, +23 us at config 3, the boundary behind it no shorter)
                    buf_st_so<T, V, STAUX>(rd, dof, j * GSTEP, x[j]);
                } else if constexpr (RDST) {  // lane offsets span several destination tiles: columns >= K masked per lane
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
        // the deflated tile goes out BEFORE the score arithmetic (left to itself the compiler sinks the stores behind the
        // score FMAs and the first butterfly level: 1.387 instead of 1.354 ms per launch at config 3)
        if constexpr (DEFL) __builtin_amdgcn_sched_barrier(0);
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
            if (i0 + V <= NV) {
                st_pack_u<T, V>(tout + i0, o);
            } else {  // the pack that straddles the last valid row (padded sweeps only)
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (i0 + e < NV) tout[i0 + e] = o.v[e];
            }
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
        if constexpr (!WEIGHTED) {
            tile += walk.step();
            continue;
        }
        tile += wstep;
        if (wshared && tile >= wswitch) {  // (uniform) the shared rounds are over
            if (!wfast) break;
            wshared = false;
            const int nfc = __builtin_popcount(wk.mask), below = __builtin_popcount(wk.mask & ((1u << (blockIdx.x & 7)) - 1u));
            tile = wswitch + (i64)(blockIdx.x >> 3) * nfc + below;
            wstep = (i64)(gridDim.x >> 3) * nfc;
        }
    }

    // epilogue: sum over the RP lanes that share a column group (low lane bits), then one lane
    // per column group writes this workgroup's partial row
    PLS_STAMP(1);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const double s = xor_range_sum<1, RP>(pacc[j]);
        const int k = cg + CG * j;
        if (rp == 0 && k < K) st_agent(part + (i64)blockIdx.x * K + k, s);  // (sc1: read by the slice's last workgroup)
    }
    ss = block_sum<NW>(ss, sred);
    if (tid == 0) st_agent(sspart + blockIdx.x, ss);
    // (the tail exists in the instantiations launched with ONE workgroup per CU only -- see slice_tail; the two-per-CU read-only
    // shapes carry neither its code nor the registers of the out-of-line update)
    constexpr bool ONE_PER_CU = !(CPT <= 16 && !DEFL) || ONEWG;
    if (ONE_PER_CU && st.cnt) {  // (uniform) the partial rows are summed inside this launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // LDS of the tail: role word and the t^T t sums in sred; the update's block sums + p_j^T w products in the score
        // exchange buffer (2 NW R doubles: the launcher checks 16 + A against it), its w in the operand vector's place
        static_assert(NT == 512, "tail_update_m1 plays the update's 1024 virtual threads with 512");
        PLS_STAMP(2);
        slice_tail<NT>(st, part, sspart, K, reinterpret_cast<int *>(sred + NW - 1), sred, &tred[0][0][0], vs);
    }
    PLS_STAMP(3);
}

// Semi-fused sweep for matrices too wide for the resident tile (K > 32 column groups' worth):
//   X' = X - t_prev p_prev^T (written),  t = X' w,  t^T t partials      -- the loading p = X'^T t then
// takes one more READ of X' (xty_kernel): 3 N K s of traffic per component instead of 4.
// Same tile access pattern as the fused pass; the column groups of a tile are streamed CPT at
// a time (nothing stays resident), the per-lane partial scores are combined once per tile.
// Dynamic LDS: 2*K doubles (w and p_prev).
// EDGE as in fused_pass_kernel.
template <typename T, int V, int R, int NT, int CPT, int EDGE = 0>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void deflate_score_kernel(
    const T *src, i64 lds_, i64 tss, T *dst, i64 ldd, i64 tsd, int rdst, i64 N, int K, const T *__restrict__ tprev,
    const double *__restrict__ pprev, const double *__restrict__ w, T *__restrict__ tout,
    double *__restrict__ sspart, i64 NV) {  // (N rows swept, NV valid score rows: see fused_pass_kernel)
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
    constexpr int CGD = EDGE ? WAVE / RP : CG;
    constexpr int LDAUX = (EDGE == 2) ? 0 : AUX_NT;
    const int cgw = EDGE ? __builtin_amdgcn_readfirstlane(cg) : 0;
    const uint32_t soff = (uint32_t)(((i64)rp * V + (i64)(cg - cgw) * lds_) * (i64)sizeof(T));
    // destination tiles may be shorter than the R rows read at a time (rdst divides R; rdst == R otherwise): the
    // lane's rows rp*V.. fall into sub-tile (rp*V)/rdst of the R/rdst destination tiles this source tile covers
    const int dsub = (rp * V) / rdst, dwithin = (rp * V) % rdst, dtiles = R / rdst;
    const uint32_t doff = (uint32_t)(((i64)dsub * tsd + dwithin + (i64)cg * ldd) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const int ngroups = (K + CG - 1) / CG;
    const __amdgpu_buffer_rsrc_t rs_tin = score_rsrc<T>(tprev, NV);
    double ss = 0.0;
    int buf = 0;
    TileWalk<EDGE == 2, R> walk(N);
    for (i64 tile = walk.first(); tile * R < walk.nlim(N); tile += walk.step(), buf ^= 1) {
        const i64 i0 = tile * R + (i64)rp * V;
        const bool rowok = (i0 < N);
        const uint32_t so = rowok ? soff : OOR, dof = rowok ? doff : OOR;
        double tp[V], tacc[V];
        {
            const Pack<T, V> tpk = buf_ld<T, V>(rs_tin, rowok ? (uint32_t)(i0 * (i64)sizeof(T)) : OOR);
#pragma unroll
            for (int e = 0; e < V; ++e) tp[e] = -(double)tpk.v[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) tacc[e] = 0.0;
        for (int g0 = 0; g0 < ngroups; g0 += CPT) {
            Pack<T, V> x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cols = min(CGD, K - CG * (g0 + j) - cgw);
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * lds_ * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(src + tile * tss + (i64)((g0 + j) * CG + cgw) * lds_), (short)0, (int)nrec, BUF_WORD3);
                x[j] = buf_ld<T, V, LDAUX>(rs, so);
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
            if (i0 + V <= NV) {
                st_pack_u<T, V>(tout + i0, o);
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (i0 + e < NV) tout[i0 + e] = o.v[e];
            }
        }
    }
    ss = block_sum<NW>(ss, sred);
    if (tid == 0) sspart[blockIdx.x] = ss;
}

// ---------------------------------------------------------------------------------------------------------------------
// The last N % V rows of a matrix (fewer than one 16-byte row pack: at most 1 row in fp64, 3 in fp32 storage).  Every
// kernel of this file works on whole row packs; its launcher hands these rows to ONE small workgroup that does the same
// step element by element and contributes one more partial row -- so the reference's "any row count" (src/pls.cpp:419-421)
// costs the hot loops nothing.  Element (i, k) of src / dst is at (i / rs) * ts + i % rs + k * ld (column-major: ts = rs).
// Steps, each optional:  x -= tprev[i] * pprev[k]  ->  dst  ;  t[i] = sum_k x v[k] -> tout (or t = tgiven[i])  ;
// part[k] = sum_i x t[i]  (or, with Y: part[k + m K] = sum_i x Y[i, m])  ;  *sspart = sum_i t[i]^2.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
struct TailArgs {
    const T *src = nullptr; i64 lds = 0, tss = 0; int rs = 1;
    T *dst = nullptr; i64 ldd = 0, tsd = 0; int rd = 1;
    const T *tprev = nullptr; const double *pprev = nullptr;
    const double *v = nullptr; T *tout = nullptr; const T *tgiven = nullptr;
    double *part = nullptr, *sspart = nullptr;
    const T *Y = nullptr; i64 ldy = 0; int M = 0;
    i64 row0 = 0; int nrows = 0, K = 0;
    int zero_rows = 0;  // rows behind the last one to clear in dst (the padding of the library's tiled copy up to a multiple of V)
};

template <typename T>
__global__ __launch_bounds__(WG) void tail_rows_kernel(TailArgs<T> a) {
    __shared__ double sm[WG / WAVE];
    constexpr int MAXR = 16 / sizeof(T) - 1;
    double t[MAXR];
    double ss = 0.0;
    for (int r = 0; r < MAXR; ++r) {
        t[r] = 0.0;
        if (r >= a.nrows) continue;  // (uniform)
        const i64 i = a.row0 + r;
        const T *srow = a.src + (i / a.rs) * a.tss + i % a.rs;
        T *drow = a.dst ? a.dst + (i / a.rd) * a.tsd + i % a.rd : nullptr;
        const double tp = a.tprev ? -(double)a.tprev[i] : 0.0;
        double sum = 0.0;
        for (int k = threadIdx.x; k < a.K; k += WG) {
            T x = srow[(i64)k * a.lds];
            if (a.tprev) x = (T)fma(tp, a.pprev[k], (double)x);
            if (drow) drow[(i64)k * a.ldd] = x;
            if (a.v) sum = fma((double)x, a.v[k], sum);
        }
        if (a.v) {
            t[r] = (double)(T)block_sum<WG / WAVE>(sum, sm);  // the score as stored
            if (threadIdx.x == 0 && a.tout) a.tout[i] = (T)t[r];
        } else if (a.tgiven) {
            t[r] = (double)a.tgiven[i];
        }
        ss = fma(t[r], t[r], ss);
    }
    if (a.dst)
        for (int r = 0; r < a.zero_rows; ++r) {
            const i64 i = a.row0 + a.nrows + r;
            T *drow = a.dst + (i / a.rd) * a.tsd + i % a.rd;
            for (int k = threadIdx.x; k < a.K; k += WG) drow[(i64)k * a.ldd] = (T)0;
        }
    if (a.part) {  // (every thread re-reads the elements it wrote itself)
        const int nm = a.Y ? a.M : 1;
        for (int k = threadIdx.x; k < a.K; k += WG)
            for (int m = 0; m < nm; ++m) {
                double p = 0.0;
                for (int r = 0; r < MAXR; ++r)
                    if (r < a.nrows) {
                        const i64 i = a.row0 + r;
                        const double x = a.dst ? (double)a.dst[(i / a.rd) * a.tsd + i % a.rd + (i64)k * a.ldd]
                                               : (double)a.src[(i / a.rs) * a.tss + i % a.rs + (i64)k * a.lds];
                        p = fma(x, a.Y ? (double)a.Y[i + (i64)m * a.ldy] : t[r], p);
                    }
                a.part[k + (i64)m * a.K] = p;
            }
    }
    if (a.sspart && threadIdx.x == 0) *a.sspart = ss;
}

template <typename T>
void launch_tail_rows(hipStream_t stream, const TailArgs<T> &a) {
    hipLaunchKernelGGL((tail_rows_kernel<T>), dim3(1), dim3(WG), 0, stream, a);
}

// Copy into row-tile-major storage AND X^T Y in the same sweep (src/pls.cpp:396 on the way into the library's tiled copy):
// what a KERNEL-plan fit needs before its first component when its passes run on the tiled copy -- wide matrices
// (1024 < K <= 4096: one read + one write instead of retile_kernel + the separate X^T Y pass, config 4), and matrices
// whose columns are not 16-byte aligned (every later pass then reads aligned tiles at the full rate).
// grid = (row chunks, column blocks of CG*CPTB columns); a workgroup walks CONSECUTIVE source tiles of R rows (256-byte
// column segments) of its column block: per tile CPTB row packs per lane in, the same packs out as one contiguous
// CG*CPTB*rdst*s-byte piece per destination tile, CPTB*MT fp64 accumulators per lane (X^T Y of the lane's rows and
// columns); part[blockIdx.x][k + m*K] = the workgroup's partial, the layout reduce_partials_kernel sums.
// YLDS (several responses): the R x MT block of Y goes through LDS once per tile -- one element per thread in, the lane's
// rows out as broadcast reads.  Loading it per lane would put MT 16-byte loads beside CPTB of X on the texture path
// (the 32 column-group lanes of a row pack all fetch the same Y values): 2.9 instead of 0.9 ms at config 4.
// EDGE as in fused_pass_kernel (2: unaligned columns -- plain loads; the consecutive tiles of a workgroup re-read the
// line a segment shares with the next tile from L2).  N % V == 0 (the tail rows are the launcher's).
// LT (destination tiles of one or two row packs, rdst <= 2 V: the copy of a matrix beyond 2048 columns): a source lane's pack
// belongs to a destination tile of its own, and stored directly a wave writes 64- or 128-byte runs (2.7 TB/s for the copy
// into row-pack tiles).  Instead the tile block is transposed through LDS -- [row pack][column] with the column index XORed
// by the row pack: conflict-free both ways -- and written out in the destination's order, 1 KB per wave-store.
// (register bounds of ONE workgroup per CU, which is what the launcher starts: config 4 0.80 -> 0.77 ms against the bounds of
// two; two tiles in flight at 8 columns per lane as well: no difference -- profiles/r4/retile_xty_wgs.txt)
template <typename T, int V, int R, int NT, int CPTB, int MT, int EDGE = 0, bool LT = false>
__global__ __launch_bounds__(NT, NT / 256) void retile_xty_kernel(const T *src, i64 lds_, const T *__restrict__ Y,
                                                                        i64 ldy, T *dst, i64 ldd, i64 tsd, int rdst, i64 N,
                                                                        int K, int M, double *__restrict__ part, int tpw) {
    constexpr int RP = R / V, CG = NT / RP;
    constexpr int LDAUX = (EDGE == 2) ? 0 : AUX_NT;
    constexpr bool YLDS = MT >= 4;
    // tiles per iteration: with few columns per lane (MT = 8 leaves room for CPTB = 4 only) two tiles' loads are in
    // flight at once -- 4 loads per lane and iteration left the sweep latency-bound (1.6 ms at config 4)
    constexpr int TU = (CPTB < 8) ? 2 : 1;
    constexpr int YE = (TU * R * MT + NT - 1) / NT;  // Y elements per thread and iteration (YLDS)
    __shared__ alignas(16) T ys[2][YLDS ? TU * MT : 1][YLDS ? R : V];
    extern __shared__ __attribute__((aligned(16))) unsigned char retile_dyn[];  // LT: [RP][CG * CPTB] packs
    Pack<T, V> *trans = reinterpret_cast<Pack<T, V> *>(retile_dyn);
    constexpr int W = CG * CPTB;  // columns of this workgroup's block
    static_assert(!LT || (W % 16 == 0 && (RP * W) % NT == 0 && RP <= 16), "transposed store shape");
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const int g0 = blockIdx.y * CPTB;  // first column group of this block
    constexpr int CGD = EDGE ? WAVE / RP : CG;
    const int cgw = EDGE ? __builtin_amdgcn_readfirstlane(cg) : 0;
    const uint32_t soff = (uint32_t)(((i64)rp * V + (i64)(cg - cgw) * lds_) * (i64)sizeof(T));
    const int dsub = (rp * V) / rdst, dwithin = (rp * V) % rdst, dtiles = R / rdst;
    const uint32_t doff = (uint32_t)(((i64)dsub * tsd + dwithin + (i64)cg * ldd) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const i64 ntiles = (N + R - 1) / R;
    const i64 tile0 = (i64)blockIdx.x * tpw, tile1 = min(ntiles, tile0 + (i64)tpw);
    double acc[CPTB][MT];
#pragma unroll
    for (int j = 0; j < CPTB; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = 0.0;
    int buf = 0;
    for (i64 tile = tile0; tile < tile1; tile += TU, buf ^= 1) {
        Pack<T, V> x[TU][CPTB];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const bool rowok = (tile + u < tile1) && ((tile + u) * R + (i64)rp * V < N);
#pragma unroll
            for (int j = 0; j < CPTB; ++j) {
                const int cols = min(CGD, K - CG * (g0 + j) - cgw);
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * lds_ * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(src + (tile + u) * R + (i64)((g0 + j) * CG + cgw) * lds_), (short)0, (int)nrec, BUF_WORD3);
                x[u][j] = buf_ld<T, V, LDAUX>(rs, rowok ? soff : OOR);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (YLDS) {  // the responses of the TU tiles' rows: one pass through LDS (rows >= N, tiles of the next chunk: zeros)
#pragma unroll
            for (int q = 0; q < YE; ++q) {
                const int idx = threadIdx.x + q * NT;  // = (u * MT + m) * R + row
                const int yr = idx % R, ym = (idx / R) % MT, yu = idx / (R * MT);
                const i64 row = (tile + yu) * R + yr;
                if (idx < TU * R * MT)
                    ys[buf][yu * MT + ym][yr] = (ym < M && tile + yu < tile1 && row < N) ? Y[row + (i64)ym * ldy] : (T)0;
            }
            __syncthreads();  // (two buffers: iteration t + 2 writes after every thread has passed this point for t + 1)
        }
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const i64 i0 = (tile + u) * R + (i64)rp * V;
            const bool rowok = (tile + u < tile1) && (i0 < N);
            const uint32_t dof = rowok ? doff : OOR;
            if (!dst) {
                // (no copy asked for: X^T Y alone -- the stand-alone product of 8 responses, which as a kernel of its own re-reads
                // the Y packs of a row chunk for every 4 columns: twice the bytes of X through the texture path)
            } else if constexpr (!LT) {
#pragma unroll
                for (int j = 0; j < CPTB; ++j) {
                    const int k = cg + CG * (g0 + j);
                    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                        dst + (tile + u) * dtiles * tsd + (i64)(g0 + j) * CG * ldd, (short)0, 0x7fffffff, BUF_WORD3);
                    buf_st<T, V, AUX_NT>(rd, (k < K) ? dof : OOR, x[u][j]);
                }
            } else {
                __syncthreads();  // the previous block has been read out
#pragma unroll
                for (int j = 0; j < CPTB; ++j) trans[rp * W + ((cg + CG * j) ^ rp)] = x[u][j];
                __syncthreads();
                const __amdgpu_buffer_rsrc_t rd =
                    __builtin_amdgcn_make_buffer_rsrc(dst + (tile + u) * dtiles * tsd, (short)0, 0x7fffffff, BUF_WORD3);
                const int Q = rdst / V;  // row packs per destination tile: 1 or 2
#pragma unroll
                for (int jj = 0; jj < RP * W / NT; ++jj) {
                    // destination order: tile d, column c, row pack q of the tile
                    const int L = jj * NT + (int)threadIdx.x;
                    const int d = L / (W * Q), rem = L - d * (W * Q);
                    const int c = (Q == 1) ? rem : (rem >> 1), q = (Q == 1) ? 0 : (rem & 1);
                    const int rp2 = d * Q + q;
                    const Pack<T, V> pk = trans[rp2 * W + (c ^ rp2)];
                    const int k = g0 * CG + c;
                    const bool ok = (tile + u < tile1) && ((tile + u) * R + (i64)rp2 * V < N) && (k < K);
                    const uint32_t off = (uint32_t)(((i64)d * tsd + (i64)q * V + (i64)k * ldd) * (i64)sizeof(T));
                    buf_st<T, V, AUX_NT>(rd, ok ? off : OOR, pk);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                Pack<T, V> yp;
                if constexpr (YLDS) {
                    yp = *reinterpret_cast<const Pack<T, V> *>(&ys[buf][u * MT + m][rp * V]);
                } else {
                    if (m < M && rowok) {
                        yp = ld_pack_u<T, V>(Y + i0 + (i64)m * ldy);
                    } else {
#pragma unroll
                        for (int e = 0; e < V; ++e) yp.v[e] = (T)0;
                    }
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const double ye = (double)yp.v[e];
#pragma unroll
                    for (int j = 0; j < CPTB; ++j) acc[j][m] = fma((double)x[u][j].v[e], ye, acc[j][m]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CPTB; ++j) {
        const int k = cg + CG * (g0 + j);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const double s = xor_range_sum<1, RP>(acc[j][m]);
            if (rp == 0 && k < K && m < M) part[(i64)blockIdx.x * ((i64)K * M) + k + (i64)m * K] = s;
        }
    }
}

// shared by the launchers: 16-byte aligned columns / element alignment; the span of the column groups behind one descriptor
template <typename T>
inline bool cols_aligned(const void *q, i64 ld) {
    constexpr int V = 16 / sizeof(T);
    return ((uintptr_t)q % 16 == 0) && (ld % V == 0);
}
template <typename T>
inline bool elem_aligned(const void *q) { return (uintptr_t)q % sizeof(T) == 0; }
// EDGE level of a source matrix (ld in elements) read with tiles of RP row lanes: 0 plain, 1 per-wave descriptors
// (long columns), 2 unaligned columns; -1 = not addressable at all
template <typename T>
inline int edge_level(const void *q, i64 ld, int cg_all, int cg_wave) {
    if (!elem_aligned<T>(q)) return -1;
    const bool fits = (i64)cg_all * ld * (i64)sizeof(T) < (1ll << 31);
    if (cols_aligned<T>(q, ld) && fits) return 0;
    if ((i64)cg_wave * ld * (i64)sizeof(T) >= (1ll << 31)) return -1;
    return cols_aligned<T>(q, ld) ? 1 : 2;
}

// rc as launch_fused_pass; *nb = partial rows written (<= max_rows).  M <= 8.  CGX: column groups of the SOURCE tile (32:
// 256-byte segments; 8 / 16: the taller tiles of narrow matrices, whose copy uses the same tile height).
template <typename T, int CGX = 32>
int launch_retile_xty(hipStream_t stream, int num_cu, const T *src, i64 lds_, const T *Y, i64 ldy, T *dst, i64 ldd, i64 tsd,
                      int rdst, i64 N, int K, int M, double *part, int max_rows, int *nb) {
    // dst == nullptr: X^T Y alone (no copy): ldd, tsd, rdst are ignored
    constexpr int V = 16 / sizeof(T);
    constexpr int R = (512 / CGX) * V, NT = 512;
    constexpr int CG = NT / (R / V);
    const bool nostore = dst == nullptr;
    if (nostore) { ldd = R; tsd = (i64)R * K; rdst = R; }
    if ((!nostore && !cols_aligned<T>(dst, ldd)) || tsd % V != 0 || rdst < V || R % rdst != 0 || N < 1 || M < 1 || M > 8 || max_rows < 2 ||
        !elem_aligned<T>(Y))
        return 1;
    const int edge = edge_level<T>(src, lds_, CG, WAVE / (R / V));
    if (edge < 0) return 1;
    if (!nostore && ((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const i64 Nf = N - N % V;
    int gx = 0;
    if (Nf > 0) {
        const i64 ntiles = (Nf + R - 1) / R;
        // columns per lane: 8, or 4 where 8 x MT accumulators and the tiles in flight do not fit 128 registers
        const int cptb = (M > 4 || (M > 2 && sizeof(T) == 4)) ? 4 : 8;
        const int nkb = (K + CG * cptb - 1) / (CG * cptb);
        // ONE workgroup per CU in total (consecutive tiles per workgroup), at most max_rows - 1 row chunks: like every
        // read+write sweep of this library the copy is faster with less in flight -- config 4 (8 responses) 0.797 -> 0.719 ms,
        // 65,536 x 8,192 fp32 1.088 -> 1.020 ms, config 3 -1.5 % (three alternating pairs, profiles/r4/retile_xty_wgs.txt)
        // (a read-only sweep -- no copy -- wants more in flight: four workgroups per CU in all)
        const i64 want = std::max<i64>(1, std::min<i64>(((i64)(nostore ? 4 : 1) * num_cu + nkb - 1) / nkb, max_rows - 1));
        const i64 tpw = (ntiles + want - 1) / want;
        gx = (int)((ntiles + tpw - 1) / tpw);
        const dim3 g((unsigned)gx, (unsigned)nkb), b(NT);
        // destination tiles of ONE row pack, one or two responses: transposed through LDS (the 32-group source tile only).
        // Measured: 87,381 x 6,144 fp64 3.14 -> 2.17 ms; with 8 responses (4 columns per lane, two tiles in flight) no gain,
        // and two-pack tiles (config 4: 128-byte runs already) lose, 0.82 -> 1.10 ms -- those keep the direct stores.
        const bool lt = CGX == 32 && rdst == V && M <= 2 && !nostore;
#define RX_LAUNCH(CPTB_, MT_, E_) \
    hipLaunchKernelGGL((retile_xty_kernel<T, V, R, NT, CPTB_, MT_, E_>), g, b, 0, stream, src, lds_, Y, ldy, dst, ldd, tsd, rdst, Nf, K, M, part, (int)tpw)
#define RX_LAUNCH_LT(CPTB_, MT_, E_)                                                                                      \
    do {                                                                                                                  \
        if constexpr (CGX == 32) {                                                                                        \
            auto kfn = &retile_xty_kernel<T, V, R, NT, CPTB_, MT_, E_, true>;                                             \
            const int dynb = (R / V) * CG * CPTB_ * 16;                                                                   \
            if (dynb > 48 * 1024 && !raise_dynamic_lds(reinterpret_cast<const void *>(kfn), dynb)) return 1;              \
            hipLaunchKernelGGL(kfn, g, b, dynb, stream, src, lds_, Y, ldy, dst, ldd, tsd, rdst, Nf, K, M, part, (int)tpw); \
        }                                                                                                                 \
    } while (0)
#define RX_CASE(CPTB_, MT_)                                                                                               \
    do {                                                                                                                  \
        if (lt) { if (edge == 2) RX_LAUNCH_LT(CPTB_, MT_, 2); else if (edge) RX_LAUNCH_LT(CPTB_, MT_, 1); else RX_LAUNCH_LT(CPTB_, MT_, 0); } \
        else if (edge == 2) RX_LAUNCH(CPTB_, MT_, 2); else if (edge) RX_LAUNCH(CPTB_, MT_, 1); else RX_LAUNCH(CPTB_, MT_, 0); \
    } while (0)
#define RX_CASE_D(CPTB_, MT_) \
    do { if (edge == 2) RX_LAUNCH(CPTB_, MT_, 2); else if (edge) RX_LAUNCH(CPTB_, MT_, 1); else RX_LAUNCH(CPTB_, MT_, 0); } while (0)
        if (M > 4) RX_CASE_D(4, 8);
        else if (M > 2 && cptb == 4) RX_CASE_D(4, 4);
        else if (M > 2) RX_CASE_D(8, 4);
        else if (M > 1) RX_CASE(8, 2);
        else RX_CASE(8, 1);
#undef RX_CASE_D
#undef RX_CASE
#undef RX_LAUNCH_LT
#undef RX_LAUNCH
    }
    if (Nf < N) {
        TailArgs<T> a;
        a.src = src; a.lds = lds_; a.tss = R; a.rs = R;
        a.dst = dst; a.ldd = ldd; a.tsd = tsd; a.rd = rdst;
        a.part = part + (i64)gx * K * M; a.Y = Y; a.ldy = ldy; a.M = M;
        a.row0 = Nf; a.nrows = (int)(N - Nf); a.K = K; a.zero_rows = V - a.nrows;
        launch_tail_rows(stream, a);
        ++gx;
    }
    *nb = gx;
    return 0;
}

// rc as launch_fused_pass; *nss = number of t^T t partials written
template <typename T>
int launch_deflate_score(hipStream_t stream, int num_cu, const T *src, i64 lds_, i64 tss, T *dst, i64 ldd, i64 tsd,
                         int rdst, i64 N, int K, const T *tprev, const double *pprev, const double *w, T *tout,
                         double *sspart, int max_rows, int *nss, bool src_padded = false) {
    // src_padded: the source is the library's own copy, whose rows up to the next multiple of V exist and hold zeros --
    // the sweep runs over them (no tail kernel); otherwise the last N % V rows go to the tail kernel
    constexpr int V = 16 / sizeof(T);
    constexpr int R = 256 / sizeof(T), NT = 512, CPT = 8;
    constexpr int CG = NT / (R / V);
    if (N < 1 || max_rows < 2 || !cols_aligned<T>(dst, ldd) || !elem_aligned<T>(tprev) || !elem_aligned<T>(tout)) return 1;
    if ((N + V) * (i64)sizeof(T) >= (1ll << 31)) return 1;  // one descriptor per score column
    const int edge = edge_level<T>(src, lds_, CG, WAVE / (R / V));
    if (edge < 0 || (edge == 0 && tss % V != 0)) return 1;
    if (tsd % V != 0 || rdst < V || R % rdst != 0) return 1;
    if (((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if ((i64)CG * ldd * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const size_t dyn = (size_t)K * 16;  // w and p_prev
    if (dyn > 72 * 1024) return 1;      // two workgroups per CU must fit the 160 KiB LDS
    auto kfn = edge == 2 ? &deflate_score_kernel<T, V, R, NT, CPT, 2>
                         : (edge == 1 ? &deflate_score_kernel<T, V, R, NT, CPT, 1> : &deflate_score_kernel<T, V, R, NT, CPT, 0>);
    if (dyn > 48 * 1024 && !raise_dynamic_lds(reinterpret_cast<const void *>(kfn), 72 * 1024)) return 1;
    const i64 Nf = src_padded ? (N + V - 1) / V * V : N - N % V;
    i64 grid = 0;
    if (Nf > 0) {
        const i64 ntiles = (Nf + R - 1) / R;
        grid = std::min<i64>(std::min<i64>(ntiles, 2 * (i64)num_cu), max_rows - 1);
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NT), (size_t)K * 16, stream, src, lds_, tss, dst, ldd, tsd, rdst, Nf,
                           K, tprev, pprev, w, tout, sspart, N);
    }
    if (Nf < N) {  // the last N % V rows
        TailArgs<T> a;
        a.src = src; a.lds = lds_; a.tss = tss; a.rs = R;
        a.dst = dst; a.ldd = ldd; a.tsd = tsd; a.rd = rdst;
        a.tprev = tprev; a.pprev = pprev; a.v = w; a.tout = tout; a.sspart = sspart + grid;
        a.row0 = Nf; a.nrows = (int)(N - Nf); a.K = K; a.zero_rows = V - a.nrows;
        launch_tail_rows(stream, a);
        ++grid;
    }
    *nss = (int)grid;
    return 0;
}

// Copy of a column-major matrix into row-tile-major storage with tiles of rdst rows (rdst divides R): read in the
// 256-byte-segment pattern, written as contiguous tile pieces.  Used once per fit by the KERNEL plan on wide
// matrices (1024 < K <= 4096), whose read-only passes then run fused on the short tiles.
template <typename T, int V, int R, int NT, int CPT, int EDGE = 0>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void retile_kernel(const T *src, i64 lds_, T *dst, i64 ldd, i64 tsd,
                                                                    int rdst, i64 N, int K) {
    constexpr int RP = R / V, CG = NT / RP;
    constexpr int CGD = EDGE ? WAVE / RP : CG;
    constexpr int LDAUX = (EDGE == 2) ? 0 : AUX_NT;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const int cgw = EDGE ? __builtin_amdgcn_readfirstlane(cg) : 0;
    const uint32_t soff = (uint32_t)(((i64)rp * V + (i64)(cg - cgw) * lds_) * (i64)sizeof(T));
    const int dsub = (rp * V) / rdst, dwithin = (rp * V) % rdst, dtiles = R / rdst;
    const uint32_t doff = (uint32_t)(((i64)dsub * tsd + dwithin + (i64)cg * ldd) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const int ngroups = (K + CG - 1) / CG;
    TileWalk<EDGE == 2, R> walk(N);
    for (i64 tile = walk.first(); tile * R < walk.nlim(N); tile += walk.step()) {
        const bool rowok = (tile * R + (i64)rp * V < N);
        const uint32_t so = rowok ? soff : OOR, dof = rowok ? doff : OOR;
        for (int g0 = 0; g0 < ngroups; g0 += CPT) {
            Pack<T, V> x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const int cols = min(CGD, K - CG * (g0 + j) - cgw);
                const uint32_t nrec = cols > 0 ? (uint32_t)((i64)cols * lds_ * (i64)sizeof(T)) : 0u;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<T *>(src + tile * R + (i64)((g0 + j) * CG + cgw) * lds_), (short)0, (int)nrec, BUF_WORD3);
                x[j] = buf_ld<T, V, LDAUX>(rs, so);
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
    if (!cols_aligned<T>(dst, ldd) || tsd % V != 0 || rdst < V || R % rdst != 0 || N < 1) return 1;
    const int edge = edge_level<T>(src, lds_, CG, WAVE / (R / V));
    if (edge < 0) return 1;
    if (((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)) return 1;
    const i64 Nf = N - N % V;
    if (Nf > 0) {
        const i64 ntiles = (Nf + R - 1) / R;
        const i64 grid = std::min<i64>(ntiles, 2 * (i64)num_cu);
        auto kfn = edge == 2 ? &retile_kernel<T, V, R, NT, CPT, 2>
                             : (edge == 1 ? &retile_kernel<T, V, R, NT, CPT, 1> : &retile_kernel<T, V, R, NT, CPT, 0>);
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(NT), 0, stream, src, lds_, dst, ldd, tsd, rdst, Nf, K);
    }
    if (Nf < N) {
        TailArgs<T> a;
        a.src = src; a.lds = lds_; a.tss = R; a.rs = R;
        a.dst = dst; a.ldd = ldd; a.tsd = tsd; a.rd = rdst;
        a.row0 = Nf; a.nrows = (int)(N - Nf); a.K = K; a.zero_rows = V - a.nrows;
        launch_tail_rows(stream, a);
    }
    return 0;
}

// Will launch_deflate_score accept every deflating pass of a fit (decided once per fit, like fused_pass_mode)?
// 0 = no, 1 = yes, 2 = yes through an EDGE instantiation (long or unaligned columns).
template <typename T>
int deflate_score_mode(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    constexpr int V = 16 / sizeof(T);
    constexpr int CG = 512 / ((256 / (int)sizeof(T)) / V);
    if (N < 1 || (size_t)K * 16 > 72 * 1024 || !elem_aligned<T>(Tm) || (N + 16) * (i64)sizeof(T) >= (1ll << 31)) return 0;
    const int e = edge_level<T>(X, ldx, CG, 4);
    return e < 0 ? 0 : (e == 0 ? 1 : 2);
}
// Can the caller's matrix be copied into short tiles (launch_retile / launch_retile_xty read it with the 256-byte-segment
// tile) and every pass then run on the copy?  The checks of deflate_score_mode without its LDS limit on K.
template <typename T>
bool wide_source_ok(const T *X, i64 ldx, i64 N, const T *Tm) {
    constexpr int V = 16 / sizeof(T);
    constexpr int CG = 512 / ((256 / (int)sizeof(T)) / V);
    if (N < 1 || !elem_aligned<T>(Tm) || (N + 16) * (i64)sizeof(T) >= (1ll << 31)) return false;
    return edge_level<T>(X, ldx, CG, 4) >= 0;
}
template <typename T>
bool deflate_score_covers(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    return deflate_score_mode<T>(X, ldx, N, K, Tm, ldt) != 0;
}

// Loading partials p_raw = X^T t for a matrix in (ld, ts) tile addressing -- the row-tile-major work buffer of
// the semi-fused plan:   part[blockIdx.x*K + k] = sum over the workgroup's tiles and rows of X[i,k] * t[i].
// grid = (row chunks of tpw tiles, column blocks of CG*CPT columns); one-shot workgroups, 16-byte accesses,
// a tile's column block is one contiguous CG*CPT*256-byte piece.
template <typename T, int V, int R, int NT, int CPT>
__global__ __launch_bounds__(NT, (NT / 256) * 2) void xty_tiled_kernel(const T *X, i64 ldx, i64 tsx, i64 N, int K,
                                                                       const T *__restrict__ t,
                                                                       double *__restrict__ part, int tpw, i64 NV) {
    constexpr int RP = R / V, CG = NT / RP;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const int g0 = blockIdx.y * CPT;  // first column group of this block
    const i64 ntiles = (N + R - 1) / R;
    const i64 tile0 = (i64)blockIdx.x * tpw, tile1 = min(ntiles, tile0 + (i64)tpw);
    const uint32_t xoff = (uint32_t)(((i64)rp * V + (i64)cg * ldx) * (i64)sizeof(T));
    constexpr uint32_t OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_t = score_rsrc<T>(t, NV);
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
        {
            const Pack<T, V> tpk = buf_ld<T, V>(rs_t, rowok ? (uint32_t)(i0 * (i64)sizeof(T)) : OOR);
#pragma unroll
            for (int e = 0; e < V; ++e) tv[e] = (double)tpk.v[e];
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
    if (!cols_aligned<T>(X, ldx) || !elem_aligned<T>(t) || tsx % V != 0 || N < 1 || max_rows < 2) return 1;
    if ((N + V) * (i64)sizeof(T) >= (1ll << 31)) return 1;  // one descriptor per score column
    if ((i64)CG * ldx * (i64)sizeof(T) >= (1ll << 31)) return 1;
    // (X is the library's own zero-padded copy: the sweep runs to the next multiple of V rows)
    const i64 Nf = (N + V - 1) / V * V;
    const i64 ntiles = (Nf + R - 1) / R;
    const int nkb = (K + CG * CPT - 1) / (CG * CPT);
    // ~8 workgroups per CU in total, at most max_rows row chunks
    const i64 want = std::max<i64>(1, std::min<i64>((8 * (i64)num_cu + nkb - 1) / nkb, max_rows));
    const i64 tpw = (ntiles + want - 1) / want;
    const i64 gx = (ntiles + tpw - 1) / tpw;
    hipLaunchKernelGGL((xty_tiled_kernel<T, V, R, NT, CPT>), dim3((unsigned)gx, (unsigned)nkb), dim3(NT), 0, stream, X,
                       ldx, tsx, Nf, K, t, part, (int)tpw, N);
    *nb = (int)gx;
    return 0;
}

// Rows per tile of the tile-resident kernels.  CGX = column groups of a 512-thread workgroup:
//   32  -> 16 lanes along the rows: 256-byte column segments (32 fp64 / 64 fp32 rows), K <= 1024 -- the shape
//          that can also read the caller's column-major matrices at full rate;
//   64, 128 -> 8 / 4 lanes along the rows (16 / 8 fp64, 32 / 16 fp32 rows), K <= 2048 / 4096: only for the
//          row-tile-major working copy, where a tile is contiguous whatever its height;
//   256, 512 -> 2 / 1 lanes along the rows at 16 columns per lane: K <= 4096 / 8192 (512, read-only: 32 per lane, K <= 16384).
template <typename T, int CGX = 32>
constexpr int tile_rows() { return (512 / CGX) * (16 / (int)sizeof(T)); }

// Will launch_fused_pass accept every pass of a fit on (X, ldx) with score columns Tm + a*ldt?  (Decided once
// per fit: the work buffer's layout depends on it.)  0 = no; 1 = yes; 2 = yes through an EDGE instantiation (columns
// that are not 16-byte aligned, or a leading dimension whose 32 column groups do not fit one descriptor -- up to
// 2^31 bytes per WAVE / RP = 4 columns there).  Any N >= 1: the last N % V rows are the tail kernel's.
// Column groups of the tile that reads the caller's matrix.  32 (16 row lanes, 256-byte segments) from 257 columns on;
// NARROW matrices take taller tiles with fewer column groups, so that a lane still has 8-16 loads in flight: 8 groups x 64
// row lanes (128 fp64 rows, 1 KB segments) up to 128 columns, 16 x 32 up to 256.  With the 32-group tile a 64-column matrix
// has 2 loads per lane and streams at 0.64-0.68 of peak (config-3-sized: 581 / 1,097 components/s), a 32-column one at 0.38-0.46.
inline int tall_groups(int K) {
    return K <= 128 ? 8 : (K <= 256 ? 16 : 32);
}

template <typename T>
int fused_pass_mode(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    const int CG = tall_groups(K);
    if (K > 32 * 32 || N < 1 || !elem_aligned<T>(Tm) || (N + 16) * (i64)sizeof(T) >= (1ll << 31)) return 0;
    const int e = edge_level<T>(X, ldx, CG, CG / 8);  // (a wave holds 64 / (512 / CG) = CG / 8 column groups)
    return e < 0 ? 0 : (e == 0 ? 1 : 2);
}
template <typename T>
bool fused_pass_covers(const T *X, i64 ldx, i64 N, int K, const T *Tm, i64 ldt) {
    return fused_pass_mode<T>(X, ldx, N, K, Tm, ldt) != 0;
}

// rc: 0 = launched, 1 = shape/alignment not covered (caller falls back to the one-product
// kernels), <0 = launch error.  grid_hint: 0 = auto.  (ldx, tsx) / (ldd, tsd): column and tile strides.
// rdst > 0 (with CGX = 32 and a deflating pass): the destination uses tiles of rdst rows.
template <typename T, int CGX = 32>
int launch_fused_pass(hipStream_t stream, int num_cu, const T *X, i64 ldx, i64 tsx, T *dst, i64 ldd, i64 tsd,
                      i64 N, int K, const double *v, const T *tprev, const double *pprev, T *tout,
                      double *part, int max_rows, double *sspart, int *nb, int *nss, int grid_hint, int rdst = 0,
                      bool src_padded = false, const SliceTail *tail = nullptr, bool *tail_used = nullptr,
                      bool *upd_done = nullptr) {
    // *upd_done: the tail ran the component update as well (tail->upd set by the caller and the shape has room for it)
    // tail (cnt, red, npush, seq, peers set by the caller): sum the partial rows inside the launch (slice_tail) when the
    // grid is large enough; *tail_used tells the caller whether reduce_partials_kernel is still to run
    // src_padded: X is the library's own copy, whose rows up to the next multiple of V exist and hold zeros -- the sweep
    // runs over them (no tail kernel); otherwise the last N % V rows go to the tail kernel
    constexpr int V = 16 / sizeof(T);
    constexpr int R = tile_rows<T, CGX>(), NT = 512;
    constexpr int CG = NT / (R / V);
    static_assert(CG == CGX, "tile shape");
    const bool defl = (tprev != nullptr);
    if (!elem_aligned<T>(tout) || (defl && (!cols_aligned<T>(dst, ldd) || !elem_aligned<T>(tprev)))) return 1;
    // the byte span of the column groups behind one descriptor (its num_records, and every lane offset) must stay below
    // 2^31: all CG groups of the workgroup, or -- EDGE -- the WAVE / RP groups of one wave
    int edge = edge_level<T>(X, ldx, CG, WAVE / (R / V));
    if (edge < 0 || (edge == 0 && tsx % V != 0) || (defl && tsd % V != 0)) return 1;
    // (512 groups: 32 columns per lane only for read-only passes -- v alone is 128 KB of LDS there, p_prev would not fit)
    if (K > CG * ((CGX == 64 || CGX == 128 || CGX == 256 || CGX < 32 || (CGX == 512 && defl)) ? 16 : 32) || N < 1 || max_rows < 2) return 1;
    if ((N + V) * (i64)sizeof(T) >= (1ll << 31)) return 1;  // one descriptor per score column
    if (defl && (i64)CG * ldd * (i64)sizeof(T) >= (1ll << 31)) return 1;
    if (rdst > 0 && (CGX != 32 || !defl || rdst < V || R % rdst != 0 ||
                     ((i64)(R / rdst) * tsd + (i64)CG * ldd) * (i64)sizeof(T) >= (1ll << 31)))
        return 1;
    if (CGX > 32 && edge != 0) return 1;  // (the short tiles only ever hold the library's own copy)
    if constexpr (CGX == 32) {
        if (rdst > 0 && K <= CG * 16) return 1;
    }
    const i64 Nf = src_padded ? (N + V - 1) / V * V : N - N % V;  // whole row packs; the rest is the tail kernel's
    i64 grid = 0;
    if (tail_used) *tail_used = false;
    if (upd_done) *upd_done = false;
    if (Nf > 0) {
        const i64 ntiles = (Nf + R - 1) / R;
        // Workgroups per CU.  Read-only passes: two (5 % faster than one on the caller's column-major X).  Read+write
        // passes on the tiled copy: ONE (2.7 % faster than two at 16 columns per lane, 4 % at 4, equal at 8 -- less
        // in flight is better for the read/write mix, tools/fused_grid_sweep.py); 32 columns per lane (256 VGPRs)
        // never fit two.
        // (512 column groups: the operand vector of a read-only pass is 64 KB of LDS, two workgroups fit; a deflating pass
        // holds p_prev as well, 128 KB)
        // (a SHORT read-only pass at <= 16 columns per lane: one per CU as well -- the ONEWG instantiations, whose tail sums the
        // partial rows and may run the update: one launch per component instead of three)
        // (from a full chip of tiles up to ~1.5 GB: below, the tail loses to the launches -- slice_tail -- above, the 5 % do)
        const bool onewg = !defl && K <= CG * 16 && CGX <= 64 && edge == 0 && tail && tail->cnt && rdst == 0 &&
                           ntiles >= (i64)num_cu && (double)ntiles * R * K * sizeof(T) <= 1.5e9;
        const int per_cu = (K <= CG * 16 && !defl && !onewg) ? 2 : 1;
        grid = grid_hint > 0 ? grid_hint : per_cu * (i64)num_cu;
        grid = std::min<i64>(std::min<i64>(grid, ntiles), max_rows - 1);
        if (grid < 1) return 1;
        const dim3 g((unsigned)grid), b(NT);
        SliceTail st;
        if (tail && tail->cnt && grid >= TAIL_MIN_WG && per_cu == 1 && grid <= (i64)num_cu) {
            st = *tail;
            st.nrows = (int)grid + (Nf < N ? 1 : 0);
            if (tail_used) *tail_used = true;
            // the update in the tail: its block sums and p_j^T w products take the score exchange buffer (2 NW R doubles)
            if (st.upd.XY && upd_done && K <= UPD1_VTHREADS && 2 * (NT / WAVE) * R >= UPD1_VWAVES + st.upd.A) {
                *upd_done = true;
            } else {
                st.upd.XY = nullptr;
                st.gx.n = 0;
            }
        }
        if (Nf < N) {  // the last N % V rows FIRST: their partial row (index `grid`) is complete when the sweep's tail sums the rows
            TailArgs<T> a;
            a.src = X; a.lds = ldx; a.tss = tsx; a.rs = R;
            if (defl) {
                a.dst = dst; a.ldd = ldd; a.tsd = tsd; a.rd = rdst > 0 ? rdst : R;
                a.tprev = tprev; a.pprev = pprev;
            }
            a.v = v; a.tout = tout; a.part = part + grid * K; a.sspart = sspart + grid;
            a.row0 = Nf; a.nrows = (int)(N - Nf); a.K = K; a.zero_rows = V - a.nrows;
            launch_tail_rows(stream, a);
        }
        // a deflating pass IN the library's tiled copy (source and destination in this tile shape: ld = R, ts = R K): the
        // one-descriptor-per-tile form (TILED); the short tiles (CGX > 32) only ever hold that copy
        // Pacing of the in-place deflating sweep: every wave sleeps 2 x 64 cycles per column of its lanes (32 x 64 cycles per
        // 128 KB tile) before it issues a tile's loads.  A read+write sweep is faster with LESS in flight (one workgroup
        // per CU beats two); throttled a little further it gains another 1.5-3 % on every shape measured -- config 3
        // 0.752 -> 0.772 / 0.774 -> 0.787 of peak on two boxes, an eighth of it 0.723 -> 0.736, config 4 0.784 -> 0.799, a
        // shard of config 5 0.773 -> 0.786 -- while 64 x 64 cycles already cost one box 2 % (profiles/r4/pace_sweep_*.txt).
        const bool tiled = defl && edge == 0 && rdst == 0 && ldx == R && ldd == R && tsx == (i64)R * K && tsd == (i64)R * K;
        // Shares of the tiles by XCD class (kernel comment, "weighted walk"): read+write sweeps at one workgroup per CU -- the
        // odd XCDs take rho times as long per tile (measured 1.05-1.10 over configs 3, 4, 5: profiles/r5/pass_stamps_shapes.txt)
        // (bit 16 of the pacing word: the operand vectors go to LDS behind the first tile's loads -- 191.5 -> 190.0 us per component
        // on an eighth of config 3, nothing at full size; pacing stays at 32 x 64 cycles per 128 KB tile on a shard too: 16 / 8 / 0
        // cost it 1.6 / 2.6 / 3.5 us per component -- profiles/r5/small_pass.txt)
        WalkWeights wk;
        {
            // rho 1.0 / 1.05 / 1.08 / 1.11 on one box: config 3 707.4 / 711.5 / 713.7 / 713.2 components/s, an eighth of it 193.3 /
            // 193.7 / 191.0 / 191.7 us per component, config 4 1,385 / 1,400 / 1,401 / 1,400, a shard of config 5 176.9 / 177.6 /
            // 178.6 / 178.4 (profiles/r5/weights_ab.txt): +1 % -- the slow XCDs catch up once the fast ones are done, the
            // sweep is bound by what the memory takes in all
            constexpr double rho = 1.08;
            constexpr unsigned fast_mask = 0x55u;  // XCDs 0, 2, 4, 6
            const int nfc = __builtin_popcount(fast_mask);
            if (defl && per_cu == 1 && grid == (i64)num_cu && grid % 8 == 0 && edge != 2 && ntiles >= 2 * grid) {
                const double share = (double)ntiles / ((double)grid * (1.0 + (rho - 1.0) * nfc / 8.0));
                wk.nfull = (int)std::max<i64>(1, std::min<i64>((i64)(share + 0.5), ntiles / grid));
                wk.mask = fast_mask;
            }
        }
        // (the one-descriptor form for READ-ONLY passes over the copy was built in round 5: 52 bytes of scratch per lane instead
        // of 24 -- the 128-register shape's spills are its 64 tile + 32 accumulator registers, not its descriptors)
        if (CGX > 32 && defl && !tiled) return 1;
#define FUSED_LAUNCH(CPT_, DEFL_, EDGE_, TILED_, dyn_)                                                                    \
    do {                                                                                                                  \
        if constexpr (!(DEFL_) && (CPT_) <= 16 && CGX <= 64 && (EDGE_) == 0) {                                            \
            if (onewg) {                                                                                                  \
                auto kfn1 = &fused_pass_kernel<T, V, R, NT, CPT_, DEFL_, AUX_NT, AUX_NT, false, EDGE_, TILED_, true>;     \
                hipLaunchKernelGGL(kfn1, g, b, dyn_, stream, X, ldx, tsx, dst, ldd, tsd, Nf, K, v, tprev, pprev, tout, part, sspart, 0, N, st, wk); \
                break;                                                                                                    \
            }                                                                                                             \
        }                                                                                                                 \
        auto kfn = &fused_pass_kernel<T, V, R, NT, CPT_, DEFL_, AUX_NT, AUX_NT, false, EDGE_, TILED_>;                   \
        if ((dyn_) > 48 * 1024 && !raise_dynamic_lds(reinterpret_cast<const void *>(kfn), (int)(dyn_))) return 1;         \
        hipLaunchKernelGGL(kfn, g, b, dyn_, stream, X, ldx, tsx, dst, ldd, tsd, Nf, K, v, tprev, pprev, tout, part, sspart, (TILED_) ? (2 * (CPT_)) | 0x10000 : 0, N, st, wk); \
    } while (0)
#define FUSED_EDGE(CPT_, DEFL_, dyn_)                                                                                     \
    do {                                                                                                                  \
        if constexpr (DEFL_) {                                                                                            \
            if (tiled) { FUSED_LAUNCH(CPT_, DEFL_, 0, true, dyn_); break; }                                               \
        }                                                                                                                 \
        if constexpr (CGX <= 32) {                                                                                        \
            if (edge == 2) { FUSED_LAUNCH(CPT_, DEFL_, 2, false, dyn_); break; }                                          \
            if (edge == 1) { FUSED_LAUNCH(CPT_, DEFL_, 1, false, dyn_); break; }                                          \
        }                                                                                                                 \
        if constexpr (CGX <= 32 || !DEFL_) FUSED_LAUNCH(CPT_, DEFL_, 0, false, dyn_);                                     \
    } while (0)
#define FUSED_CASE(CPT_)                                                                                                  \
    do {                                                                                                                  \
        const size_t dyn = ((size_t)2 * CG * CPT_ * sizeof(double) > 48 * 1024) ? (size_t)2 * CG * CPT_ * sizeof(double) : 0; \
        if (defl) FUSED_EDGE(CPT_, true, dyn);                                                                            \
        else FUSED_EDGE(CPT_, false, (CGX >= 512 ? dyn / 2 : dyn));   /* (no p_prev: the second half is never touched) */ \
    } while (0)
        if constexpr (CGX == 32) {
            if (rdst > 0) {  // first deflation into shorter tiles: only the 32-columns-per-lane shape needs it
                auto kfn = edge == 2 ? &fused_pass_kernel<T, V, R, NT, 32, true, AUX_NT, AUX_NT, true, 2>
                                     : (edge == 1 ? &fused_pass_kernel<T, V, R, NT, 32, true, AUX_NT, AUX_NT, true, 1>
                                                  : &fused_pass_kernel<T, V, R, NT, 32, true, AUX_NT, AUX_NT, true, 0>);
                hipLaunchKernelGGL(kfn, g, b, 0, stream, X, ldx, tsx, dst, ldd, tsd, Nf, K, v, tprev, pprev, tout, part, sspart, rdst, N, st, wk);
            } else if (K <= CG * 4) FUSED_CASE(4);
            else if (K <= CG * 8) FUSED_CASE(8);
            else if (K <= CG * 12) FUSED_CASE(12);  // (the in-between shapes: a matrix just above a power-of-two boundary would
            else if (K <= CG * 16) FUSED_CASE(16);  //  run the next shape half empty -- profiles/r4/width_scan.txt)
            else if (K <= CG * 24) FUSED_CASE(24);
            else FUSED_CASE(32);
        } else if constexpr (CGX == 64 || CGX == 128) {
            if (K <= CG * 12) FUSED_CASE(12);
            else FUSED_CASE(16);  // (32 columns per lane on these tiles was the round-1 shape: slower, deleted in round 4)
        } else if constexpr (CGX == 8) {  // narrow matrices: 128-row (fp64) tiles
            if (K <= CG * 4) FUSED_CASE(4);
            else if (K <= CG * 8) FUSED_CASE(8);
            else if (K <= CG * 12) FUSED_CASE(12);
            else FUSED_CASE(16);
        } else if constexpr (CGX == 16) {
            if (K <= CG * 12) FUSED_CASE(12);
            else FUSED_CASE(16);
        } else if constexpr (CGX == 512) {
            // 512 x 1: a tile = ONE row pack of every column, 16 bytes per column (the copy is row-pack-major): K <= 8192
            // at 16 columns per lane; read-only passes up to K = 16384 at 32 (256 VGPRs, one workgroup per CU)
            if (K <= CG * 12) FUSED_CASE(12);
            else if (K <= CG * 16) FUSED_CASE(16);
            else FUSED_EDGE(32, false, (size_t)CG * 32 * sizeof(double));
        } else {
            if (K <= CG * 12) FUSED_CASE(12);
            else FUSED_CASE(16);  // 256 column groups x 2 row lanes: K <= 4096 at 16 columns per lane
        }
#undef FUSED_CASE
#undef FUSED_EDGE
#undef FUSED_LAUNCH
    }
    if (Nf < N) {  // the last N % V rows: one more partial row
        if (Nf <= 0) {  // (fewer rows than one pack: nothing but them)
            TailArgs<T> a;
            a.src = X; a.lds = ldx; a.tss = tsx; a.rs = R;
            if (defl) {
                a.dst = dst; a.ldd = ldd; a.tsd = tsd; a.rd = rdst > 0 ? rdst : R;
                a.tprev = tprev; a.pprev = pprev;
            }
            a.v = v; a.tout = tout; a.part = part + grid * K; a.sspart = sspart + grid;
            a.row0 = Nf; a.nrows = (int)(N - Nf); a.K = K; a.zero_rows = V - a.nrows;
            launch_tail_rows(stream, a);
        }
        ++grid;
    }
    *nb = (int)grid;
    *nss = (int)grid;
    return 0;
}

}  // namespace plsk
