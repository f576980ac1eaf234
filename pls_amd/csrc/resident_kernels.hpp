// resident_kernels.hpp -- the whole fit of a MID-SIZE single-response problem in ONE launch (round 5).
//
// Between the single-launch kernels (tiny_kernels.hpp: N <= 1024 rows, one workgroup) and matrices whose pass takes longer than
// its launches (~50 MB) a component of the general plan costs 13-16 us of launches whatever the data (5,000 x 128, ten
// components: 0.17 ms, of which the passes are 0.01).  Here the single-launch kernel's workgroup -- 1024 threads, its rows of X
// in REGISTERS for the life of the fit, everything K-sized in LDS -- is one of G <= 256, each with a block of RW = 64 wps rows;
// what a component sums over ALL rows (X^T t, t^T t; once X^T Y) meets in one grid-wide exchange per component:
//     every workgroup stores its partial [K + 1] sc1, arrives at ONE counter (agent-scope add), waits until the counter shows
//     everyone (sc1 poll, bounded), and sums the G partials ITSELF in a fixed order -- every workgroup the same bits, so the
//     K-sized bookkeeping (q, the XY deflation, w, the r recurrence: src/pls.cpp:403-416, :427-433) runs replicated, no
//     second exchange; two parities of the partial rows (a workgroup can only be one exchange ahead of the slowest).
// The hand-off is the first row of MI355X_MICROARCH.md's "Valid forms" table (hipMalloc memory, one workgroup per CU -- 1024
// threads at 128 registers fill a CU --, 8-byte sc1 stores and loads, one lane signalling for its workgroup behind a barrier).
// The wait needs all G workgroups RESIDENT: G <= the CU count, and the launcher serialises resident fits of one process; a wait
// beyond the time limit (another process holding CUs) raises a status word that pls_hip_synchronize reports, never a hang.
// M = 1, KERNEL form (X read-only).  Layout inside a workgroup as tiny_fit_kernel: lane = row of a 64-row block, a wave =
// (row block rb, column slice s); K <= (16 / wps) * TINY_RC.
#pragma once
#include "tiny_kernels.hpp"

namespace plsk {

struct ResidentSync {
    unsigned *bar = nullptr;     // the arrival counter of THIS launch, zero at its start
    unsigned *bar_next = nullptr;  // the next launch's counter: zeroed by workgroup 0 (launches alternate between two, so that no
                                   // memset sits in front of a fit and a launch that ended in a time-out leaves nothing behind)
    double *part = nullptr;      // [2 parities][G][LP] partial vectors
    int *status = nullptr;       // host-mapped: set to 1 by a wait that ran out
    long long limit = 0;         // ticks of the wall clock (100 MHz)
    int LP = 0;                  // row stride of part: K + 1 rounded up to 8
};

constexpr int RESIDENT_MAX_WG = 256;

// rows per workgroup = 64 * wps; 0: the shape is not covered
inline int resident_wps(i64 N, int K, int M, int A, i64 ldx, size_t es, int num_cu) {
    if (M != 1 || N < 1 || A > K || K < 1 || (i64)TINY_KMAX * ldx * (i64)es >= (1ll << 31)) return 0;
    if (tiny_fit_covers(N, K, M, A, ldx, es) || micro_fit_covers(N, K, M, A, ldx, es)) return 0;  // (one workgroup / one wave does it)
    if ((size_t)2 * K * A * 8 > TINY_LDS_MAX) return 0;
    // the TALLEST row block whose column slices still hold K columns: the fewest workgroups -- a workgroup's work per component
    // is its 26 register values per thread whatever the block's shape, while the exchange costs by the number of partial vectors
    for (int wps = UPD_WAVES; wps >= 1; wps /= 2) {
        const i64 G = (N + (i64)WAVE * wps - 1) / ((i64)WAVE * wps);
        if (K <= (UPD_WAVES / wps) * TINY_RC && G <= std::min(num_cu, RESIDENT_MAX_WG)) return wps;
    }
    return 0;
}

// grid-wide sum of loc[0 .. L): every workgroup ends with the same totals in tot[0 .. L).  scratch: >= UPD_THREADS doubles.
// phase: this exchange's number, 0, 1, ... (the counter is monotonic: phase p waits for (p + 1) * G arrivals).
// wait = false (an earlier wait of this workgroup ran out): arrive, do not wait again.
__device__ __forceinline__ bool resident_grid_sum(const double *loc, int L, const ResidentSync &sy, unsigned phase, double *tot,
                                                  double *scratch, int *flag, bool wait) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, G = gridDim.x;
    double *mine = sy.part + ((i64)(phase & 1) * G + blockIdx.x) * sy.LP;
    for (int j = tid; j < L; j += UPD_THREADS) st_agent(mine + j, loc[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave, before the barrier behind which one lane signals
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(sy.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (phase + 1u) * (unsigned)G;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(sy.bar, (short)0, 4, BUF_WORD3);
        const long long t0 = wall_clock64();
        int ok = wait ? 1 : 0;
        while (wait && (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rb, 0, 0, AUX_SC1) < target) {
            if (wall_clock64() - t0 > sy.limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *flag = ok;
    }
    __syncthreads();  // (the polling lane has seen everyone: the other waves load behind this barrier)
    const bool ok = *flag != 0;
    if (!ok && tid == 0) __hip_atomic_store(sy.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // P = 1024 / LW partial sums per value (LW = L rounded up to whole waves): thread (j, h) adds the workgroups h, h + P,
    // h + 2 P, ... in that order, then the P partial sums meet in order -- the same association in every workgroup
    const int LW = (L + WAVE - 1) / WAVE * WAVE;
    const int P = UPD_THREADS / LW, j = tid % LW, h = tid / LW;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        sy.part + (i64)(phase & 1) * G * sy.LP, (short)0, (int)((i64)G * sy.LP * 8), BUF_WORD3);
    double s = 0.0;
    if (j < L && h < P) {
        int g = h;
        for (; g + 15 * P < G; g += 16 * P) {  // sixteen loads in flight per lane
            double x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rp, (uint32_t)(((i64)(g + u * P) * sy.LP + j) * 8), 0, AUX_SC1);
                __builtin_memcpy(&x[u], &raw, 8);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += x[u];
        }
        for (; g < G; g += P) {
            const u32x2 raw = __builtin_amdgcn_raw_buffer_load_b64(rp, (uint32_t)(((i64)g * sy.LP + j) * 8), 0, AUX_SC1);
            double x;
            __builtin_memcpy(&x, &raw, 8);
            s += x;
        }
    }
    scratch[tid] = s;
    lds_barrier();
    if (tid < L) {
        double t = 0.0;
        for (int q = 0; q < P; ++q) t += scratch[q * LW + tid];
        tot[tid] = ok ? t : __builtin_nan("");
    }
    lds_barrier();
    return ok;
}

// X: N x K (ld ldx), Y: N x 1; W, P, R: K x A; Q: 1 x A; Tm: N x A (ld ldt); B: K x 1 or null.  grid = G workgroups of
// 64 * wps rows; dynamic LDS: 2 * K * A doubles.
template <typename T>
__global__ __launch_bounds__(UPD_THREADS) void resident_fit_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 N,
                                                                   int K, int A, double *__restrict__ W, double *__restrict__ P,
                                                                   double *__restrict__ Q, double *__restrict__ R,
                                                                   T *__restrict__ Tm, i64 ldt, double *__restrict__ B, int wps,
                                                                   const ResidentSync sy) {
    extern __shared__ double dyn[];
    double *Pl = dyn, *Rl = dyn + (i64)K * A;  // P[:, j], R[:, j] as they are produced
    __shared__ double tp[UPD_THREADS], colp[UPD_WAVES][TINY_RC], praw[TINY_KMAX + 8], tot[TINY_KMAX + 8], wl[TINY_KMAX], vsl[TINY_KMAX];
    __shared__ double cs[TINY_KMAX], ql[TINY_KMAX], sred[2 * UPD_WAVES];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    TinyShape shp(1);
    shp.wps = wps;
    shp.S = UPD_WAVES / wps;
    const i64 row0 = (i64)blockIdx.x * WAVE * wps;
    const int s = wv / shp.wps, rb = wv % shp.wps, il = rb * WAVE + lane;
    const i64 i = row0 + il;
    const bool act = s < shp.S && i < N;
    const int k = tid;
    const bool kok = k < K;
    const bool lead = blockIdx.x == 0;  // writes the K-sized outputs (every workgroup holds the same)
    const int slot = (k % shp.S) * TINY_RC + k / shp.S;

    const uint32_t nrec = (uint32_t)(((i64)(K - 1) * ldx + N) * (i64)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), (short)0, (int)nrec, BUF_WORD3);
    const uint32_t voff = act ? (uint32_t)((i + (i64)s * ldx) * (i64)sizeof(T)) : 0x80000000u;
    const uint32_t cstep = (uint32_t)((i64)shp.S * ldx * (i64)sizeof(T));
    double x[TINY_RC];
#pragma unroll
    for (int j = 0; j < TINY_RC; ++j) x[j] = tiny_ld<T>(rs, voff, (uint32_t)j * cstep);
    for (int c = tid; c < TINY_KMAX; c += UPD_THREADS) vsl[c] = 0.0;
    const double yv = act ? (double)Y[i] : 0.0;
    unsigned phase = 0;
    bool ok = true;
    // outputs through buffer descriptors (tiny_kernels.hpp, out_rsrc): only the first workgroup writes the K-sized ones
    const i64 ka8 = lead ? (i64)K * A * 8 : 0;
    const __amdgpu_buffer_rsrc_t rW = out_rsrc(W, ka8), rP = out_rsrc(P, ka8), rR = out_rsrc(R, ka8);
    const __amdgpu_buffer_rsrc_t rT = out_rsrc(Tm + row0, ((i64)(A - 1) * ldt + min((i64)WAVE * wps, N - row0)) * (i64)sizeof(T));
    const uint32_t kof = kok ? (uint32_t)k * 8u : 0x80000000u;
    const uint32_t tof = (act && s == 0) ? (uint32_t)il * (uint32_t)sizeof(T) : 0x80000000u;

    tiny_column_sums(x, yv, colp, K, shp, praw);  // this workgroup's rows of XY = X^T Y (:396)
    lds_barrier();
    if (lead && tid == 0) __hip_atomic_store(sy.bar_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = resident_grid_sum(praw, K, sy, phase++, tot, tp, &flag, ok);
    double xyk = kok ? tot[k] : 0.0;
    {  // w_0 = XY / |XY| (:404, :411), r_0 = w_0
        const double w = xyk / sqrt(tiny_block_sum(xyk * xyk, sred));
        st_out(rW, kof, 0, w);
        st_out(rR, kof, 0, w);
        if (kok) {
            Rl[k] = w;
            vsl[slot] = w;
        }
        const double c = wave_sum(kok ? w * xyk : 0.0);  // r_0^T XY: see the end of the loop
        if (lane == 0) sred[UPD_WAVES + wv] = c;
    }
    const double *vs = vsl + s * TINY_RC;
    for (int a = 0; a < A; ++a) {
        lds_barrier();  // r_a complete
        double acc = 0.0;  // t = X r (:419)
#pragma unroll
        for (int j = 0; j < TINY_RC; ++j) {
            acc = fma(x[j], vs[j], acc);
            if (j % 8 == 7) asm volatile("" ::: "memory");
        }
        tp[tid] = acc;
        lds_barrier();
        double ti = 0.0;
        if (act)
            for (int q = 0; q < shp.S; ++q) ti += tp[(q * shp.wps + rb) * WAVE + lane];
        ti = (double)(T)ti;  // the score as stored
        st_score<T>(rT, tof, (uint32_t)((i64)a * ldt * (i64)sizeof(T)), ti);
        {  // this workgroup's rows of t^T t (:420): summed by waves here, added up behind the barrier inside the column sums
            const double c = wave_sum((act && s == 0) ? ti * ti : 0.0);
            if (lane == 0) sred[wv] = c;
        }
        tiny_column_sums<false>(x, ti, colp, K, shp, praw);                              // ... and of X^T t (:427); colp, praw: last read barriers ago
        if (tid == K) {  // (thread j of the exchange stores praw[j]: every thread reads what it wrote itself -- no barrier in between)
            double ttl = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < UPD_WAVES; ++w2) ttl += sred[w2];
            praw[K] = ttl;
        }
        ok = resident_grid_sum(praw, K + 1, sy, phase++, tot, tp, &flag, ok);
        const double tt = tot[K];
        const double p = kok ? tot[k] / tt : 0.0;                                                      // (:427)
        double rxy = 0.0;  // r^T XY: its wave sums were left in sred[UPD_WAVES ..] when r_a was formed
#pragma unroll
        for (int w2 = 0; w2 < UPD_WAVES; ++w2) rxy += sred[UPD_WAVES + w2];
        const double q = rxy / tt;                                                                     // q = r^T XY / tt (:428)
        st_out(rP, kof, (uint32_t)a * (uint32_t)K * 8u, p);
        if (kok) Pl[k + (i64)a * K] = p;
        if (tid == 0) {
            if (lead) Q[a] = q;
            ql[a] = q;
        }
        xyk -= (p * q) * tt;  // XY -= (p q^T) tt (:429)
        const int n = a + 1;
        if (n >= A) break;
        // |XY|^2 is one more wave's sum beside the p_j^T XY (no reduction of its own: two barriers fewer per component); the norm
        // then divides both w = XY / |XY| (:404, :411) and the c_j = p_j^T w (:415)
        if (kok) wl[k] = xyk;
        lds_barrier();
        for (int j = wv; j <= n; j += UPD_WAVES) {
            double c = 0.0;
            if (j < n) {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
            } else {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
            }
            c = wave_sum(c);
            if (lane == 0) (j < n ? cs[j] : sred[0]) = c;
        }
        lds_barrier();
        const double inv = 1.0 / sqrt(sred[0]);
        const double w = xyk * inv;
        st_out(rW, kof, (uint32_t)n * (uint32_t)K * 8u, w);
        double r = w;
        for (int j = 0; j < n; ++j) r -= (cs[j] * inv) * Rl[(kok ? k : 0) + (i64)j * K];  // the reference's order (:412-416)
        st_out(rR, kof, (uint32_t)n * (uint32_t)K * 8u, r);
        if (kok) {
            Rl[k + (i64)n * K] = r;
            vsl[slot] = r;
        }
        {  // r_n^T XY, the numerator of the next q (:428): summed by waves now, added up behind the barriers of the next exchange
            const double c = wave_sum(kok ? r * xyk : 0.0);
            if (lane == 0) sred[UPD_WAVES + wv] = c;
        }
    }
    lds_barrier();
    if (B && kok && lead) {  // B = R Q^T (:444-451)
        double b = 0.0;
        for (int a = 0; a < A; ++a) b = fma(Rl[k + (i64)a * K], ql[a], b);
        B[k] = b;
    }
    (void)ok;
}


// ---------------------------------------------------------------------------------------------------------------------
// The same one launch for 2 <= M <= 8 responses: tiny_fit_m_kernel's workgroup (XY of K x M in LDS, the direction by the one-wave
// eigen solver) x G, with X^T Y exchanged once -- the K M values in pieces of at most 1024 -- and [X^T t, t^T t] per component.
// Dynamic LDS: (2 K + M) A doubles.
// ---------------------------------------------------------------------------------------------------------------------
inline int resident_m_wps(i64 N, int K, int M, int A, i64 ldx, size_t es, int num_cu) {
    if (M < 2 || M > 8 || N < 1 || A > K || K < 1 || (i64)TINY_KMAX * ldx * (i64)es >= (1ll << 31)) return 0;
    if (tiny_fit_m_covers(N, K, M, A, ldx, es) || micro_fit_covers(N, K, M, A, ldx, es)) return 0;
    if ((size_t)(2 * K + M) * A * 8 > TINY_LDS_MAX) return 0;
    for (int wps = UPD_WAVES; wps >= 1; wps /= 2) {
        const i64 G = (N + (i64)WAVE * wps - 1) / ((i64)WAVE * wps);
        if (K <= (UPD_WAVES / wps) * TINY_RC && G <= std::min(num_cu, RESIDENT_MAX_WG)) return wps;
    }
    return 0;
}

template <typename T, int MM>
__global__ __launch_bounds__(UPD_THREADS) void resident_fit_m_kernel(const T *__restrict__ X, i64 ldx, const T *__restrict__ Y, i64 ldy,
                                                                     i64 N, int K, int M, int A, int power_iters,
                                                                     double *__restrict__ W, double *__restrict__ P,
                                                                     double *__restrict__ Q, double *__restrict__ R,
                                                                     T *__restrict__ Tm, i64 ldt, double *__restrict__ B, int wps,
                                                                     const ResidentSync sy) {
    static_assert(MM * MM <= WAVE, "one wave solves the eigenproblem");
    extern __shared__ double dyn[];
    double *Pl = dyn, *Rl = dyn + (i64)K * A, *Ql = dyn + 2 * (i64)K * A;  // P[:, j], R[:, j], Q[:, j] as they are produced
    __shared__ double tp[UPD_THREADS], colp[UPD_WAVES][TINY_RC], praw[TINY_KMAX + 8], tot[UPD_THREADS], xy[MM][TINY_KMAX], wl[TINY_KMAX],
        vsl[TINY_KMAX];
    __shared__ double cs[TINY_KMAX], sred[UPD_WAVES], Gs[MM * MM], Bs[MM * MM], Cs[MM * MM], qs[MM], qa[MM];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    TinyShape shp(1);
    shp.wps = wps;
    shp.S = UPD_WAVES / wps;
    const i64 row0 = (i64)blockIdx.x * WAVE * wps;
    const int s = wv / shp.wps, rb = wv % shp.wps, il = rb * WAVE + lane;
    const i64 i = row0 + il;
    const bool act = s < shp.S && i < N;
    const int k = tid;
    const bool kok = k < K;
    const bool lead = blockIdx.x == 0;
    const int slot = (k % shp.S) * TINY_RC + k / shp.S;

    const uint32_t nrec = (uint32_t)(((i64)(K - 1) * ldx + N) * (i64)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(X), (short)0, (int)nrec, BUF_WORD3);
    const uint32_t voff = act ? (uint32_t)((i + (i64)s * ldx) * (i64)sizeof(T)) : 0x80000000u;
    const uint32_t cstep = (uint32_t)((i64)shp.S * ldx * (i64)sizeof(T));
    double x[TINY_RC];
#pragma unroll
    for (int j = 0; j < TINY_RC; ++j) x[j] = tiny_ld<T>(rs, voff, (uint32_t)j * cstep);
    for (int c = tid; c < TINY_KMAX; c += UPD_THREADS) vsl[c] = 0.0;
    unsigned phase = 0;
    bool ok = true;
    if (lead && tid == 0) __hip_atomic_store(sy.bar_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // XY = X^T Y (:396): this workgroup's rows, response by response; then the K M values meet in pieces of <= 1024
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m < M) {
            const double yv = act ? (double)Y[i + (i64)m * ldy] : 0.0;
            tiny_column_sums(x, yv, colp, K, shp, xy[m]);
        } else if (kok) {
            xy[m][k] = 0.0;
        }
    }
    lds_barrier();
    {
        const int L = K * M;
        for (int f0 = 0; f0 < L; f0 += UPD_THREADS) {
            const int n = min(UPD_THREADS, L - f0);
            if (tid < n) tot[tid] = xy[(f0 + tid) / K][(f0 + tid) % K];  // (the exchange reads its input before it writes its output)
            lds_barrier();
            ok = resident_grid_sum(tot, n, sy, phase++, tot, tp, &flag, ok);
            if (tid < n) xy[(f0 + tid) / K][(f0 + tid) % K] = tot[tid];
            lds_barrier();
        }
    }
    const double *vs = vsl + s * TINY_RC;
    for (int a = 0; a < A; ++a) {
        // ---- direction (:403-411): G = XY^T XY, one wave per pair (i <= j)
        lds_barrier();  // XY complete
        for (int pr = wv; pr < MM * (MM + 1) / 2; pr += UPD_WAVES) {
            int gi = 0, rem = pr;
            while (rem >= MM - gi) { rem -= MM - gi; ++gi; }
            const int gj = gi + rem;
            double g = 0.0;
            for (int kk = lane; kk < K; kk += WAVE) g = fma(xy[gi][kk], xy[gj][kk], g);
            g = wave_sum(g);
            if (lane == 0) { Gs[gi + gj * MM] = g; Gs[gj + gi * MM] = g; }
        }
        lds_barrier();
        if (wv == 0) dominant_eigvec_wave<MM>(Gs, Bs, Cs, qs, power_iters);
        lds_barrier();
        double wk = 0.0;
#pragma unroll
        for (int m = 0; m < MM; ++m) wk = fma(kok ? xy[m][k] : 0.0, qs[m], wk);  // w = XY q (:408)
        // |XY q|^2 is one more wave's sum beside the p_j^T (XY q) (no reduction of its own: two barriers fewer per component); the
        // norm then divides both w (:411) and the c_j = p_j^T w (:415)
        if (kok) wl[k] = wk;
        lds_barrier();
        for (int j = wv; j <= a; j += UPD_WAVES) {
            double c = 0.0;
            if (j < a) {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(Pl[kk + (i64)j * K], wl[kk], c);
            } else {
                for (int kk = lane; kk < K; kk += WAVE) c = fma(wl[kk], wl[kk], c);
            }
            c = wave_sum(c);
            if (lane == 0) (j < a ? cs[j] : sred[0]) = c;
        }
        lds_barrier();
        const double inv = 1.0 / sqrt(sred[0]);
        wk *= inv;
        if (kok && lead) W[k + (i64)a * K] = wk;
        double r = wk;
        for (int j = 0; j < a; ++j) r -= (cs[j] * inv) * Rl[(kok ? k : 0) + (i64)j * K];  // the reference's order (:412-416)
        if (kok) {
            if (lead) R[k + (i64)a * K] = r;
            Rl[k + (i64)a * K] = r;
            vsl[slot] = r;
        }
        lds_barrier();  // r_a complete
        // ---- score, loading (:419-427)
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < TINY_RC; ++j) {
            acc = fma(x[j], vs[j], acc);
            if (j % 8 == 7) asm volatile("" ::: "memory");
        }
        tp[tid] = acc;
        lds_barrier();
        double ti = 0.0;
        if (act)
            for (int q = 0; q < shp.S; ++q) ti += tp[(q * shp.wps + rb) * WAVE + lane];
        ti = (double)(T)ti;  // the score as stored
        if (act && s == 0) Tm[i + (i64)a * ldt] = (T)ti;
        {  // this workgroup's rows of t^T t (:420): summed by waves here, added up behind the barrier inside the column sums
            const double c = wave_sum((act && s == 0) ? ti * ti : 0.0);
            if (lane == 0) sred[wv] = c;
        }
        tiny_column_sums<false>(x, ti, colp, K, shp, praw);                       // ... and of X^T t (:421); colp, praw: last read barriers ago
        if (tid == K) {  // (thread j of the exchange stores praw[j]: every thread reads what it wrote itself -- no barrier in between)
            double ttl = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < UPD_WAVES; ++w2) ttl += sred[w2];
            praw[K] = ttl;
        }
        ok = resident_grid_sum(praw, K + 1, sy, phase++, tot, tp, &flag, ok);
        const double tt = tot[K];
        const double p = kok ? tot[k] / tt : 0.0;  // (:427)
        if (kok) {
            if (lead) P[k + (i64)a * K] = p;
            Pl[k + (i64)a * K] = p;
        }
        // ---- q = XY^T r / tt (:428): one wave per response
        if (wv < MM) {
            double c = 0.0;
            for (int kk = lane; kk < K; kk += WAVE) c = fma(Rl[kk + (i64)a * K], xy[wv][kk], c);
            c = wave_sum(c) / tt;
            if (lane == 0) {
                qa[wv] = c;
                if (wv < M) {
                    Ql[wv + (i64)a * M] = c;
                    if (lead) Q[wv + (i64)a * M] = c;
                }
            }
        }
        lds_barrier();
        if (kok) {
#pragma unroll
            for (int m = 0; m < MM; ++m) xy[m][k] -= (p * qa[m]) * tt;  // XY -= (p q^T) tt (:429)
        }
    }
    lds_barrier();
    if (B && kok && lead)  // B = R Q^T (:444-447)
        for (int m = 0; m < M; ++m) {
            double b = 0.0;
            for (int a = 0; a < A; ++a) b = fma(Rl[k + (i64)a * K], Ql[m + (i64)a * M], b);
            B[k + (i64)m * K] = b;
        }
    (void)ok;
}

}  // namespace plsk
