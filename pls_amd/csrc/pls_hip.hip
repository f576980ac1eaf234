// pls_hip.hip -- the C-ABI of include/pls_hip.h over the gfx950 kernels in this directory.
// Host side of the library: argument checks, workspace, launch geometry, the A-loop of
// Model::plsr (src/pls.cpp:390-437) enqueued on one HIP stream with no host round trip,
// the injected all-reduce for row-sharded fits, HIP-event profiling.
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pls_hip.h"
#include "fused_kernels.hpp"
#include "defer_kernels.hpp"
#include "small_kernels.hpp"
#include "coop_update.hpp"
#include "wide1_update.hpp"
#include "largem_kernels.hpp"
#include "tiny_kernels.hpp"
#include "stream_kernels.hpp"
#include "syrk_kernels.hpp"
#include "cv_kernels.hpp"
#include "synth_kernels.hpp"
#include "host_pipeline.hpp"
#include "exchange_kernels.hpp"

using plsk::i64;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct Launch {
    int fam;
    i64 bytes;
    hipEvent_t e0, e1;
};

}  // namespace

struct pls_hip_context {
    int device = 0;
    hipStream_t stream = nullptr;
    pls_hip_allreduce_fn reducer = nullptr;
    void *reducer_user = nullptr;
    int rank = 0, nranks = 1;
    double *user_red = nullptr;
    i64 user_red_count = 0;
    i64 opt_algo = PLS_HIP_ALGO_KERNEL, opt_fuse = 1, opt_profile = 0, opt_power_iters = 48;
    i64 opt_fused_grid = 0, opt_work_layout = 1, opt_defer = 1, opt_graph = 0;
    // PLS_HIP_OPT_GRAPH: the last repeated device-memory fit as an executable graph (one entry: a loop of identical fits)
    std::vector<uint64_t> graph_key, graph_seen;
    hipGraphExec_t graph_exec = nullptr;
    DevBuf zeros, part, sspart, xbpart, wide1, red, red2, xx, xyp, praw, xy, v, cs, coop, lm, gxx, gxy, tab, work, cvidx, cvx, cvy, cvws, cve, cvtx, cvty, cvtt, cvm, cvkeep, hX, hY, hT, hW, hP, hQ, hR, hB, hIn, hOut;
    std::string err;
    // profiling
    std::vector<hipEvent_t> ev_pool;  // grows until pls_hip_get_timing harvests and recycles it
    size_t ev_used = 0;
    std::vector<Launch> launches;     // every bracketed launch since the last harvest
    std::vector<Launch> fits;         // one bracket per pls_hip_fit since the last harvest
    Launch cur_fit{};
    bool fit_timed = false;
    int num_cu = 256;
    // host <-> device staging (host_pipeline.hpp): pinned double buffer + copy threads, created on first large transfer
    plsh::Stager stager;
    int copy_threads = 0;  // 0 = default (PLS_HIP_COPY_THREADS or min(16, cores/2))
    hipStream_t copy_stream = nullptr;  // transfers that run beside kernels of `stream` (upload_accumulate)
    // X^T X (K x K) and X^T Y (K x M) of THIS member's rows, already formed while the rows were uploaded
    // (upload_accumulate): a fit that may use the Gram plan takes them instead of two passes over X
    const double *pre_xx = nullptr, *pre_xy = nullptr;
    // Members of a group: workspace that has to grow is not freed on the spot -- hipFree waits for the whole DEVICE, and a
    // member that shares its GPU with others (virtual shards) would wait for a peer's exchange kernel that in turn waits for
    // this member's next collective.  The old blocks are released when no member is running (run_members).
    bool defer_free = false;
    std::vector<void *> graveyard;
    // cross-process device-side exchange (pls_hip_xchg_*, exchange_kernels.hpp)
    struct XchgIpc *xchg = nullptr;
    // This member's view of the device-side exchange it belongs to, whoever owns the inboxes (XchgIpc: one process per
    // GPU; pls_hip_group: the members of one process).  With it the per-component collective of a fused fit is not a
    // call of the reducer: the push rides in the tail of the pass, the gather is the prologue of the component update.
    struct XchgEndpoint {
        bool on = false;
        int n = 0, rank = 0;
        double *const *inbox = nullptr;             // [n] the members' inboxes (peer-visible addresses)
        unsigned long long *const *flags = nullptr;  // [n] their flags
        unsigned long long *seq = nullptr;           // this member's running collective number
        int *status = nullptr, *host_status = nullptr;
        const long long *limit = nullptr;
    } xep;
    DevBuf tailcnt;  // arrival counters of slice_tail (fused_kernels.hpp)
    // replica guard of sharded fits (small_kernels.hpp): host-mapped flag "the ranks derived different W/P/Q/R/B"
    int *diverged = nullptr, *diverged_dev = nullptr;
    DevBuf guard;
};

namespace {

#define HIPCHK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);               \
            return PLS_HIP_ERR_DEVICE;                                                     \
        }                                                                                  \
    } while (0)

#define CHK(expr)                      \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != PLS_HIP_OK) return rc__; \
    } while (0)

int fail(pls_hip_context *c, int code, const std::string &msg) {
    c->err = msg;
    return code;
}

int ensure(pls_hip_context *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.bytes && b.p) return PLS_HIP_OK;
    if (b.p) {
        if (c->defer_free) {
            c->graveyard.push_back(b.p);
        } else {
            HIPCHK(c, hipStreamSynchronize(c->stream));  // earlier launches may still read it
            HIPCHK(c, hipFree(b.p));
        }
        b.p = nullptr;
        b.bytes = 0;
    }
    bytes = std::max<size_t>(bytes, 256);
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        return fail(c, PLS_HIP_ERR_ALLOC, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    b.bytes = bytes;
    return PLS_HIP_OK;
}

size_t esize(int dtype) { return dtype == PLS_HIP_F64 ? 8 : 4; }

// ---- profiling ------------------------------------------------------------------------
hipEvent_t take_event(pls_hip_context *c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}
struct Scope {  // brackets one launch with events when profiling is on
    pls_hip_context *c;
    Launch l{};
    bool on;
    // profile level 1 brackets only the streaming kernels over X (the K-sized bookkeeping kernels
    // run unbracketed, so that event records do not widen the gaps of the A-loop); level 2: all
    Scope(pls_hip_context *ctx, int fam, i64 bytes)
        : c(ctx), on(ctx->opt_profile >= 2 || (ctx->opt_profile == 1 && fam != PLS_HIP_FAM_SMALL)) {
        if (!on) return;
        l.fam = fam;
        l.bytes = bytes;
        l.e0 = take_event(c);
        l.e1 = take_event(c);
        if (!l.e0 || !l.e1) { on = false; return; }
        (void)hipEventRecord(l.e0, c->stream);
    }
    ~Scope() {
        if (!on) return;
        (void)hipEventRecord(l.e1, c->stream);
        c->launches.push_back(l);
    }
};

// ---- tracing: roctx ranges around the phases of a fit -----------------------------------------------------
// The reference has no tracing.  With PLS_HIP_ROCTX=1 in the environment every fit is wrapped
// in roctx ranges -- "pls_hip_fit", "X^T Y", "X^T X (SYRK)", "component a", "upload" -- which `rocprofv3 --marker-trace`
// shows next to the kernels.  The marker library (librocprofiler-sdk-roctx.so) is looked up at run time: the product
// has no link-time dependency on it and the ranges cost nothing when the switch is off.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("PLS_HIP_ROCTX");
        if (!e || atoi(e) == 0) return;
        void *lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
inline Roctx &roctx() {
    static Roctx r;
    return r;
}
struct Range {  // RAII range; `text` must outlive the call only
    bool on;
    explicit Range(const char *text) : on(roctx().push != nullptr) {
        if (on) roctx().push(text);
    }
    Range(const char *prefix, int n) : on(roctx().push != nullptr) {
        if (on) {
            char buf[64];
            std::snprintf(buf, sizeof(buf), "%s %d", prefix, n);
            roctx().push(buf);
        }
    }
    ~Range() {
        if (on) roctx().pop();
    }
};

#define LAUNCH_CHECK(ctx)                                                           \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e__);   \
            return PLS_HIP_ERR_DEVICE;                                              \
        }                                                                           \
    } while (0)

template <typename T>
bool vec_ok(const void *p, i64 ld, int vec) {
    return ((uintptr_t)p % (sizeof(T) * vec) == 0) && (ld % vec == 0);
}

// ---- geometry -------------------------------------------------------------------------
constexpr int XTY_KCMT = 32;  // accumulators per lane in xty_kernel
constexpr int DEFL_KC = 32;

struct XtyGeom {
    int G;    // row groups = number of partial rows (same for every m-tile of one product)
    int nkg;  // column groups of this m-tile
};
// All m-tiles of one X^T Y write the same number of partial rows G.  It is derived from the column groups of the
// product's FIRST tile (kc_first columns each; the first tile is the widest in m, i.e. the one with the most column
// groups): with G from the 32-column shape instead, the 8-response tile of config 4 ran 16,384 workgroups of 8 row
// chunks each and spent half its time in their 32 butterfly sums (1.0 ms = 2.1 TB/s, fp32 and fp64 alike).
XtyGeom xty_geom(i64 N, int K, int KC, int vec, int target_wgs, int kc_first) {
    XtyGeom g;
    g.nkg = (K + KC - 1) / KC;
    const int nkg32 = (K + kc_first - 1) / kc_first;
    const i64 nch = (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec);
    i64 G = std::max<i64>(1, target_wgs / nkg32);
    G = std::min<i64>(G, std::max<i64>(nch, 1));
    g.G = (int)G;
    return g;
}
// upper bound of partial rows any product of this fit can write
i64 max_partial_rows(pls_hip_context *c, i64 N, int K) {
    const int nkg32 = (K + XTY_KCMT - 1) / XTY_KCMT;
    const i64 nch = (N + plsk::WG - 1) / plsk::WG;  // vec = 1 is the worst case
    const i64 G = std::min<i64>(std::max<i64>(1, (8 * c->num_cu) / nkg32), std::max<i64>(nch, 1));
    return std::max<i64>(G, std::max<i64>(8 * (i64)c->num_cu, c->opt_fused_grid));
}

// ---- typed launchers --------------------------------------------------------------------
template <typename T, int VEC, int MT>
void launch_xb_t(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const double *Bm, i64 ldb,
                 int ncols, T *out, i64 ldo, double *sspart, int *nss) {
    const i64 per = (i64)plsk::WG * VEC;
    const int nblk = (int)((N + per - 1) / per);
    if (sspart && MT == 1) {
        hipLaunchKernelGGL((plsk::xb_kernel<T, VEC, 1, true>), dim3(nblk), dim3(plsk::WG), 0,
                           c->stream, X, ldx, N, K, Bm, ldb, ncols, out, ldo, sspart);
        *nss = nblk;
    } else {
        hipLaunchKernelGGL((plsk::xb_kernel<T, VEC, MT, false>), dim3(nblk), dim3(plsk::WG), 0,
                           c->stream, X, ldx, N, K, Bm, ldb, ncols, out, ldo, (double *)nullptr);
    }
}

// out(N x C) = X * Bm ; optionally sum of squares partials of column 0 (C must be 1 then)
template <typename T>
int launch_xb(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const double *Bm, i64 ldb,
              int C, T *out, i64 ldo, double *sspart, int *nss) {
    constexpr int FV = 16 / sizeof(T);
    bool wide = vec_ok<T>(X, ldx, FV) && vec_ok<T>(out, ldo, FV);
    // keep >= ~4 workgroups per CU in flight: narrow the per-lane access on short matrices
    if (wide && N / ((i64)FV * plsk::WG) < 4 * (i64)c->num_cu) wide = false;
    // One score column of a short, wide matrix: the rows alone give fewer workgroups than there are CUs -- split the
    // columns as well (xb_split_kernel); ~3 workgroups per CU, at least 128 columns each.
    static const bool split_on = !(getenv("PLS_HIP_XB_SPLIT") && atoi(getenv("PLS_HIP_XB_SPLIT")) == 0);
    if (split_on && N > 0 && K >= 1024) {
        const bool v2 = vec_ok<T>(X, ldx, FV) && (N + (i64)FV * plsk::WG - 1) / ((i64)FV * plsk::WG) >= 8;
        const i64 per = (i64)plsk::WG * (v2 ? FV : 1);
        const i64 rg = (N + per - 1) / per;
        // taken while one row per lane cannot give every CU a workgroup (fp32: two -- its 4-byte accesses stream worse);
        // measured per shape, tools/xb_split_sweep.py: beyond that the row-parallel kernel is as fast or faster
        const i64 rg1 = (N + plsk::WG - 1) / plsk::WG;
        if (rg1 <= (i64)(sizeof(T) == 4 ? 2 : 1) * c->num_cu) {
            int KS = (int)std::min<i64>(K / 128, (3 * (i64)c->num_cu + rg - 1) / rg);
            const int kper = (K + KS - 1) / KS;
            KS = (K + kper - 1) / kper;
            const i64 ldp = (N + 63) / 64 * 64;
            const int mt = C > 2 ? 4 : (C > 1 ? 2 : 1);  // columns per sweep of X
            if (KS >= 2 && KS <= 65535 && ensure(c, c->xbpart, (size_t)KS * mt * ldp * 8) == PLS_HIP_OK) {
                double *xp = (double *)c->xbpart.p;
                const int fb = (int)((N + 63) / 64);
                for (int c0 = 0; c0 < C; c0 += mt) {
                    const int use = std::min(mt, C - c0);
                    const double *b = Bm + (i64)c0 * ldb;
                    Scope s(c, PLS_HIP_FAM_XB, (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8);
                    const dim3 g((unsigned)rg, (unsigned)KS), blk(plsk::WG);
#define XS_CASE(V_, M_) hipLaunchKernelGGL((plsk::xb_split_kernel<T, V_, M_>), g, blk, 0, c->stream, X, ldx, N, K, kper, b, ldb, use, xp, ldp)
                    if (v2) { if (mt == 4) XS_CASE(FV, 4); else if (mt == 2) XS_CASE(FV, 2); else XS_CASE(FV, 1); }
                    else { if (mt == 4) XS_CASE(1, 4); else if (mt == 2) XS_CASE(1, 2); else XS_CASE(1, 1); }
#undef XS_CASE
                    LAUNCH_CHECK(c);
                    hipLaunchKernelGGL((plsk::xb_split_finish_kernel<T>), dim3(fb, use), blk, 0, c->stream, (const double *)xp, ldp, KS, mt,
                                       N, out + (i64)c0 * ldo, ldo, C == 1 ? sspart : (double *)nullptr);
                    LAUNCH_CHECK(c);
                }
                if (C == 1 && sspart && nss) *nss = fb;
                return PLS_HIP_OK;
            }
            c->err.clear();
        }
    }
    int c0 = 0;
    while (c0 < C) {
        const int rem = C - c0;
        const double *b = Bm + (i64)c0 * ldb;
        T *o = out + (i64)c0 * ldo;
        const int cap = wide ? (FV == 2 ? 32 : 8) : 32;  // fp32 x 4 rows per lane: 8 columns = 32 fp64 accumulators
        if (sizeof(T) == 4 && rem > 8 && vec_ok<T>(X, ldx, FV)) {
            // fp32 storage, many columns: up to 32 per pass on the matrix cores (xb_mfma_kernel) -- the LDS-staged
            // VALU kernel below holds only 8 columns of fp64 accumulators per pass at 4 rows per lane.  (For fp64
            // storage, where it takes 32 columns per pass, it is the faster one: 0.86 vs 1.04 ms at 20 columns.)
            const int use = std::min(rem, 32);
            const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
            Scope s(c, PLS_HIP_FAM_XB, bytes);
            const i64 per = (i64)(plsk::WG / plsk::WAVE) * 16 * FV;  // rows per workgroup
            const dim3 grid((unsigned)((N + per - 1) / per)), blk(plsk::WG);
            if (use > 16)
                hipLaunchKernelGGL((plsk::xb_mfma_kernel<T, FV, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
            else
                hipLaunchKernelGGL((plsk::xb_mfma_kernel<T, FV, 1>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo);
            LAUNCH_CHECK(c);
            c0 += use;
            continue;
        }
        if (rem > 4) {
            // many columns: Bm through LDS, up to `cap` columns per pass over X; the tile is the column
            // count rounded up to a multiple of 4 (every extra column costs VEC fp64 FMAs per element)
            const int use = std::min(rem, cap);
            const int mtc = (use + 3) & ~3;
            const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
            Scope s(c, PLS_HIP_FAM_XB, bytes);
            // fp64, 13..20 columns on a large matrix: two row packs per lane (one LDS read of a B value feeds 4 FMAs)
            static const int np_env = getenv("PLS_HIP_XB_NP") ? atoi(getenv("PLS_HIP_XB_NP")) : 2;
            const bool two = wide && FV == 2 && mtc >= 16 && mtc <= 20 && np_env == 2 && N >= (i64)c->num_cu * 4 * plsk::WG * FV * 2;
            const i64 per = (i64)plsk::WG * (wide ? FV : 1) * (two ? 2 : 1);
            const dim3 grid((unsigned)((N + per - 1) / per)), blk(plsk::WG);
            if (two) {
                if constexpr (FV == 2) {
                    switch (mtc) {
                        case 16: hipLaunchKernelGGL((plsk::xb_wide_kernel<T, 2, 16, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo); break;
                        default: hipLaunchKernelGGL((plsk::xb_wide_kernel<T, 2, 20, 2>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo); break;
                    }
                }
                LAUNCH_CHECK(c);
                c0 += use;
                continue;
            }
#define XW_CASE(V, M_) hipLaunchKernelGGL((plsk::xb_wide_kernel<T, V, M_>), grid, blk, 0, c->stream, X, ldx, N, K, b, ldb, use, o, ldo)
#define XW_SWITCH(V)                                   \
    switch (mtc) {                                     \
        case 8: XW_CASE(V, 8); break;                  \
        case 12: XW_CASE(V, 12); break;                \
        case 16: XW_CASE(V, 16); break;                \
        case 20: XW_CASE(V, 20); break;                \
        case 24: XW_CASE(V, 24); break;                \
        case 28: XW_CASE(V, 28); break;                \
        default: XW_CASE(V, 32); break;                \
    }
            if (wide) {
                if constexpr (FV == 2) {
                    XW_SWITCH(FV)
                } else {
                    XW_CASE(FV, 8);
                }
            } else {
                XW_SWITCH(1)
            }
#undef XW_SWITCH
#undef XW_CASE
            LAUNCH_CHECK(c);
            c0 += use;
            continue;
        }
        const int mt = rem > 2 ? 4 : rem > 1 ? 2 : 1;
        const int use = std::min(mt, rem);
        const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * use * sizeof(T) + (i64)K * use * 8;
        Scope s(c, PLS_HIP_FAM_XB, bytes);
#define XB_CASE(V, M_) launch_xb_t<T, V, M_>(c, X, ldx, N, K, b, ldb, use, o, ldo, sspart, nss)
        if (wide) {
            if (mt == 4) XB_CASE(FV, 4); else if (mt == 2) XB_CASE(FV, 2); else XB_CASE(FV, 1);
        } else {
            if (mt == 4) XB_CASE(1, 4); else if (mt == 2) XB_CASE(1, 2); else XB_CASE(1, 1);
        }
#undef XB_CASE
        LAUNCH_CHECK(c);
        c0 += use;
    }
    return PLS_HIP_OK;
}

template <typename T, int VEC, int KC, int MT>
void launch_xty_t(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
                  int m0, double *part, const XtyGeom &g) {
    hipLaunchKernelGGL((plsk::xty_kernel<T, VEC, KC, MT>), dim3(g.G, g.nkg), dim3(plsk::WG), 0,
                       c->stream, X, ldx, Y, ldy, N, K, M, m0, part);
}

// part[G][K*M] = per-row-group partials of X^T Y; returns G through *nb
template <typename T>
int launch_xty(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
               double *part, int *nb) {
    constexpr int FV = 16 / sizeof(T);
    const bool wide = vec_ok<T>(X, ldx, FV) && vec_ok<T>(Y, ldy, FV);
    // workgroups per CU aimed at: 8; 4 with 8-response tiles, whose 32 butterfly sums per workgroup want longer walks
    // (config 4, fp32: 0.55 ms at 4, 0.59 at 8, 0.77 at 32 -- tools/xty_m8.py)
    static const int tgt_env = getenv("PLS_HIP_XTY_TGT") ? atoi(getenv("PLS_HIP_XTY_TGT")) : 0;
    const int target = (tgt_env > 0 ? tgt_env : (M >= 8 ? 4 : 8)) * c->num_cu;
    int m0 = 0;
    const int kc_first = XTY_KCMT / ((M >= 8) ? 8 : (M >= 4 ? 4 : (M >= 2 ? 2 : 1)));
    while (m0 < M) {
        const int mt = (M - m0 >= 8) ? 8 : (M - m0 >= 4 ? 4 : (M - m0 >= 2 ? 2 : 1));
        // 8 responses: 8 columns per workgroup (64 accumulators per lane) -- the Y packs of a row chunk are loaded once
        // per column group, so 4 columns meant twice as many bytes of Y as of X through L2 (1.0 ms = 2.1 TB/s at config 4)
        static const int kc8 = getenv("PLS_HIP_XTY_KC8") ? atoi(getenv("PLS_HIP_XTY_KC8")) : 4;
        static const bool xty8 = !(getenv("PLS_HIP_XTY8") && atoi(getenv("PLS_HIP_XTY8")) == 0);
        const int kc = mt == 8 ? kc8 : XTY_KCMT / mt;
        const XtyGeom g = xty_geom(N, K, kc, wide ? FV : 1, target, kc_first);
        *nb = g.G;
        const i64 bytes = (i64)N * K * sizeof(T) + (i64)N * mt * sizeof(T) + (i64)K * mt * 8;
        Scope s(c, PLS_HIP_FAM_XTY, bytes);
#define XTY_CASE(V, KC_, M_) launch_xty_t<T, V, KC_, M_>(c, X, ldx, Y, ldy, N, K, M, m0, part, g)
        if (wide && mt == 8 && kc == 4 && K % 4 == 0 && xty8) {
            hipLaunchKernelGGL((plsk::xty8_kernel<T, FV>), dim3(g.G, g.nkg), dim3(plsk::WG), 0, c->stream, X, ldx, Y, ldy, N, K,
                               M, m0, part);
        } else if (wide) {
            if (mt == 8 && kc == 8) XTY_CASE(FV, 8, 8); else if (mt == 8 && kc == 16) XTY_CASE(FV, 16, 8);
            else if (mt == 8) XTY_CASE(FV, 4, 8); else if (mt == 4) XTY_CASE(FV, 8, 4);
            else if (mt == 2) XTY_CASE(FV, 16, 2); else XTY_CASE(FV, 32, 1);
        } else {
            if (mt == 8 && kc != 4) XTY_CASE(1, 8, 8); else if (mt == 8) XTY_CASE(1, 4, 8); else if (mt == 4) XTY_CASE(1, 8, 4);
            else if (mt == 2) XTY_CASE(1, 16, 2); else XTY_CASE(1, 32, 1);
        }
#undef XTY_CASE
        LAUNCH_CHECK(c);
        m0 += mt;
    }
    return PLS_HIP_OK;
}

template <typename T>
int launch_deflate(pls_hip_context *c, const T *src, i64 lds, T *dst, i64 ldd, i64 N, int K,
                   const T *t, const double *p) {
    constexpr int FV = 16 / sizeof(T);
    const bool wide = vec_ok<T>(src, lds, FV) && vec_ok<T>(dst, ldd, FV) && vec_ok<T>(t, FV, FV);
    const int nkg = (K + DEFL_KC - 1) / DEFL_KC;
    const int vec = wide ? FV : 1;
    const i64 nch = (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec);
    const i64 G = std::min<i64>(std::max<i64>(nch, 1), std::max<i64>(1, (16 * c->num_cu) / nkg));
    const i64 bytes = 2 * (i64)N * K * sizeof(T) + (i64)N * sizeof(T) + (i64)K * 8;
    Scope s(c, PLS_HIP_FAM_DEFLATE, bytes);
    const i64 nrb = (N + (i64)plsk::WG * FV - 1) / ((i64)plsk::WG * FV);
    if (wide && K <= 65535 && nrb >= 1 && nrb < (1ll << 31)) {  // one 4 KB column piece per workgroup
        hipLaunchKernelGGL((plsk::deflate_piece_kernel<T, FV>), dim3((unsigned)nrb, (unsigned)K), dim3(plsk::WG), 0,
                           c->stream, src, lds, dst, ldd, N, t, p);
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    if (wide)
        hipLaunchKernelGGL((plsk::deflate_kernel<T, FV, DEFL_KC>), dim3((unsigned)G, nkg),
                           dim3(plsk::WG), 0, c->stream, src, lds, dst, ldd, N, K, t, p);
    else
        hipLaunchKernelGGL((plsk::deflate_kernel<T, 1, DEFL_KC>), dim3((unsigned)G, nkg),
                           dim3(plsk::WG), 0, c->stream, src, lds, dst, ldd, N, K, t, p);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

int launch_reduce(pls_hip_context *c, const double *part, int nb, int L, const double *sspart,
                  int nss, double *red, i64 out_stride = 0) {
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)nb * L + nss + (i64)plsk::RED_SLICES * (L + 1)) * 8);
    hipLaunchKernelGGL(plsk::reduce_partials_kernel, dim3((L + 63) / 64, plsk::RED_SLICES),
                       dim3(plsk::WG), 0, c->stream, part, nb, L, sspart, nss, red, out_stride);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

// n*K (values of P and of R the r update must read) above which it is split over many workgroups
constexpr i64 ROTATE_SPLIT_MIN = 16384;

// The component update for M > 32 responses (largem_kernels.hpp): M-sized data in global memory, plain multi-workgroup
// kernels, the eigenvector by `power_iters` squarings without early exit.  Scratch in c->lm.
int launch_update_large(pls_hip_context *c, const double *red, double *XY, double *W, double *P, double *Q, double *R,
                        double *v, int K, int M, int A, int a, int nip) {
    const i64 MM = (i64)M * M;
    const int nparts = (K + plsk::WG - 1) / plsk::WG;
    const i64 prows = max_partial_rows(c, K, M);
    const size_t need = (size_t)(3 * MM + 2 * M + nparts + 8 + K + prows * MM + (i64)plsk::RED_SLICES * MM) * 8;
    CHK(ensure(c, c->lm, need));
    double *G = (double *)c->lm.p, *Bm = G + MM, *Cm = Bm + MM, *qe = Cm + MM, *qv = qe + M, *ssp = qv + M;
    double *tr = ssp + nparts, *wraw = tr + 8, *xpart = wraw + K, *xred = xpart + prows * MM;
    const dim3 blk(plsk::WG);
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 4 + MM * 3 * c->opt_power_iters) * 8);
    if (a >= 0) {
        hipLaunchKernelGGL(plsk::lm_pq_kernel, dim3(M + nparts), blk, 0, c->stream, red, (const double *)XY, (const double *)R, P,
                           Q, qv, K, M, a);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::lm_deflate_kernel, dim3((unsigned)(((i64)K * M + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           red, XY, (const double *)P, (const double *)qv, K, M, a);
        LAUNCH_CHECK(c);
    } else {
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)(((i64)K * M + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           red, K * M, XY);
        LAUNCH_CHECK(c);
    }
    const int n = a + 1;
    if (n >= A) return PLS_HIP_OK;
    {  // G = XY^T XY (:405) with the column-reduction kernels: "X" = XY (K rows, M columns), "Y" = XY
        int nb = 0;
        CHK(launch_xty<double>(c, XY, K, XY, K, K, M, M, xpart, &nb));
        CHK(launch_reduce(c, xpart, nb, (int)MM, nullptr, 0, xred));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((MM + plsk::WG - 1) / plsk::WG)), blk, 0, c->stream,
                           (const double *)xred, (int)MM, G);
        LAUNCH_CHECK(c);
    }
    if (M <= plsk::MMAX) {
        // up to 32 responses: the whole direction solve in one workgroup's LDS, one launch
        hipLaunchKernelGGL(plsk::lm_eig_lds_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, (const double *)G, M,
                           (int)c->opt_power_iters, qe);
        LAUNCH_CHECK(c);
    } else {
        // dominant eigenvector: B_0 = G / tr G, B_{j+1} = B_j^2 / tr(B_j^2)
        const dim3 sq((M + 15) / 16, (M + 15) / 16), sqb(16, 16);
        const unsigned nmm = (unsigned)((MM + plsk::WG - 1) / plsk::WG);
        hipLaunchKernelGGL(plsk::lm_trace_kernel, dim3(1), blk, 0, c->stream, (const double *)G, M, tr);
        hipLaunchKernelGGL(plsk::lm_scale_kernel, dim3(nmm), blk, 0, c->stream, (const double *)G, (const double *)tr, MM, Bm);
        for (int it = 0; it < (int)c->opt_power_iters; ++it) {
            hipLaunchKernelGGL(plsk::lm_square_kernel, sq, sqb, 0, c->stream, (const double *)Bm, M, Cm);
            hipLaunchKernelGGL(plsk::lm_trace_kernel, dim3(1), blk, 0, c->stream, (const double *)Cm, M, tr);
            hipLaunchKernelGGL(plsk::lm_scale_kernel, dim3(nmm), blk, 0, c->stream, (const double *)Cm, (const double *)tr, MM, Bm);
        }
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::lm_eig_finish_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, (const double *)G,
                           (const double *)Bm, M, qe);
        LAUNCH_CHECK(c);
    }
    hipLaunchKernelGGL(plsk::lm_w_kernel, dim3(nparts), blk, 0, c->stream, (const double *)XY, (const double *)qe, K, M, wraw, ssp);
    LAUNCH_CHECK(c);
    double *wn = W + (i64)n * K;
    hipLaunchKernelGGL(plsk::lm_normalize_kernel, dim3(nparts), blk, 0, c->stream, (const double *)wraw, (const double *)ssp,
                       nparts, K, wn, n == 0 ? R : (double *)nullptr, n == 0 ? v : (double *)nullptr);
    LAUNCH_CHECK(c);
    if (n > 0) {  // r = w - sum_j (p_j^T w) r_j (:412-416)
        double *cs = (double *)c->cs.p;
        hipLaunchKernelGGL(plsk::rotate_dots_kernel, dim3(n), blk, 0, c->stream, P, W, K, n, cs);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::rotate_apply_kernel, dim3(nparts), blk, 0, c->stream, W, R, cs, v, K, n, nip);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

// Will launch_update run the ONE-workgroup kernel for this shape (the form that can take the gather of a sharded fit's
// collective as its prologue)?  The conditions of the branches in launch_update, in their order.
bool update_is_single(int K, int M, int A, int a) {
    static const bool mid_on = !(getenv("PLS_HIP_MID_UPDATE") && atoi(getenv("PLS_HIP_MID_UPDATE")) == 0);
    static const int wide1_min = getenv("PLS_HIP_WIDE1_MIN") ? atoi(getenv("PLS_HIP_WIDE1_MIN")) : 4097;
    static const bool widem_on = !(getenv("PLS_HIP_WIDEM_UPDATE") && atoi(getenv("PLS_HIP_WIDEM_UPDATE")) == 0);
    static const int widem_min = getenv("PLS_HIP_WIDEM_MIN") ? atoi(getenv("PLS_HIP_WIDEM_MIN")) : plsk::COOP_MAXG * plsk::COOP_WG + 1;
    static const bool coop_on = !(getenv("PLS_HIP_COOP_UPDATE") && atoi(getenv("PLS_HIP_COOP_UPDATE")) == 0);
    int g = 0, e = 0;
    if (M > plsk::MMAX || (mid_on && M > 8 && (i64)K * M >= 16384)) return false;
    if (M == 1 && K >= wide1_min && plsk::wide1_geometry(K, &g, &e)) return false;
    if (widem_on && M >= 2 && M <= 8 && K >= widem_min && (i64)K * M >= 16384 && A <= 4096 && plsk::wide1_geometry(K, &g, &e)) return false;
    if (coop_on && plsk::coop_update_covers(K, M) && A <= 4096) return false;
    const int n = a + 1;
    return !(n < A && n > 0 && (i64)n * K >= ROTATE_SPLIT_MIN);  // (the r recurrence on several workgroups: two more launches)
}

// nip: 0 = KERNEL algo (next pass is X r), 1 = NIPALS (next pass X_a w)
// gx (only where update_is_single says yes): the gather of the component's collective as the kernel's prologue; red is
// then written (slice 0) instead of read
int launch_update(pls_hip_context *c, double *red, double *XY, double *W, double *P,
                  double *Q, double *R, double *v, int K, int M, int A, int a, int nip, const plsk::XchgGather *gx = nullptr) {
    const int n = a + 1;
    // 9 <= M <= 32 responses on many columns: the one-workgroup kernel walks K x M values several times and forms the
    // M (M + 1) / 2 Gram entries one wave per pair (208 us per component at K = 4096, M = 16; 630 us at M = 32 -- more
    // than the 0.32 ms pass); the multi-workgroup kernels of the many-response path with the LDS eigen solve: ~12 launches
    static const bool mid_on = !(getenv("PLS_HIP_MID_UPDATE") && atoi(getenv("PLS_HIP_MID_UPDATE")) == 0);
    if (M > plsk::MMAX || (mid_on && M > 8 && (i64)K * M >= 16384))
        return launch_update_large(c, (const double *)red, XY, W, P, Q, R, v, K, M, A, a, nip);
    // One response on very many columns: element-wise work and K-long sums on up to 128 workgroups, two launches
    // (wide1_update.hpp) instead of one workgroup walking K (+ one workgroup per p_j^T w of the r recurrence)
    static const int wide1_min = getenv("PLS_HIP_WIDE1_MIN") ? atoi(getenv("PLS_HIP_WIDE1_MIN")) : 4097;  // (up to 4096 columns the one-workgroup kernel keeps XY in registers)
    int w1g = 0, w1e = 0;
    if (M == 1 && K >= wide1_min && plsk::wide1_geometry(K, &w1g, &w1e)) {
        CHK(ensure(c, c->wide1, (size_t)((i64)(A + 2) * w1g + A + 1) * 8));
        double *w1part = (double *)c->wide1.p, *w1q = w1part + (i64)(A + 1) * w1g, *w1tot = w1q + w1g;
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
        hipLaunchKernelGGL(plsk::wide1_a_kernel, dim3(w1g), dim3(plsk::W1_WG), (size_t)plsk::W1_WG * w1e * 8, c->stream, red, XY, P, Q,
                           K, A, a, w1e, (const double *)w1q, w1part);
        LAUNCH_CHECK(c);
        if (n < A && n + 1 <= 4096) {
            hipLaunchKernelGGL(plsk::wide1_b_kernel<false>, dim3(w1g), dim3(plsk::W1_WG), (size_t)(n + 1) * 8, c->stream,
                               (const double *)XY, W, R, v, K, n, w1e, nip, (const double *)w1part, w1q, (const double *)nullptr);
            LAUNCH_CHECK(c);
        } else if (n < A) {  // more totals than a workgroup's LDS holds: formed once, read from global memory
            hipLaunchKernelGGL(plsk::wide1_totals_kernel, dim3((n + 4) / 4), dim3(plsk::W1_WG), 0, c->stream, (const double *)w1part, n,
                               w1g, w1tot);
            hipLaunchKernelGGL(plsk::wide1_b_kernel<true>, dim3(w1g), dim3(plsk::W1_WG), 0, c->stream, (const double *)XY, W, R, v, K,
                               n, w1e, nip, (const double *)w1part, w1q, (const double *)w1tot);
            LAUNCH_CHECK(c);
        }
        return PLS_HIP_OK;
    }
    // 2..8 responses beyond the cooperative kernel's 16,384 columns: the same arithmetic cut at its two exchanges, three launches
    static const bool widem_on = !(getenv("PLS_HIP_WIDEM_UPDATE") && atoi(getenv("PLS_HIP_WIDEM_UPDATE")) == 0);
    static const int widem_min = getenv("PLS_HIP_WIDEM_MIN") ? atoi(getenv("PLS_HIP_WIDEM_MIN")) : plsk::COOP_MAXG * plsk::COOP_WG + 1;
    if (widem_on && M >= 2 && M <= 8 && K >= widem_min && (i64)K * M >= 16384 && A <= 4096 && plsk::wide1_geometry(K, &w1g, &w1e)) {
        CHK(ensure(c, c->wide1, (size_t)((i64)(A + plsk::WM_GSTRIDE + plsk::WM_QSTRIDE) * w1g) * 8));
        double *gp = (double *)c->wide1.p, *cp = gp + (i64)plsk::WM_GSTRIDE * w1g, *qp = cp + (i64)A * w1g;
        const dim3 g(w1g), b(plsk::W1_WG);
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
#define WM_CASE(MM_)                                                                                                              \
    do {                                                                                                                          \
        hipLaunchKernelGGL((plsk::widem_a_kernel<MM_>), g, b, 0, c->stream, red, XY, P, Q, K, M, A, a, w1e, (const double *)qp, gp);  \
        if (n < A) {                                                                                                              \
            hipLaunchKernelGGL((plsk::widem_b_kernel<MM_>), g, b, (size_t)plsk::W1_WG * w1e * 8, c->stream, (const double *)XY, W,      \
                               (const double *)P, K, M, n, w1e, (int)c->opt_power_iters, (const double *)gp, cp);                  \
            hipLaunchKernelGGL((plsk::widem_c_kernel<MM_>), g, b, (size_t)(n + 1) * 8, c->stream, (const double *)XY,                  \
                               (const double *)W, R, v, K, M, n, w1e, nip, (const double *)cp, qp);                                \
        }                                                                                                                         \
    } while (0)
        if (M <= 2) WM_CASE(2); else if (M <= 4) WM_CASE(4); else WM_CASE(8);
#undef WM_CASE
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // PLS_HIP_COOP_UPDATE=0 in the environment keeps the single-workgroup kernel (A/B measurements only)
    static const bool coop_on = !(getenv("PLS_HIP_COOP_UPDATE") && atoi(getenv("PLS_HIP_COOP_UPDATE")) == 0);
    if (coop_on && plsk::coop_update_covers(K, M) && A <= 4096) {
        // several workgroups, two in-launch exchanges, r included (coop_update.hpp): one launch per component
        const size_t need = plsk::coop_scratch_bytes(A);
        if (c->coop.bytes < need) {  // the exchange counters start from zero
            CHK(ensure(c, c->coop, need));
            HIPCHK(c, hipMemsetAsync(c->coop.p, 0, c->coop.bytes, c->stream));
        }
        unsigned *cnt = (unsigned *)c->coop.p;
        // every fit starts from zeroed exchange counters, whatever an earlier (failed) fit left behind
        if (a < 0) HIPCHK(c, hipMemsetAsync(cnt, 0, 256, c->stream));
        double *qraw = (double *)((char *)c->coop.p + 256), *gpart = qraw + plsk::COOP_MAXG * plsk::COOP_QSTRIDE;
        double *cpart = gpart + plsk::COOP_MAXG * plsk::COOP_GSTRIDE;
        const dim3 grid((K + plsk::COOP_WG - 1) / plsk::COOP_WG), blk(plsk::COOP_WG);
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
#define COOP_CASE(MM_) hipLaunchKernelGGL((plsk::coop_update_kernel<MM_>), grid, blk, (size_t)A * sizeof(double), c->stream, \
                                          red, XY, W, P, Q, R, v, K, M, A, a, nip, (int)c->opt_power_iters, cnt, qraw, gpart, cpart)
        if (M <= 2) COOP_CASE(2); else if (M <= 4) COOP_CASE(4); else COOP_CASE(8);
#undef COOP_CASE
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // (the in-kernel r recurrence stages its p_j^T w products in min(A, 4096) doubles of LDS: beyond 4096 components
    // K > 4096 as well, so the multi-workgroup form below takes over from the fourth component on)
    const bool split = n < A && n > 0 && (i64)n * K >= ROTATE_SPLIT_MIN;
    Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + K) * 8);
    if (gx) {
        if (split) return fail(c, PLS_HIP_ERR_DEVICE, "internal: gather prologue on a split update");
        hipLaunchKernelGGL(plsk::component_update_gather_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                           (size_t)std::min(A, 4096) * sizeof(double), c->stream, *gx, red, XY, W, P, Q, R, v, K, M, A, a, nip,
                           (int)c->opt_power_iters, 0);
    } else {
        hipLaunchKernelGGL(plsk::component_update_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                           (size_t)std::min(A, 4096) * sizeof(double), c->stream, (const double *)red, XY, W, P, Q, R, v, K, M, A, a, nip,
                           (int)c->opt_power_iters, (int)split);
    }
    LAUNCH_CHECK(c);
    if (split) {
        double *cs = (double *)c->cs.p;
        hipLaunchKernelGGL(plsk::rotate_dots_kernel, dim3(n), dim3(plsk::WG), 0, c->stream, P, W, K, n, cs);
        LAUNCH_CHECK(c);
        hipLaunchKernelGGL(plsk::rotate_apply_kernel, dim3((K + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0,
                           c->stream, W, R, cs, v, K, n, nip);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

// Every collective of the library is a SLICED message: RED_SLICES slices of count / RED_SLICES values each, whose
// consumers add the slices in index order (an element-wise all-reduce keeps that layout; the device-side exchanges
// leave the total in slice 0 and zeros behind it) -- include/pls_hip.h, pls_hip_allreduce_fn.
int do_allreduce(pls_hip_context *c, double *buf, i64 count) {
    if (!c->reducer) return PLS_HIP_OK;  // an installed reducer is called even for one rank
    if (count % plsk::RED_SLICES != 0) return fail(c, PLS_HIP_ERR_REDUCER, "internal: collective of " + std::to_string(count) + " values is not sliced");
    const int rc = c->reducer(c->reducer_user, buf, count, (void *)c->stream);
    if (rc != 0) return fail(c, PLS_HIP_ERR_REDUCER, "all-reduce callback returned " + std::to_string(rc));
    return PLS_HIP_OK;
}

// XX(K x K, fp64) = X^T X summed over ranks: matrix-core SYRK when the layout allows it, otherwise the
// column-reduction kernel in 32-column blocks.  Uses c->part / c->red2 as scratch.
// Whichever kernels the LOCAL shard takes (its row count, alignment and leading dimension decide, and an
// empty shard runs none), the exchange is always ONE all-reduce of RED_SLICES*K*K values in the same layout:
// the ranks of a sharded fit can never disagree on the sequence of collectives.
// compute_xx_local: this rank's X^T X into the slices of c->red2 (no collective).  With Y and xy_red given, the matrix-core
// path forms X^T Y in the same sweep (its diagonal workgroups, syrk_kernels.hpp) and leaves it, reduced into slices, in
// xy_red; *xy_done says whether it did (the caller runs the separate X^T Y kernel otherwise).
// compute_xx_finish: the one all-reduce of the slices and their sum -> XX.
template <typename T>
int compute_xx_local(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, const T *Y = nullptr, i64 ldy = 0, int M = 0,
                     double *xy_red = nullptr, bool *xy_done = nullptr) {
    constexpr int CB = 32;
    const i64 KK = (i64)K * K;
    if (xy_done) *xy_done = false;
    CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8));
    double *red2 = (double *)c->red2.p;
    bool have = false;
    if (N > 0) {
        // matrix-core path: 128 x 128 blocks on v_mfma_f64_16x16x4_f64, row-split partial blocks
        const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
        const i64 S = std::max<i64>(1, (16 * (i64)c->num_cu + nbk * (nbk + 1) / 2 - 1) / (nbk * (nbk + 1) / 2));  // capacity bound
        if (ensure(c, c->part, (size_t)S * KK * 8) == PLS_HIP_OK) {
            double *part = (double *)c->part.p;
            int nb = 0, nb_xy = 0;
            int rc;
            {
                Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) * ((nbk + 1)) + KK * 8);
                if (!c->zeros.p) {  // source of out-of-range rows for the LDS-DMA panels
                    CHK(ensure(c, c->zeros, 256));
                    HIPCHK(c, hipMemsetAsync(c->zeros.p, 0, 256, c->stream));
                }
                // PLS_HIP_SYRK_GLDS=0 in the environment selects the register-staged kernel (A/B measurements only)
                static const bool glds = !(getenv("PLS_HIP_SYRK_GLDS") && atoi(getenv("PLS_HIP_SYRK_GLDS")) == 0);
                static const bool fuse_xy = !(getenv("PLS_HIP_SYRK_XY") && atoi(getenv("PLS_HIP_SYRK_XY")) == 0);
                const i64 xycap = 64 * (i64)K * std::max(M, 1);  // at most 64 row splits of a diagonal block
                const bool want_xy = fuse_xy && glds && Y && xy_red && M >= 1 && M <= 8 &&
                                     ensure(c, c->xyp, (size_t)xycap * 8) == PLS_HIP_OK;
                if (!want_xy) c->err.clear();
                rc = plsk::launch_syrk<T>(c->stream, c->num_cu, X, ldx, N, K, part, S * KK, &nb,
                                          glds ? c->zeros.p : nullptr, want_xy ? Y : nullptr, ldy, M,
                                          want_xy ? (double *)c->xyp.p : nullptr, xycap, want_xy ? &nb_xy : nullptr);
                if (rc != 0) s.on = false;
            }
            if (rc == 0) {
                LAUNCH_CHECK(c);
                CHK(launch_reduce(c, part, nb, (int)KK, nullptr, 0, red2));
                have = true;
                if (nb_xy > 0) {
                    CHK(launch_reduce(c, (const double *)c->xyp.p, nb_xy, K * M, nullptr, 0, xy_red));
                    *xy_done = true;
                }
            }
        } else {
            c->err.clear();  // no room for the partial blocks: the column-block path needs far less
        }
        if (!have) {  // unaligned layouts: column blocks of X^T X through the column-reduction kernel, into the same slices
            const i64 prow = max_partial_rows(c, N, K);
            CHK(ensure(c, c->part, (size_t)prow * (size_t)K * CB * 8));
            double *part = (double *)c->part.p;
            for (int c0 = 0; c0 < K; c0 += CB) {
                const int cb = std::min(CB, K - c0);
                int nb = 0;
                CHK(launch_xty<T>(c, X, ldx, X + (i64)c0 * ldx, ldx, N, K, cb, part, &nb));
                CHK(launch_reduce(c, part, nb, K * cb, nullptr, 0, red2 + (i64)c0 * K, KK));
            }
        }
    } else {
        HIPCHK(c, hipMemsetAsync(red2, 0, (size_t)plsk::RED_SLICES * KK * 8, c->stream));
    }
    return PLS_HIP_OK;
}

int compute_xx_finish(pls_hip_context *c, int K, double *XX) {
    const i64 KK = (i64)K * K;
    double *red2 = (double *)c->red2.p;
    CHK(do_allreduce(c, red2, (i64)plsk::RED_SLICES * KK));
    hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                       c->stream, (const double *)red2, (int)KK, XX);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

template <typename T>
int compute_xx(pls_hip_context *c, const T *X, i64 ldx, i64 N, int K, double *XX) {
    CHK(compute_xx_local<T>(c, X, ldx, N, K));
    return compute_xx_finish(c, K, XX);
}

// Sharded fits: every rank must have derived the same bits (small_kernels.hpp, "replica guard").  Two small launches and one
// 512-byte all-reduce per fit; the verdict lands in a host-mapped flag that pls_hip_synchronize (and the host-memory entry)
// turn into PLS_HIP_ERR_REDUCER.
int replica_guard(pls_hip_context *c, const double *W, const double *P, const double *Q, const double *R, const double *B,
                  int K, int M, int A) {
    if (!c->reducer || c->nranks < 2 || c->nranks > 1024) return PLS_HIP_OK;
    static const bool on = !(getenv("PLS_HIP_REPLICA_GUARD") && atoi(getenv("PLS_HIP_REPLICA_GUARD")) == 0);
    if (!on) return PLS_HIP_OK;
    if (!c->diverged) {
        if (hipHostMalloc((void **)&c->diverged, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&c->diverged_dev, c->diverged, 0) != hipSuccess) {
            (void)hipGetLastError();
            c->diverged = nullptr;
            return PLS_HIP_OK;  // (no mapped host memory: the guard is an extra, not a precondition)
        }
        *c->diverged = 0;
    }
    CHK(ensure(c, c->guard, (size_t)plsk::RED_SLICES * 8 * 8));
    double *g = (double *)c->guard.p;
    hipLaunchKernelGGL(plsk::replica_checksum_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream, W, P, R, Q, B, (i64)K * A,
                       (i64)M * A, (i64)K * M, g);
    LAUNCH_CHECK(c);
    CHK(do_allreduce(c, g, (i64)plsk::RED_SLICES * 8));
    hipLaunchKernelGGL(plsk::replica_verify_kernel, dim3(1), dim3(plsk::WAVE), 0, c->stream, (const double *)g, c->nranks,
                       c->diverged_dev);
    LAUNCH_CHECK(c);
    return PLS_HIP_OK;
}

int check_xchg(pls_hip_context *c);
void xchg_release(pls_hip_context *c);

int check_diverged(pls_hip_context *c) {  // (the stream has been synchronised)
    CHK(check_xchg(c));
    if (c->diverged && *c->diverged) {
        *c->diverged = 0;
        return fail(c, PLS_HIP_ERR_REDUCER, "the ranks of the sharded fit derived different W / P / Q / R / B: the reducer did not "
                                            "leave identical sums on every rank");
    }
    return PLS_HIP_OK;
}

// ---- the fit on device pointers -----------------------------------------------------------
template <typename T>
int fit_device(pls_hip_context *c, const T *X, i64 ldx, const T *Y, i64 ldy, i64 N, int K, int M,
               int A, int method, double *W, double *P, double *Q, double *R, T *Tm, i64 ldt, double *B) {
    // GRAM plan for a KERNEL_TYPE1 request: the K-sized loop runs on XX = X^T X exactly as KERNEL_TYPE2
    // does (no pass over X per component), then the scores are formed in one pass, T = X R.
    // AUTO: pick between the read-only pass plan and the Gram plan from a bandwidth / matrix-core cost
    // model (measured rates on MI355X: ~6 TB/s streaming reads, ~59 TFLOP/s executed in the fp64 SYRK of
    // which the symmetric half is computed).  GRAM pays off for A >~ K/60.
    // A single-response problem small enough for one workgroup's registers: the whole fit in ONE launch (tiny_kernels.hpp) instead of
    // three launches per component, whose dispatch latency would be the entire cost.  The reference's sequence, so the
    // KERNEL plan and AUTO (also when X^T X came with the upload: one launch beats the K x K loop's sixty);
    // an explicit NIPALS or GRAM request keeps its own kernels.
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && plsk::tiny_fit_covers(N, K, M, A, ldx, sizeof(T)) &&
        !(getenv("PLS_HIP_TINY") && atoi(getenv("PLS_HIP_TINY")) == 0)) {
        const size_t lds = (size_t)2 * K * A * 8;
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_kernel<T>, (int)plsk::TINY_LDS_MAX)  /* raised once per device: to the most any fit asks for */)
            return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");
        Range r_fit("pls_hip_fit (single launch)");
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
        hipLaunchKernelGGL((plsk::tiny_fit_kernel<T>), dim3(1), dim3(plsk::UPD_THREADS), lds, c->stream, X, ldx, Y, (int)N, K, A,
                           W, P, Q, R, Tm, ldt, B, (const i64 *)nullptr, 0, (i64)0, (double *)nullptr);
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    // ... and the same for 2..8 responses (tiny_fit_m_kernel: the reference's own example, README.md:23, is such a fit)
    if (method == PLS_HIP_KERNEL_TYPE1 && (c->opt_algo == PLS_HIP_ALGO_KERNEL || c->opt_algo == PLS_HIP_ALGO_AUTO) && c->opt_fuse &&
        !c->reducer && plsk::tiny_fit_m_covers(N, K, M, A, ldx, sizeof(T)) &&
        !(getenv("PLS_HIP_TINY") && atoi(getenv("PLS_HIP_TINY")) == 0)) {
        const size_t lds = (size_t)(2 * K + M) * A * 8;
        Range r_fit("pls_hip_fit (single launch)");
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)N * K + (i64)N * M + (i64)N * A) * (i64)sizeof(T) + (3 * (i64)K + M) * A * 8);
#define TINY_M(MM_)                                                                                                          \
    do {                                                                                                                     \
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_m_kernel<T, MM_>, (int)plsk::TINY_LDS_MAX))               \
            return fail(c, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");            \
        hipLaunchKernelGGL((plsk::tiny_fit_m_kernel<T, MM_>), dim3(1), dim3(plsk::UPD_THREADS), lds, c->stream, X, ldx, Y, ldy,  \
                           (int)N, K, M, A, (int)c->opt_power_iters, W, P, Q, R, Tm, ldt, B, (const i64 *)nullptr, 0, (i64)0,  \
                           (double *)nullptr);                                                                               \
    } while (0)
        if (M <= 2) TINY_M(2); else if (M <= 4) TINY_M(4); else TINY_M(8);
#undef TINY_M
        LAUNCH_CHECK(c);
        return PLS_HIP_OK;
    }
    i64 algo = c->opt_algo;
    const bool have_pre = c->pre_xx && c->pre_xy && K <= 32768;
    if (algo == PLS_HIP_ALGO_AUTO && have_pre) {
        algo = PLS_HIP_ALGO_GRAM;  // X^T X is already there: the component loop needs no pass over X at all
    } else if (algo == PLS_HIP_ALGO_AUTO) {
        const double pass_s = (double)N * K * sizeof(T) / 6.0e12;
        const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
        // tiles of 16 x 16 the SYRK executes: the blocks above the diagonal in full, 40 of 64 in a diagonal block (syrk_kernels.hpp)
        const double tile_frac = (32.0 * nbk * (nbk - 1) + 40.0 * nbk) / (64.0 * nbk * nbk);
        const double syrk_s = 2.0 * N * (double)K * K * tile_frac / 59.0e12;
        // ranks of a sharded fit see different N: they must not disagree on the plan -> KERNEL there
        const bool gram_ok = K <= 2048 && N >= 4096 && !c->reducer;
        algo = (gram_ok && (1 + A) * pass_s > syrk_s + 2.5 * pass_s + A * 25e-6) ? PLS_HIP_ALGO_GRAM
                                                                               : PLS_HIP_ALGO_KERNEL;
    }
    const bool gram = (method == PLS_HIP_KERNEL_TYPE1) && (algo == PLS_HIP_ALGO_GRAM);
    const bool type2 = (method == PLS_HIP_KERNEL_TYPE2) || gram;
    const bool nipals = !type2 && (algo == PLS_HIP_ALGO_NIPALS);
    const int nip = nipals ? 1 : 0;
    const i64 L0 = (i64)K * M;
    const i64 redn = (i64)plsk::RED_SLICES * std::max<i64>(L0, K + 1);
    const i64 prow = max_partial_rows(c, N, K);
    CHK(ensure(c, c->part, (size_t)prow * (size_t)std::max<i64>(L0, K) * 8));
    // t^T t partials: one per workgroup of whichever kernel forms the scores (narrow X*v: N/256; tile kernels: their grid)
    // (the split score kernel of short, wide matrices leaves one partial per 64 rows)
    const i64 ssmax = std::max<i64>(std::max<i64>(K >= 1024 ? (N + 63) / 64 : (N + plsk::WG - 1) / plsk::WG, 1), max_partial_rows(c, N, K));
    CHK(ensure(c, c->sspart, (size_t)ssmax * 8));
    CHK(ensure(c, c->xy, (size_t)L0 * 8));
    CHK(ensure(c, c->v, (size_t)K * 8));
    CHK(ensure(c, c->cs, (size_t)A * 8));
    double *red;
    if (c->user_red) {
        if (c->user_red_count < redn) return fail(c, PLS_HIP_ERR_INVALID, "reduce buffer too small");
        red = c->user_red;
    } else {
        CHK(ensure(c, c->red, (size_t)redn * 8));
        red = (double *)c->red.p;
    }
    // NIPALS keeps the deflated matrix in a library-owned buffer.  When the tile-resident pass covers the fit it
    // is stored row-tile-major (every R x K tile one contiguous block: fused_kernels.hpp), otherwise column-major
    // with ld = N for the one-product kernels.
    // the tile that reads the caller's matrix: 32 column groups x 16 row lanes, taller with fewer groups for narrow matrices
    const int tall_cg = plsk::tall_groups(K);
    const i64 TR = (512 / tall_cg) * (i64)(16 / sizeof(T));
    // Any row count (the last N % V rows go to a tail kernel), any alignment of the columns and leading dimensions up to
    // 2^31 / 4 bytes (mode 2: the EDGE instantiations) -- the same one-sweep traffic for every matrix the reference
    // accepts (src/pls.cpp:419-421)
    const int fused_mode = (c->opt_fuse && N > 0) ? plsk::fused_pass_mode<T>(X, ldx, N, K, Tm, ldt) : 0;
    const bool fused_fit = fused_mode != 0;
    const int wide_mode = (c->opt_fuse && !fused_fit && N > 0) ? plsk::deflate_score_mode<T>(X, ldx, N, K, Tm, ldt) : 0;
    // wide matrices (no resident tile): deflation + score in one sweep, loading in a second read
    const bool semi_fit = nipals && wide_mode != 0;
    // Beyond the semi-fused sweep's reach (its w and p_prev need 16 K bytes of LDS: K <= 4608) and up to 8192 columns the
    // copy's tiles are ONE row pack high (512 column groups x 16 columns per lane): the copy is made in the X^T Y sweep
    // (retile_xty) and every component runs fused on it -- instead of 4 N K s (NIPALS) / 2 N K s (KERNEL) per component
    // through the one-product kernels.
    static const bool wide16 = !(getenv("PLS_HIP_WIDE16") && atoi(getenv("PLS_HIP_WIDE16")) == 0);
    static const bool wide512 = !(getenv("PLS_HIP_WIDE512") && atoi(getenv("PLS_HIP_WIDE512")) == 0);  // A/B measurements only
    // (read-only passes take 32 columns per lane there: the KERNEL plan up to 16384 columns)
    const bool wide_src = wide512 && wide16 && c->opt_fuse && !fused_fit && N > 0 && K <= 512 * (nipals ? 16 : 32) &&
                          plsk::wide_source_ok<T>(X, ldx, N, Tm);
    const bool wide_only = nipals && wide_src && K > 256 * 16 && K <= 512 * 16 && M <= 8 && A >= 3 && c->opt_work_layout != 0 &&
                           !(getenv("PLS_HIP_RETILE_XTY") && atoi(getenv("PLS_HIP_RETILE_XTY")) == 0);
    const bool tiled_work = nipals && (fused_fit || semi_fit || wide_only) && c->opt_work_layout != 0;
    // Row-tile-major tiles are contiguous whatever their height, so for 1024 < K <= 4096 the working copy uses
    // SHORTER tiles (8-32 rows) that do fit the registers of a CU: from the third component on the fully fused
    // pass runs again (2 N K s per component instead of the semi-fused 3 N K s).  Only the first deflation has
    // to read the caller's column-major X in 256-byte pieces (deflate_score, writing the short tiles).
    // KERNEL plan on such a matrix: X is copied ONCE into short tiles (read + write), after which every component
    // is one fused read-only pass instead of two one-product passes -- pays from the third component on.
    // ... and on a matrix whose columns are not 16-byte aligned (ld odd, a base pointer at 8 mod 16): the one-sweep pass can
    // read it (EDGE level 2) but every 256-byte segment then shares a line with its neighbours and the pass runs at 0.52
    // instead of 0.73 of peak; with the copy (formed in the same sweep as X^T Y: retile_xty_kernel) every component reads
    // aligned tiles.  Costs one write of X; pays from the fourth component on (PLS_HIP_COPY_MIN).
    constexpr int FVX = 16 / (int)sizeof(T);
    // The same copy pays for ALIGNED matrices once there are enough components: a read-only pass over the tiled copy, one
    // contiguous 128 KB block per tile, runs at 0.85 of peak (0.63 ms at config 3) against 0.72 (0.74 ms) over the caller's
    // column-major matrix in 256-byte segments; the copy costs 0.86 ms more than the X^T Y pass it replaces
    // (PLS_HIP_COPY_MIN_ALIGNED, default 10 components).
    // Beyond 512 columns the direct pass (32 columns per lane, one workgroup per CU) already reads at 0.81 of peak and the
    // copy only pays from ~30 components on (one shard of config 5: 2.65 -> 2.52 ms per pass against 4.1 ms for the copy).
    static const int copy_min = getenv("PLS_HIP_COPY_MIN") ? atoi(getenv("PLS_HIP_COPY_MIN")) : 4;
    static const int copy_min_al = getenv("PLS_HIP_COPY_MIN_ALIGNED") ? atoi(getenv("PLS_HIP_COPY_MIN_ALIGNED")) : 10;
    const bool copy_fit = fused_fit && M <= 8 &&
                          A >= (vec_ok<T>(X, ldx, FVX) ? (K <= 32 * 16 ? copy_min_al : 3 * copy_min_al + 2) : copy_min);
    bool retile_fit = !nipals && !type2 && A >= 3 && c->opt_work_layout != 0 &&
                      ((wide_mode != 0 && K <= 128 * 32) || (wide_src && K > 128 * 32) || copy_fit);
    // column groups of the short tiles of a wide matrix (1024 < K <= 4096): 16 columns per lane in 128 / 256 groups
    // (8-row fp32 / 4-row fp64 tiles at K <= 4096) -- the register shape of the headline kernel, two workgroups per CU
    // on read-only passes.  Config 4: read+write pass 0.766 -> 0.710 ms (0.70 -> 0.76 of peak), read-only pass
    // 0.364 -> 0.324 ms (0.74 -> 0.83) against 32 columns per lane in 64 / 128 groups (PLS_HIP_WIDE16=0, the round-1 shape).
    const int wide_groups = fused_fit ? (K <= 32 * 16 ? tall_cg : 64)  // (the copy of a matrix the resident tile covers)
                            : wide16 ? (K <= 128 * 16 ? 128 : (K <= 256 * 16 ? 256 : ((wide_src && (!nipals || wide_only)) ? 512 : 0)))
                                     : (K <= 64 * 32 ? 64 : (K <= 128 * 32 ? 128 : 0));
    if (retile_fit) {  // the copy is optional: without room for it the one-product kernels do the job
        const i64 wr = (512 / wide_groups) * (i64)(16 / sizeof(T));
        if (ensure(c, c->work, (size_t)((N + wr - 1) / wr) * wr * K * sizeof(T)) != PLS_HIP_OK) {
            retile_fit = false;
            c->err.clear();
        }
    }
    const int wide_cg = ((semi_fit && tiled_work) || retile_fit || wide_only) ? wide_groups : 0;
    // 512 < K <= 1024: the resident tile of the caller's layout needs 32 columns per lane (one 8-wave workgroup per
    // CU, 5.7-5.85 TB/s); on the tiled copy the same K fits half-height tiles at 16 columns per lane (6.0 TB/s).
    // Component 0 reads X with the tall tile, the first deflation reads X tall and writes the short tiles (rdst),
    // every later pass runs on the short tiles.
    const int mid_cg = (nipals && fused_fit && tiled_work && K > 32 * 16 && A > 2) ? 64 : 0;
    // opt-in deferred write-back (defer_kernels.hpp): up to `defer` rank-1 updates pending per stored matrix
    const int defer = (nipals && fused_mode == 1 && tall_cg == 32 && N % FVX == 0 && plsk::cols_aligned<T>(Tm, ldt) && tiled_work && K <= 32 * 16)
                          ? (int)c->opt_defer : 1;
    const int work_cg = wide_cg ? wide_cg : (mid_cg ? mid_cg : tall_cg);
    const i64 WR = (512 / work_cg) * (i64)(16 / sizeof(T));  // rows per tile of the working copy
    if ((nipals && A > 1 && N > 0) || retile_fit)
        CHK(ensure(c, c->work, (tiled_work || retile_fit) ? (size_t)((N + WR - 1) / WR) * WR * K * sizeof(T)
                                                          : (size_t)((N + 3) & ~(i64)3) * K * sizeof(T)));
    double *part = (double *)c->part.p, *sspart = (double *)c->sspart.p;
    double *XY = (double *)c->xy.p, *v = (double *)c->v.p;
    T *work = (T *)c->work.p;
    // The partial rows of a fused pass are summed in the tail of the pass itself (slice_tail: no reduce launch behind it),
    // and with the device-side exchange attached the push of a sharded component rides there too, the gather in front of
    // the update: pass -> update.  PLS_HIP_TAIL=0: the launches of round 3 (A/B measurements).
    static const bool tail_on = !(getenv("PLS_HIP_TAIL") && atoi(getenv("PLS_HIP_TAIL")) == 0);
    plsk::SliceTail tail;
    if (tail_on && c->opt_fuse && N > 0) {
        if (!c->tailcnt.p) CHK(ensure(c, c->tailcnt, 256));
        HIPCHK(c, hipMemsetAsync(c->tailcnt.p, 0, 256, c->stream));  // (whatever an earlier, failed fit left behind)
        tail.cnt = (unsigned *)c->tailcnt.p;
        tail.red = red;
    }
    const bool xep_ok = c->xep.on && c->reducer && (i64)K + 1 <= plsk::XCHG_CAP;

    // prologue: XY = X^T Y (src/pls.cpp:396), summed over ranks
    Range r_fit("pls_hip_fit");
    std::unique_ptr<Range> r_phase(new Range("X^T Y"));
    const bool use_pre = have_pre && (gram || method == PLS_HIP_KERNEL_TYPE2);
    bool xy_from_syrk = false, xx_local_done = false, retiled = false;
    // NIPALS on a wide matrix takes the same first sweep: its working copy is then complete before the first component,
    // which runs fused on it, as every later one does in place (instead of two one-product passes over X for component 0
    // and the semi-fused sweep + a loading pass for component 1: config 4 36.6 -> 36.0 ms per fit)
    const bool nip_copy = nipals && (semi_fit || wide_only) && tiled_work && wide_cg != 0 && A >= 3;
    if (!use_pre && N > 0 && (retile_fit || nip_copy) && M <= 8) {
        // the copy into tiles and X^T Y in ONE sweep over the caller's matrix (instead of retile_kernel + the X^T Y pass)
        static const bool rx_on = !(getenv("PLS_HIP_RETILE_XTY") && atoi(getenv("PLS_HIP_RETILE_XTY")) == 0);
        int nb = 0, rc = 1;
        if (rx_on) {
            Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T) + (i64)N * M * sizeof(T) + L0 * 8);
            // (the source tile is as tall as the copy's for narrow matrices, the 256-byte-segment tile otherwise)
#define RX_CALL(CG_) plsk::launch_retile_xty<T, CG_>(c->stream, c->num_cu, X, ldx, Y, ldy, work, WR, WR * (i64)K, (int)WR, N, K, M, part, (int)prow, &nb)
            rc = (fused_fit && tall_cg == 8) ? RX_CALL(8) : ((fused_fit && tall_cg == 16) ? RX_CALL(16) : RX_CALL(32));
#undef RX_CALL
            if (rc != 0) s.on = false;
        }
        if (rc == 0) {
            LAUNCH_CHECK(c);
            CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
            retiled = true;
        }
    }
    if (wide_only && !retiled && N > 0) return fail(c, PLS_HIP_ERR_DEVICE, "copy into row-pack tiles failed");
    if (retiled) {
    } else if (use_pre) {
        hipLaunchKernelGGL(plsk::fill_slices_kernel, dim3((unsigned)((L0 + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, c->pre_xy, (int)L0, red);
        LAUNCH_CHECK(c);
    } else if (N > 0) {
        // KERNEL_TYPE2 / GRAM: the SYRK's diagonal workgroups form X^T Y on the way (no pass over X of its own); its
        // partial blocks take c->part, so it runs first and `part` is read again afterwards
        if (type2) {
            CHK(compute_xx_local<T>(c, X, ldx, N, K, Y, ldy, M, red, &xy_from_syrk));
            xx_local_done = true;
            part = (double *)c->part.p;
        }
        if (!xy_from_syrk) {
            int nb = 0;
            CHK(launch_xty<T>(c, X, ldx, Y, ldy, N, K, M, part, &nb));
            CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
        }
    } else {
        HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * L0 * 8, c->stream));
    }
    CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * L0));
    CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, -1, nip));
    r_phase.reset();

    if (type2) {
        // KERNEL_TYPE2 (src/pls.cpp:398, :422-425): XX = X^T X once, then the A-loop never touches X:
        // tt = r^T XX r, p = XX r / tt; T is not computed.  XX is formed in 32-column blocks with the
        // same column-reduction kernel as X^T Y (functional; an MFMA SYRK is the planned fast form).
        CHK(ensure(c, c->xx, (size_t)K * K * 8));
        CHK(ensure(c, c->praw, (size_t)K * 8));
        double *XX = (double *)c->xx.p, *praw = (double *)c->praw.p;
        r_phase.reset(new Range("X^T X (SYRK)"));
        if (use_pre) {  // this member's X^T X came with the upload: present it as slice 0, sum over the members
            const i64 KK = (i64)K * K;
            CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8));
            hipLaunchKernelGGL(plsk::fill_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                               c->stream, c->pre_xx, (int)KK, (double *)c->red2.p);
            LAUNCH_CHECK(c);
            CHK(do_allreduce(c, (double *)c->red2.p, (i64)plsk::RED_SLICES * KK));
            hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                               c->stream, (const double *)c->red2.p, (int)KK, XX);
            LAUNCH_CHECK(c);
        } else {
            if (!xx_local_done) CHK(compute_xx_local<T>(c, X, ldx, N, K));  // (an empty shard: zero slices)
            CHK(compute_xx_finish(c, K, XX));
        }
        r_phase.reset();
        for (int a = 0; a < A; ++a) {
            Range r_comp("component", a);
            {
                Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * K + 2 * K) * 8);
                hipLaunchKernelGGL(plsk::symv_kernel, dim3((K + 3) / 4), dim3(plsk::WG), 0, c->stream,
                                   (const double *)XX, (const double *)v, K, praw);  // XX symmetric: XX r
                LAUNCH_CHECK(c);
            }
            if (update_is_single(K, M, A, a)) {  // tt = r^T praw and the packing: the prologue of the one-workgroup update
                Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * M * 3 + (i64)K * (2 * (a + 2)) + 2 * K) * 8);
                hipLaunchKernelGGL(plsk::component_update_type2_kernel, dim3(1), dim3(plsk::UPD_THREADS),
                                   (size_t)std::min(A, 4096) * sizeof(double), c->stream, (const double *)praw, (const double *)v, red,
                                   XY, W, P, Q, R, v, K, M, A, a, (int)c->opt_power_iters, 0);
                LAUNCH_CHECK(c);
                continue;
            }
            hipLaunchKernelGGL(plsk::type2_pack_kernel, dim3(1), dim3(plsk::UPD_THREADS), 0, c->stream,
                               (const double *)praw, (const double *)v, K, red);
            LAUNCH_CHECK(c);
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, 0));
        }
        if (B) {
            const int nblk = (int)((L0 + plsk::WG - 1) / plsk::WG);
            hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, c->stream, R, Q, K, M, A, B);
            LAUNCH_CHECK(c);
        }
        if (gram && N > 0) {  // T = X R (src/pls.cpp:439-442 applied to the training data)
            Range r_t("T = X R");
            int nss = 0;
            CHK(launch_xb<T>(c, X, ldx, N, K, R, K, A, Tm, ldt, nullptr, &nss));
        }
        return replica_guard(c, W, P, Q, R, B, K, M, A);
    }

    const T *Xc = X;
    i64 ldc = ldx, tsc = TR;  // column stride and tile stride of the current matrix
    // (a column-major working copy keeps 16-byte columns whatever N is)
    constexpr i64 PV = 16 / (i64)sizeof(T);
    const i64 ldw = (tiled_work || retile_fit) ? WR : (N + PV - 1) / PV * PV, tsw = (tiled_work || retile_fit) ? WR * (i64)K : TR;
    bool cur_tiled = false;  // Xc is the row-tile-major working copy
    int defer_b = 0;         // deferred write-back: index of the stored matrix X_b
    for (int a = 0; a < A; ++a) {
        Range r_comp("component", a);
        bool tail_used = false;
        // sharded over the device-side exchange, the one-workgroup update behind the pass: the pass pushes (if its tail
        // runs), the update gathers.  The collective number is drawn only when the pass did push.
        const bool want_push = tail.cnt && xep_ok && update_is_single(K, M, A, a);
        tail.npush = 0;
        if (want_push) {
            const unsigned long long seq = *c->xep.seq + 1;
            const int par = (int)(seq & 1);
            tail.npush = c->xep.n;
            tail.seq = seq;
            for (int j = 0; j < c->xep.n; ++j) {
                tail.peers.slot[j] = c->xep.inbox[j] + ((i64)par * c->xep.n + c->xep.rank) * plsk::XCHG_CAP;
                tail.peers.flag[j] = c->xep.flags[j] + par * c->xep.n + c->xep.rank;
            }
        }
        if (N > 0) {
            bool done = false;
            if (fused_fit && defer > 1 && a > 0) {
                // deferred write-back: the stored matrix is X_b (the caller's X for b = 0, the working copy after the
                // first store); apply the a - b pending updates in registers, store X_a when `defer` are pending
                int nb = 0, nss = 0, rc;
                const int np = a - defer_b;
                const bool store = (np == defer) && (a + 1 < A);
                plsk::PendingUpdates<T> pend{};
                for (int n = 0; n < np; ++n) {
                    pend.t[n] = Tm + (i64)(defer_b + n) * ldt;
                    pend.p[n] = P + (i64)(defer_b + n) * K;
                }
                {
                    const i64 bytes = (store ? 2 : 1) * (i64)N * K * sizeof(T) + (np + 1) * (i64)N * sizeof(T) +
                                      (np + 2) * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
                    rc = plsk::launch_fused_defer<T>(c->stream, c->num_cu, Xc, ldc, tsc, work, ldw, tsw, N, K, v, np, pend,
                                                     store, Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss);
                    if (rc != 0) s.on = false;
                }
                if (rc != 0) return fail(c, PLS_HIP_ERR_DEVICE, "deferred fused pass launch failed");
                LAUNCH_CHECK(c);
                done = true;
                if (store) { Xc = work; ldc = ldw; tsc = tsw; cur_tiled = true; defer_b = a; }
                CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            } else if (fused_fit && !retile_fit) {
                // tile-resident pass: [deflate with (t_{a-1}, p_{a-1}) +] t_a = X v, X^T t_a partials
                int nb = 0, nss = 0;
                const T *tprev = (nipals && a > 0) ? Tm + (i64)(a - 1) * ldt : nullptr;
                const double *pprev = (nipals && a > 0) ? P + (i64)(a - 1) * K : nullptr;
                int rc;
                {
                    const i64 bytes = (tprev ? 2 : 1) * (i64)N * K * sizeof(T) +
                                      (tprev ? 2 : 1) * (i64)N * sizeof(T) + (tprev ? 3 : 2) * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
                    if (mid_cg && a >= 2)  // half-height tiles of the working copy, in place
                        rc = plsk::launch_fused_pass<T, 64>(c->stream, c->num_cu, work, ldw, tsw, work, ldw, tsw, N, K, v,
                                                            tprev, pprev, Tm + (i64)a * ldt, part, (int)prow, sspart,
                                                            &nb, &nss, (int)c->opt_fused_grid, 0, true, &tail, &tail_used);
                    else {  // (a == 1 with mid_cg: X in 256-byte segments -> half-height tiles)
#define TALL_PASS(CG_) plsk::launch_fused_pass<T, CG_>(c->stream, c->num_cu, Xc, ldc, tsc, tprev ? work : nullptr, ldw, tsw, N, K, v, tprev, \
                                                       pprev, Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss,                        \
                                                       (int)c->opt_fused_grid, (mid_cg && tprev) ? (int)WR : 0, Xc == work, &tail, &tail_used)
                        rc = tall_cg == 8 ? TALL_PASS(8) : (tall_cg == 16 ? TALL_PASS(16) : TALL_PASS(32));
#undef TALL_PASS
                    }
                    if (rc != 0) s.on = false;  // nothing was launched: drop the event pair
                }
                if (rc == 0) {
                    LAUNCH_CHECK(c);
                    done = true;
                    if (tprev) { Xc = work; ldc = ldw; tsc = tsw; cur_tiled = tiled_work; }
                    if (!tail_used) CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
                } else {
                    return fail(c, PLS_HIP_ERR_DEVICE, "fused pass launch failed");
                }
            } else if (wide_cg && (nipals ? (a >= 2 || retiled) : true)) {
                // short-tile fused pass on the working copy: NIPALS deflates it in place, KERNEL only reads it
                int nb = 0, nss = 0, rc;
                const T *tprev = (nipals && a > 0) ? Tm + (i64)(a - 1) * ldt : nullptr;
                const double *pprev = (nipals && a > 0) ? P + (i64)(a - 1) * K : nullptr;
                if (!nipals && a == 0 && !retiled) {  // the one-time copy into short tiles (before the first component: it runs fused, too)
                    Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T));
                    if (plsk::launch_retile<T>(c->stream, c->num_cu, X, ldx, work, ldw, tsw, (int)WR, N, K) != 0) {
                        s.on = false;
                        return fail(c, PLS_HIP_ERR_DEVICE, "retile launch failed");
                    }
                    LAUNCH_CHECK(c);
                }
                {
                    const i64 bytes = (tprev ? 2 : 1) * ((i64)N * K * sizeof(T) + (i64)N * sizeof(T)) + 3 * (i64)K * 8;
                    Scope s(c, PLS_HIP_FAM_FUSED, bytes);
#define WIDE_PASS(CG_) plsk::launch_fused_pass<T, CG_>(c->stream, c->num_cu, work, ldw, tsw, work, ldw, tsw, N, K, v, tprev, pprev, \
                                                        Tm + (i64)a * ldt, part, (int)prow, sspart, &nb, &nss, (int)c->opt_fused_grid, 0, true, \
                                                        &tail, &tail_used)
                    rc = wide_cg == 8 ? WIDE_PASS(8) : (wide_cg == 16 ? WIDE_PASS(16) : (wide_cg == 32 ? WIDE_PASS(32)
                         : (wide_cg == 64 ? WIDE_PASS(64) : (wide_cg == 128 ? WIDE_PASS(128) : (wide_cg == 256 ? WIDE_PASS(256) : WIDE_PASS(512))))));
#undef WIDE_PASS
                    if (rc != 0) s.on = false;
                }
                if (rc != 0) return fail(c, PLS_HIP_ERR_DEVICE, "short-tile fused pass launch failed");
                LAUNCH_CHECK(c);
                done = true;
                if (!tail_used) CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            }
            if (!done) {
                int nss = 0, nb = 0;
                bool have_t = false;
                if (nipals && a > 0) {  // X_a = X_{a-1} - t_{a-1} p_{a-1}^T (first one out of place)
                    const T *tprev = Tm + (i64)(a - 1) * ldt;
                    const double *pprev = P + (i64)(a - 1) * K;
                    if (semi_fit) {  // wide matrices: deflation and score in one sweep (3NK instead of 4NK)
                        const i64 bytes = 2 * (i64)N * K * sizeof(T) + 2 * (i64)N * sizeof(T) + 2 * (i64)K * 8;
                        Scope s(c, PLS_HIP_FAM_DEFLATE, bytes);
                        const int rc = plsk::launch_deflate_score<T>(c->stream, c->num_cu, Xc, ldc, tsc, work, ldw, tsw,
                                                                     (int)(tiled_work ? WR : TR), N, K, tprev, pprev, v,
                                                                     Tm + (i64)a * ldt, sspart, (int)ssmax, &nss, Xc == work);
                        if (rc != 0) {
                            s.on = false;
                            return fail(c, PLS_HIP_ERR_DEVICE, "deflate+score launch failed");
                        }
                        LAUNCH_CHECK(c);
                        have_t = true;
                    }
                    if (!have_t) CHK(launch_deflate<T>(c, Xc, ldc, work, ldw, N, K, tprev, pprev));
                    Xc = work;
                    ldc = ldw;
                    tsc = tsw;
                    cur_tiled = tiled_work;
                }
                if (!have_t)
                    CHK(launch_xb<T>(c, Xc, ldc, N, K, v, K, 1, Tm + (i64)a * ldt, ldt, sspart, &nss));  // :419-420
                if (cur_tiled) {  // row-tile-major work buffer: the loading in tile addressing (:421)
                    Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + (i64)N * sizeof(T) + (i64)K * 8);
                    const T *ta = Tm + (i64)a * ldt;
                    const int xrc =
                        wide_cg == 64    ? plsk::launch_xty_tiled<T, 64>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                        : wide_cg == 128 ? plsk::launch_xty_tiled<T, 128>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                        : wide_cg == 256 ? plsk::launch_xty_tiled<T, 256>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb)
                                         : plsk::launch_xty_tiled<T, 32>(c->stream, c->num_cu, Xc, ldc, tsc, N, K, ta, part, (int)prow, &nb);
                    if (xrc != 0) {
                        s.on = false;
                        return fail(c, PLS_HIP_ERR_DEVICE, "tiled loading launch failed");
                    }
                    LAUNCH_CHECK(c);
                } else {
                    CHK(launch_xty<T>(c, Xc, ldc, Tm + (i64)a * ldt, ldt, N, K, 1, part, &nb));  // :421
                }
                CHK(launch_reduce(c, part, nb, K, sspart, nss, red));
            }
        } else {
            HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * (K + 1) * 8, c->stream));
        }
        if (tail_used && want_push) {  // pushed from the tail of the pass: gather in the prologue of the update
            const unsigned long long seq = ++*c->xep.seq;
            const int par = (int)(seq & 1);
            plsk::XchgGather gx;
            gx.inbox = c->xep.inbox[c->xep.rank] + (i64)par * c->xep.n * plsk::XCHG_CAP;
            gx.flags = c->xep.flags[c->xep.rank] + par * c->xep.n;
            gx.n = c->xep.n; gx.cap = plsk::XCHG_CAP; gx.seq = seq;
            gx.status = c->xep.status; gx.host_status = c->xep.host_status; gx.limit = *c->xep.limit;
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, nip, &gx));
        } else {
            CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * (K + 1)));
            CHK(launch_update(c, red, XY, W, P, Q, R, v, K, M, A, a, nip));  // :427-433 and :403-416 of a+1
        }
    }
    if (B) {
        Scope s(c, PLS_HIP_FAM_SMALL, ((i64)K * A + (i64)M * A + (i64)K * M) * 8);
        const int nblk = (int)((L0 + plsk::WG - 1) / plsk::WG);
        hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, c->stream, R, Q,
                           K, M, A, B);
        LAUNCH_CHECK(c);
    }
    return replica_guard(c, W, P, Q, R, B, K, M, A);
}

int check_handle(pls_hip_handle h) { return h ? PLS_HIP_OK : PLS_HIP_ERR_INVALID; }

int set_device(pls_hip_context *c) {
    HIPCHK(c, hipSetDevice(c->device));
    return PLS_HIP_OK;
}

void begin_fit_timing(pls_hip_context *c) {
    c->fit_timed = false;
    if (!c->opt_profile) return;
    c->cur_fit.e0 = take_event(c);
    c->cur_fit.e1 = take_event(c);
    if (c->cur_fit.e0 && c->cur_fit.e1) {
        (void)hipEventRecord(c->cur_fit.e0, c->stream);
        c->fit_timed = true;
    }
}
void end_fit_timing(pls_hip_context *c) {
    if (!c->fit_timed) return;
    (void)hipEventRecord(c->cur_fit.e1, c->stream);
    c->fits.push_back(c->cur_fit);
    c->fit_timed = false;
}

// host <-> device staging of a column-major matrix with leading dimension.  Matrices of a few MB and more go
// through the pinned double-buffer pipeline of host_pipeline.hpp (the caller's pageable memory is repacked by host
// threads while the DMA engine moves the previous tile); small ones are one plain copy.
constexpr size_t PIPELINE_MIN_BYTES = (size_t)4 << 20;
int h2d(pls_hip_context *c, void *dst, i64 ldd, const void *src, i64 lds, i64 rows, i64 cols, size_t es) {
    if (rows == 0 || cols == 0) return PLS_HIP_OK;
    if ((size_t)rows * (size_t)cols * es >= PIPELINE_MIN_BYTES) {
        HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
        HIPCHK(c, plsh::upload(c->stager, c->stream, dst, ldd, src, lds, rows, cols, es));
        return PLS_HIP_OK;
    }
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ldd * es, src, (size_t)lds * es, (size_t)rows * es,
                               (size_t)cols, hipMemcpyHostToDevice, c->stream));
    return PLS_HIP_OK;
}
int d2h(pls_hip_context *c, void *dst, i64 ldd, const void *src, i64 lds, i64 rows, i64 cols, size_t es) {
    if (rows == 0 || cols == 0) return PLS_HIP_OK;
    if ((size_t)rows * (size_t)cols * es >= PIPELINE_MIN_BYTES) {
        HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
        HIPCHK(c, plsh::download(c->stager, c->stream, dst, ldd, src, lds, rows, cols, es));
        return PLS_HIP_OK;
    }
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)ldd * es, src, (size_t)lds * es, (size_t)rows * es,
                               (size_t)cols, hipMemcpyDeviceToHost, c->stream));
    return PLS_HIP_OK;
}

// Upload of X (host, rows x K) in ROW blocks with X^T X and X^T Y accumulated block by block on the compute stream
// while the DMA engine already moves the next block: the matrix-core SYRK of a block (2 rb K^2 flops) takes a
// fraction of the block's PCIe time (K * 2e-4 of it), so by the time the last rows have arrived the Gram matrix of
// the whole shard is complete and a Gram-plan fit needs no further pass over X for its component loop.
// dY: the member's rows of Y, already on the device.  *ok = false: the layout does not allow it (plain upload done).
template <typename T>
int upload_accumulate(pls_hip_context *c, T *dX, i64 ldd, const T *hX, i64 ldx, i64 N, int K, const T *dY, i64 ldy,
                      int M, double *XXacc, double *XYacc, bool *ok) {
    constexpr int FV = 16 / sizeof(T);
    const size_t es = sizeof(T);
    *ok = false;
    const i64 KK = (i64)K * K, L0 = (i64)K * M;
    i64 rb = (i64)(plsh::STAGE_BYTES / ((size_t)K * es)) & ~(i64)63;
    const int nbk = (K + plsk::SYRK_TB - 1) / plsk::SYRK_TB;
    const i64 S = std::max<i64>(1, (2 * (i64)c->num_cu) / (nbk * (nbk + 1) / 2));
    if (N < 1 || K > 4096 || M > plsk::LM_MAX || rb < 64 || !vec_ok<T>(dX, ldd, FV) || !vec_ok<T>(dY, ldy, FV) ||
        ensure(c, c->red2, (size_t)plsk::RED_SLICES * KK * 8) != PLS_HIP_OK ||
        ensure(c, c->part, std::max<size_t>((size_t)S * KK, (size_t)max_partial_rows(c, rb, K) * L0) * 8) != PLS_HIP_OK ||
        ensure(c, c->red, (size_t)plsk::RED_SLICES * std::max<i64>(L0, K + 1) * 8) != PLS_HIP_OK) {
        c->err.clear();
        return h2d(c, dX, ldd, hX, ldx, N, K, es);
    }
    HIPCHK(c, c->stager.ensure(c->copy_threads > 0 ? c->copy_threads : plsh::default_copy_threads(), c->device));
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->zeros.p) {
        CHK(ensure(c, c->zeros, 256));
        HIPCHK(c, hipMemsetAsync(c->zeros.p, 0, 256, c->stream));
    }
    HIPCHK(c, hipMemsetAsync(XXacc, 0, (size_t)KK * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(XYacc, 0, (size_t)L0 * 8, c->stream));
    plsh::Stager &st = c->stager;
    double *part = (double *)c->part.p, *red2 = (double *)c->red2.p, *red = (double *)c->red.p;
    bool acc = true;
    Range r_up("upload + X^T X / X^T Y accumulation");
    for (i64 r0 = 0; r0 < N; r0 += rb) {
        const i64 rbn = std::min(rb, N - r0);
        const int s = st.slot;
        if (st.busy[s]) HIPCHK(c, hipEventSynchronize(st.ev[s]));
        plsh::repack(*st.pool, (char *)st.buf[s], (char *)const_cast<T *>(hX), ldx, r0, 0, rbn, K, es, true);
        HIPCHK(c, hipMemcpy2DAsync(dX + r0, (size_t)ldd * es, st.buf[s], (size_t)rbn * es, (size_t)rbn * es, (size_t)K,
                                   hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipEventRecord(st.ev[s], c->copy_stream));
        st.busy[s] = true;
        st.slot ^= 1;
        HIPCHK(c, hipStreamWaitEvent(c->stream, st.ev[s], 0));  // kernels of this block (and of the fit) after its rows
        if (!acc) continue;
        int nb = 0;
        if (plsk::launch_syrk<T>(c->stream, c->num_cu, dX + r0, ldd, rbn, K, part, S * KK, &nb, c->zeros.p) != 0) {
            acc = false;  // ragged block the matrix-core kernel declines: the fit forms X^T X itself
            continue;
        }
        LAUNCH_CHECK(c);
        CHK(launch_reduce(c, part, nb, (int)KK, nullptr, 0, red2));
        hipLaunchKernelGGL(plsk::accumulate_slices_kernel, dim3((unsigned)((KK + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, (const double *)red2, (int)KK, XXacc);
        LAUNCH_CHECK(c);
        CHK(launch_xty<T>(c, dX + r0, ldd, dY + r0, ldy, rbn, K, M, part, &nb));
        CHK(launch_reduce(c, part, nb, (int)L0, nullptr, 0, red));
        hipLaunchKernelGGL(plsk::accumulate_slices_kernel, dim3((unsigned)((L0 + plsk::WG - 1) / plsk::WG)), dim3(plsk::WG), 0,
                           c->stream, (const double *)red, (int)L0, XYacc);
        LAUNCH_CHECK(c);
    }
    *ok = acc;
    return PLS_HIP_OK;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int pls_hip_abi_version(void) { return PLS_HIP_ABI_VERSION; }

int pls_hip_create(pls_hip_handle *out, int device, void *stream) {
    if (!out) return PLS_HIP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        (void)hipGetLastError();
        return PLS_HIP_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return PLS_HIP_ERR_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PLS_HIP_ERR_DEVICE;  // no other target
    pls_hip_context *c = new (std::nothrow) pls_hip_context();
    if (!c) return PLS_HIP_ERR_ALLOC;
    c->device = device;
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipSetDevice(device) != hipSuccess) { delete c; return PLS_HIP_ERR_DEVICE; }
    c->stream = (hipStream_t)stream;  // NULL = the device's default (null) stream
    *out = c;
    return PLS_HIP_OK;
}

int pls_hip_destroy(pls_hip_handle h) {
    if (!h) return PLS_HIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    DevBuf *bufs[] = {&h->tailcnt, &h->zeros, &h->part, &h->sspart, &h->xbpart, &h->wide1, &h->red, &h->red2, &h->xx, &h->xyp, &h->praw, &h->xy, &h->v, &h->cs, &h->coop, &h->lm, &h->gxx, &h->gxy, &h->tab,
                      &h->cvidx, &h->cvx, &h->cvy, &h->cvws, &h->cve, &h->cvtx, &h->cvty, &h->cvtt, &h->cvm, &h->cvkeep, &h->work, &h->hX, &h->hY,
                      &h->hT, &h->hW, &h->hP, &h->hQ, &h->hR, &h->hB, &h->hIn, &h->hOut};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (void *q : h->graveyard) (void)hipFree(q);
    if (h->xchg) xchg_release(h);
    if (h->guard.p) (void)hipFree(h->guard.p);
    if (h->diverged) (void)hipHostFree(h->diverged);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    h->stager.release();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    delete h;
    return PLS_HIP_OK;
}

int pls_hip_set_stream(pls_hip_handle h, void *stream) {
    CHK(check_handle(h));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)stream;
    return PLS_HIP_OK;
}

int pls_hip_set_option(pls_hip_handle h, int option, int64_t value) {
    CHK(check_handle(h));
    switch (option) {
        case PLS_HIP_OPT_ALGO:
            if (value != PLS_HIP_ALGO_KERNEL && value != PLS_HIP_ALGO_NIPALS && value != PLS_HIP_ALGO_GRAM &&
                value != PLS_HIP_ALGO_AUTO)
                return fail(h, PLS_HIP_ERR_INVALID, "unknown algo");
            h->opt_algo = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSE: h->opt_fuse = value ? 1 : 0; return PLS_HIP_OK;
        case PLS_HIP_OPT_PROFILE: h->opt_profile = value < 0 ? 0 : (value > 2 ? 2 : value); return PLS_HIP_OK;
        case PLS_HIP_OPT_POWER_ITERS:
            if (value < 1 || value > 4096) return fail(h, PLS_HIP_ERR_INVALID, "power iters out of range");
            h->opt_power_iters = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSED_GRID:
            if (value < 0 || value > (1 << 20)) return fail(h, PLS_HIP_ERR_INVALID, "fused grid out of range");
            h->opt_fused_grid = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_WORK_LAYOUT: h->opt_work_layout = value ? 1 : 0; return PLS_HIP_OK;
        case PLS_HIP_OPT_DEFER:
            if (value < 1 || value > plsk::DEFER_MAX) return fail(h, PLS_HIP_ERR_INVALID, "defer out of range");
            h->opt_defer = value;
            return PLS_HIP_OK;
        case PLS_HIP_OPT_GRAPH: h->opt_graph = value ? 1 : 0; return PLS_HIP_OK;
        default: return fail(h, PLS_HIP_ERR_INVALID, "unknown option");
    }
}

int pls_hip_get_option(pls_hip_handle h, int option, int64_t *value) {
    CHK(check_handle(h));
    if (!value) return PLS_HIP_ERR_INVALID;
    switch (option) {
        case PLS_HIP_OPT_ALGO: *value = h->opt_algo; return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSE: *value = h->opt_fuse; return PLS_HIP_OK;
        case PLS_HIP_OPT_PROFILE: *value = h->opt_profile; return PLS_HIP_OK;
        case PLS_HIP_OPT_POWER_ITERS: *value = h->opt_power_iters; return PLS_HIP_OK;
        case PLS_HIP_OPT_FUSED_GRID: *value = h->opt_fused_grid; return PLS_HIP_OK;
        case PLS_HIP_OPT_GRAPH: *value = h->opt_graph; return PLS_HIP_OK;
        case PLS_HIP_OPT_WORK_LAYOUT: *value = h->opt_work_layout; return PLS_HIP_OK;
        case PLS_HIP_OPT_DEFER: *value = h->opt_defer; return PLS_HIP_OK;
        default: return fail(h, PLS_HIP_ERR_INVALID, "unknown option");
    }
}

int pls_hip_set_reducer(pls_hip_handle h, pls_hip_allreduce_fn fn, void *user, int rank, int nranks) {
    CHK(check_handle(h));
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn))
        return fail(h, PLS_HIP_ERR_INVALID, "bad reducer arguments");
    h->reducer = fn;
    h->reducer_user = user;
    h->rank = rank;
    h->nranks = nranks;
    return PLS_HIP_OK;
}

int pls_hip_set_reduce_buffer(pls_hip_handle h, void *buf, int64_t count) {
    CHK(check_handle(h));
    if ((buf && count <= 0) || (!buf && count != 0)) return fail(h, PLS_HIP_ERR_INVALID, "bad reduce buffer");
    h->user_red = (double *)buf;
    h->user_red_count = count;
    return PLS_HIP_OK;
}

int pls_hip_synchronize(pls_hip_handle h) {
    CHK(check_handle(h));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_diverged(h);
}

const char *pls_hip_last_error(pls_hip_handle h) { return h ? h->err.c_str() : "null handle"; }

int pls_hip_get_timing(pls_hip_handle h, pls_hip_timing *out) {
    CHK(check_handle(h));
    if (!out) return PLS_HIP_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    CHK(set_device(h));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (const Launch &l : h->fits) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, l.e0, l.e1));
        out->fit_ms += ms;
        out->fits += 1;
    }
    for (const Launch &l : h->launches) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, l.e0, l.e1));
        out->fam_ms[l.fam] += ms;
        out->fam_launches[l.fam] += 1;
        out->fam_bytes[l.fam] += l.bytes;
    }
    h->launches.clear();  // harvested: recycle the event pool
    h->fits.clear();
    h->ev_used = 0;
    return PLS_HIP_OK;
}

int pls_hip_fit(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                int64_t K, int64_t M, int64_t A, int method, int dtype, int mem, double *W, double *P,
                double *Q, double *R, void *T, int64_t ldt, double *B) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    if (method != PLS_HIP_KERNEL_TYPE1 && method != PLS_HIP_KERNEL_TYPE2)
        return fail(h, PLS_HIP_ERR_INVALID, "bad method");
    const bool sharded = h->nranks > 1;
    if (N < 0 || (N == 0 && !sharded) || K < 1 || M < 1 || A < 1 || A > K)
        return fail(h, PLS_HIP_ERR_INVALID, "bad shape: need N>=1, K>=1, M>=1, 1<=A<=K");
    if (K > (1 << 30) || M > plsk::LM_MAX)
        return fail(h, PLS_HIP_ERR_UNSUPPORTED, "more than 1024 responses (or K > 2^30) not supported on the device");
    if (N > 0 && (!X || !Y || (!T && method == PLS_HIP_KERNEL_TYPE1))) return fail(h, PLS_HIP_ERR_INVALID, "null X/Y/T");
    if ((method == PLS_HIP_KERNEL_TYPE2 || h->opt_algo == PLS_HIP_ALGO_GRAM) && K > 32768)  // (AUTO never picks GRAM there)
        return fail(h, PLS_HIP_ERR_UNSUPPORTED, "KERNEL_TYPE2 keeps a K x K matrix and 8 reduction slices of it (77 GB at K = 32768): K <= 32768");
    if (!W || !P || !Q || !R) return fail(h, PLS_HIP_ERR_INVALID, "null W/P/Q/R");
    if (ldx < std::max<i64>(N, 1) || ldy < std::max<i64>(N, 1) || (T && ldt < std::max<i64>(N, 1)))
        return fail(h, PLS_HIP_ERR_INVALID, "leading dimension smaller than N");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const int Ki = (int)K, Mi = (int)M, Ai = (int)A;

    const void *dX = X, *dY = Y;
    void *dT = T;
    double *dW = W, *dP = P, *dQ = Q, *dR = R, *dB = B;
    i64 dldx = ldx, dldy = ldy, dldt = ldt;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = std::max<i64>(N, 1) + (std::max<i64>(N, 1) & 1);  // even ld keeps 16-B columns
        CHK(ensure(h, h->hX, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(ensure(h, h->hT, (size_t)ldn * A * es));
        CHK(ensure(h, h->hW, (size_t)K * A * 8));
        CHK(ensure(h, h->hP, (size_t)K * A * 8));
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)K * M * 8));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        // A plan that works from X^T X (AUTO, GRAM, KERNEL_TYPE2) gets it for free: accumulated on the matrix cores
        // row block by row block while X crosses PCIe (upload_accumulate).  Single rank only: the ranks of a sharded
        // fit must not differ in their plan.
        bool pre = false;
        const bool wants_gram = h->opt_algo == PLS_HIP_ALGO_AUTO || h->opt_algo == PLS_HIP_ALGO_GRAM || method == PLS_HIP_KERNEL_TYPE2;
        const bool single_launch = method == PLS_HIP_KERNEL_TYPE1 && h->opt_algo == PLS_HIP_ALGO_AUTO && h->opt_fuse &&
                                   (plsk::tiny_fit_covers(N, Ki, Mi, Ai, ldn, es) ||
                                    plsk::tiny_fit_m_covers(N, Ki, Mi, Ai, ldn, es));  // (no use for X^T X there)
        if (wants_gram && !single_launch && !h->reducer && N > 0 && K <= 4096 && ensure(h, h->gxx, (size_t)K * K * 8) == PLS_HIP_OK &&
            ensure(h, h->gxy, (size_t)K * M * 8) == PLS_HIP_OK) {
            if (dtype == PLS_HIP_F64)
                CHK(upload_accumulate<double>(h, (double *)h->hX.p, ldn, (const double *)X, ldx, N, Ki, (const double *)h->hY.p, ldn,
                                              Mi, (double *)h->gxx.p, (double *)h->gxy.p, &pre));
            else
                CHK(upload_accumulate<float>(h, (float *)h->hX.p, ldn, (const float *)X, ldx, N, Ki, (const float *)h->hY.p, ldn, Mi,
                                             (double *)h->gxx.p, (double *)h->gxy.p, &pre));
        } else {
            h->err.clear();
            CHK(h2d(h, h->hX.p, ldn, X, ldx, N, K, es));
        }
        if (pre) {
            h->pre_xx = (const double *)h->gxx.p;
            h->pre_xy = (const double *)h->gxy.p;
        }
        dX = h->hX.p; dY = h->hY.p; dT = h->hT.p;
        dW = (double *)h->hW.p; dP = (double *)h->hP.p; dQ = (double *)h->hQ.p; dR = (double *)h->hR.p;
        dB = B ? (double *)h->hB.p : nullptr;
        dldx = dldy = dldt = ldn;
    }
    // PLS_HIP_OPT_GRAPH: a repeated device-memory fit is replayed as one graph launch.  First occurrence of a call: eager (it
    // also sizes every workspace); second: the same enqueue sequence under stream capture (nothing allocates any more),
    // instantiated and launched; from the third on: hipGraphLaunch.  The kernel arguments are baked into the graph, so the key
    // is everything they derive from.
    const bool graphable = h->opt_graph && mem == PLS_HIP_MEM_DEVICE && !h->reducer && h->opt_profile == 0 && !h->user_red &&
                           h->stream != nullptr;  // (the legacy default stream cannot be captured)
    std::vector<uint64_t> key;
    if (graphable) {
        key = {(uint64_t)(uintptr_t)X, (uint64_t)ldx, (uint64_t)(uintptr_t)Y, (uint64_t)ldy, (uint64_t)N, (uint64_t)K, (uint64_t)M,
               (uint64_t)A, (uint64_t)method, (uint64_t)dtype, (uint64_t)(uintptr_t)W, (uint64_t)(uintptr_t)P, (uint64_t)(uintptr_t)Q,
               (uint64_t)(uintptr_t)R, (uint64_t)(uintptr_t)T, (uint64_t)ldt, (uint64_t)(uintptr_t)B, (uint64_t)h->opt_algo,
               (uint64_t)h->opt_fuse, (uint64_t)h->opt_power_iters, (uint64_t)h->opt_fused_grid, (uint64_t)h->opt_work_layout,
               (uint64_t)h->opt_defer, (uint64_t)(uintptr_t)h->stream, (uint64_t)(uintptr_t)h->pre_xx, (uint64_t)(uintptr_t)h->pre_xy};
        if (h->graph_exec && key == h->graph_key) {
            HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
            return PLS_HIP_OK;
        }
    }
    const bool capture = graphable && key == h->graph_seen;
    if (capture) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; h->graph_key.clear(); }
        HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    }
    begin_fit_timing(h);
    int rc;
    if (dtype == PLS_HIP_F64)
        rc = fit_device<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, Ki, Mi, Ai, method,
                                dW, dP, dQ, dR, (double *)dT, dldt, dB);
    else
        rc = fit_device<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, Ki, Mi, Ai, method, dW,
                               dP, dQ, dR, (float *)dT, dldt, dB);
    end_fit_timing(h);
    if (capture) {
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(h->stream, &graph);
        if (rc == PLS_HIP_OK && ce == hipSuccess && graph && hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            h->graph_key = key;
            (void)hipGraphDestroy(graph);
            HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));  // this call's work
        } else {
            if (graph) (void)hipGraphDestroy(graph);
            h->graph_exec = nullptr;
            h->graph_seen.clear();
            (void)hipGetLastError();
            if (rc != PLS_HIP_OK) return rc;
            return fail(h, PLS_HIP_ERR_DEVICE, "stream capture of the fit failed");
        }
    } else if (graphable) {
        h->graph_seen = key;
    }
    if (mem == PLS_HIP_MEM_HOST) h->pre_xx = h->pre_xy = nullptr;
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, W, K, dW, K, K, A, 8));
        CHK(d2h(h, P, K, dP, K, K, A, 8));
        CHK(d2h(h, R, K, dR, K, K, A, 8));
        CHK(d2h(h, Q, M, dQ, M, M, A, 8));
        if (B) CHK(d2h(h, B, K, dB, K, K, M, 8));
        if (method == PLS_HIP_KERNEL_TYPE1) CHK(d2h(h, T, ldt, dT, dldt, N, A, es));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        CHK(check_diverged(h));
    }
    return PLS_HIP_OK;
}

int pls_hip_coefficients(pls_hip_handle h, const double *R, const double *Q, int64_t K, int64_t M,
                         int64_t A, int64_t cc, int mem, double *B) {
    CHK(check_handle(h));
    if (!R || !Q || !B || K < 1 || M < 1 || A < 1 || cc < 0 || cc > A || K > (1 << 30))
        return fail(h, PLS_HIP_ERR_INVALID, "bad coefficients arguments");  // comp <= A: src/pls.cpp:445
    CHK(set_device(h));
    const double *dR = R, *dQ = Q;
    double *dB = B;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)K * M * 8));
        CHK(h2d(h, h->hR.p, K, R, K, K, A, 8));
        CHK(h2d(h, h->hQ.p, M, Q, M, M, A, 8));
        dR = (double *)h->hR.p; dQ = (double *)h->hQ.p; dB = (double *)h->hB.p;
    } else if (mem != PLS_HIP_MEM_DEVICE) {
        return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    }
    const int nblk = (int)((K * M + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::coefficients_kernel, dim3(nblk), dim3(plsk::WG), 0, h->stream, dR, dQ,
                       (int)K, (int)M, (int)cc, dB);
    LAUNCH_CHECK(h);
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, B, K, dB, K, K, M, 8));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_xb(pls_hip_handle h, const void *X, int64_t ldx, int64_t N, int64_t K, const double *Bm,
               int64_t ldb, int64_t C, int dtype, int mem, void *out, int64_t ldo) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 0 || K < 1 || C < 1 || K > (1 << 30) || C > (1 << 20) || ldb < K ||
        ldx < std::max<i64>(N, 1) || ldo < std::max<i64>(N, 1) || !Bm || (N > 0 && (!X || !out)))
        return fail(h, PLS_HIP_ERR_INVALID, "bad xb arguments");
    if (N == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const void *dX = X;
    const double *dBm = Bm;
    void *dO = out;
    i64 dldx = ldx, dldo = ldo, dldb = ldb;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = N + (N & 1);
        CHK(ensure(h, h->hIn, (size_t)ldn * K * es));
        CHK(ensure(h, h->hOut, (size_t)ldn * C * es));
        CHK(ensure(h, h->hB, (size_t)K * C * 8));
        CHK(h2d(h, h->hIn.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hB.p, K, Bm, ldb, K, C, 8));
        dX = h->hIn.p; dO = h->hOut.p; dBm = (const double *)h->hB.p;
        dldx = dldo = ldn; dldb = K;
    } else if (mem != PLS_HIP_MEM_DEVICE) {
        return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    }
    int nss = 0;
    if (dtype == PLS_HIP_F64)
        CHK(launch_xb<double>(h, (const double *)dX, dldx, N, (int)K, dBm, dldb, (int)C, (double *)dO, dldo, nullptr, &nss));
    else
        CHK(launch_xb<float>(h, (const float *)dX, dldx, N, (int)K, dBm, dldb, (int)C, (float *)dO, dldo, nullptr, &nss));
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, out, ldo, dO, dldo, N, C, es));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_xty(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                int64_t K, int64_t M, int dtype, double *XY) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || K < 1 || M < 1 || K > (1 << 30) || M > (1 << 20) || !X || !Y || !XY || ldx < N || ldy < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad xty arguments");
    CHK(set_device(h));
    CHK(ensure(h, h->part, (size_t)max_partial_rows(h, N, (int)K) * (size_t)(K * M) * 8));
    int nb = 0;
    if (dtype == PLS_HIP_F64)
        CHK(launch_xty<double>(h, (const double *)X, ldx, (const double *)Y, ldy, N, (int)K, (int)M, (double *)h->part.p, &nb));
    else
        CHK(launch_xty<float>(h, (const float *)X, ldx, (const float *)Y, ldy, N, (int)K, (int)M, (double *)h->part.p, &nb));
    CHK(ensure(h, h->red, (size_t)plsk::RED_SLICES * (size_t)(K * M) * 8));
    CHK(launch_reduce(h, (const double *)h->part.p, nb, (int)(K * M), nullptr, 0, (double *)h->red.p));
    const int nblk = (int)((K * M + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3(nblk), dim3(plsk::WG), 0, h->stream,
                       (const double *)h->red.p, (int)(K * M), XY);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

int pls_hip_deflate(pls_hip_handle h, const void *src, int64_t lds, void *dst, int64_t ldd, int64_t N,
                    int64_t K, const void *t, const double *p, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || K < 1 || K > (1 << 30) || !src || !dst || !t || !p || lds < N || ldd < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad deflate arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return launch_deflate<double>(h, (const double *)src, lds, (double *)dst, ldd, N, (int)K, (const double *)t, p);
    return launch_deflate<float>(h, (const float *)src, lds, (float *)dst, ldd, N, (int)K, (const float *)t, p);
}

}  // extern "C"

namespace {

// column mean / sd / z-scores on the device (two column-reduction passes + one scale pass)
template <typename T>
int zscores_device(pls_hip_context *c, const T *X, i64 ldx, i64 N, i64 n_total, int K, T *Z, i64 ldz,
                   double *mean, double *sd) {
    constexpr int FV = 16 / sizeof(T);
    constexpr int KC = 16;
    const bool wide = vec_ok<T>(X, ldx, FV) && (!Z || vec_ok<T>(Z, ldz, FV));
    const int vec = wide ? FV : 1;
    const int nkg = (K + KC - 1) / KC;
    const i64 nch = std::max<i64>(1, (N + (i64)plsk::WG * vec - 1) / ((i64)plsk::WG * vec));
    const int G = (int)std::min<i64>(nch, std::max<i64>(1, (8 * c->num_cu) / nkg));
    CHK(ensure(c, c->part, (size_t)G * K * 8));
    CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * K * 8));
    double *part = (double *)c->part.p, *red = (double *)c->red2.p;
    const dim3 grid(G, nkg), blk(plsk::WG);
    // Mean and sd from ONE sweep (colmoments_kernel: per-wave shifted sums merged pairwise) -- two sweeps of X
    // (statistics, scale) for the z-scores instead of the reference's three.  Row-sharded: the shards' (count, mean, M2)
    // triples meet in two all-reduces of K sums (colmoments_shard_kernel), as many as the two-pass form needs.  Every
    // rank takes this branch or none (the environment decides, not the shard), an empty shard contributes zeros.
    static const bool one_pass = !(getenv("PLS_HIP_ZSCORE_ONE_PASS") && atoi(getenv("PLS_HIP_ZSCORE_ONE_PASS")) == 0);
    const bool shifted = one_pass && (c->reducer || (N > 0 && n_total == N));
    if (shifted) {
        // all workgroups resident at once (5 per CU at 84 VGPRs): one round, no tail
        const int G1 = (int)std::min<i64>(nch, std::max<i64>(1, (5 * c->num_cu) / nkg));
        const bool sharded = c->reducer != nullptr;
        // (the shards' sums travel in the layout every collective of the library has -- RED_SLICES slices of K values, the
        // values in slice 0, zeros behind: the device-side exchanges sum the slices of a message into slice 0)
        CHK(ensure(c, c->part, (size_t)(G1 * (3 * (i64)K + 1) + (sharded ? (3 + (i64)plsk::RED_SLICES) * K + 1 : 0)) * 8));
        part = (double *)c->part.p;
        double *cnt = part + (i64)G1 * 3 * K;
        double *tri = sharded ? cnt + G1 : nullptr, *buf = sharded ? tri + 3 * (i64)K + 1 : nullptr;
        if (sharded) HIPCHK(c, hipMemsetAsync(buf + K, 0, (size_t)(plsk::RED_SLICES - 1) * K * 8, c->stream));
        const dim3 gk((K + plsk::WG - 1) / plsk::WG);
        if (N > 0) {
            {
                Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + 2 * (i64)K * 8);
                const dim3 g1(G1, nkg);
                if (wide) hipLaunchKernelGGL((plsk::colmoments_kernel<T, FV, KC>), g1, blk, 0, c->stream, X, ldx, N, K, part, cnt);
                else hipLaunchKernelGGL((plsk::colmoments_kernel<T, 1, KC>), g1, blk, 0, c->stream, X, ldx, N, K, part, cnt);
                LAUNCH_CHECK(c);
            }
            hipLaunchKernelGGL(plsk::colmoments_finish_kernel, gk, blk, 0, c->stream, (const double *)part, (const double *)cnt, G1, K,
                               mean, sd, tri);
            LAUNCH_CHECK(c);
        } else {
            HIPCHK(c, hipMemsetAsync(tri, 0, (size_t)(3 * (i64)K + 1) * 8, c->stream));
        }
        if (sharded) {
            for (int step = 0; step < 3; ++step) {
                hipLaunchKernelGGL(plsk::colmoments_shard_kernel, gk, blk, 0, c->stream, (const double *)tri, K, (double)n_total, step,
                                   buf, mean, sd);
                LAUNCH_CHECK(c);
                if (step < 2) CHK(do_allreduce(c, buf, (i64)plsk::RED_SLICES * K));
            }
        }
    }
    for (int mode = shifted ? 2 : 0; mode < 2; ++mode) {
        if (N > 0) {
            Scope s(c, PLS_HIP_FAM_XTY, (i64)N * K * sizeof(T) + (i64)K * 8);
#define CS_CASE(V_, M_) hipLaunchKernelGGL((plsk::colstat_kernel<T, V_, KC, M_>), grid, blk, 0, c->stream, X, ldx, N, K, mean, part)
            if (wide) { if (mode == 0) CS_CASE(FV, 0); else CS_CASE(FV, 1); }
            else { if (mode == 0) CS_CASE(1, 0); else CS_CASE(1, 1); }
#undef CS_CASE
            LAUNCH_CHECK(c);
            CHK(launch_reduce(c, part, G, K, nullptr, 0, red));
        } else {
            HIPCHK(c, hipMemsetAsync(red, 0, (size_t)plsk::RED_SLICES * K * 8, c->stream));
        }
        CHK(do_allreduce(c, red, (i64)plsk::RED_SLICES * K));
        hipLaunchKernelGGL(plsk::colstat_finish_kernel, dim3((K + plsk::WG - 1) / plsk::WG), blk, 0, c->stream,
                           (const double *)red, K, (double)n_total, mode, mode == 0 ? mean : sd);
        LAUNCH_CHECK(c);
    }
    if (Z && N > 0) {
        Scope s(c, PLS_HIP_FAM_DEFLATE, 2 * (i64)N * K * sizeof(T) + 2 * (i64)K * 8);
        const dim3 g2((unsigned)std::min<i64>(nch, std::max<i64>(1, (16 * c->num_cu) / nkg)), nkg);
        static const bool piece = !(getenv("PLS_HIP_ZS_PIECE") && atoi(getenv("PLS_HIP_ZS_PIECE")) == 0);
        if (wide && piece && K <= 65535)
            hipLaunchKernelGGL((plsk::zscale_piece_kernel<T, FV>), dim3((unsigned)nch, K), blk, 0, c->stream, X, ldx, Z, ldz, N, mean, sd);
        else if (wide) hipLaunchKernelGGL((plsk::zscale_kernel<T, FV, KC>), g2, blk, 0, c->stream, X, ldx, Z, ldz, N, K, mean, sd);
        else hipLaunchKernelGGL((plsk::zscale_kernel<T, 1, KC>), g2, blk, 0, c->stream, X, ldx, Z, ldz, N, K, mean, sd);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

template <typename T>
int sse_device(pls_hip_context *c, const T *S, i64 lds, const T *Y, i64 ldy, i64 N, int A, int M,
               const double *Q, double *SSE) {
    // ranges of component counts with at most 1024 running sums each (the sweep keeps them per wave in LDS)
    const int step = std::max(1, 1024 / M);
    const int G = (int)std::min<i64>(std::max<i64>(1, (N + plsk::WG - 1) / plsk::WG), 4 * (i64)c->num_cu);
    for (int c_lo = 0; c_lo < A; c_lo += step) {
        const int c_hi = std::min(A, c_lo + step), AM = (c_hi - c_lo) * M;
        CHK(ensure(c, c->part, (size_t)G * AM * 8));
        CHK(ensure(c, c->red2, (size_t)plsk::RED_SLICES * AM * 8));
        {
            Scope s(c, PLS_HIP_FAM_XB, (i64)N * (c_hi + M) * sizeof(T) + (i64)AM * 8);
            hipLaunchKernelGGL((plsk::sse_components_kernel<T>), dim3(G), dim3(plsk::WG),
                               (size_t)(plsk::WG / plsk::WAVE) * AM * 8, c->stream, S, lds, Y, ldy, N, c_lo, c_hi, M, Q,
                               (double *)c->part.p);
            LAUNCH_CHECK(c);
        }
        CHK(launch_reduce(c, (const double *)c->part.p, G, AM, nullptr, 0, (double *)c->red2.p));
        CHK(do_allreduce(c, (double *)c->red2.p, (i64)plsk::RED_SLICES * AM));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((AM + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0, c->stream,
                           (const double *)c->red2.p, AM, SSE + (i64)c_lo * M);
        LAUNCH_CHECK(c);
    }
    return PLS_HIP_OK;
}

}  // namespace

extern "C" {

int pls_hip_colwise_z_scores(pls_hip_handle h, const void *X, int64_t ldx, int64_t N, int64_t n_total,
                             int64_t K, int dtype, void *Z, int64_t ldz, double *mean, double *sd) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 0 || n_total < N || n_total < 1 || K < 1 || K > (1 << 30) || !mean || !sd || (N > 0 && !X) ||
        ldx < std::max<i64>(N, 1) || (Z && ldz < std::max<i64>(N, 1)))
        return fail(h, PLS_HIP_ERR_INVALID, "bad z-score arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return zscores_device<double>(h, (const double *)X, ldx, N, n_total, (int)K, (double *)Z, ldz, mean, sd);
    return zscores_device<float>(h, (const float *)X, ldx, N, n_total, (int)K, (float *)Z, ldz, mean, sd);
}

int pls_hip_sse_by_components(pls_hip_handle h, const void *S, int64_t lds, const void *Y, int64_t ldy,
                              int64_t N, int64_t A, int64_t M, const double *Q, int dtype, double *SSE) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (N < 1 || A < 1 || M < 1 || M > 1024 || A > (1 << 20) || !S || !Y || !Q || !SSE || lds < N || ldy < N)
        return fail(h, PLS_HIP_ERR_INVALID, "bad sse arguments");
    CHK(set_device(h));
    if (dtype == PLS_HIP_F64)
        return sse_device<double>(h, (const double *)S, lds, (const double *)Y, ldy, N, (int)A, (int)M, Q, SSE);
    return sse_device<float>(h, (const float *)S, lds, (const float *)Y, ldy, N, (int)A, (int)M, Q, SSE);
}

}  // extern "C"

namespace {

// device part of pls_hip_cv_folds on storage type T (X, Y device pointers; E device pointer)
template <typename T>
int cv_folds_device(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                    const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    const i64 K = Ki, M = Mi, A = Ai;
    const plsk::CvLayout L(Ki, Mi, Ai, ts);
    CHK(ensure(h, h->xx, (size_t)K * K * 8));
    CHK(ensure(h, h->xy, (size_t)K * M * 8));
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    CHK(ensure(h, h->cvx, (size_t)nobs * K * 8));
    CHK(ensure(h, h->cvy, (size_t)nobs * M * 8));
    CHK(ensure(h, h->cvws, (size_t)num_folds * (size_t)L.total * 8));
    double *XX = (double *)h->xx.p, *XYd = (double *)h->xy.p;
    // XX and XY of the whole matrix, once (or taken from the upload that already formed them)
    if (h->pre_xx && h->pre_xy) {
        HIPCHK(h, hipMemcpyAsync(XX, h->pre_xx, (size_t)K * K * 8, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(XYd, h->pre_xy, (size_t)K * M * 8, hipMemcpyDeviceToDevice, h->stream));
    } else {
        CHK(compute_xx<T>(h, dX, dldx, N, Ki, XX));
    }
    if (!(h->pre_xx && h->pre_xy)) {
        CHK(ensure(h, h->part, (size_t)max_partial_rows(h, N, Ki) * (size_t)(K * M) * 8));
        CHK(ensure(h, h->red, (size_t)plsk::RED_SLICES * std::max<i64>(K * M, K + 1) * 8));
        int nb = 0;
        CHK(launch_xty<T>(h, dX, dldx, dY, dldy, N, Ki, Mi, (double *)h->part.p, &nb));
        CHK(launch_reduce(h, (const double *)h->part.p, nb, Ki * Mi, nullptr, 0, (double *)h->red.p));
        hipLaunchKernelGGL(plsk::sum_slices_kernel, dim3((Ki * Mi + plsk::WG - 1) / plsk::WG), dim3(plsk::WG), 0,
                           h->stream, (const double *)h->red.p, Ki * Mi, XYd);
        LAUNCH_CHECK(h);
    }
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((plsk::cv_gather_kernel<T>), dim3((unsigned)nobs), dim3(plsk::WG), 0, h->stream, dX, dldx, dY,
                       dldy, Ki, Mi, (const i64 *)h->cvidx.p, (double *)h->cvx.p, (double *)h->cvy.p);
    LAUNCH_CHECK(h);
    {
        Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * A * ((i64)K * K + 4 * K) * 8);
        hipLaunchKernelGGL(plsk::cv_folds_kernel, dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), (size_t)A * 8,
                           h->stream, (const double *)XX, (const double *)XYd, (const double *)h->cvx.p,
                           (const double *)h->cvy.p, Ki, Mi, Ai, ts, (double *)h->cvws.p, dE, (int)h->opt_power_iters);
        LAUNCH_CHECK(h);
    }
    return PLS_HIP_OK;
}

// Small single-response data (the reference's examples): every fold is a single-launch fit (tiny_kernels.hpp) on the whole X
// with its held-out rows masked, one workgroup per fold -- no X^T X at all, which for N < K is the smaller object anyway.
template <typename T>
int cv_folds_tiny(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 N, int Ki, int Ai, const int64_t *test_idx, int ts,
                  i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    const size_t lds = (size_t)2 * Ki * Ai * 8;
    if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_kernel<T>, (int)plsk::TINY_LDS_MAX)  /* raised once per device: to the most any fit asks for */)
        return fail(h, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");
    Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * N * Ki * (i64)sizeof(T));
    hipLaunchKernelGGL((plsk::tiny_fit_kernel<T>), dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), lds, h->stream, dX, dldx, dY,
                       (int)N, Ki, Ai, (double *)nullptr, (double *)nullptr, (double *)nullptr, (double *)nullptr, (T *)nullptr,
                       (i64)0, (double *)nullptr, (const i64 *)h->cvidx.p, ts, nobs, dE);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

// ... the same for 2..8 responses (tiny_fit_m_kernel in fold mode)
template <typename T>
int cv_folds_tiny_m(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                    const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    const size_t lds = (size_t)(2 * Ki + Mi) * Ai * 8;
    Scope s(h, PLS_HIP_FAM_SMALL, (i64)num_folds * N * Ki * (i64)sizeof(T));
#define TINY_M(MM_)                                                                                                          \
    do {                                                                                                                     \
        if (!plsk::raise_dynamic_lds((const void *)plsk::tiny_fit_m_kernel<T, MM_>, (int)plsk::TINY_LDS_MAX))               \
            return fail(h, PLS_HIP_ERR_DEVICE, "dynamic LDS limit of the single-launch fit could not be raised");            \
        hipLaunchKernelGGL((plsk::tiny_fit_m_kernel<T, MM_>), dim3((unsigned)num_folds), dim3(plsk::UPD_THREADS), lds, h->stream, dX, \
                           dldx, dY, dldy, (int)N, Ki, Mi, Ai, (int)h->opt_power_iters, (double *)nullptr, (double *)nullptr,  \
                           (double *)nullptr, (double *)nullptr, (T *)nullptr, (i64)0, (double *)nullptr,                    \
                           (const i64 *)h->cvidx.p, ts, nobs, dE);                                                           \
    } while (0)
    if (Mi <= 2) TINY_M(2); else if (Mi <= 4) TINY_M(4); else TINY_M(8);
#undef TINY_M
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

// The general form of the same call: one refit per fold on the rows that are not in its test set -- what the reference
// does (src/pls.cpp:478-488, :524-545), with the training rows gathered on the device and the fit running under the
// handle's own plan.  Serves the shapes the batched kernel declines (M > 32, A > 4096, K > 16384, a workspace that does
// not fit); costs num_folds fits.
template <typename T>
int cv_folds_refit(pls_hip_context *h, const T *dX, i64 dldx, const T *dY, i64 dldy, i64 N, int Ki, int Mi, int Ai,
                   const int64_t *test_idx, int ts, i64 num_folds, double *dE) {
    const i64 nobs = num_folds * ts;
    const i64 K = Ki, M = Mi, A = Ai;
    const i64 ldtr = (N + 3) & ~(i64)3;
    CHK(ensure(h, h->cvidx, (size_t)nobs * 8));
    CHK(ensure(h, h->cvx, (size_t)nobs * K * 8));
    CHK(ensure(h, h->cvy, (size_t)nobs * M * 8));
    CHK(ensure(h, h->cvkeep, (size_t)N * 8));
    CHK(ensure(h, h->cvtx, (size_t)ldtr * K * sizeof(T)));
    CHK(ensure(h, h->cvty, (size_t)ldtr * M * sizeof(T)));
    CHK(ensure(h, h->cvtt, (size_t)ldtr * A * sizeof(T)));
    CHK(ensure(h, h->cvm, (size_t)(3 * K * A + M * A + (i64)ts * A) * 8));
    double *Wf = (double *)h->cvm.p, *Pf = Wf + K * A, *Rf = Pf + K * A, *Qf = Rf + K * A, *us = Qf + M * A;
    HIPCHK(h, hipMemcpyAsync(h->cvidx.p, test_idx, (size_t)nobs * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL((plsk::cv_gather_kernel<T>), dim3((unsigned)nobs), dim3(plsk::WG), 0, h->stream, dX, dldx, dY,
                       dldy, Ki, Mi, (const i64 *)h->cvidx.p, (double *)h->cvx.p, (double *)h->cvy.p);
    LAUNCH_CHECK(h);
    const double *saved_xx = h->pre_xx, *saved_xy = h->pre_xy;  // products of ALL rows: not a fold's
    h->pre_xx = h->pre_xy = nullptr;
    std::vector<char> held(N, 0);
    std::vector<int64_t> keep(N);
    int rc = PLS_HIP_OK;
    for (i64 f = 0; f < num_folds && rc == PLS_HIP_OK; ++f) {
        for (int i = 0; i < ts; ++i) held[test_idx[f * ts + i]] = 1;
        i64 ntr = 0;
        for (i64 r = 0; r < N; ++r)
            if (!held[r]) keep[ntr++] = r;
        for (int i = 0; i < ts; ++i) held[test_idx[f * ts + i]] = 0;
        if (ntr < 1 || A > K) { rc = fail(h, PLS_HIP_ERR_INVALID, "cv_folds: a fold leaves no training rows"); break; }
        if (hipMemcpyAsync(h->cvkeep.p, keep.data(), (size_t)ntr * 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            rc = fail(h, PLS_HIP_ERR_DEVICE, "cv_folds: upload of the training row list failed");
            break;
        }
        const unsigned gx = (unsigned)((ntr + plsk::WG - 1) / plsk::WG);
        hipLaunchKernelGGL((plsk::gather_rows_kernel<T>), dim3(gx, (unsigned)std::min<i64>(K, 1024)), dim3(plsk::WG), 0, h->stream,
                           dX, dldx, (const i64 *)h->cvkeep.p, ntr, Ki, (T *)h->cvtx.p, ldtr);
        hipLaunchKernelGGL((plsk::gather_rows_kernel<T>), dim3(gx, (unsigned)std::min<i64>(M, 1024)), dim3(plsk::WG), 0, h->stream,
                           dY, dldy, (const i64 *)h->cvkeep.p, ntr, Mi, (T *)h->cvty.p, ldtr);
        rc = fit_device<T>(h, (const T *)h->cvtx.p, ldtr, (const T *)h->cvty.p, ldtr, ntr, Ki, Mi, Ai, PLS_HIP_KERNEL_TYPE1,
                           Wf, Pf, Qf, Rf, (T *)h->cvtt.p, ldtr, nullptr);
        if (rc != PLS_HIP_OK) break;
        hipLaunchKernelGGL(plsk::cv_refit_residuals_kernel, dim3((unsigned)ts), dim3(plsk::WG), 0, h->stream,
                           (const double *)h->cvx.p + f * ts * K, (const double *)h->cvy.p + f * ts * M, (const double *)Rf,
                           (const double *)Qf, Ki, Mi, Ai, ts, f, nobs, us, dE);
        // `keep` is rewritten for the next fold: its copy must have been consumed
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess)
            rc = fail(h, PLS_HIP_ERR_DEVICE, "cv_folds: a fold's refit failed on the device");
    }
    h->pre_xx = saved_xx;
    h->pre_xy = saved_xy;
    return rc;
}

// the batched kernel's shapes (cv_kernels.hpp): everything M-sized in one workgroup's LDS, X^T X resident
bool cv_batched_covers(i64 K, i64 M, i64 A) {
    const bool force_refit = getenv("PLS_HIP_CV_REFIT") && atoi(getenv("PLS_HIP_CV_REFIT")) != 0;
    return !force_refit && A <= 4096 && K <= 16384 && (M == 1 || M <= plsk::MMAX);
}

}  // namespace

extern "C" {

int pls_hip_cv_folds(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                     int64_t K, int64_t M, int64_t A, const int64_t *test_idx, int64_t test_size,
                     int64_t num_folds, int dtype, int mem, double *E) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    if (h->reducer) return fail(h, PLS_HIP_ERR_UNSUPPORTED, "cv_folds: not available on a sharded handle");
    if (N < 2 || K < 1 || M < 1 || A < 1 || A > K || K > (1 << 30) || !X || !Y ||
        !test_idx || !E || test_size < 1 || test_size >= N || num_folds < 1 || ldx < N || ldy < N ||
        num_folds > (1 << 22) || test_size > (1 << 20))
        return fail(h, PLS_HIP_ERR_INVALID, "bad cv_folds arguments");
    if (M > plsk::LM_MAX) return fail(h, PLS_HIP_ERR_UNSUPPORTED, "more than 1024 responses not supported on the device");
    const i64 nobs = num_folds * test_size;
    for (i64 j = 0; j < nobs; ++j)
        if (test_idx[j] < 0 || test_idx[j] >= N) return fail(h, PLS_HIP_ERR_INVALID, "cv_folds: test index out of range");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const void *dX = X, *dY = Y;
    i64 dldx = ldx, dldy = ldy;
    if (mem == PLS_HIP_MEM_HOST) {
        const i64 ldn = N + ((-N) & 3);  // 16-byte columns for either type
        CHK(ensure(h, h->hX, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(h2d(h, h->hX.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        dX = h->hX.p; dY = h->hY.p;
        dldx = dldy = ldn;
    }
    CHK(ensure(h, h->cve, (size_t)nobs * A * M * 8));
    double *dE = (mem == PLS_HIP_MEM_HOST) ? (double *)h->cve.p : E;
    int rc = PLS_HIP_ERR_ALLOC;
    const bool tiny = M == 1 && plsk::tiny_fit_covers(N, (int)K, 1, (int)A, dldx, es) && !getenv("PLS_HIP_CV_REFIT") &&
                      !(getenv("PLS_HIP_TINY") && atoi(getenv("PLS_HIP_TINY")) == 0);
    const bool tiny_m = plsk::tiny_fit_m_covers(N, (int)K, (int)M, (int)A, dldx, es) && !getenv("PLS_HIP_CV_REFIT") &&
                        !(getenv("PLS_HIP_TINY") && atoi(getenv("PLS_HIP_TINY")) == 0);
    if (tiny) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_tiny<double>(h, (const double *)dX, dldx, (const double *)dY, N, (int)K, (int)A, test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_tiny<float>(h, (const float *)dX, dldx, (const float *)dY, N, (int)K, (int)A, test_idx, (int)test_size, num_folds, dE);
    } else if (tiny_m) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_tiny_m<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                         (int)test_size, num_folds, dE);
        else
            rc = cv_folds_tiny_m<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A, test_idx,
                                        (int)test_size, num_folds, dE);
    } else if (cv_batched_covers(K, M, A)) {
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_device<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A,
                                         test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_device<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A,
                                        test_idx, (int)test_size, num_folds, dE);
    }
    if (rc == PLS_HIP_ERR_ALLOC) {  // declined, or the per-fold workspaces of the batched form do not fit: one refit per fold
        h->err.clear();
        if (dtype == PLS_HIP_F64)
            rc = cv_folds_refit<double>(h, (const double *)dX, dldx, (const double *)dY, dldy, N, (int)K, (int)M, (int)A,
                                        test_idx, (int)test_size, num_folds, dE);
        else
            rc = cv_folds_refit<float>(h, (const float *)dX, dldx, (const float *)dY, dldy, N, (int)K, (int)M, (int)A,
                                       test_idx, (int)test_size, num_folds, dE);
    }
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST)
        HIPCHK(h, hipMemcpyAsync(E, dE, (size_t)nobs * A * M * 8, hipMemcpyDeviceToHost, h->stream));
    // the index list is host memory of the caller: the copy above must have consumed it before we return
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PLS_HIP_OK;
}

int pls_hip_model_sse(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy, int64_t N,
                      int64_t K, int64_t M, int64_t A, const double *R, const double *Q, int dtype, int mem,
                      double *SSE) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (mem != PLS_HIP_MEM_HOST && mem != PLS_HIP_MEM_DEVICE) return fail(h, PLS_HIP_ERR_INVALID, "bad mem kind");
    const bool empty_member = (N == 0 && h->nranks > 1);  // an empty shard still takes part in the reduction
    if (N < 0 || (N == 0 && !empty_member) || K < 1 || M < 1 || A < 1 || M > 1024 || A > (1 << 20) || K > (1 << 30) ||
        (N > 0 && (!X || !Y)) || !R || !Q || !SSE || ldx < std::max<i64>(N, 1) || ldy < std::max<i64>(N, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad model_sse arguments");
    CHK(set_device(h));
    const size_t es = esize(dtype);
    const i64 ldn = std::max<i64>(N, 1) + (std::max<i64>(N, 1) & 1);
    const void *dX = X, *dY = Y;
    const double *dR = R, *dQ = Q;
    double *dE = SSE;
    i64 dldx = ldx, dldy = ldy;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(ensure(h, h->hIn, (size_t)ldn * K * es));
        CHK(ensure(h, h->hY, (size_t)ldn * M * es));
        CHK(ensure(h, h->hR, (size_t)K * A * 8));
        CHK(ensure(h, h->hQ, (size_t)M * A * 8));
        CHK(ensure(h, h->hB, (size_t)M * A * 8));
        CHK(h2d(h, h->hIn.p, ldn, X, ldx, N, K, es));
        CHK(h2d(h, h->hY.p, ldn, Y, ldy, N, M, es));
        CHK(h2d(h, h->hR.p, K, R, K, K, A, 8));
        CHK(h2d(h, h->hQ.p, M, Q, M, M, A, 8));
        dX = h->hIn.p; dY = h->hY.p; dR = (const double *)h->hR.p; dQ = (const double *)h->hQ.p;
        dE = (double *)h->hB.p;
        dldx = dldy = ldn;
    }
    CHK(ensure(h, h->hOut, (size_t)ldn * A * es));  // the scores S = X R stay on the device
    int nss = 0, rc;
    if (dtype == PLS_HIP_F64) {
        if (N > 0) CHK(launch_xb<double>(h, (const double *)dX, dldx, N, (int)K, dR, K, (int)A, (double *)h->hOut.p, ldn, nullptr, &nss));
        rc = sse_device<double>(h, (const double *)h->hOut.p, ldn, (const double *)dY, dldy, N, (int)A, (int)M, dQ, dE);
    } else {
        if (N > 0) CHK(launch_xb<float>(h, (const float *)dX, dldx, N, (int)K, dR, K, (int)A, (float *)h->hOut.p, ldn, nullptr, &nss));
        rc = sse_device<float>(h, (const float *)h->hOut.p, ldn, (const float *)dY, dldy, N, (int)A, (int)M, dQ, dE);
    }
    if (rc != PLS_HIP_OK) return rc;
    if (mem == PLS_HIP_MEM_HOST) {
        CHK(d2h(h, SSE, M, dE, M, M, A, 8));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return PLS_HIP_OK;
}

int pls_hip_synth_x(pls_hip_handle h, void *X, int64_t ldx, int64_t row0, int64_t nrows, int64_t K,
                    uint64_t seed, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (nrows < 0 || row0 < 0 || K < 1 || K > (1 << 28) || (nrows > 0 && !X) || ldx < std::max<i64>(nrows, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad synth_x arguments");
    if (nrows == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    CHK(ensure(h, h->tab, (size_t)K * (plsk::SYN_F + 1) * 8));
    const uint64_t sE = plsk::mix64(seed), sZ = plsk::mix64(seed + 1), sL = plsk::mix64(seed + 2);
    const int ntab = (int)((K * plsk::SYN_F + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::synth_table_kernel, dim3(ntab), dim3(plsk::WG), 0, h->stream,
                       (double *)h->tab.p, (int)K, sL, 0, plsk::mix64(seed + 5));
    LAUNCH_CHECK(h);
    constexpr int KC = 64;
    const dim3 grid((unsigned)((nrows + plsk::WG - 1) / plsk::WG), (unsigned)((K + KC - 1) / KC));
    if (dtype == PLS_HIP_F64)
        hipLaunchKernelGGL((plsk::synth_x_kernel<double, KC>), grid, dim3(plsk::WG), 0, h->stream,
                           (double *)X, ldx, row0, nrows, (int)K, sE, sZ, (const double *)h->tab.p);
    else
        hipLaunchKernelGGL((plsk::synth_x_kernel<float, KC>), grid, dim3(plsk::WG), 0, h->stream,
                           (float *)X, ldx, row0, nrows, (int)K, sE, sZ, (const double *)h->tab.p);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

int pls_hip_synth_y(pls_hip_handle h, void *Y, int64_t ldy, int64_t row0, int64_t nrows, int64_t M,
                    uint64_t seed, int dtype) {
    CHK(check_handle(h));
    if (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32) return fail(h, PLS_HIP_ERR_INVALID, "bad dtype");
    if (nrows < 0 || row0 < 0 || M < 1 || M > (1 << 20) || (nrows > 0 && !Y) || ldy < std::max<i64>(nrows, 1))
        return fail(h, PLS_HIP_ERR_INVALID, "bad synth_y arguments");
    if (nrows == 0) return PLS_HIP_OK;
    CHK(set_device(h));
    // the Y table shares the workspace with the X table: generate Y before or after X, both
    // are stream-ordered
    CHK(ensure(h, h->tab, (size_t)std::max<i64>(M, 1) * plsk::SYN_F * 8));
    const uint64_t sZ = plsk::mix64(seed + 1), sC = plsk::mix64(seed + 3), sN = plsk::mix64(seed + 4);
    const int ntab = (int)((M * plsk::SYN_F + plsk::WG - 1) / plsk::WG);
    hipLaunchKernelGGL(plsk::synth_table_kernel, dim3(ntab), dim3(plsk::WG), 0, h->stream,
                       (double *)h->tab.p, (int)M, sC, 1, (uint64_t)0);
    LAUNCH_CHECK(h);
    const dim3 grid((unsigned)((nrows + plsk::WG - 1) / plsk::WG));
    if (dtype == PLS_HIP_F64)
        hipLaunchKernelGGL((plsk::synth_y_kernel<double>), grid, dim3(plsk::WG), 0, h->stream,
                           (double *)Y, ldy, row0, nrows, (int)M, sZ, sN, (const double *)h->tab.p);
    else
        hipLaunchKernelGGL((plsk::synth_y_kernel<float>), grid, dim3(plsk::WG), 0, h->stream, (float *)Y,
                           ldy, row0, nrows, (int)M, sZ, sN, (const double *)h->tab.p);
    LAUNCH_CHECK(h);
    return PLS_HIP_OK;
}

}  // extern "C"

// =============================================================================================
// One process per GPU without RCCL: the device-side exchange across processes (include/pls_hip.h, pls_hip_xchg_*)
// =============================================================================================
struct XchgIpc {
    int rank = 0, n = 1, device = 0;
    double *inbox[plsk::XCHG_MAX] = {nullptr};
    unsigned long long *flags[plsk::XCHG_MAX] = {nullptr};
    bool opened[plsk::XCHG_MAX] = {false};
    int *status = nullptr, *host_status = nullptr, *host_status_dev = nullptr;
    unsigned long long seq = 0;
    long long limit = 0;
    bool connected = false;
};

namespace {

struct XchgBlob {  // what a rank publishes (PLS_HIP_XCHG_HANDLE_BYTES)
    hipIpcMemHandle_t inbox, flags;
    int64_t pid;
    int32_t rank, nranks;
    char pad[PLS_HIP_XCHG_HANDLE_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 16];
};
static_assert(sizeof(XchgBlob) == PLS_HIP_XCHG_HANDLE_BYTES, "exchange blob size");

int ipc_allreduce(void *user, void *buf, int64_t count, void *stream) {
    pls_hip_context *c = static_cast<pls_hip_context *>(user);
    XchgIpc *x = c->xchg;
    if (!x || !x->connected || count % plsk::RED_SLICES != 0) return 20;
    const i64 L = count / plsk::RED_SLICES;
    for (i64 j0 = 0; j0 < L; j0 += plsk::XCHG_CAP) {  // (long messages -- X^T X of KERNEL_TYPE2 -- in pieces)
        const int Lc = (int)std::min<i64>(plsk::XCHG_CAP, L - j0);
        const int rc = plsk::xchg_launch_piece((hipStream_t)stream, x->n, x->rank, x->inbox, x->flags, (double *)buf, L, j0, Lc,
                                               plsk::RED_SLICES, ++x->seq, x->status, x->host_status_dev, x->limit);
        if (rc != 0) return rc;
    }
    return 0;
}

void xchg_release(pls_hip_context *c) {
    XchgIpc *x = c->xchg;
    if (!x) return;
    c->xep = pls_hip_context::XchgEndpoint();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int j = 0; j < x->n; ++j)
        if (x->opened[j]) {
            (void)hipIpcCloseMemHandle(x->inbox[j]);
            (void)hipIpcCloseMemHandle(x->flags[j]);
        }
    if (x->inbox[x->rank]) (void)hipFree(x->inbox[x->rank]);
    if (x->flags[x->rank]) (void)hipFree(x->flags[x->rank]);
    if (x->host_status) (void)hipHostFree(x->host_status);
    (void)hipGetLastError();
    delete x;
    c->xchg = nullptr;
}

int check_xchg(pls_hip_context *c) {
    if (c->xchg && c->xchg->host_status && *c->xchg->host_status)
        return fail(c, PLS_HIP_ERR_REDUCER, "device-side exchange: this rank waited longer than the time limit for its peers' partial "
                                            "sums (a rank failed or fell out of step); the exchange is unusable until it is set up again");
    return PLS_HIP_OK;
}

}  // namespace

extern "C" {

int pls_hip_xchg_create(pls_hip_handle h, int rank, int nranks, void *mine) {
    CHK(check_handle(h));
    if (!mine || nranks < 2 || nranks > plsk::XCHG_MAX || rank < 0 || rank >= nranks)
        return fail(h, PLS_HIP_ERR_INVALID, "bad exchange arguments (2..16 ranks)");
    CHK(set_device(h));
    if (h->xchg) xchg_release(h);
    XchgIpc *x = new (std::nothrow) XchgIpc();
    if (!x) return PLS_HIP_ERR_ALLOC;
    x->rank = rank; x->n = nranks; x->device = h->device;
    h->xchg = x;
    const size_t ib = (size_t)2 * nranks * plsk::XCHG_CAP * 8, fb = (size_t)2 * nranks * 8 + 64;
    XchgBlob blob;
    std::memset(&blob, 0, sizeof(blob));
    if (hipExtMallocWithFlags((void **)&x->inbox[rank], ib, hipDeviceMallocFinegrained) != hipSuccess ||
        hipExtMallocWithFlags((void **)&x->flags[rank], fb, hipDeviceMallocFinegrained) != hipSuccess ||
        hipMemset(x->flags[rank], 0, fb) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipHostMalloc((void **)&x->host_status, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&x->host_status_dev, x->host_status, 0) != hipSuccess ||
        hipIpcGetMemHandle(&blob.inbox, x->inbox[rank]) != hipSuccess || hipIpcGetMemHandle(&blob.flags, x->flags[rank]) != hipSuccess) {
        const std::string why = hipGetErrorString(hipGetLastError());
        xchg_release(h);
        return fail(h, PLS_HIP_ERR_DEVICE, "exchange memory could not be set up or exported: " + why);
    }
    *x->host_status = 0;
    x->status = reinterpret_cast<int *>(x->flags[rank] + 2 * nranks);
    x->limit = plsk::xchg_time_limit(h->device);
    blob.pid = (int64_t)getpid();
    blob.rank = rank; blob.nranks = nranks;
    std::memcpy(mine, &blob, sizeof(blob));
    return PLS_HIP_OK;
}

int pls_hip_xchg_connect(pls_hip_handle h, const void *all) {
    CHK(check_handle(h));
    XchgIpc *x = h->xchg;
    if (!x || !all) return fail(h, PLS_HIP_ERR_INVALID, "pls_hip_xchg_create first");
    CHK(set_device(h));
    const XchgBlob *blobs = static_cast<const XchgBlob *>(all);
    for (int j = 0; j < x->n; ++j) {
        if (blobs[j].rank != j || blobs[j].nranks != x->n) return fail(h, PLS_HIP_ERR_INVALID, "exchange handles are not in rank order");
        if (j == x->rank) continue;
        if (hipIpcOpenMemHandle((void **)&x->inbox[j], blobs[j].inbox, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
            hipIpcOpenMemHandle((void **)&x->flags[j], blobs[j].flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            const std::string why = hipGetErrorString(hipGetLastError());
            return fail(h, PLS_HIP_ERR_DEVICE, "inbox of rank " + std::to_string(j) + " could not be opened: " + why);
        }
        x->opened[j] = true;
    }
    x->connected = true;
    h->xep.on = true;
    h->xep.n = x->n; h->xep.rank = x->rank;
    h->xep.inbox = x->inbox; h->xep.flags = x->flags;
    h->xep.seq = &x->seq; h->xep.status = x->status; h->xep.host_status = x->host_status_dev; h->xep.limit = &x->limit;
    return pls_hip_set_reducer(h, ipc_allreduce, h, x->rank, x->n);
}

int pls_hip_xchg_selftest(pls_hip_handle h) {
    CHK(check_handle(h));
    XchgIpc *x = h->xchg;
    if (!x || !x->connected) return fail(h, PLS_HIP_ERR_INVALID, "pls_hip_xchg_connect first");
    CHK(set_device(h));
    CHK(ensure(h, h->guard, (size_t)plsk::RED_SLICES * 8 * 8));
    double host[plsk::RED_SLICES * 8] = {0};
    for (int j = 0; j < 8; ++j) host[j] = (x->rank + 1.0) * (j + 1);  // slice 0; the other slices stay zero
    HIPCHK(h, hipMemcpyAsync(h->guard.p, host, sizeof(host), hipMemcpyHostToDevice, h->stream));
    const long long keep = x->limit;
    x->limit = std::min(keep, plsk::xchg_time_limit(h->device) / 6 + 1);  // a short limit for the probe round (5 s by default)
    const int rc = ipc_allreduce(h, h->guard.p, plsk::RED_SLICES * 8, (void *)h->stream);
    x->limit = keep;
    if (rc != 0) return fail(h, PLS_HIP_ERR_REDUCER, "exchange launch failed");
    HIPCHK(h, hipMemcpyAsync(host, h->guard.p, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    CHK(check_xchg(h));
    for (int j = 0; j < 8; ++j)
        if (host[j] != 0.5 * x->n * (x->n + 1) * (j + 1)) return fail(h, PLS_HIP_ERR_REDUCER, "exchange self-test: wrong sums");
    return PLS_HIP_OK;
}

int pls_hip_xchg_destroy(pls_hip_handle h) {
    CHK(check_handle(h));
    if (h->xchg) {
        xchg_release(h);
        return pls_hip_set_reducer(h, nullptr, nullptr, 0, 1);
    }
    return PLS_HIP_OK;
}

}  // extern "C"

#include "group_impl.hpp"
