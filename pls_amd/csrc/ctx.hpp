// ctx.hpp -- the handle (pls_hip_context), its workspace buffers, status / launch-check macros, HIP-event and roctx brackets.
// Part of libpls_hip.so: included by pls_hip.hip (one translation unit), in the order given there.
#pragma once

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
    DevBuf resident;  // resident_fit_kernel: two arrival counters (256 bytes) + [2][G][LP] partial vectors
    DevBuf rgflags;   // resident_gram_fit_kernel: one arrival word per workgroup (1 KB), values grow from launch to launch
    unsigned rg_epoch = 0;
    unsigned long long resident_launches = 0;
    // The environment switches of the library, read ONCE when the handle is created (INTEGRATION.md lists them):
    //   PLS_HIP_TINY=0           small fits on the general plan instead of the single-launch kernels (tests compare the two)
    //   PLS_HIP_CV_REFIT=1       pls_hip_cv_folds as one device refit per fold (the general form; tests compare)
    //   PLS_HIP_TAIL=0           partial rows summed by reduce_partials_kernel behind the pass, sharded collectives through
    //                            the reducer call (the launches of round 3; A/B measurements)
    //   PLS_HIP_TAIL=1 / 2       the one-response update as the last act of the tail -- ONE launch per component, sharded over
    //                            the device-side exchange as well (push, wait for the peers, update) -- never (1) / behind every
    //                            pass (2).  Default (unset): behind READ-ONLY passes only; behind a deflating sweep the tail's loads
    //                            queue behind the write drain and the update is faster as a launch of its own
    //                            (profiles/r5/tail_ab.txt)
    //   PLS_HIP_RESIDENT=0       mid-size single-response fits on the general plan instead of the one-launch resident fit
    //   PLS_HIP_REPLICA_GUARD=0  no replica-divergence check after a sharded fit (must be the same on every rank)
    struct Env {
        bool tiny = true, cv_refit = false, tail = true, replica_guard = true, resident = true;
        int tail_update = 1;  // 0: never, 1: in the tail of READ-ONLY passes (default), 2: of every pass
        int xb4 = 1;          // PLS_HIP_XB4=0: X B with 5..32 columns on the older kernels
        int resident_gram = 1;  // PLS_HIP_RESIDENT_GRAM=0: AUTO's mid-size fits on the per-component resident kernels; 2: also under KERNEL (measurements); 4: the row form of its first phase everywhere
    } env;
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

}  // namespace
