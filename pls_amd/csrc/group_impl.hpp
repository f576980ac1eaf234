// group_impl.hpp -- pls_hip_group: one process, one handle + one host thread per GPU (include/pls_hip.h, "one
// process, several GPUs").  Included at the end of pls_hip.hip: it drives the members through the public entry
// points on their own handles and adds (a) the row partition and the resident matrices, (b) the in-process
// all-reduce: the device-side exchange of exchange_kernels.hpp (every member writes its partial sums into its peers'
// inboxes; default when every member has its own GPU) or, for members that share a GPU and for messages beyond 512 KB,
// the host-synchronised form in which every member reads the other members' buffers over peer access.
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace {

constexpr int GROUP_MAX = plsk::XCHG_MAX;
using plsk::XCHG_CAP;

struct PeerPtrs {
    const double *p[GROUP_MAX];
};

// out[i] = sum over the members' buffers in rank order: every member computes the same bits
__global__ __launch_bounds__(256) void peer_sum_kernel(PeerPtrs src, int n, i64 count, double *__restrict__ out) {
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
        double s = src.p[0][i];
        for (int j = 1; j < n; ++j) s += src.p[j][i];
        out[i] = s;
    }
}

// host barrier of the member threads; abort() releases every waiter with `false` (a member failed)
class GroupBarrier {
public:
    void reset(int n) {
        std::lock_guard<std::mutex> lk(mu_);
        n_ = n;
        count_ = 0;
        broken_ = false;
    }
    bool arrive_and_wait() {
        std::unique_lock<std::mutex> lk(mu_);
        if (broken_) return false;
        const uint64_t gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
            return true;
        }
        cv_.wait(lk, [&] { return gen_ != gen || broken_; });
        return gen_ != gen;  // released by the last arriver (not by an abort)
    }
    void abort() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            broken_ = true;
        }
        cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int n_ = 1, count_ = 0;
    uint64_t gen_ = 0;
    bool broken_ = false;
};

}  // namespace

struct pls_hip_matrix_s {
    int dtype = PLS_HIP_F64;
    i64 N = 0, K = 0;
    std::vector<void *> data;  // per member: device block (nrows x K, ld)
    std::vector<i64> ld, row0, nrows;
    // pls_hip_group_upload_xy: per member X^T X (K x K) and X^T Y (K x M) of its rows, formed during the upload
    std::vector<double *> gram_xx, gram_xy;
    uint64_t id = 0, gram_with = 0;  // serial number of this matrix / of the Y the products were formed with
    bool gram_ok = false;
};

struct pls_hip_group_s;
namespace {
struct Member {
    pls_hip_group_s *g = nullptr;
    int rank = 0;
};
}  // namespace

struct pls_hip_group_s {
    int n = 0;
    std::vector<int> dev;
    std::vector<pls_hip_context *> h;
    std::vector<hipStream_t> stream;
    std::vector<Member> member;
    // in-process all-reduce
    GroupBarrier bar;
    std::vector<void *> bufptr;
    std::vector<i64> bufcount;
    std::vector<double *> scratch;
    std::vector<i64> scratch_count;
    std::vector<hipEvent_t> ready[2], done[2];
    std::vector<int> phase;
    // device-side exchange (see above): per member inbox, flags, status word, collectives issued so far
    bool xchg = false;
    std::vector<double *> inbox;
    std::vector<unsigned long long *> xflags;
    std::vector<int *> xstatus;
    int *xhost = nullptr, *xhost_dev = nullptr;  // one host-mapped word per member: a wait timed out
    std::vector<unsigned long long> xseq;
    std::vector<long long> xlimit;  // time limit of a wait in wall-clock ticks of the member's device
    std::string err;
    // freed blocks of resident matrices, kept for the next allocation of the same size (a Model that is rebuilt on
    // data of the same shape, cross-validation refits): hipMalloc / hipFree of multi-GB blocks cost 50-250 ms when
    // the runtime returns the memory to the driver in between
    struct FreeBlock { void *p; size_t bytes; };
    std::vector<std::vector<FreeBlock>> freelist;  // per member, oldest first
};

namespace {

// rows of member r: whole quads of 4 rows (16-byte row packs in either storage type), the same split for every
// matrix of N rows
void row_block(i64 N, int n, int r, i64 *row0, i64 *nrows) {
    const i64 quads = (N + 3) / 4;
    const i64 a = std::min<i64>(N, 4 * (quads * r / n)), b = std::min<i64>(N, 4 * (quads * (r + 1) / n));
    *row0 = a;
    *nrows = b - a;
}

int gfail(pls_hip_group_s *g, int code, const std::string &msg) {
    g->err = msg;
    return code;
}

// the two launches of one device-side collective of member r (its device current); buf: `slices` x L doubles
int xchg_launch(pls_hip_group_s *g, int r, double *buf, int L, int slices, hipStream_t stream, long long limit) {
    return plsk::xchg_launch_piece(stream, g->n, r, g->inbox.data(), g->xflags.data(), buf, L, 0, L, slices, ++g->xseq[r],
                                   g->xstatus[r], g->xhost_dev + r, limit);
}

// drain every member's stream and start the sequence numbers over (after a time-out or a failed member)
void xchg_reset(pls_hip_group_s *g) {
    for (int r = 0; r < g->n; ++r)
        if (hipSetDevice(g->dev[r]) == hipSuccess) (void)hipStreamSynchronize(g->stream[r]);
    for (int r = 0; r < g->n; ++r) {
        if (hipSetDevice(g->dev[r]) != hipSuccess) continue;
        (void)hipMemset(g->xflags[r], 0, (size_t)2 * g->n * 8 + 64);
        (void)hipDeviceSynchronize();
        g->xhost[r] = 0;
        g->xseq[r] = 0;
    }
    (void)hipSetDevice(g->dev[0]);
}

// One round of the exchange on known values when the group is created: member r contributes r + 1, every member must end
// up with n (n + 1) / 2 within two seconds.  Anything else (no peer writes into fine-grained memory on this system, members
// that share a GPU without a hardware queue each, ...) switches the group to the host-synchronised exchange for good.
bool xchg_selftest(pls_hip_group_s *g) {
    const int n = g->n;
    std::vector<double *> buf(n, nullptr);
    bool ok = true;
    for (int r = 0; r < n && ok; ++r) {
        double host[plsk::RED_SLICES] = {0};
        host[0] = r + 1.0;
        ok = hipSetDevice(g->dev[r]) == hipSuccess && hipMalloc((void **)&buf[r], sizeof(host)) == hipSuccess &&
             hipMemcpy(buf[r], host, sizeof(host), hipMemcpyHostToDevice) == hipSuccess;
    }
    for (int r = 0; r < n && ok; ++r) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, g->dev[r]) != hipSuccess || khz <= 0) khz = 100000;
        ok = hipSetDevice(g->dev[r]) == hipSuccess &&
             xchg_launch(g, r, buf[r], 1, plsk::RED_SLICES, g->stream[r], (long long)(2.0 * 1e3 * khz)) == 0;
    }
    for (int r = 0; r < n; ++r) {
        if (hipSetDevice(g->dev[r]) != hipSuccess) { ok = false; continue; }
        double got = 0.0;
        if (hipStreamSynchronize(g->stream[r]) != hipSuccess ||
            (buf[r] && hipMemcpy(&got, buf[r], 8, hipMemcpyDeviceToHost) != hipSuccess) || got != 0.5 * n * (n + 1) || g->xhost[r] != 0)
            ok = false;
        if (buf[r]) (void)hipFree(buf[r]);
    }
    (void)hipGetLastError();
    if (!ok) xchg_reset(g);
    return ok;
}

// pls_hip_allreduce_fn of a group member (called on the member's thread, device current)
int group_allreduce(void *user, void *buf, int64_t count, void *stream_) {
    Member *m = static_cast<Member *>(user);
    pls_hip_group_s *g = m->g;
    const int r = m->rank, n = g->n;
    if (n == 1) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (g->xchg && count % plsk::RED_SLICES == 0 && count / plsk::RED_SLICES <= XCHG_CAP)
        return xchg_launch(g, r, (double *)buf, (int)(count / plsk::RED_SLICES), plsk::RED_SLICES, stream, g->xlimit[r]);
    const int ph = (g->phase[r] ^= 1);
    if (g->scratch_count[r] < count) {
        if (g->scratch[r]) {
            g->h[r]->graveyard.push_back(g->scratch[r]);  // (released when no member is running)
            g->scratch[r] = nullptr;
        }
        if (hipMalloc((void **)&g->scratch[r], (size_t)count * 8) != hipSuccess) return 3;
        g->scratch_count[r] = count;
    }
    g->bufptr[r] = buf;
    g->bufcount[r] = count;
    if (hipEventRecord(g->ready[ph][r], stream) != hipSuccess) return 4;
    if (!g->bar.arrive_and_wait()) return 5;  // every member has published its buffer and recorded `ready`
    PeerPtrs src;
    for (int j = 0; j < n; ++j) {
        if (g->bufcount[j] != count) {  // the members disagree on the collective: a bug, never a data condition
            g->bar.abort();
            return 6;
        }
        src.p[j] = static_cast<const double *>(g->bufptr[j]);
        if (j != r && hipStreamWaitEvent(stream, g->ready[ph][j], 0) != hipSuccess) return 7;
    }
    const unsigned grid = (unsigned)std::min<i64>(1024, (count + 255) / 256);
    hipLaunchKernelGGL(peer_sum_kernel, dim3(grid), dim3(256), 0, stream, src, n, (i64)count, g->scratch[r]);
    if (hipGetLastError() != hipSuccess) return 8;
    if (hipEventRecord(g->done[ph][r], stream) != hipSuccess) return 9;
    if (!g->bar.arrive_and_wait()) return 10;  // every member has enqueued its reads of all the buffers
    for (int j = 0; j < n; ++j)
        if (j != r && hipStreamWaitEvent(stream, g->done[ph][j], 0) != hipSuccess) return 11;
    if (hipMemcpyAsync(buf, g->scratch[r], (size_t)count * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 12;
    return 0;
}

// run fn(rank) on one host thread per member; a failing member releases the others from the reducer's barrier
template <typename F>
int run_members(pls_hip_group_s *g, F fn) {
    std::vector<int> rc(g->n, PLS_HIP_OK);
    g->bar.reset(g->n);
    for (int r = 0; r < g->n; ++r)  // workspace the members outgrew in earlier calls (no member is running now)
        if (!g->h[r]->graveyard.empty() && hipSetDevice(g->dev[r]) == hipSuccess) {
            for (void *q : g->h[r]->graveyard) (void)hipFree(q);
            g->h[r]->graveyard.clear();
        }
    auto body = [&](int r) {
        int code = PLS_HIP_ERR_DEVICE;
        if (hipSetDevice(g->dev[r]) == hipSuccess) code = fn(r);
        rc[r] = code;
        if (code != PLS_HIP_OK) g->bar.abort();
    };
    std::vector<std::thread> th;
    for (int r = 1; r < g->n; ++r) th.emplace_back(body, r);
    body(0);
    for (std::thread &t : th) t.join();
    if (g->xchg) {
        // device-side exchange: a wait that timed out (host-mapped status words), or a member that failed and left the
        // others out of step -> drain every stream, start the sequence numbers over
        bool timed_out = false, failed = false;
        for (int r = 0; r < g->n; ++r) {
            timed_out = timed_out || g->xhost[r] != 0;
            failed = failed || rc[r] != PLS_HIP_OK;
        }
        if (timed_out || failed) {
            for (int r = 0; r < g->n; ++r)
                if (hipSetDevice(g->dev[r]) == hipSuccess) (void)hipStreamSynchronize(g->stream[r]);
            std::string state;
            for (int r = 0; r < g->n; ++r) {
                timed_out = timed_out || g->xhost[r] != 0;
                if (hipSetDevice(g->dev[r]) != hipSuccess) continue;
                std::vector<unsigned long long> fl(2 * g->n, 0);  // what the member had received when it gave up
                (void)hipMemcpy(fl.data(), g->xflags[r], fl.size() * 8, hipMemcpyDeviceToHost);
                state += " member " + std::to_string(r) + " (collective " + std::to_string(g->xseq[r]) + ") has";
                for (unsigned long long f : fl) state += " " + std::to_string(f);
            }
            xchg_reset(g);
            if (timed_out && !failed) {
                g->err = "device-side exchange: a member waited longer than the time limit for its peers' partial sums;" + state;
                return PLS_HIP_ERR_REDUCER;
            }
        }
    }
    for (int r = 0; r < g->n; ++r)
        if (rc[r] != PLS_HIP_OK) {
            // prefer the message of a member that failed on its own over one that was released by the abort
            int first = r;
            for (int q = 0; q < g->n; ++q)
                if (rc[q] != PLS_HIP_OK && rc[q] != PLS_HIP_ERR_REDUCER) { first = q; break; }
            g->err = "member " + std::to_string(first) + ": " + g->h[first]->err;
            return rc[first];
        }
    return PLS_HIP_OK;
}

constexpr size_t FREELIST_MAX_BLOCKS = 6;
constexpr size_t FREELIST_MAX_BYTES = (size_t)96 << 30;  // per member; a third of the 288 GB of an MI355X

void *block_alloc(pls_hip_group_s *g, int r, size_t bytes) {
    std::vector<pls_hip_group_s::FreeBlock> &fl = g->freelist[r];
    for (size_t i = 0; i < fl.size(); ++i)
        if (fl[i].bytes == bytes) {
            void *p = fl[i].p;
            fl.erase(fl.begin() + (long)i);
            return p;
        }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        for (auto &b : fl) (void)hipFree(b.p);  // make room and try once more
        fl.clear();
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return p;
}

void block_free(pls_hip_group_s *g, int r, void *p, size_t bytes) {
    std::vector<pls_hip_group_s::FreeBlock> &fl = g->freelist[r];
    fl.push_back({p, bytes});
    size_t total = 0;
    for (auto &b : fl) total += b.bytes;
    while (!fl.empty() && (fl.size() > FREELIST_MAX_BLOCKS || total > FREELIST_MAX_BYTES)) {
        total -= fl.front().bytes;
        (void)hipFree(fl.front().p);
        fl.erase(fl.begin());
    }
}

bool same_partition(const pls_hip_group_s *g, const pls_hip_matrix_s *m) { return m && (int)m->data.size() == g->n; }

}  // namespace

extern "C" {

int pls_hip_group_create(pls_hip_group *out, int n, const int *devices) {
    if (!out) return PLS_HIP_ERR_INVALID;
    *out = nullptr;
    if (n < 1 || n > GROUP_MAX || !devices) return PLS_HIP_ERR_INVALID;
    std::unique_ptr<pls_hip_group_s> g(new (std::nothrow) pls_hip_group_s());
    if (!g) return PLS_HIP_ERR_ALLOC;
    g->n = n;
    g->dev.assign(devices, devices + n);
    g->h.assign(n, nullptr);
    g->stream.assign(n, nullptr);
    g->member.resize(n);
    g->bufptr.assign(n, nullptr);
    g->bufcount.assign(n, 0);
    g->scratch.assign(n, nullptr);
    g->scratch_count.assign(n, 0);
    g->phase.assign(n, 0);
    g->inbox.assign(n, nullptr);
    g->xflags.assign(n, nullptr);
    g->xstatus.assign(n, nullptr);
    g->xseq.assign(n, 0);
    g->xlimit.assign(n, 0);
    {
        // PLS_HIP_GROUP_EXCHANGE = device | host.  Default: device when every member has a GPU of its own; members that
        // SHARE a GPU (virtual shards: tests, rehearsals) wait for each other's kernels on one device, which needs as
        // many hardware queues as members (GPU_MAX_HW_QUEUES) -- there the host-synchronised exchange stays the default.
        bool distinct = true;
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b) distinct = distinct && devices[a] != devices[b];
        const char *e = getenv("PLS_HIP_GROUP_EXCHANGE");
        g->xchg = n > 1 && (e ? std::strcmp(e, "device") == 0 : distinct);
        if (g->xchg && (hipHostMalloc((void **)&g->xhost, GROUP_MAX * sizeof(int), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
                        hipHostGetDevicePointer((void **)&g->xhost_dev, g->xhost, 0) != hipSuccess)) {
            (void)hipGetLastError();
            g->xchg = false;
        }
        if (g->xchg) std::memset(g->xhost, 0, GROUP_MAX * sizeof(int));
    }
    g->freelist.resize(n);
    for (int p = 0; p < 2; ++p) {
        g->ready[p].assign(n, nullptr);
        g->done[p].assign(n, nullptr);
    }
    int rc = PLS_HIP_OK;
    for (int r = 0; r < n && rc == PLS_HIP_OK; ++r) {
        if (hipSetDevice(devices[r]) != hipSuccess) { (void)hipGetLastError(); rc = PLS_HIP_ERR_DEVICE; break; }
        for (int q = 0; q < n; ++q) {  // direct loads of the other members' partial buffers over xGMI
            if (devices[q] == devices[r]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[r], devices[q]) != hipSuccess || !can) { rc = PLS_HIP_ERR_DEVICE; break; }
            const hipError_t e = hipDeviceEnablePeerAccess(devices[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { rc = PLS_HIP_ERR_DEVICE; break; }
            (void)hipGetLastError();
        }
        if (rc != PLS_HIP_OK) break;
        if (hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess) { rc = PLS_HIP_ERR_DEVICE; break; }
        rc = pls_hip_create(&g->h[r], devices[r], g->stream[r]);
        if (rc != PLS_HIP_OK) break;
        g->h[r]->defer_free = true;
        for (int p = 0; p < 2; ++p)
            if (hipEventCreateWithFlags(&g->ready[p][r], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&g->done[p][r], hipEventDisableTiming) != hipSuccess)
                rc = PLS_HIP_ERR_DEVICE;
        if (rc == PLS_HIP_OK && n > 1 && g->xchg) {
            const size_t ib = (size_t)2 * n * XCHG_CAP * 8, fb = (size_t)2 * n * 8;
            if (hipExtMallocWithFlags((void **)&g->inbox[r], ib, hipDeviceMallocFinegrained) != hipSuccess ||
                hipExtMallocWithFlags((void **)&g->xflags[r], fb + 64, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                g->xchg = false;  // (no fine-grained memory: the host-synchronised exchange)
            } else {
                g->xstatus[r] = reinterpret_cast<int *>(g->xflags[r] + 2 * n);
                g->xlimit[r] = plsk::xchg_time_limit(devices[r]);
                if (hipMemset(g->xflags[r], 0, fb + 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = PLS_HIP_ERR_DEVICE;
            }
        }
        g->member[r].g = g.get();
        g->member[r].rank = r;
        if (rc == PLS_HIP_OK && n > 1) {
            rc = pls_hip_set_reducer(g->h[r], group_allreduce, &g->member[r], r, n);
            // the members share the host cores for their staging copies
            g->h[r]->copy_threads = std::max(2, plsh::default_copy_threads() * 2 / n);
        }
    }
    if (rc != PLS_HIP_OK) {
        pls_hip_group_destroy(g.release());
        return rc;
    }
    if (g->xchg && !xchg_selftest(g.get())) g->xchg = false;
    if (g->xchg)  // the members' view of the exchange: the push of a fused component rides in the pass, the gather in the update
        for (int r = 0; r < n; ++r) {
            pls_hip_context::XchgEndpoint &e = g->h[r]->xep;
            e.on = true;
            e.n = n; e.rank = r;
            e.inbox = g->inbox.data(); e.flags = g->xflags.data();
            e.seq = &g->xseq[r]; e.status = g->xstatus[r]; e.host_status = g->xhost_dev + r; e.limit = &g->xlimit[r];
        }
    *out = g.release();
    return PLS_HIP_OK;
}

int pls_hip_group_destroy(pls_hip_group g) {
    if (!g) return PLS_HIP_OK;
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (g->h[r]) (void)pls_hip_destroy(g->h[r]);  // synchronises the member's stream
        if (g->scratch[r]) (void)hipFree(g->scratch[r]);
        if (r < (int)g->inbox.size() && g->inbox[r]) (void)hipFree(g->inbox[r]);
        if (r < (int)g->xflags.size() && g->xflags[r]) (void)hipFree(g->xflags[r]);
        if (r < (int)g->freelist.size())
            for (auto &b : g->freelist[r]) (void)hipFree(b.p);
        for (int p = 0; p < 2; ++p) {
            if (g->ready[p][r]) (void)hipEventDestroy(g->ready[p][r]);
            if (g->done[p][r]) (void)hipEventDestroy(g->done[p][r]);
        }
        if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
    }
    if (g->xhost) (void)hipHostFree(g->xhost);
    delete g;
    return PLS_HIP_OK;
}

int pls_hip_group_size(pls_hip_group g) { return g ? g->n : 0; }

int pls_hip_group_exchange(pls_hip_group g) { return (g && g->xchg) ? 1 : 0; }

int pls_hip_group_handle(pls_hip_group g, int rank, pls_hip_handle *out) {
    if (!g || !out || rank < 0 || rank >= g->n) return PLS_HIP_ERR_INVALID;
    *out = g->h[rank];
    return PLS_HIP_OK;
}

int pls_hip_group_set_option(pls_hip_group g, int option, int64_t value) {
    if (!g) return PLS_HIP_ERR_INVALID;
    for (int r = 0; r < g->n; ++r) {
        const int rc = pls_hip_set_option(g->h[r], option, value);
        if (rc != PLS_HIP_OK) return gfail(g, rc, g->h[r]->err);
    }
    return PLS_HIP_OK;
}

const char *pls_hip_group_last_error(pls_hip_group g) { return g ? g->err.c_str() : "null group"; }

int pls_hip_group_alloc(pls_hip_group g, int64_t N, int64_t K, int dtype, pls_hip_matrix *out) {
    if (!g || !out) return PLS_HIP_ERR_INVALID;
    *out = nullptr;
    if (N < 1 || K < 1 || (dtype != PLS_HIP_F64 && dtype != PLS_HIP_F32)) return gfail(g, PLS_HIP_ERR_INVALID, "bad matrix shape");
    std::unique_ptr<pls_hip_matrix_s> m(new (std::nothrow) pls_hip_matrix_s());
    if (!m) return PLS_HIP_ERR_ALLOC;
    static std::atomic<uint64_t> serial{0};
    m->id = ++serial;
    m->dtype = dtype;
    m->N = N;
    m->K = K;
    m->data.assign(g->n, nullptr);
    m->ld.assign(g->n, 1);
    m->row0.assign(g->n, 0);
    m->nrows.assign(g->n, 0);
    const size_t es = esize(dtype);
    for (int r = 0; r < g->n; ++r) {
        row_block(N, g->n, r, &m->row0[r], &m->nrows[r]);
        m->ld[r] = std::max<i64>(4, (m->nrows[r] + 3) & ~(i64)3);
        if (hipSetDevice(g->dev[r]) != hipSuccess ||
            !(m->data[r] = block_alloc(g, r, (size_t)m->ld[r] * (size_t)K * es))) {
            (void)hipGetLastError();
            pls_hip_group_free(g, m.release());
            return gfail(g, PLS_HIP_ERR_ALLOC, "hipMalloc of a resident matrix block failed");
        }
    }
    *out = m.release();
    return PLS_HIP_OK;
}

int pls_hip_group_free(pls_hip_group g, pls_hip_matrix m) {
    if (!m) return PLS_HIP_OK;
    if (!g) return PLS_HIP_ERR_INVALID;
    for (int r = 0; r < (int)m->data.size() && r < g->n; ++r) {
        (void)hipSetDevice(g->dev[r]);
        (void)hipStreamSynchronize(g->stream[r]);  // launches of this member may still use the block
        if (m->data[r]) block_free(g, r, m->data[r], (size_t)m->ld[r] * (size_t)m->K * esize(m->dtype));
        if (r < (int)m->gram_xx.size() && m->gram_xx[r]) (void)hipFree(m->gram_xx[r]);
        if (r < (int)m->gram_xy.size() && m->gram_xy[r]) (void)hipFree(m->gram_xy[r]);
    }
    delete m;
    return PLS_HIP_OK;
}

int pls_hip_matrix_shape(pls_hip_matrix m, int64_t *N, int64_t *K, int *dtype) {
    if (!m) return PLS_HIP_ERR_INVALID;
    if (N) *N = m->N;
    if (K) *K = m->K;
    if (dtype) *dtype = m->dtype;
    return PLS_HIP_OK;
}

int pls_hip_matrix_block(pls_hip_matrix m, int rank, void **data, int64_t *ld, int64_t *row0, int64_t *nrows) {
    if (!m || rank < 0 || rank >= (int)m->data.size()) return PLS_HIP_ERR_INVALID;
    if (data) *data = m->data[rank];
    if (ld) *ld = m->ld[rank];
    if (row0) *row0 = m->row0[rank];
    if (nrows) *nrows = m->nrows[rank];
    return PLS_HIP_OK;
}

int pls_hip_group_upload(pls_hip_group g, const void *host, int64_t ld, int64_t N, int64_t K, int dtype,
                         pls_hip_matrix *out) {
    if (!g || !out) return PLS_HIP_ERR_INVALID;
    *out = nullptr;
    if (!host || N < 1 || ld < N) return gfail(g, PLS_HIP_ERR_INVALID, "bad upload arguments");
    pls_hip_matrix m = nullptr;
    CHK(pls_hip_group_alloc(g, N, K, dtype, &m));
    const size_t es = esize(dtype);
    const int rc = run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        CHK(h2d(c, m->data[r], m->ld[r], (const char *)host + (size_t)m->row0[r] * es, ld, m->nrows[r], K, es));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PLS_HIP_OK;
    });
    if (rc != PLS_HIP_OK) {
        pls_hip_group_free(g, m);
        return rc;
    }
    *out = m;
    return PLS_HIP_OK;
}

int pls_hip_group_upload_xy(pls_hip_group g, const void *hostX, int64_t ldx, const void *hostY, int64_t ldy, int64_t N,
                            int64_t K, int64_t M, int dtype, pls_hip_matrix *Xo, pls_hip_matrix *Yo) {
    if (!g || !Xo || !Yo) return PLS_HIP_ERR_INVALID;
    *Xo = *Yo = nullptr;
    if (!hostX || !hostY || N < 1 || ldx < N || ldy < N || K < 1 || M < 1)
        return gfail(g, PLS_HIP_ERR_INVALID, "bad upload_xy arguments");
    pls_hip_matrix X = nullptr, Y = nullptr;
    CHK(pls_hip_group_alloc(g, N, K, dtype, &X));
    int rc = pls_hip_group_alloc(g, N, M, dtype, &Y);
    if (rc != PLS_HIP_OK) {
        pls_hip_group_free(g, X);
        return rc;
    }
    X->gram_xx.assign(g->n, nullptr);
    X->gram_xy.assign(g->n, nullptr);
    const size_t es = esize(dtype);
    std::vector<char> okv(g->n, 0);
    // data small enough for the single-launch fit (tiny_kernels.hpp; any A that fits its LDS): X^T X would never be read
    const bool single_launch_data = g->n == 1 && plsk::tiny_fit_covers(N, (int)K, (int)M, 1, X->ld[0], es);
    rc = run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        if (single_launch_data || hipMalloc((void **)&X->gram_xx[r], (size_t)K * K * 8) != hipSuccess ||
            hipMalloc((void **)&X->gram_xy[r], (size_t)K * M * 8) != hipSuccess) {
            (void)hipGetLastError();  // no room for the products: a plain upload
            CHK(h2d(c, Y->data[r], Y->ld[r], (const char *)hostY + (size_t)Y->row0[r] * es, ldy, Y->nrows[r], M, es));
            CHK(h2d(c, X->data[r], X->ld[r], (const char *)hostX + (size_t)X->row0[r] * es, ldx, X->nrows[r], K, es));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return PLS_HIP_OK;
        }
        CHK(h2d(c, Y->data[r], Y->ld[r], (const char *)hostY + (size_t)Y->row0[r] * es, ldy, Y->nrows[r], M, es));
        bool ok = false;
        if (X->nrows[r] == 0) {  // an empty member contributes zeros
            HIPCHK(c, hipMemsetAsync(X->gram_xx[r], 0, (size_t)K * K * 8, c->stream));
            HIPCHK(c, hipMemsetAsync(X->gram_xy[r], 0, (size_t)K * M * 8, c->stream));
            ok = true;
        } else if (dtype == PLS_HIP_F64) {
            CHK(upload_accumulate<double>(c, (double *)X->data[r], X->ld[r], (const double *)hostX + X->row0[r], ldx,
                                          X->nrows[r], (int)K, (const double *)Y->data[r], Y->ld[r], (int)M,
                                          X->gram_xx[r], X->gram_xy[r], &ok));
        } else {
            CHK(upload_accumulate<float>(c, (float *)X->data[r], X->ld[r], (const float *)hostX + X->row0[r], ldx,
                                         X->nrows[r], (int)K, (const float *)Y->data[r], Y->ld[r], (int)M, X->gram_xx[r],
                                         X->gram_xy[r], &ok));
        }
        okv[r] = ok ? 1 : 0;
        HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller's memory has been read, the products are complete
        return PLS_HIP_OK;
    });
    if (rc != PLS_HIP_OK) {
        pls_hip_group_free(g, X);
        pls_hip_group_free(g, Y);
        return rc;
    }
    bool all = true;  // the members must agree on the plan of a later fit: all of them hold the products, or none is used
    for (int r = 0; r < g->n; ++r) all = all && okv[r];
    X->gram_ok = all;
    X->gram_with = Y->id;
    *Xo = X;
    *Yo = Y;
    return PLS_HIP_OK;
}

int pls_hip_group_download(pls_hip_group g, pls_hip_matrix m, int64_t col0, int64_t ncols, void *host, int64_t ld) {
    if (!g || !same_partition(g, m) || !host || col0 < 0 || ncols < 0 || col0 + ncols > m->K || ld < m->N)
        return g ? gfail(g, PLS_HIP_ERR_INVALID, "bad download arguments") : PLS_HIP_ERR_INVALID;
    if (ncols == 0) return PLS_HIP_OK;
    const size_t es = esize(m->dtype);
    return run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        CHK(d2h(c, (char *)host + (size_t)m->row0[r] * es, ld, (const char *)m->data[r] + (size_t)col0 * m->ld[r] * es,
                m->ld[r], m->nrows[r], ncols, es));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PLS_HIP_OK;
    });
}

int pls_hip_group_fit(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A, int method, double *W,
                      double *P, double *Q, double *R, pls_hip_matrix T, double *B) {
    if (!g) return PLS_HIP_ERR_INVALID;
    if (!same_partition(g, X) || !same_partition(g, Y) || X->N != Y->N || X->dtype != Y->dtype || !W || !P || !Q || !R ||
        A < 1 || A > X->K)
        return gfail(g, PLS_HIP_ERR_INVALID, "bad group_fit arguments");
    if (method == PLS_HIP_KERNEL_TYPE1 && (!same_partition(g, T) || T->N != X->N || T->K < A || T->dtype != X->dtype))
        return gfail(g, PLS_HIP_ERR_INVALID, "group_fit: T must be a resident N x A matrix of the storage type of X");
    const i64 K = X->K, M = Y->K;
    // every member's small outputs come back to the host: member 0's go to the caller, the others are compared
    std::vector<std::vector<double>> scratch(g->n > 1 ? g->n - 1 : 0);
    const size_t nW = (size_t)K * A, nQ = (size_t)M * A, nB = B ? (size_t)K * M : 0;
    for (auto &v : scratch) v.resize(3 * nW + nQ + nB);
    const int rc = run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        CHK(ensure(c, c->hW, nW * 8));
        CHK(ensure(c, c->hP, nW * 8));
        CHK(ensure(c, c->hR, nW * 8));
        CHK(ensure(c, c->hQ, nQ * 8));
        if (B) CHK(ensure(c, c->hB, nB * 8));
        const bool t1 = (method == PLS_HIP_KERNEL_TYPE1);
        if (X->gram_ok && X->gram_with == Y->id) {  // X^T X and X^T Y of this member's rows came with the upload
            c->pre_xx = X->gram_xx[r];
            c->pre_xy = X->gram_xy[r];
        }
        const int frc = pls_hip_fit(c, X->data[r], X->ld[r], Y->data[r], Y->ld[r], X->nrows[r], K, M, A, method, X->dtype,
                                    PLS_HIP_MEM_DEVICE, (double *)c->hW.p, (double *)c->hP.p, (double *)c->hQ.p,
                                    (double *)c->hR.p, t1 ? T->data[r] : nullptr, t1 ? T->ld[r] : 1,
                                    B ? (double *)c->hB.p : nullptr);
        c->pre_xx = c->pre_xy = nullptr;
        CHK(frc);
        double *w = r == 0 ? W : scratch[r - 1].data();
        double *p = r == 0 ? P : w + nW, *rr = r == 0 ? R : w + 2 * nW, *q = r == 0 ? Q : w + 3 * nW;
        double *b = r == 0 ? B : w + 3 * nW + nQ;
        CHK(d2h(c, w, K, c->hW.p, K, K, A, 8));
        CHK(d2h(c, p, K, c->hP.p, K, K, A, 8));
        CHK(d2h(c, rr, K, c->hR.p, K, K, A, 8));
        CHK(d2h(c, q, M, c->hQ.p, M, M, A, 8));
        if (B) CHK(d2h(c, b, K, c->hB.p, K, K, M, 8));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PLS_HIP_OK;
    });
    if (rc != PLS_HIP_OK) return rc;
    for (int r = 1; r < g->n; ++r) {  // lock-step check: replicated arithmetic on identical reduced partials
        const double *w = scratch[r - 1].data();
        if (std::memcmp(w, W, nW * 8) || std::memcmp(w + nW, P, nW * 8) || std::memcmp(w + 2 * nW, R, nW * 8) ||
            std::memcmp(w + 3 * nW, Q, nQ * 8) || (B && std::memcmp(w + 3 * nW + nQ, B, nB * 8)))
            return gfail(g, PLS_HIP_ERR_REDUCER, "members " + std::to_string(r) + " and 0 derived different W/P/Q/R/B");
    }
    return PLS_HIP_OK;
}

int pls_hip_group_xb(pls_hip_group g, pls_hip_matrix X, const double *Bm, int64_t ldb, int64_t C, pls_hip_matrix out) {
    if (!g) return PLS_HIP_ERR_INVALID;
    if (!same_partition(g, X) || !same_partition(g, out) || !Bm || C < 1 || ldb < X->K || out->N != X->N || out->K < C ||
        out->dtype != X->dtype)
        return gfail(g, PLS_HIP_ERR_INVALID, "bad group_xb arguments");
    const i64 K = X->K;
    return run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        if (X->nrows[r] == 0) return PLS_HIP_OK;
        CHK(ensure(c, c->hB, (size_t)K * C * 8));
        CHK(h2d(c, c->hB.p, K, Bm, ldb, K, C, 8));
        CHK(pls_hip_xb(c, X->data[r], X->ld[r], X->nrows[r], K, (const double *)c->hB.p, K, C, X->dtype, PLS_HIP_MEM_DEVICE,
                       out->data[r], out->ld[r]));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PLS_HIP_OK;
    });
}

int pls_hip_group_model_sse(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A, const double *R,
                            const double *Q, double *SSE) {
    if (!g) return PLS_HIP_ERR_INVALID;
    if (!same_partition(g, X) || !same_partition(g, Y) || X->N != Y->N || X->dtype != Y->dtype || !R || !Q || !SSE || A < 1)
        return gfail(g, PLS_HIP_ERR_INVALID, "bad group_model_sse arguments");
    const i64 K = X->K, M = Y->K;
    return run_members(g, [&](int r) -> int {
        pls_hip_context *c = g->h[r];
        CHK(ensure(c, c->hR, (size_t)K * A * 8));
        CHK(ensure(c, c->hQ, (size_t)M * A * 8));
        CHK(ensure(c, c->hB, (size_t)M * A * 8));
        CHK(h2d(c, c->hR.p, K, R, K, K, A, 8));
        CHK(h2d(c, c->hQ.p, M, Q, M, M, A, 8));
        CHK(pls_hip_model_sse(c, X->data[r], X->ld[r], Y->data[r], Y->ld[r], X->nrows[r], K, M, A, (const double *)c->hR.p,
                              (const double *)c->hQ.p, X->dtype, PLS_HIP_MEM_DEVICE, (double *)c->hB.p));
        if (r == 0) CHK(d2h(c, SSE, M, c->hB.p, M, M, A, 8));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return PLS_HIP_OK;
    });
}

int pls_hip_group_cv_folds(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A, const int64_t *test_idx,
                           int64_t test_size, int64_t num_folds, double *E) {
    if (!g) return PLS_HIP_ERR_INVALID;
    if (g->n != 1) return gfail(g, PLS_HIP_ERR_UNSUPPORTED, "group_cv_folds: single-member groups only");
    if (!same_partition(g, X) || !same_partition(g, Y) || X->N != Y->N || X->dtype != Y->dtype || !E)
        return gfail(g, PLS_HIP_ERR_INVALID, "bad group_cv_folds arguments");
    pls_hip_context *c = g->h[0];
    if (hipSetDevice(g->dev[0]) != hipSuccess) return gfail(g, PLS_HIP_ERR_DEVICE, "hipSetDevice failed");
    const i64 nobs = num_folds * test_size;
    if (nobs < 1 || A < 1) return gfail(g, PLS_HIP_ERR_INVALID, "bad group_cv_folds arguments");
    // the residuals land in a device buffer of the member (pls_hip_cv_folds returns synchronised) and are read back
    DevBuf tmp;
    int rc = ensure(c, tmp, (size_t)nobs * A * Y->K * 8);
    if (rc == PLS_HIP_OK) {
        if (X->gram_ok && X->gram_with == Y->id) {  // the folds downdate X^T X and X^T Y: both came with the upload
            c->pre_xx = X->gram_xx[0];
            c->pre_xy = X->gram_xy[0];
        }
        rc = pls_hip_cv_folds(c, X->data[0], X->ld[0], Y->data[0], Y->ld[0], X->N, X->K, Y->K, A, test_idx, test_size,
                              num_folds, X->dtype, PLS_HIP_MEM_DEVICE, (double *)tmp.p);
        c->pre_xx = c->pre_xy = nullptr;
    }
    if (rc == PLS_HIP_OK && hipMemcpy(E, tmp.p, (size_t)nobs * A * Y->K * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        c->err = "hipMemcpy of the fold residuals failed";
        rc = PLS_HIP_ERR_DEVICE;
    }
    if (tmp.p) (void)hipFree(tmp.p);
    if (rc != PLS_HIP_OK) return gfail(g, rc, c->err);
    return PLS_HIP_OK;
}

}  // extern "C"
