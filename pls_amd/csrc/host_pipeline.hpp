// Host side of the transfers between caller-owned (pageable) host matrices and device memory.
//
// The reference's entry point takes host matrices (Model(const Mat2D&, ...), src/pls.cpp:340-353), so a drop-in
// pays one transfer of X over PCIe per fit: 4.3 GB at BASELINE config 3, several times the 15 ms the fit itself
// takes.  A plain hipMemcpy from pageable memory stages the data through the runtime's own bounce buffers with one
// host thread (measured round 1: 47 GB/s).  Here the staging is explicit: two pinned buffers per handle, a small
// pool of host threads that repacks tile i+1 (strided column pieces -> one contiguous pinned tile) while the DMA
// engine transfers tile i, so the transfer runs at min(host copy rate, PCIe rate) and the caller's memory is read
// exactly once.  Device -> host runs the same pipeline in reverse.
#pragma once
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace plsh {

// A fixed set of worker threads that run the jobs 0..njobs-1 of ONE parallel_for at a time (the owner -- a handle --
// is single-threaded by contract).  The calling thread works too.
class CopyPool {
public:
    // cpus: CPUs the workers may run on (empty = wherever the scheduler puts them)
    explicit CopyPool(int nthreads, const std::vector<int> &cpus = std::vector<int>())
        : caller_works_(cpus.empty() || nthreads < 3) {
        // with workers bound to the device's NUMA node the calling thread (which may sit on another socket, where every
        // piece it copied would cross the inter-socket link twice) only hands out the work and waits
        for (int i = caller_works_ ? 1 : 0; i < nthreads; ++i)
            workers_.emplace_back([this, cpus] {
                if (!cpus.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus)
                        if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // best effort
                }
                loop();
            });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    CopyPool(const CopyPool &) = delete;
    CopyPool &operator=(const CopyPool &) = delete;

    void parallel_for(int njobs, const std::function<void(int)> &fn) {
        if (njobs <= 0) return;
        if (workers_.empty() || njobs == 1) {
            for (int j = 0; j < njobs; ++j) fn(j);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            njobs_ = njobs;
            next_.store(0);
            finished_ = 0;
            ++epoch_;
        }
        cv_.notify_all();
        if (caller_works_) run();
        // every worker acknowledges the epoch: none of them can still be inside run() when the next one is set up,
        // and a job that was taken has been completed by its taker before that thread left run()
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return finished_ == (int)workers_.size(); });
        fn_ = nullptr;
    }

private:
    void run() {
        for (;;) {
            const int j = next_.fetch_add(1);
            if (j >= njobs_) break;
            (*fn_)(j);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            run();
            {
                std::lock_guard<std::mutex> lk(mu_);
                ++finished_;
            }
            done_cv_.notify_one();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    int njobs_ = 0, finished_ = 0;
    std::atomic<int> next_{0};
    uint64_t epoch_ = 0;
    bool stop_ = false;
    const bool caller_works_;
};

inline int default_copy_threads() {  // PLS_HIP_COPY_THREADS (read once per process), else min(16, cores / 2)
    static const int cached = [] {
        if (const char *e = std::getenv("PLS_HIP_COPY_THREADS")) {
            const int v = std::atoi(e);
            if (v >= 1 && v <= 256) return v;
        }
        const unsigned hw = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(16u, (hw ? hw : 1u) / 2));
    }();
    return cached;
}

// CPUs of the NUMA node closest to `device` (hipDeviceAttributeHostNumaId + the node's sysfs cpulist, "0-63,128-191");
// empty when the platform does not say.  The pinned staging buffers live on that node; copy threads running there
// repack at the local memory rate whatever core the caller's thread happens to be on (measured on a two-socket host:
// 47-52 GB/s staged from the device's socket, 28-34 GB/s from the other one).
inline std::vector<int> device_numa_cpus(int device) {
    std::vector<int> cpus;
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device) != hipSuccess || node < 0) {
        (void)hipGetLastError();
        return cpus;
    }
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string spec;
    if (!f.is_open() || !std::getline(f, spec)) return cpus;
    size_t pos = 0;
    while (pos < spec.size()) {
        size_t end = spec.find(',', pos);
        if (end == std::string::npos) end = spec.size();
        const std::string tok = spec.substr(pos, end - pos);
        const size_t dash = tok.find('-');
        const int a = std::atoi(tok.c_str()), b = dash == std::string::npos ? a : std::atoi(tok.c_str() + dash + 1);
        for (int c = a; c <= b && c - a < 4096; ++c) cpus.push_back(c);
        pos = end + 1;
    }
    return cpus;
}

inline size_t env_size(const char *name, size_t dflt, size_t lo, size_t hi) {
    if (const char *e = std::getenv(name)) {
        const long v = std::atol(e);
        if (v >= (long)lo && v <= (long)hi) return (size_t)v;
    }
    return dflt;
}
// per pinned buffer (two per handle), MiB: PLS_HIP_STAGE_MB (read once per process); one copy job: 512 KiB
static const size_t STAGE_BYTES = env_size("PLS_HIP_STAGE_MB", 32, 1, 1024) << 20;
static const size_t PIECE_BYTES = (size_t)512 << 10;

struct Stager {
    void *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    CopyPool *pool = nullptr;
    int slot = 0;

    hipError_t ensure(int threads, int device) {
        if (buf[0]) return hipSuccess;
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&buf[i], STAGE_BYTES, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        pool = new CopyPool(threads, device_numa_cpus(device));
        return hipSuccess;
    }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (busy[i]) (void)hipEventSynchronize(ev[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
            buf[i] = nullptr;
            ev[i] = nullptr;
            busy[i] = false;
        }
        delete pool;
        pool = nullptr;
    }
};

// tile geometry shared by both directions: tiles of rb rows x cb columns, each at most STAGE_BYTES
struct Tiling {
    int64_t rb, cb;
    Tiling(int64_t rows, int64_t cols, size_t es) {
        rb = std::min<int64_t>(rows, (int64_t)(STAGE_BYTES / es));
        cb = std::max<int64_t>(1, std::min<int64_t>(cols, (int64_t)(STAGE_BYTES / ((size_t)rb * es))));
    }
};

// pinned tile (rbn x cbn, ld = rbn) <-> the caller's column-major matrix, in pieces of at most PIECE_BYTES
inline void repack(CopyPool &pool, char *tile, char *host, int64_t ldh, int64_t r0, int64_t k0, int64_t rbn,
                   int64_t cbn, size_t es, bool to_tile) {
    const size_t colbytes = (size_t)rbn * es;
    const int pieces = (int)((colbytes + PIECE_BYTES - 1) / PIECE_BYTES);
    // short columns: several whole columns per job
    const int cols_per_job = pieces == 1 ? (int)std::max<size_t>(1, PIECE_BYTES / std::max<size_t>(colbytes, 1)) : 1;
    const int njobs = pieces == 1 ? (int)((cbn + cols_per_job - 1) / cols_per_job) : (int)(cbn * pieces);
    pool.parallel_for(njobs, [&](int j) {
        if (pieces == 1) {
            const int64_t c0 = (int64_t)j * cols_per_job, c1 = std::min<int64_t>(cbn, c0 + cols_per_job);
            for (int64_t c = c0; c < c1; ++c) {
                char *t = tile + (size_t)c * colbytes;
                char *h = host + ((size_t)(k0 + c) * (size_t)ldh + (size_t)r0) * es;
                if (to_tile) std::memcpy(t, h, colbytes);
                else std::memcpy(h, t, colbytes);
            }
        } else {
            const int64_t c = j / pieces;
            const size_t off = (size_t)(j % pieces) * PIECE_BYTES, len = std::min(PIECE_BYTES, colbytes - off);
            char *t = tile + (size_t)c * colbytes + off;
            char *h = host + ((size_t)(k0 + c) * (size_t)ldh + (size_t)r0) * es + off;
            if (to_tile) std::memcpy(t, h, len);
            else std::memcpy(h, t, len);
        }
    });
}

// dst (device, ld = ldd elements) <- src (host, ld = lds): rows x cols elements of es bytes, on `stream`.
// On return the caller's memory has been read completely (the last tiles may still be in flight from the pinned
// buffers; they are ordered on the stream).
inline hipError_t upload(Stager &st, hipStream_t stream, void *dst, int64_t ldd, const void *src, int64_t lds,
                         int64_t rows, int64_t cols, size_t es) {
    const Tiling tl(rows, cols, es);
    for (int64_t r0 = 0; r0 < rows; r0 += tl.rb)
        for (int64_t k0 = 0; k0 < cols; k0 += tl.cb) {
            const int64_t rbn = std::min(tl.rb, rows - r0), cbn = std::min(tl.cb, cols - k0);
            const int s = st.slot;
            if (st.busy[s]) {
                hipError_t e = hipEventSynchronize(st.ev[s]);
                if (e != hipSuccess) return e;
            }
            repack(*st.pool, (char *)st.buf[s], (char *)const_cast<void *>(src), lds, r0, k0, rbn, cbn, es, true);
            char *d = (char *)dst + ((size_t)k0 * (size_t)ldd + (size_t)r0) * es;
            hipError_t e = (rbn == ldd)
                               ? hipMemcpyAsync(d, st.buf[s], (size_t)rbn * cbn * es, hipMemcpyHostToDevice, stream)
                               : hipMemcpy2DAsync(d, (size_t)ldd * es, st.buf[s], (size_t)rbn * es, (size_t)rbn * es,
                                                  (size_t)cbn, hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
            e = hipEventRecord(st.ev[s], stream);
            if (e != hipSuccess) return e;
            st.busy[s] = true;
            st.slot ^= 1;
        }
    return hipSuccess;
}

// dst (host) <- src (device); returns with the data in the caller's memory
inline hipError_t download(Stager &st, hipStream_t stream, void *dst, int64_t ldd, const void *src, int64_t lds,
                           int64_t rows, int64_t cols, size_t es) {
    const Tiling tl(rows, cols, es);
    struct Pending { int slot; int64_t r0, k0, rbn, cbn; bool on = false; } prev;
    auto drain = [&](Pending &p) -> hipError_t {
        if (!p.on) return hipSuccess;
        hipError_t e = hipEventSynchronize(st.ev[p.slot]);
        if (e != hipSuccess) return e;
        repack(*st.pool, (char *)st.buf[p.slot], (char *)dst, ldd, p.r0, p.k0, p.rbn, p.cbn, es, false);
        st.busy[p.slot] = false;
        p.on = false;
        return hipSuccess;
    };
    for (int64_t r0 = 0; r0 < rows; r0 += tl.rb)
        for (int64_t k0 = 0; k0 < cols; k0 += tl.cb) {
            const int64_t rbn = std::min(tl.rb, rows - r0), cbn = std::min(tl.cb, cols - k0);
            const int s = st.slot;
            if (st.busy[s]) {  // a tile of an earlier upload still in flight from this buffer
                hipError_t e = hipEventSynchronize(st.ev[s]);
                if (e != hipSuccess) return e;
                st.busy[s] = false;
            }
            const char *d = (const char *)src + ((size_t)k0 * (size_t)lds + (size_t)r0) * es;
            hipError_t e = (rbn == lds)
                               ? hipMemcpyAsync(st.buf[s], d, (size_t)rbn * cbn * es, hipMemcpyDeviceToHost, stream)
                               : hipMemcpy2DAsync(st.buf[s], (size_t)rbn * es, d, (size_t)lds * es, (size_t)rbn * es,
                                                  (size_t)cbn, hipMemcpyDeviceToHost, stream);
            if (e != hipSuccess) return e;
            e = hipEventRecord(st.ev[s], stream);
            if (e != hipSuccess) return e;
            st.busy[s] = true;
            e = drain(prev);  // unpack the previous tile while this one is in flight
            if (e != hipSuccess) return e;
            prev.slot = s; prev.r0 = r0; prev.k0 = k0; prev.rbn = rbn; prev.cbn = cbn; prev.on = true;
            st.slot ^= 1;
        }
    return drain(prev);
}

}  // namespace plsh
