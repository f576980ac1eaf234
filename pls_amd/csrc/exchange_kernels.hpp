// exchange_kernels.hpp -- the device-side exchange of the small partial sums of a row-sharded fit, shared by the in-process
// group (group_impl.hpp: members = handles of one process) and the cross-process form (pls_hip_xchg_*: one process per GPU,
// the peers' inboxes opened over IPC).  No RCCL, no host thread, event or copy per collective.
//
// Every member owns an INBOX on its device: [2 parities][n members][XCHG_CAP] doubles + [2][n] sequence flags, in
// fine-grained memory (remote writes become visible to a kernel that is already running).  Collective number q of a
// member (every member issues the same sequence of collectives):
//   push   (one workgroup per destination):  the member's 8 reduction slices summed to ONE vector, written into slot
//          [q & 1][rank] of every member's inbox over xGMI, then -- system-scope release -- flag[q & 1][rank] = q;
//   gather (same stream, right behind):  spins (system-scope acquire) until its own n flags show q, sums the n slots in
//          rank order into slice 0 of the member's buffer and clears slices 1..7 (the consumers add the 8 slices).
// Two parities suffice: a member can only reach collective q + 2 after it has gathered q + 1, i.e. after every peer has
// PUSHED q + 1, which each peer does behind its own gather of q.  A wait that lasts longer than the time limit (a member
// that failed or fell out of step) sets the member's status words and leaves the loop; later waits of the same member
// return at once.  Messages longer than XCHG_CAP go in pieces (or, in the group, through the host-synchronised path).
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "common.hpp"

namespace plsk {

constexpr int XCHG_MAX = 16;             // members
constexpr i64 XCHG_CAP = 1 << 16;        // doubles per message piece (512 KB)
constexpr int XCHG_THREADS = 1024;
constexpr double XCHG_TIMEOUT_S = 30.0;  // (a peer's first fit may still be loading code objects or allocating; PLS_HIP_XCHG_TIMEOUT_S)

struct XchgPeers {
    double *slot[XCHG_MAX];                 // inbox slot [parity][my rank] of every member
    unsigned long long *flag[XCHG_MAX];     // its flag
};

// The consumer side of one collective as kernel arguments: what a kernel needs to wait for the members' pushes of
// collective `seq` and to read their vectors (the gather as the PROLOGUE of the component update, small_kernels.hpp)
struct XchgGather {
    const double *inbox = nullptr;              // this member's slots of the parity seq & 1: [n][cap]
    const unsigned long long *flags = nullptr;  // its n flags of that parity
    int n = 0;                                  // 0: no gather (the reduced slices are in `red`)
    i64 cap = 0;
    unsigned long long seq = 0;
    int *status = nullptr, *host_status = nullptr;
    long long limit = 0;
};

// buf: `slices` slices of Ltot doubles; the piece [j0, j0 + L) of their sum goes out
__global__ __launch_bounds__(XCHG_THREADS) void xchg_push_kernel(XchgPeers peers, const double *__restrict__ buf, i64 Ltot, i64 j0,
                                                                 int L, int slices, unsigned long long seq) {
    double *dst = peers.slot[blockIdx.x];
    for (int j = threadIdx.x; j < L; j += XCHG_THREADS) {
        double sum = buf[j0 + j];
        for (int sl = 1; sl < slices; ++sl) sum += buf[(i64)sl * Ltot + j0 + j];
        __hip_atomic_store(dst + j, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(peers.flag[blockIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void xchg_gather_kernel(const double *inbox, const unsigned long long *flags, int n, i64 cap,
                                                          i64 Ltot, i64 j0, int L, int slices, unsigned long long seq,
                                                          double *__restrict__ buf, int *status, int *host_status, long long limit) {
    __shared__ int ok;
    if (threadIdx.x == 0) ok = (*status == 0);  // an earlier wait of this member timed out: do not wait again
    __syncthreads();
    if ((int)threadIdx.x < n && ok) {
        const long long t0 = wall_clock64();  // limit: ticks of the device's wall clock (hipDeviceAttributeWallClockRate)
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (!ok && threadIdx.x == 0) {
        *status = 1;
        __hip_atomic_store(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (host-mapped: read without a copy)
    }
    for (int j = blockIdx.x * 256 + threadIdx.x; j < L; j += gridDim.x * 256) {
        double sum = ok ? __hip_atomic_load(inbox + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __builtin_nan("");
        for (int m = 1; m < n && ok; ++m) sum += __hip_atomic_load(inbox + (i64)m * cap + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[j0 + j] = sum;
        for (int sl = 1; sl < slices; ++sl) buf[(i64)sl * Ltot + j0 + j] = 0.0;
    }
}

// Push and gather of one SHORT piece in ONE launch: block 0 pushes to every member (the body of xchg_push_kernel with the
// destinations in a loop), the blocks behind it gather (xchg_gather_kernel).  The gather blocks wait for the member's own
// flag like for every other; block 0 publishes its flags only after it has read all of buf and issued every store, so a
// gather block never overwrites an element the push still has to read.  Block 0 is dispatched first; the grid is at most
// 5 workgroups.  One launch boundary fewer per collective (long pieces take the two launches, one push workgroup per
// destination).
constexpr int XCHG_FUSED_MAX = 4096;  // doubles
__global__ __launch_bounds__(XCHG_THREADS) void xchg_push_gather_kernel(XchgPeers peers, int do_push, const double *inbox,
                                                                        const unsigned long long *flags, int n, i64 cap, i64 Ltot,
                                                                        i64 j0, int L, int slices, unsigned long long seq, double *buf,
                                                                        int *status, int *host_status, long long limit) {
    if (blockIdx.x == 0) {
        if (!do_push) return;
        for (int j = threadIdx.x; j < L; j += XCHG_THREADS) {
            double sum = buf[j0 + j];
            for (int sl = 1; sl < slices; ++sl) sum += buf[(i64)sl * Ltot + j0 + j];
            for (int d = 0; d < n; ++d) __hip_atomic_store(peers.slot[d] + j, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();
        __syncthreads();
        if ((int)threadIdx.x < n) __hip_atomic_store(peers.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const int gb = blockIdx.x - 1, ngb = gridDim.x - 1;
    __shared__ int ok;
    if (threadIdx.x == 0) ok = (*status == 0);
    __syncthreads();
    if ((int)threadIdx.x < n && ok) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (wall_clock64() - t0 > limit) {
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (!ok && threadIdx.x == 0 && gb == 0) {
        *status = 1;
        __hip_atomic_store(host_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int j = gb * XCHG_THREADS + threadIdx.x; j < L; j += ngb * XCHG_THREADS) {
        double sum = ok ? __hip_atomic_load(inbox + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __builtin_nan("");
        for (int m = 1; m < n && ok; ++m) sum += __hip_atomic_load(inbox + (i64)m * cap + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[j0 + j] = sum;
        for (int sl = 1; sl < slices; ++sl) buf[(i64)sl * Ltot + j0 + j] = 0.0;
    }
}

// the launch(es) of one collective piece; seq: the member's running collective number (already incremented)
inline int xchg_launch_piece(hipStream_t stream, int n, int rank, double *const *inboxes, unsigned long long *const *flagsv,
                             double *buf, i64 Ltot, i64 j0, int L, int slices, unsigned long long seq, int *status,
                             int *host_status, long long limit) {
    const int par = (int)(seq & 1);
    XchgPeers peers;
    for (int j = 0; j < n; ++j) {
        peers.slot[j] = inboxes[j] + ((i64)par * n + rank) * XCHG_CAP;
        peers.flag[j] = flagsv[j] + par * n + rank;
    }
    bool skip = false;
#ifdef PLS_HIP_TESTING
    // fault injection for the tests of the time-out path -- compiled into testing/libpls_hip.so ONLY (make testing; the
    // production library has no such branch): PLS_HIP_TEST_DROP_PUSH="rank:collective" makes that rank skip that one push,
    // once per process (its peers -- and itself -- then wait for a flag that never comes).  The flag is atomic: the members
    // of a group call this from concurrent host threads.
    static const char *drop = getenv("PLS_HIP_TEST_DROP_PUSH");
    static std::atomic<bool> dropped{false};  // (once per process)
    if (drop && !dropped.load(std::memory_order_relaxed)) {
        int dr = -1;
        unsigned long long dq = 0;
        if (sscanf(drop, "%d:%llu", &dr, &dq) == 2 && dr == rank && dq == seq) skip = !dropped.exchange(true);
    }
#endif
    if (L <= XCHG_FUSED_MAX) {
        const int ngb = (int)std::min<i64>(4, (L + XCHG_THREADS - 1) / XCHG_THREADS);
        hipLaunchKernelGGL(xchg_push_gather_kernel, dim3(1 + ngb), dim3(XCHG_THREADS), 0, stream, peers, skip ? 0 : 1,
                           (const double *)(inboxes[rank] + (i64)par * n * XCHG_CAP), (const unsigned long long *)(flagsv[rank] + par * n),
                           n, XCHG_CAP, Ltot, j0, L, slices, seq, buf, status, host_status, limit);
        return hipGetLastError() == hipSuccess ? 0 : 13;
    }
    if (!skip)
        hipLaunchKernelGGL(xchg_push_kernel, dim3(n), dim3(XCHG_THREADS), 0, stream, peers, (const double *)buf, Ltot, j0, L, slices, seq);
    hipLaunchKernelGGL(xchg_gather_kernel, dim3((unsigned)std::min<i64>(16, (L + 255) / 256)), dim3(256), 0, stream,
                       (const double *)(inboxes[rank] + (i64)par * n * XCHG_CAP), (const unsigned long long *)(flagsv[rank] + par * n),
                       n, XCHG_CAP, Ltot, j0, L, slices, seq, buf, status, host_status, limit);
    return hipGetLastError() == hipSuccess ? 0 : 13;
}

inline long long xchg_time_limit(int device) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;
    const char *te = getenv("PLS_HIP_XCHG_TIMEOUT_S");
    const double secs = (te && atof(te) > 0) ? atof(te) : XCHG_TIMEOUT_S;
    return (long long)(secs * 1e3 * khz);
}

}  // namespace plsk
