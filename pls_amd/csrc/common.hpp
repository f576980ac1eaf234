// Shared device helpers for the gfx950 PLS kernels.  Wave = 64 lanes, workgroup = 256 threads
// unless a kernel says otherwise.  All reductions are fixed-order (no float atomics) so a
// fit is bit-reproducible run to run and bit-identical across the ranks of a sharded fit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

namespace plsk {

typedef int64_t i64;

constexpr int WAVE = 64;
constexpr int WG = 256;

// VEC consecutive rows of one column = one 16-byte (or narrower) global access per lane.
template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T v[V];
};

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld_pack(const T *p) {
    return *reinterpret_cast<const Pack<T, V> *>(p);
}
template <typename T, int V>
__device__ __forceinline__ void st_pack(T *p, const Pack<T, V> &x) {
    *reinterpret_cast<Pack<T, V> *>(p) = x;
}

// Streaming (nt) forms for the big matrix: X is touched once per pass and never fits a cache, so
// its traffic should not evict the vectors that ARE reused (scores, partial sums).
template <typename T, int V>
struct NtVec { typedef T type __attribute__((ext_vector_type(V))); };
template <typename T>
struct NtVec<T, 1> { typedef T type; };

template <typename T, int V>
__device__ __forceinline__ Pack<T, V> ld_pack_nt(const T *p) {
    typedef typename NtVec<T, V>::type VT;
    const VT r = __builtin_nontemporal_load(reinterpret_cast<const VT *>(p));
    Pack<T, V> o;
    __builtin_memcpy(&o, &r, sizeof(o));
    return o;
}
template <typename T, int V>
__device__ __forceinline__ void st_pack_nt(T *p, const Pack<T, V> &x) {
    typedef typename NtVec<T, V>::type VT;
    VT r;
    __builtin_memcpy(&r, &x, sizeof(r));
    __builtin_nontemporal_store(r, reinterpret_cast<VT *>(p));
}

__device__ __forceinline__ double shfl_xor_f64(double x, int mask) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __shfl_xor(lo, mask, WAVE);
    hi = __shfl_xor(hi, mask, WAVE);
    return __hiloint2double(hi, lo);
}

// butterfly sum over the 64 lanes of a wave: every lane ends with the same bits
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += shfl_xor_f64(x, m);
    return x;
}

// Sum over a workgroup of NW waves (blockDim.x = 64*NW); result valid in every thread.
// smem: >= NW doubles.  Fixed order: butterfly inside the wave, then waves 0..NW-1.
template <int NW>
__device__ __forceinline__ double block_sum(double x, double *smem) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smem[w] = x;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += smem[i];
    return s;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that drives several
// GPUs (pls_hip_group) must raise it once on each of them.  true = the kernel may be launched with `bytes` of
// dynamic LDS on the current device.
inline bool raise_dynamic_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.insert({fn, dev});
    return true;
}

}  // namespace plsk
