// Shared device helpers for the gfx950 PLS kernels.  Wave = 64 lanes, workgroup = 256 threads
// unless a kernel says otherwise.  All reductions are fixed-order (no float atomics) so a
// fit is bit-reproducible run to run and bit-identical across the ranks of a sharded fit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <map>
#include <utility>

namespace plsk {

typedef int64_t i64;

constexpr int WAVE = 64;
constexpr int WG = 256;
// slices of every reduced vector (reduce_partials_kernel, slice_tail): enough workgroups take part in a reduction, and the
// consumers add the slices of a value in index order
constexpr int RED_SLICES = 8;

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

// ---- cross-lane exchanges on the VALU (gfx950): DPP moves inside a row of 16 lanes, v_permlane{16,32}_swap across rows.
// __shfl_xor compiles to ds_bpermute_b32 -- an LDS-crossbar round trip of ~100 cycles per 32-bit half; the forms below are
// ordinary vector instructions.  dpp_ctrl: 0xB1 / 0x4E quad_perm [1,0,3,2] / [2,3,0,1] (lane ^ 1, lane ^ 2),
// 0x141 row_half_mirror (l <-> 7-l inside 8 lanes), 0x140 row_mirror (l <-> 15-l inside a row).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// a, b -> (a', b'): ROWS16 = false: a' = [a.lanes 0-31, b.lanes 0-31], b' = [a.lanes 32-63, b.lanes 32-63] (v_permlane32_swap);
// ROWS16 = true: the same with the odd 16-lane rows of a and the even rows of b (v_permlane16_swap).  Either way
// a' + b' = (a + a's partner across the split) in the lanes below the split, (b + b's partner) in the lanes above it.
template <bool ROWS16>
__device__ __forceinline__ void permlane_swap_f64(double &a, double &b) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    u32x2 l, h;
    if constexpr (ROWS16) {
        l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    } else {
        l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    }
    a = __hiloint2double((int)h.x, (int)l.x);
    b = __hiloint2double((int)h.y, (int)l.y);
}
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// x + (x of lane ^ MASK): one level of a butterfly.  MASK 32 / 16: the permlane swap of (x, x); 8: row_ror:8; 4: row_shl:4 for
// the lanes with bit 2 clear (banks 0, 2) and row_shr:4 for the others; 2 / 1: quad_perm.
template <int MASK>
__device__ __forceinline__ double xor_add_f64(double x) {
    static_assert(MASK == 32 || MASK == 16 || MASK == 8 || MASK == 4 || MASK == 2 || MASK == 1, "lane distance");
    if constexpr (MASK >= 16) {
        double a = x, b = x;
        permlane_swap_f64<MASK == 16>(a, b);
        return a + b;
    } else if constexpr (MASK == 4) {
        const int lo = __double2loint(x), hi = __double2hiint(x);
        int rl = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false), rh = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
        rl = __builtin_amdgcn_update_dpp(rl, lo, 0x114, 0xf, 0xa, false);
        rh = __builtin_amdgcn_update_dpp(rh, hi, 0x114, 0xf, 0xa, false);
        return x + __hiloint2double(rh, rl);
    } else {
        return x + dpp_mov_f64<MASK == 8 ? 0x128 : (MASK == 2 ? 0x4E : 0xB1)>(x);
    }
}
// sum over the lanes that differ in the bits LO, 2 LO, ... below HI (powers of two), every lane of the group ends with it
template <int LO, int HI>
__device__ __forceinline__ double xor_range_sum(double x) {
    if constexpr (LO >= HI) {
        return x;
    } else {
        return xor_range_sum<LO * 2, HI>(xor_add_f64<LO>(x));
    }
}

// Sum over the 64 lanes, every lane (and the scalar unit) ends with the same bits: four DPP levels give every row of 16 its
// total, the four row totals are read as scalars and added in row order.  (The ds_bpermute butterfly this replaces cost
// six dependent LDS round trips: config 2's single-launch fit 93 -> 63 us, profiles/r2/small_fits.json.)
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_mov_f64<0xB1>(x);
    x += dpp_mov_f64<0x4E>(x);
    x += dpp_mov_f64<0x141>(x);
    x += dpp_mov_f64<0x140>(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}

// Workgroup barrier for data exchanged through LDS ONLY.  __syncthreads() is a workgroup-scope fence as well: the
// compiler puts s_waitcnt vmcnt(0) in front of it, i.e. every barrier that follows a global store waits for the store's
// acknowledgement from memory.  Where nothing the workgroup stores to global memory is read back by it, the barrier only
// has to order LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// 8-byte store that leaves this XCD's L2 at once (global_store_dwordx2 sc1): the producer side of a hand-off to another
// workgroup of the same launch (MI355X_MICROARCH.md, "Valid forms")
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): a process that drives several
// GPUs (pls_hip_group) must raise it once on each of them.  true = the kernel may be launched with `bytes` of
// dynamic LDS on the current device.  The size raised to is remembered: a later call that asks for MORE raises again
// (callers whose size depends on the problem -- lm_eig_lds_big_kernel: (2 M^2 + M) 8 bytes -- must not be pinned to the
// first problem's), one that asks for less is a look-up.
inline bool raise_dynamic_lds(const void *fn, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    std::lock_guard<std::mutex> lock(mu);
    const auto it = done.find({fn, dev});
    if (it != done.end() && it->second >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done[{fn, dev}] = bytes;
    return true;
}

}  // namespace plsk
