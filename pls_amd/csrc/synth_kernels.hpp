// Device generator of the synthetic benchmark inputs (spec: DESIGN.md "Synthetic inputs").
// Counter-based, libm-free, exact in fp64 (24-bit dyadic uniforms times {0, +-1/2, +-1}), so the
// device output can be checked bit for bit against a host implementation of the same spec.
//   X[i,k] = amp(k) * E(i,k) + sum_{f<8} z(i,f) * L(f,k),   amp(k) = 2^-(h%4) * (8 + (h/4)%8)/32, h = mix64(sA ^ k)
//            (column-dependent noise amplitude: keeps every component of a tall matrix well determined)
//   Y[i,j] = 2^-(j%16) * sum_{f<8} z(i,f) * C(f,j) + 1/8 * Nz(i,j)
// i is the GLOBAL row index, so row shards of one matrix can be generated independently.
#pragma once
#include "common.hpp"

namespace plsk {

constexpr int SYN_F = 8;

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u24(uint64_t stream, uint64_t idx) {  // dyadic uniform in [-1,1)
    const uint64_t h = mix64(stream ^ idx);
    return (double)((int64_t)(h >> 40) - 8388608) * (1.0 / 8388608.0);
}

// tab[f*n + c]: X loadings L in {-1,-1/2,0,1/2,1} (mode 0) or Y loadings C in {-1,0,1} (mode 1);
// mode 0 only: tab[SYN_F*n + c] = noise amplitude of column c (stream sA)
__global__ __launch_bounds__(WG) void synth_table_kernel(double *__restrict__ tab, int n,
                                                         uint64_t stream, int mode, uint64_t sA) {
    const int idx = blockIdx.x * WG + threadIdx.x;
    if (idx >= n * SYN_F) return;
    const int c = idx / SYN_F, f = idx % SYN_F;
    const uint64_t h = mix64(stream ^ (uint64_t)(c * SYN_F + f));
    tab[f * n + c] = mode == 0 ? ((double)(int)(h % 5) - 2.0) * 0.5 : (double)((int)(h % 3) - 1);
    if (mode == 0 && f == 0) {
        const uint64_t ha = mix64(sA ^ (uint64_t)c);
        // 2^-(5 + ha%4) * (8 + (ha/4)%8): exponent field built directly, no libm
        tab[SYN_F * n + c] = (double)(8 + (int)((ha >> 2) & 7)) * __hiloint2double((1023 - 5 - (int)(ha & 3)) << 20, 0);
    }
}

// grid = (row blocks of WG rows, column groups of KC); one thread per row, coalesced stores
template <typename T, int KC>
__global__ __launch_bounds__(WG) void synth_x_kernel(T *__restrict__ X, i64 ldx, i64 row0,
                                                     i64 nrows, int K, uint64_t sE, uint64_t sZ,
                                                     const double *__restrict__ Ltab) {
    const i64 ii = (i64)blockIdx.x * WG + threadIdx.x;
    if (ii >= nrows) return;
    const uint64_t i = (uint64_t)(row0 + ii);
    double z[SYN_F];
#pragma unroll
    for (int f = 0; f < SYN_F; ++f) z[f] = u24(sZ, i * SYN_F + f);
    const int k0 = blockIdx.y * KC, k1 = min(K, k0 + KC);
    for (int k = k0; k < k1; ++k) {
        double s = Ltab[SYN_F * K + k] * u24(sE, i * (uint64_t)K + (uint64_t)k);
#pragma unroll
        for (int f = 0; f < SYN_F; ++f) s += z[f] * Ltab[f * K + k];
        X[ii + (i64)k * ldx] = (T)s;
    }
}

template <typename T>
__global__ __launch_bounds__(WG) void synth_y_kernel(T *__restrict__ Y, i64 ldy, i64 row0,
                                                     i64 nrows, int M, uint64_t sZ, uint64_t sN,
                                                     const double *__restrict__ Ctab) {
    const i64 ii = (i64)blockIdx.x * WG + threadIdx.x;
    if (ii >= nrows) return;
    const uint64_t i = (uint64_t)(row0 + ii);
    double z[SYN_F];
#pragma unroll
    for (int f = 0; f < SYN_F; ++f) z[f] = u24(sZ, i * SYN_F + f);
    for (int j = 0; j < M; ++j) {
        double s = 0.0;
#pragma unroll
        for (int f = 0; f < SYN_F; ++f) s += z[f] * Ctab[f * M + j];
        const double scale = __hiloint2double((1023 - (j % 16)) << 20, 0);  // 2^-(j%16)
        Y[ii + (i64)j * ldy] = (T)(scale * s + 0.125 * u24(sN, i * (uint64_t)M + (uint64_t)j));
    }
}

}  // namespace plsk
