// xty_mfma_probe -- evidence for the "MFMA or VALU?" choice on the tall-skinny contractions
// XY = X^T Y (src/pls.cpp:396) and p = X^T t (:421).  Times, on the same column-major fp64 data,
//   (a) the product's VALU kernel shape (lane = 2 consecutive rows of one column, 1 KiB contiguous per
//       wave-load, per-lane FMA chains, one butterfly per workgroup) and
//   (b) a v_mfma_f64_16x16x4_f64 kernel (A = 16 columns of X x 4 rows, B = 4 rows x 16 responses,
//       of which only M are real),
// and checks that both give the same numbers.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o xty_mfma_probe xty_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long i64;
struct alignas(16) P2 { double v[2]; };
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ P2 ldnt(const double* p) {
    const d2 r = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    P2 o; o.v[0] = r.x; o.v[1] = r.y; return o;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double shx(double x, int m) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __shfl_xor(lo, m, 64); hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}

// (a) VALU: grid (G, K/KC); KC*MT accumulators per lane
template <int KC, int MT>
__global__ __launch_bounds__(256) void xty_valu(const double* __restrict__ X, i64 ldx, const double* __restrict__ Y, i64 ldy,
                                                i64 N, int K, int M, double* __restrict__ part) {
    __shared__ double red[4][KC * MT];
    const int k0 = blockIdx.y * KC;
    double acc[KC][MT];
#pragma unroll
    for (int a = 0; a < KC; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[a][m] = 0;
    for (i64 c = blockIdx.x; c * 512 < N; c += gridDim.x) {
        const i64 i0 = c * 512 + 2 * threadIdx.x;
        P2 y[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) y[m] = *reinterpret_cast<const P2*>(Y + i0 + (i64)m * ldy);
#pragma unroll
        for (int kb = 0; kb < KC; kb += (KC < 8 ? KC : 8)) {
            P2 x[(KC < 8 ? KC : 8)];
#pragma unroll
            for (int u = 0; u < (KC < 8 ? KC : 8); ++u)
                x[u] = ldnt(X + i0 + (i64)(k0 + kb + u) * ldx);
#pragma unroll
            for (int u = 0; u < (KC < 8 ? KC : 8); ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[kb + u][m] = fma(x[u].v[1], y[m].v[1], fma(x[u].v[0], y[m].v[0], acc[kb + u][m]));
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int a = 0; a < KC; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double s = acc[a][m];
            for (int mm = 32; mm >= 1; mm >>= 1) s += shx(s, mm);
            if (lane == 0) red[w][a * MT + m] = s;
        }
    __syncthreads();
    if (threadIdx.x < KC * MT) {
        const int a = threadIdx.x / MT, m = threadIdx.x % MT;
        part[(i64)blockIdx.x * ((i64)K * M) + (k0 + a) + (i64)m * K] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// (b) MFMA f64 16x16x4: one wave = 16 columns of X; lane l: column l&15, row slot l>>4, response l&15
template <int U>
__global__ __launch_bounds__(256) void xty_mfma(const double* __restrict__ X, i64 ldx, const double* __restrict__ Y, i64 ldy,
                                                i64 N, int K, int M, double* __restrict__ part) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kbase = blockIdx.y * 64 + wv * 16;
    const int ci = lane & 15, rg = lane >> 4;
    const double* xp = X + (i64)(kbase + ci) * ldx + 2 * rg;
    const double* yp = Y + (i64)(ci < M ? ci : 0) * ldy + 2 * rg;
    const bool ym = ci < M;
    d4 acc = {0, 0, 0, 0};
    for (i64 c = blockIdx.x; c * (8 * U) < N; c += gridDim.x) {
        const i64 i0 = c * (8 * U);
        P2 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = ldnt(xp + i0 + 8 * u);
            y[u] = *reinterpret_cast<const P2*>(yp + i0 + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].v[0], ym ? y[u].v[0] : 0.0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].v[1], ym ? y[u].v[1] : 0.0, acc, 0, 0, 0);
        }
    }
    // D[i][j]: j = lane & 15 (response), i = (lane >> 4) + 4*reg (column of X)
    if (ci < M)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            part[(i64)blockIdx.x * ((i64)K * M) + (kbase + rg + 4 * r) + (i64)ci * K] = acc[r];
}

__global__ void reduce_rows(const double* part, int nb, int L, double* out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= L) return;
    double s = 0;
    for (int b = 0; b < nb; ++b) s += part[(i64)b * L + j];
    out[j] = s;
}
__global__ void fill(double* p, i64 n, unsigned long long seed) {
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = seed + (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = (double)((long long)(z >> 40) - 8388608) * (1.0 / 8388608.0);
}

template <typename F>
double time_ms(F&& launch, int reps = 9) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const i64 N = 1 << 20; const int K = 512;
    const int only = argc > 1 ? atoi(argv[1]) : 0;  // 1: VALU only, 2: MFMA only (for PMC runs)
    double *X, *Y, *part, *o1, *o2;
    CK(hipMalloc(&X, N * K * 8)); CK(hipMalloc(&Y, N * 8 * 8)); CK(hipMalloc(&part, 2048ll * K * 8 * 8));
    CK(hipMalloc(&o1, K * 8 * 8)); CK(hipMalloc(&o2, K * 8 * 8));
    hipLaunchKernelGGL(fill, dim3((unsigned)((N * K + 255) / 256)), dim3(256), 0, 0, X, N * K, 1ull);
    hipLaunchKernelGGL(fill, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, 0, Y, N * 8, 2ull);
    CK(hipDeviceSynchronize());
    std::vector<double> h1(K * 8), h2(K * 8);
    for (int M : {1, 8}) {
        const double bytes = (double)N * K * 8 + (double)N * M * 8;
        if (only != 2) {
            const int G = 128;
            double ms = M == 1 ? time_ms([&] { hipLaunchKernelGGL((xty_valu<32, 1>), dim3(G, K / 32), dim3(256), 0, 0, X, N, Y, N, N, K, M, part); })
                               : time_ms([&] { hipLaunchKernelGGL((xty_valu<4, 8>), dim3(16, K / 4), dim3(256), 0, 0, X, N, Y, N, N, K, M, part); });
            hipLaunchKernelGGL(reduce_rows, dim3((K * M + 255) / 256), dim3(256), 0, 0, part, M == 1 ? G : 16, K * M, o1);
            printf("M=%d VALU  %.3f ms  %.0f GB/s\n", M, ms, bytes / ms / 1e6);
        }
        if (only != 1) {
            for (int G : {256, 512, 1024}) {
                double ms = time_ms([&] { hipLaunchKernelGGL((xty_mfma<4>), dim3(G, K / 64), dim3(256), 0, 0, X, N, Y, N, N, K, M, part); });
                double ms8 = time_ms([&] { hipLaunchKernelGGL((xty_mfma<8>), dim3(G, K / 64), dim3(256), 0, 0, X, N, Y, N, N, K, M, part); });
                printf("M=%d MFMA  G=%d  U4 %.3f ms %.0f GB/s   U8 %.3f ms %.0f GB/s\n", M, G, ms, bytes / ms / 1e6, ms8, bytes / ms8 / 1e6);
            }
            hipLaunchKernelGGL((xty_mfma<4>), dim3(256, K / 64), dim3(256), 0, 0, X, N, Y, N, N, K, M, part);
            hipLaunchKernelGGL(reduce_rows, dim3((K * M + 255) / 256), dim3(256), 0, 0, part, 256, K * M, o2);
        }
        if (only == 0) {
            CK(hipMemcpy(h1.data(), o1, K * M * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, K * M * 8, hipMemcpyDeviceToHost));
            double d = 0, n = 0;
            for (int i = 0; i < K * M; ++i) { d += (h1[i] - h2[i]) * (h1[i] - h2[i]); n += h1[i] * h1[i]; }
            printf("M=%d  |VALU - MFMA| / |VALU| = %.2e\n", M, std::sqrt(d / n));
        }
    }
    return 0;
}
