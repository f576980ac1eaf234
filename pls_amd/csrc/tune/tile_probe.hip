// tile_probe -- how fast can MI355X stream a COLUMN-MAJOR N x K fp64 matrix when a workgroup
// must hold a full-width row tile (R rows x all K columns) resident, as the fused
// score+loading(+deflation) pass needs?  A tile of R rows is K separate R*8-byte segments, one
// per column (column stride N*8 bytes), so R sets the contiguous segment length.
// Prints achieved GB/s (algorithmic bytes / event time) for read-only and read+write sweeps.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tile_probe tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef long long i64;
struct alignas(16) P2 { double v[2]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// R rows per tile, NT threads.  lane layout: rp = tid % (R/2) (row pair), cg = tid / (R/2).
template <int R, int NT, int K, bool WRITE>
__global__ __launch_bounds__(NT) void tile_stream(const double* __restrict__ X, double* __restrict__ Xo,
                                                  i64 N, double* __restrict__ sink, double scale) {
    constexpr int RP = R / 2, CG = NT / RP, CPT = K / CG;
    static_assert(K % CG == 0 && NT % RP == 0, "shape");
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    double acc = 0.0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 i0 = t * R + 2 * rp;
        P2 x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = *reinterpret_cast<const P2*>(X + i0 + (i64)(cg + CG * j) * N);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            acc += x[j].v[0] + x[j].v[1];
            if (WRITE) {
                x[j].v[0] *= scale; x[j].v[1] *= scale;
                *reinterpret_cast<P2*>(Xo + i0 + (i64)(cg + CG * j) * N) = x[j];
            }
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// plain contiguous copy: the chip's read+write ceiling for comparison
template <int U, int MODE>  // MODE 0 plain, 1 nontemporal store, 2 nontemporal load+store
__global__ __launch_bounds__(256) void copy_stream(const P2* __restrict__ A, P2* __restrict__ B, i64 n) {
    const i64 stride = (i64)gridDim.x * 256;
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        P2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 2) { x[u].v[0] = __builtin_nontemporal_load(&A[i + u * stride].v[0]); x[u].v[1] = __builtin_nontemporal_load(&A[i + u * stride].v[1]); }
            else x[u] = A[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE >= 1) { __builtin_nontemporal_store(x[u].v[0], &B[i + u * stride].v[0]); __builtin_nontemporal_store(x[u].v[1], &B[i + u * stride].v[1]); }
            else B[i + u * stride] = x[u];
        }
    }
    for (; i < n; i += stride) B[i] = A[i];
}

template <int R, int NT, int K, int MODE>
__global__ __launch_bounds__(NT) void tile_rw_nt(const double* __restrict__ X, double* __restrict__ Xo,
                                                  i64 N, double* __restrict__ sink, double scale) {
    constexpr int RP = R / 2, CG = NT / RP, CPT = K / CG;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    double acc = 0.0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 i0 = t * R + 2 * rp;
        P2 x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const double* p = X + i0 + (i64)(cg + CG * j) * N;
            if (MODE == 2) { x[j].v[0] = __builtin_nontemporal_load(p); x[j].v[1] = __builtin_nontemporal_load(p + 1); }
            else x[j] = *reinterpret_cast<const P2*>(p);
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            acc += x[j].v[0] + x[j].v[1];
            x[j].v[0] *= scale; x[j].v[1] *= scale;
            double* q = Xo + i0 + (i64)(cg + CG * j) * N;
            __builtin_nontemporal_store(x[j].v[0], q); __builtin_nontemporal_store(x[j].v[1], q + 1);
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// column streaming baseline: thread = row pair of a 2*NT-row chunk, walks KC columns
template <int NT, int KC, bool WRITE>
__global__ __launch_bounds__(NT) void col_stream(const double* __restrict__ X, double* __restrict__ Xo,
                                                 i64 N, int K, double* __restrict__ sink, double scale) {
    double acc = 0.0;
    const int k0 = blockIdx.y * KC;
    for (i64 c = blockIdx.x; c * (2 * NT) < N; c += gridDim.x) {
        const i64 i0 = c * (2 * NT) + 2 * threadIdx.x;
#pragma unroll
        for (int kb = 0; kb < KC; kb += 8) {
            P2 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const P2*>(X + i0 + (i64)(k0 + kb + u) * N);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc += x[u].v[0] + x[u].v[1];
                if (WRITE) { x[u].v[0] *= scale; x[u].v[1] *= scale;
                    *reinterpret_cast<P2*>(Xo + i0 + (i64)(k0 + kb + u) * N) = x[u]; }
            }
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

template <typename F>
double time_ms(F&& launch, int reps = 7) {
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

template <int R, int NT, bool WRITE>
void run_tile(const char* tag, const double* X, double* Xo, i64 N, double* sink, int wgs) {
    constexpr int K = 512;
    double ms = time_ms([&] { hipLaunchKernelGGL((tile_stream<R, NT, K, WRITE>), dim3(wgs), dim3(NT), 0, 0, X, Xo, N, sink, 1.0); });
    double bytes = (WRITE ? 2.0 : 1.0) * N * K * 8;
    printf("%-10s R=%3d NT=%4d wgs=%5d %s  %.3f ms  %.0f GB/s\n", tag, R, NT, wgs, WRITE ? "rw" : "ro", ms, bytes / ms / 1e6);
}

int main() {
    const i64 N = 1 << 20; const int K = 512;
    double *X, *Xo, *sink;
    CK(hipMalloc(&X, N * K * 8)); CK(hipMalloc(&Xo, N * K * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(X, 0x3c, N * K * 8)); CK(hipMemset(Xo, 0, N * K * 8));
    for (int wr = 0; wr < 2; ++wr) {
        for (int G : {64, 128, 256}) {
            double ms = wr ? time_ms([&] { hipLaunchKernelGGL((col_stream<256, 32, true>), dim3(G, K / 32), dim3(256), 0, 0, X, Xo, N, K, sink, 1.0); })
                           : time_ms([&] { hipLaunchKernelGGL((col_stream<256, 32, false>), dim3(G, K / 32), dim3(256), 0, 0, X, Xo, N, K, sink, 1.0); });
            printf("col_stream G=%d %s %.3f ms %.0f GB/s\n", G, wr ? "rw" : "ro", ms, (wr ? 2.0 : 1.0) * N * K * 8 / ms / 1e6);
        }
        // in place variant of rw
        if (wr) {
            double ms = time_ms([&] { hipLaunchKernelGGL((col_stream<256, 32, true>), dim3(128, K / 32), dim3(256), 0, 0, X, X, N, K, sink, 1.0); });
            printf("col_stream G=128 rw-inplace %.3f ms %.0f GB/s\n", ms, 2.0 * N * K * 8 / ms / 1e6);
        }
    }
    {
        const i64 n = N * K / 2;
        for (int g : {1024, 2048, 4096, 8192}) {
            double m0 = time_ms([&] { hipLaunchKernelGGL((copy_stream<4, 0>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, n); });
            double m1 = time_ms([&] { hipLaunchKernelGGL((copy_stream<4, 1>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, n); });
            double m2 = time_ms([&] { hipLaunchKernelGGL((copy_stream<4, 2>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, n); });
            double m3 = time_ms([&] { hipLaunchKernelGGL((copy_stream<8, 0>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, n); });
            double m4 = time_ms([&] { hipLaunchKernelGGL((copy_stream<4, 0>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)X, n); });
            printf("copy g=%d: plain %.0f  nt-store %.0f  nt-both %.0f  U8 %.0f  inplace %.0f GB/s\n", g, 2.0*N*K*8/m0/1e6, 2.0*N*K*8/m1/1e6, 2.0*N*K*8/m2/1e6, 2.0*N*K*8/m3/1e6, 2.0*N*K*8/m4/1e6);
        }
        double ms = time_ms([&] { CK(hipMemcpyAsync(Xo, X, N * K * 8, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D %.0f GB/s\n", 2.0 * N * K * 8 / ms / 1e6);
        for (int wgs : {512, 2048}) {
            double a1 = time_ms([&] { hipLaunchKernelGGL((tile_rw_nt<32, 512, 512, 1>), dim3(wgs), dim3(512), 0, 0, X, Xo, N, sink, 1.0); });
            double a2 = time_ms([&] { hipLaunchKernelGGL((tile_rw_nt<32, 512, 512, 2>), dim3(wgs), dim3(512), 0, 0, X, Xo, N, sink, 1.0); });
            double a3 = time_ms([&] { hipLaunchKernelGGL((tile_rw_nt<32, 512, 512, 1>), dim3(wgs), dim3(512), 0, 0, X, X, N, sink, 1.0); });
            double a4 = time_ms([&] { hipLaunchKernelGGL((tile_rw_nt<64, 1024, 512, 1>), dim3(wgs), dim3(1024), 0, 0, X, X, N, sink, 1.0); });
            printf("tile_rw_nt wgs=%d: R32 nt-store %.0f  nt-both %.0f  inplace nt-store %.0f  R64 inplace nt-store %.0f GB/s\n", wgs, 2.0*N*K*8/a1/1e6, 2.0*N*K*8/a2/1e6, 2.0*N*K*8/a3/1e6, 2.0*N*K*8/a4/1e6);
        }
    }
    for (int wgs : {2048}) {
        run_tile<16, 256, false>("tile", X, Xo, N, sink, wgs);
        run_tile<32, 256, false>("tile", X, Xo, N, sink, wgs);
        run_tile<32, 512, false>("tile", X, Xo, N, sink, wgs);
        run_tile<64, 512, false>("tile", X, Xo, N, sink, wgs);
        run_tile<64, 1024, false>("tile", X, Xo, N, sink, wgs);
        run_tile<128, 1024, false>("tile", X, Xo, N, sink, wgs);
        run_tile<32, 512, true>("tile", X, Xo, N, sink, wgs);
        run_tile<64, 1024, true>("tile", X, Xo, N, sink, wgs);
        run_tile<128, 1024, true>("tile", X, Xo, N, sink, wgs);
        run_tile<32, 512, true>("tile-inpl", X, X, N, sink, wgs);
        run_tile<64, 1024, true>("tile-inpl", X, X, N, sink, wgs);
    }
    return 0;
}
