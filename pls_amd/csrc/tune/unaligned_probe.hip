// unaligned_probe -- how should a tile-resident sweep read a column-major fp64 matrix whose columns are NOT 16-byte
// aligned (ld odd: every second column starts at 8 mod 16)?  Read-only tile stream, R = 32 rows x K = 512 columns per
// tile (256-byte segments), 512 threads, nt loads; prints the time per sweep for several ways of forming the loads.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o unaligned_probe unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long i64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0/1: one 16-byte load per lane at the natural address (aligned or not, as ld makes it)
// MODE 2  : two 8-byte loads per lane (rows 2rp, 2rp+1)
// MODE 4  : aligned superset: the lane's 16-byte load is shifted back by the column's misalignment m (0/1 elements), the
//           last row lane of a misaligned column also loads the pack behind the tile; values realigned with a DPP-free shuffle
// CHUNK   : a workgroup walks CONSECUTIVE tiles (the line a segment shares with the next tile is re-read by the same CU)
// CHUNK 2: XCD-contiguous: workgroup b runs on XCD b % 8 (round-robin dispatch); every XCD takes one contiguous eighth of the
//          tiles and its workgroups walk it cyclically, so the two tiles that share a 128-byte line are read at about the same
//          time by workgroups behind the SAME L2
// NT: streaming (nt) policy on the loads
template <int MODE, int CHUNK, bool NT = true>
__global__ __launch_bounds__(512, 2) void tile_ro(const double* __restrict__ X, i64 ld, i64 N, double* __restrict__ sink) {
    constexpr int R = 32, RP = 16, CG = 32, CPT = 16;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    double acc = 0.0;
    i64 t0 = blockIdx.x, t1 = ntiles, ts = gridDim.x;
    if (CHUNK == 1) { const i64 per = (ntiles + gridDim.x - 1) / gridDim.x; t0 = blockIdx.x * per; t1 = std::min(ntiles, t0 + per); ts = 1; }
    if (CHUNK == 2) {
        const i64 per = (ntiles + 7) / 8;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        t0 = xcd * per + slot; t1 = std::min(ntiles, (xcd + 1) * per); ts = gridDim.x >> 3;
    }
    for (i64 t = t0; t < t1; t += ts) {
        const i64 i0 = t * R + 2 * rp;
        d2 x[CPT];
        if (MODE <= 1) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const d2u* p = reinterpret_cast<const d2u*>(X + i0 + (i64)(cg + CG * j) * ld);
                x[j] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const double* p = X + i0 + (i64)(cg + CG * j) * ld;
                x[j].x = __builtin_nontemporal_load(p); x[j].y = __builtin_nontemporal_load(p + 1);
            }
        } else {
            const int m = (int)(((i64)cg * ld) & 1);  // (CG even: the misalignment of a lane's columns is the same for all j)
            d2 e[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const double* p = X + i0 + (i64)(cg + CG * j) * ld - m;
                x[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
                if (m && rp == RP - 1) e[j] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + 2));
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                // natural rows (2rp, 2rp+1) of a misaligned column = (own.y, next lane's x)
                double nx = __shfl_down(x[j].x, 1, 16);
                if (rp == RP - 1) nx = e[j].x;
                if (m) { x[j].x = x[j].y; x[j].y = nx; }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc += x[j].x * 1.5 + x[j].y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// MODE 3: one ROW per lane (8-byte loads, 32 lanes along the rows, 16 column groups x 32 columns per lane)
__global__ __launch_bounds__(512, 1) void tile_ro_v1(const double* __restrict__ X, i64 ld, i64 N, double* __restrict__ sink) {
    constexpr int R = 32, RP = 32, CG = 16, CPT = 32;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    double acc = 0.0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 i0 = t * R + rp;
        double x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = __builtin_nontemporal_load(X + i0 + (i64)(cg + CG * j) * ld);
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc += x[j] * 1.5;
    }
    if (acc == 1.2345e300) sink[0] = acc;
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

int main() {
    const i64 N = 1 << 20; const int K = 512;
    double *X, *sink;
    CK(hipMalloc(&X, (N + 8) * K * 8 + 64)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(X, 0x3c, (N + 8) * K * 8 + 64));
    const double gb = (double)N * K * 8 / 1e6;
    for (int wgs : {512, 256}) {
        for (i64 ld : {N, N + 1}) {
            const char* tag = (ld & 1) ? "odd ld " : "even ld";
            double a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 0>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  16-byte loads at the natural address     %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 1>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  ... consecutive tiles per workgroup        %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<2, 0>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  two 8-byte loads per row pair              %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<4, 0>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  aligned superset loads + realignment       %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<4, 1>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  ... consecutive tiles per workgroup        %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 2>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  16-byte loads, XCD-contiguous tiles, nt        %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 2, false>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  16-byte loads, XCD-contiguous tiles, plain     %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 0, false>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  16-byte loads, cyclic tiles, plain             %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<1, 1, false>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  16-byte loads, consecutive tiles, plain        %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL((tile_ro<4, 2>), dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  aligned superset, XCD-contiguous tiles, nt     %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
            a = time_ms([&] { hipLaunchKernelGGL(tile_ro_v1, dim3(wgs), dim3(512), 0, 0, X, ld, N, sink); });
            printf("wgs=%d %s  one row per lane (8-byte loads, 32 columns) %.3f ms %5.0f GB/s\n", wgs, tag, a, gb / a);
        }
    }
    return 0;
}
