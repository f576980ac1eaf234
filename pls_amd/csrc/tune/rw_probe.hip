// rw_probe -- what limits a read+write sweep of a 4.3 GB fp64 matrix on MI355X?  torch's elementwise
// add reaches 5.9 TB/s (out of place) / 5.5 TB/s (in place) on this chip while every persistent
// (grid = 2 x CUs, grid-stride) kernel of this library and hipMemcpyDtoD stay at 4.8-5.1 TB/s.
// Variables probed: one-shot workgroups (one chunk or tile per workgroup, hardware-dispatched) vs
// persistent loops, bytes per workgroup, contiguous chunks vs column-major tiles (256-byte column
// segments), in place vs out of place, nt vs plain stores.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o rw_probe rw_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef long long i64;
struct alignas(16) P2 { double v[2]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one-shot contiguous: workgroup b handles U*256 packs starting at b*U*256
template <int U, bool NT>
__global__ __launch_bounds__(256) void oneshot_add(const P2* __restrict__ A, P2* B, double s) {
    const i64 base = (i64)blockIdx.x * (U * 256) + threadIdx.x;
    P2 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = A[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        x[u].v[0] += s; x[u].v[1] += s;
        if (NT) { __builtin_nontemporal_store(x[u].v[0], &B[base + u * 256].v[0]); __builtin_nontemporal_store(x[u].v[1], &B[base + u * 256].v[1]); }
        else B[base + u * 256] = x[u];
    }
}
// same work, persistent grid-stride loop over the chunks
template <int U, bool NT>
__global__ __launch_bounds__(256) void persist_add(const P2* __restrict__ A, P2* B, double s, i64 nchunks) {
    for (i64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const i64 base = c * (U * 256) + threadIdx.x;
        P2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = A[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u].v[0] += s; x[u].v[1] += s;
            if (NT) { __builtin_nontemporal_store(x[u].v[0], &B[base + u * 256].v[0]); __builtin_nontemporal_store(x[u].v[1], &B[base + u * 256].v[1]); }
            else B[base + u * 256] = x[u];
        }
    }
}
// column-major tile, R rows x K columns, tiles [t0, t0 + TPW) per workgroup (one-shot) or grid-stride (persistent)
template <int R, int NTH, int K, bool NT, bool PERSIST>
__global__ __launch_bounds__(NTH) void tile_add(const double* __restrict__ X, double* Xo, i64 N, double s, int tpw) {
    constexpr int RP = R / 2, CG = NTH / RP, CPT = K / CG;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    i64 t = PERSIST ? blockIdx.x : (i64)blockIdx.x * tpw;
    const i64 tend = PERSIST ? ntiles : std::min<i64>(ntiles, t + tpw);
    const i64 step = PERSIST ? gridDim.x : 1;
    for (; t < tend; t += step) {
        const i64 i0 = t * R + 2 * rp;
        P2 x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = *reinterpret_cast<const P2*>(X + i0 + (i64)(cg + CG * j) * N);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            x[j].v[0] += s; x[j].v[1] += s;
            double* q = Xo + i0 + (i64)(cg + CG * j) * N;
            if (NT) { __builtin_nontemporal_store(x[j].v[0], q); __builtin_nontemporal_store(x[j].v[1], q + 1); }
            else *reinterpret_cast<P2*>(q) = x[j];
        }
    }
}

// tile kernel with a rolling window of W loads in flight per lane: store j is issued as soon as load j has
// returned and load j + W follows it; all CPT packs stay in registers (the fused pass needs them after the
// score is known).  W = CPT is tile_add.
template <int R, int NTH, int K, int W, bool PERSIST>
__global__ __launch_bounds__(NTH) void tile_add_window(const double* __restrict__ X, double* Xo, i64 N, double s, int tpw, double* sink) {
    constexpr int RP = R / 2, CG = NTH / RP, CPT = K / CG;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    i64 t = PERSIST ? blockIdx.x : (i64)blockIdx.x * tpw;
    const i64 tend = PERSIST ? ntiles : std::min<i64>(ntiles, t + tpw);
    const i64 step = PERSIST ? gridDim.x : 1;
    double acc = 0.0;
    for (; t < tend; t += step) {
        const i64 i0 = t * R + 2 * rp;
        P2 x[CPT];
#pragma unroll
        for (int j = 0; j < W; ++j) x[j] = *reinterpret_cast<const P2*>(X + i0 + (i64)(cg + CG * j) * N);
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            x[j].v[0] += s; x[j].v[1] += s;
            *reinterpret_cast<P2*>(Xo + i0 + (i64)(cg + CG * j) * N) = x[j];
            if (j + W < CPT) x[j + W] = *reinterpret_cast<const P2*>(X + i0 + (i64)(cg + CG * (j + W)) * N);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc += x[j].v[0] * x[j].v[1];  // keeps the tile live to the end
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// one-shot workgroup = R rows x (CG*CPT) columns of a tile: blockIdx.x = tile, blockIdx.y = column block
template <int R, int NTH, int CPT, bool XFAST>
__global__ __launch_bounds__(NTH) void tile_colblock(const double* __restrict__ X, double* Xo, i64 N, double s) {
    constexpr int RP = R / 2, CG = NTH / RP;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 t = XFAST ? blockIdx.x : blockIdx.y;
    const int cb = XFAST ? blockIdx.y : blockIdx.x;
    const i64 i0 = t * R + 2 * rp;
    P2 x[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) x[j] = *reinterpret_cast<const P2*>(X + i0 + (i64)(cb * CG * CPT + cg + CG * j) * N);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        x[j].v[0] += s; x[j].v[1] += s;
        *reinterpret_cast<P2*>(Xo + i0 + (i64)(cb * CG * CPT + cg + CG * j) * N) = x[j];
    }
}

// rank-1 deflation dst = src - t p^T with one-shot workgroups: NTH threads = NTH/256 columns x 512 rows (2 rows
// per thread), CPT column sets per thread.  blockIdx.x = row block (fastest), blockIdx.y = column block.
template <int NTH, int CPT>
__global__ __launch_bounds__(NTH) void defl_oneshot(const double* __restrict__ X, double* Xo, i64 N, const double* __restrict__ t,
                                                    const double* __restrict__ p) {
    constexpr int CW = NTH / 256;  // columns side by side in the workgroup
    const int rp = threadIdx.x % 256, cw = threadIdx.x / 256;
    const i64 i0 = (i64)blockIdx.x * 512 + 2 * rp;
    const P2 tv = *reinterpret_cast<const P2*>(t + i0);
    P2 x[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) x[j] = *reinterpret_cast<const P2*>(X + i0 + (i64)((blockIdx.y * CPT + j) * CW + cw) * N);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const double pk = p[(blockIdx.y * CPT + j) * CW + cw];
        x[j].v[0] = fma(-tv.v[0], pk, x[j].v[0]); x[j].v[1] = fma(-tv.v[1], pk, x[j].v[1]);
        *reinterpret_cast<P2*>(Xo + i0 + (i64)((blockIdx.y * CPT + j) * CW + cw) * N) = x[j];
    }
}

// read-only sweeps: one-shot contiguous chunks of U*4 KB per workgroup, and persistent grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void oneshot_read(const P2* __restrict__ A, double* sink) {
    const i64 base = (i64)blockIdx.x * (U * 256) + threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        P2 x;
        if (NT) { x.v[0] = __builtin_nontemporal_load(&A[base + u * 256].v[0]); x.v[1] = __builtin_nontemporal_load(&A[base + u * 256].v[1]); }
        else x = A[base + u * 256];
        acc += x.v[0] * x.v[1];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void persist_read(const P2* __restrict__ A, double* sink, i64 nchunks) {
    double acc = 0.0;
    for (i64 c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const i64 base = c * (U * 256) + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            P2 x;
            if (NT) { x.v[0] = __builtin_nontemporal_load(&A[base + u * 256].v[0]); x.v[1] = __builtin_nontemporal_load(&A[base + u * 256].v[1]); }
            else x = A[base + u * 256];
            acc += x.v[0] * x.v[1];
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

int main() {
    const i64 N = 1 << 20; const int K = 512;
    const double bytes = 2.0 * N * K * 8;
    double *X, *Xo;
    CK(hipMalloc(&X, N * K * 8)); CK(hipMalloc(&Xo, N * K * 8));
    CK(hipMemset(X, 0, N * K * 8)); CK(hipMemset(Xo, 0, N * K * 8));
    const i64 npacks = N * K / 2;
    if (getenv("RW_PROBE_READS")) {
        const double rbytes = 1.0 * N * K * 8;
#define RD(U)                                                                                                             \
        {                                                                                                                 \
            const i64 g = npacks / (U * 256);                                                                             \
            double a = time_ms([&] { hipLaunchKernelGGL((oneshot_read<U, false>), dim3(g), dim3(256), 0, 0, (const P2*)X, Xo); }); \
            double b = time_ms([&] { hipLaunchKernelGGL((oneshot_read<U, true>), dim3(g), dim3(256), 0, 0, (const P2*)X, Xo); });  \
            printf("read oneshot %3d KB/WG: plain %.0f nt %.0f GB/s |", U * 4, rbytes / a / 1e6, rbytes / b / 1e6);       \
            for (int pg : {512, 2048, 8192}) {                                                                            \
                double e = time_ms([&] { hipLaunchKernelGGL((persist_read<U, false>), dim3(pg), dim3(256), 0, 0, (const P2*)X, Xo, g); }); \
                double f = time_ms([&] { hipLaunchKernelGGL((persist_read<U, true>), dim3(pg), dim3(256), 0, 0, (const P2*)X, Xo, g); });  \
                printf(" persistent %d: plain %.0f nt %.0f |", pg, rbytes / e / 1e6, rbytes / f / 1e6);                   \
            }                                                                                                             \
            printf("\n");                                                                                                 \
        }
        RD(1) RD(2) RD(4) RD(8) RD(16) RD(32)
        return 0;
    }
#define ONESHOT(U)                                                                                                        \
    {                                                                                                                     \
        const i64 g = npacks / (U * 256);                                                                                 \
        double a = time_ms([&] { hipLaunchKernelGGL((oneshot_add<U, false>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, 0.0); }); \
        double b = time_ms([&] { hipLaunchKernelGGL((oneshot_add<U, true>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, 0.0); });  \
        double c = time_ms([&] { hipLaunchKernelGGL((oneshot_add<U, false>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)X, 0.0); });  \
        double d = time_ms([&] { hipLaunchKernelGGL((oneshot_add<U, true>), dim3(g), dim3(256), 0, 0, (const P2*)X, (P2*)X, 0.0); });   \
        printf("oneshot contiguous %3d KB/WG: out-of-place plain %.0f nt %.0f | in-place plain %.0f nt %.0f GB/s\n", U * 4, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6); \
        for (int pg : {512, 2048, 8192}) {                                                                                \
            double e = time_ms([&] { hipLaunchKernelGGL((persist_add<U, false>), dim3(pg), dim3(256), 0, 0, (const P2*)X, (P2*)Xo, 0.0, g); }); \
            double f = time_ms([&] { hipLaunchKernelGGL((persist_add<U, false>), dim3(pg), dim3(256), 0, 0, (const P2*)X, (P2*)X, 0.0, g); });  \
            printf("   persistent grid %5d: out-of-place %.0f | in-place %.0f GB/s\n", pg, bytes / e / 1e6, bytes / f / 1e6); \
        }                                                                                                                 \
    }
    ONESHOT(1) ONESHOT(4) ONESHOT(16)
#define TILE(R, NTH, NT_, tag)                                                                                            \
    for (int tpw : {1, 2, 4, 8}) {                                                                                        \
        const i64 g = (N / R + tpw - 1) / tpw;                                                                            \
        double a = time_ms([&] { hipLaunchKernelGGL((tile_add<R, NTH, 512, NT_, false>), dim3(g), dim3(NTH), 0, 0, X, Xo, N, 0.0, tpw); }); \
        double b = time_ms([&] { hipLaunchKernelGGL((tile_add<R, NTH, 512, NT_, false>), dim3(g), dim3(NTH), 0, 0, X, X, N, 0.0, tpw); });  \
        printf("oneshot tile R=%d NT=%d %s tiles/WG=%d: out-of-place %.0f | in-place %.0f GB/s\n", R, NTH, tag, tpw, bytes / a / 1e6, bytes / b / 1e6); \
    }                                                                                                                     \
    {                                                                                                                     \
        double a = time_ms([&] { hipLaunchKernelGGL((tile_add<R, NTH, 512, NT_, true>), dim3(512), dim3(NTH), 0, 0, X, Xo, N, 0.0, 0); }); \
        double b = time_ms([&] { hipLaunchKernelGGL((tile_add<R, NTH, 512, NT_, true>), dim3(512), dim3(NTH), 0, 0, X, X, N, 0.0, 0); });  \
        printf("persistent tile R=%d NT=%d %s grid 512: out-of-place %.0f | in-place %.0f GB/s\n", R, NTH, tag, bytes / a / 1e6, bytes / b / 1e6); \
    }
    TILE(32, 512, false, "plain")
    {
        double *tvec, *pvec;
        CK(hipMalloc(&tvec, N * 8)); CK(hipMalloc(&pvec, K * 8)); CK(hipMemset(tvec, 0, N * 8)); CK(hipMemset(pvec, 0, K * 8));
#define DEFL(NTH, CPT)                                                                                                    \
        {                                                                                                                 \
            const dim3 g((unsigned)(N / 512), K / ((NTH / 256) * CPT));                                                   \
            double a = time_ms([&] { hipLaunchKernelGGL((defl_oneshot<NTH, CPT>), g, dim3(NTH), 0, 0, X, Xo, N, tvec, pvec); }); \
            double b = time_ms([&] { hipLaunchKernelGGL((defl_oneshot<NTH, CPT>), g, dim3(NTH), 0, 0, X, X, N, tvec, pvec); });  \
            printf("defl_oneshot NT=%d CPT=%d (%d KB/WG): out-of-place %.0f | in-place %.0f GB/s\n", NTH, CPT, 4 * (NTH / 256) * CPT, bytes / a / 1e6, bytes / b / 1e6); \
        }
        DEFL(256, 1) DEFL(256, 2) DEFL(256, 4) DEFL(512, 1) DEFL(512, 2) DEFL(1024, 1) DEFL(1024, 2)
    }
#define CB(R, NTH, CPT)                                                                                                   \
    {                                                                                                                     \
        constexpr int CGc = NTH / (R / 2);                                                                                \
        const dim3 g1((unsigned)(N / R), K / (CGc * CPT)), g2(K / (CGc * CPT), (unsigned)(N / R));                        \
        double a = time_ms([&] { hipLaunchKernelGGL((tile_colblock<R, NTH, CPT, true>), g1, dim3(NTH), 0, 0, X, Xo, N, 0.0); }); \
        double b = time_ms([&] { hipLaunchKernelGGL((tile_colblock<R, NTH, CPT, true>), g1, dim3(NTH), 0, 0, X, X, N, 0.0); });  \
        double c = N / R > 65535 ? 0.0 : time_ms([&] { hipLaunchKernelGGL((tile_colblock<R, NTH, CPT, false>), g2, dim3(NTH), 0, 0, X, Xo, N, 0.0); }); \
        double d = N / R > 65535 ? 0.0 : time_ms([&] { hipLaunchKernelGGL((tile_colblock<R, NTH, CPT, false>), g2, dim3(NTH), 0, 0, X, X, N, 0.0); });  \
        printf("colblock R=%d NT=%d CPT=%d (%d KB/WG): tile-fastest out %.0f in %.0f | column-block-fastest out %.0f in %.0f GB/s\n", R, NTH, CPT, R * CGc * CPT * 8 / 1024, bytes / a / 1e6, bytes / b / 1e6, c > 0 ? bytes / c / 1e6 : 0.0, d > 0 ? bytes / d / 1e6 : 0.0); \
    }
    CB(32, 256, 1) CB(32, 256, 2) CB(32, 256, 4) CB(32, 512, 1) CB(32, 512, 4) CB(128, 256, 1) CB(128, 256, 4) CB(512, 256, 1) CB(512, 256, 4)
#define WIN(R, NTH, W)                                                                                                    \
    {                                                                                                                     \
        double a = time_ms([&] { hipLaunchKernelGGL((tile_add_window<R, NTH, 512, W, true>), dim3(512), dim3(NTH), 0, 0, X, Xo, N, 0.0, 0, Xo); }); \
        double b = time_ms([&] { hipLaunchKernelGGL((tile_add_window<R, NTH, 512, W, true>), dim3(512), dim3(NTH), 0, 0, X, X, N, 0.0, 0, Xo); });  \
        const i64 g = N / R;                                                                                              \
        double c = time_ms([&] { hipLaunchKernelGGL((tile_add_window<R, NTH, 512, W, false>), dim3(g), dim3(NTH), 0, 0, X, Xo, N, 0.0, 1, Xo); }); \
        double d = time_ms([&] { hipLaunchKernelGGL((tile_add_window<R, NTH, 512, W, false>), dim3(g), dim3(NTH), 0, 0, X, X, N, 0.0, 1, Xo); });  \
        printf("window tile R=%d NT=%d W=%2d: persistent(512) out %.0f in %.0f | oneshot out %.0f in %.0f GB/s\n", R, NTH, W, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6); \
    }
    WIN(32, 512, 1) WIN(32, 512, 2) WIN(32, 512, 4) WIN(32, 512, 8) WIN(32, 512, 16)
    WIN(32, 1024, 1) WIN(32, 1024, 2) WIN(32, 1024, 4) WIN(32, 1024, 8)
    WIN(64, 1024, 2) WIN(64, 1024, 4) WIN(64, 1024, 16)
    return 0;
}
