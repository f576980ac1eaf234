// xb4_tune -- pacing / tile mapping / workgroup size / batch depth of xb_mfma4_kernel (../xb_mfma4.hpp) on config 3's shape
// (1,048,576 x 512 fp64, 20 and 8 columns): HIP-event time per launch and the fraction of 8 TB/s (X once + the output).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I.. -o xb4_tune xb4_tune.hip
#include "../xb_mfma4.hpp"
#include <cstdio>
#include <vector>
using plsk::i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill_full_kernel(double *p, i64 n, unsigned seed) {  // every mantissa bit random, |x| < 1
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
        h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        p[i] = (double)(long long)h * (1.0 / 9223372036854775808.0);
    }
}
__global__ void fill_kernel(double *p, i64 n, unsigned seed) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (double)(h & 0xFFFF) / 65536.0 - 0.5;
    }
}
__global__ void check_kernel(const double *X, i64 ldx, int K, const double *B, int C, const double *out, i64 ldo, i64 N, double *maxerr) {
    const i64 r = blockIdx.x < 64 ? N - 1 - blockIdx.x : ((i64)blockIdx.x * 7919 * 131) % N;
    const int c = threadIdx.x;
    if (c >= C) return;
    double s = 0;
    for (int k = 0; k < K; ++k) s += X[r + (i64)k * ldx] * B[k + (i64)c * K];
    const double e = fabs(s - out[r + (i64)c * ldo]);
    atomicMax((unsigned long long *)maxerr, __double_as_longlong(e));
}

// calibration: the plain read sweep of tune/tile_probe (32-row tiles, all K columns per workgroup) -- what THIS box streams
struct alignas(16) P2 { double v[2]; };
template <int R, int NT, int KK>
__global__ __launch_bounds__(NT) void tile_stream(const double *__restrict__ X, i64 N, double *__restrict__ sink) {
    constexpr int RP = R / 2, CG = NT / RP, CPT = KK / CG;
    const int rp = threadIdx.x % RP, cg = threadIdx.x / RP;
    const i64 ntiles = N / R;
    double acc = 0.0;
    for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const i64 i0 = t * R + 2 * rp;
        P2 x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = *reinterpret_cast<const P2 *>(X + i0 + (i64)(cg + CG * j) * N);
#pragma unroll
        for (int j = 0; j < CPT; ++j) acc += x[j].v[0] + x[j].v[1];
    }
    if (acc == 1.2345e300) sink[0] = acc;
}
void calibrate(const double *X, i64 N, double *sink, int cus) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float sum = 0;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((tile_stream<32, 512, 512>), dim3(cus * 2), dim3(512), 0, 0, X, N, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) sum += ms;
    }
    printf("calibration: plain read sweep of X %.4f ms = %.3f of peak\n", sum / 5, (double)N * 512 * 8 / (sum / 5 * 1e-3) / 8e12);
    fflush(stdout);
}

template <int NCG, int MAP, int WGT, int UU, int AUXL = 2, int STAUX = 2, int BAR = 1>
void run(const char *name, const double *X, i64 ldx, i64 N, int K, const double *B, int C, double *out, double *maxerr, int cus, int per_cu) {
    auto fn = plsk::xb_mfma4_kernel<double, 2, NCG, MAP, WGT, UU, AUXL, STAUX, BAR>;
    const int U = UU ? UU : plsk::xb4_u(2, NCG);
    const size_t lds = (size_t)plsk::xb4_kp(K, U) * plsk::xb4_stride(NCG) * 8;
    CK(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(maxerr, 0, 8));
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(fn, dim3(cus * per_cu), dim3(WGT), lds, 0, X, ldx, N, K, B, (i64)K, C, out, ldx);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    hipLaunchKernelGGL(check_kernel, dim3(256), dim3(32), 0, 0, X, ldx, K, B, C, out, ldx, N, maxerr);
    double err; CK(hipMemcpy(&err, maxerr, 8, hipMemcpyDeviceToHost));
    const double by = (double)N * K * 8 + (double)N * C * 8;
    printf("%-60s C=%2d  avg %.4f ms (best %.4f)  %.3f of peak  max err %.1e\n", name, C, sum / reps, best, by / (sum / reps * 1e-3) / 8e12, err);
    fflush(stdout);
}

int main() {
    const i64 N = 1 << 20; const int K = 512;
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    double *X, *B, *out, *maxerr;
    CK(hipMalloc(&X, N * K * 8)); CK(hipMalloc(&B, K * 32 * 8)); CK(hipMalloc(&out, N * 32 * 8)); CK(hipMalloc(&maxerr, 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, N * K, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, B, (i64)K * 32, 2u);
    CK(hipDeviceSynchronize());
    calibrate(X, N, maxerr, cus);
    i64 LDX = N, NN = N;
#define RUN(NCG, MAP, WGT, UU, STAUX, BAR) run<NCG, MAP, WGT, UU, 2, STAUX, BAR>("ncg=" #NCG " map=" #MAP " wg=" #WGT " U=" #UU " staux=" #STAUX " bar=" #BAR, X, LDX, NN, K, B, 4 * NCG, out, maxerr, cus, 1)
    // (MAP = 0: the waves of a workgroup a grid apart; BAR = 1: one barrier per round ahead of the stores; earlier states of this
    // file also timed the kernel without its stores, pacing, two row packs per lane: profiles/r5/xb4_tune.txt)
    RUN(5, 0, 1024, 0, 2, 1);   // the shipped configuration
    RUN(5, 0, 1024, 0, 2, 0);   // no barrier: the stores of the 16 waves trickle
    RUN(5, 16, 1024, 0, 2, 1);  // the workgroup's 16 tiles contiguous (4 KB per column)
    RUN(5, 0, 512, 0, 2, 1);
    RUN(5, 0, 1024, 8, 2, 1);   // eight column steps per batch
    RUN(5, 0, 1024, 2, 2, 1);
    RUN(5, 0, 1024, 0, 0, 1);   // plain stores
    NN = N - 1;                   // 31 rows in the partial last tile
    RUN(5, 0, 1024, 0, 2, 1);
    NN = N;
    RUN(2, 0, 1024, 0, 2, 1);
    RUN(8, 0, 1024, 0, 2, 1);
    calibrate(X, N, maxerr, cus);
    return 0;
}
