// syrk_stream_probe.hip -- the SYRK's off-diagonal workgroup (128 x 128 block of X^T X, eight waves of 32 x 64, operands
// streamed into LDS by LDS-DMA) with the DEPTH of the stream as a parameter: slabs of RB rows, NBUF LDS buffers, the DMA of slab
// s + NBUF - 1 issued while slab s is computed, a counted vmcnt wait for slab s.  The shipped kernel is RB = 16, NBUF = 2
// (one slab ahead).  Every workgroup computes a full block (no diagonal blocks, no X^T Y, no result stores that matter):
// TFLOP/s of executed MFMAs against 78.6.
//   hipcc --offload-arch=gfx950 -O3 -o syrk_stream_probe syrk_stream_probe.hip && ./syrk_stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef long long i64;
constexpr int TB = 128;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one slab of both panels: RB rows of 128 + 128 columns.  A wave instruction moves 1 KB = (1024 / (RB * 8)) columns.
template <int RB>
__device__ __forceinline__ void dma_slab(double *buf, const double *X, i64 ldx, int bi, int bj, i64 s, int wv, int lane) {
    constexpr int CPI = 1024 / (RB * 8);          // columns per wave instruction: 8 (RB = 16), 16 (RB = 8)
    constexpr int CH = RB / 2;                     // 16-byte chunks per column: 8 / 4
    constexpr int IPP = TB / CPI;                  // instructions per panel: 16 / 8
    constexpr int PANEL = TB * RB;
    const int lc = lane / CH, pos = lane % CH;
#pragma unroll
    for (int j = 0; j < IPP / 8; ++j) {
        const int i = wv + 8 * j;
        const int col = CPI * i + lc;
        const int key = (RB == 16) ? ((col >> 1) & 7) : ((col >> 2) & 3);
        const int q = pos ^ key;
        const i64 row = s * RB + (i64)q * 2;
        __builtin_amdgcn_global_load_lds(X + row + (i64)(bi * TB + col) * ldx, buf + i * (CPI * RB), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(X + row + (i64)(bj * TB + col) * ldx, buf + PANEL + i * (CPI * RB), 16, 0, 0);
    }
}

// CHUNK: a row split owns a CONTIGUOUS range of slabs (the workgroups in flight then read rows N / nsplit apart) instead of
// every nsplit-th slab (all workgroups within nsplit consecutive slabs of each other)
template <int RB, int NBUF, bool CHUNK = false>
__global__ __launch_bounds__(512, 4) void stream_kernel(const double *__restrict__ X, i64 ldx, i64 N, int nsplit_, double *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    double *lds = reinterpret_cast<double *>(raw);
    constexpr int PANEL = TB * RB, CS = RB;
    constexpr int LPS = 2 * (TB / (1024 / (RB * 8))) / 8;  // DMA instructions per wave and slab: 4 (RB = 16), 2 (RB = 8)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int a0 = (wv >> 1) * 32, b0 = (wv & 1) * 64;
    const int pair = blockIdx.x % 6, split = blockIdx.x / 6;
    const int bi = pair < 3 ? 0 : (pair < 5 ? 1 : 2), bj = pair < 3 ? pair + 1 : (pair < 5 ? pair - 1 : 3);
    f64x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
    const i64 nall = N / RB;
    const i64 per = (nall + nsplit_ - 1) / nsplit_;
    const i64 nslabs = CHUNK ? ((split + 1) * per < nall ? (split + 1) * per : nall) : nall;
    const i64 nsplit = CHUNK ? 1 : nsplit_;
    i64 s = CHUNK ? split * per : split;
#pragma unroll
    for (int d = 0; d < NBUF - 1; ++d)
        if (s + (i64)d * nsplit < nslabs) dma_slab<RB>(lds + d * 2 * PANEL, X, ldx, bi, bj, s + (i64)d * nsplit, wv, lane);
    int buf = 0;
    for (; s < nslabs; s += nsplit) {
        // slab s has landed once at most (NBUF - 2) later slabs' loads of this wave are outstanding
        // (the last NBUF - 2 iterations wait a little early; harmless)
        wait_vm<(NBUF - 2) * LPS>();
        __syncthreads();
        const double *As = lds + (size_t)buf * 2 * PANEL, *Bs = As + PANEL;
        {
            const int nb = (buf + NBUF - 1) % NBUF;  // the buffer read in the previous iteration: free since the barrier above
            const i64 sn = s + (i64)(NBUF - 1) * nsplit;
            if (sn < nslabs) dma_slab<RB>(lds + nb * 2 * PANEL, X, ldx, bi, bj, sn, wv, lane);
        }
#pragma unroll
        for (int kk = 0; kk < RB; kk += 4) {
            const int r = kk + lq;
            double a[2], b[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int col = a0 + 16 * m + li;
                const int key = (RB == 16) ? ((col >> 1) & 7) : ((col >> 2) & 3);
                a[m] = As[col * CS + (((r / 2) ^ key) * 2) + (r % 2)];
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = b0 + 16 * n + li;
                const int key = (RB == 16) ? ((col >> 1) & 7) : ((col >> 2) & 3);
                b[n] = Bs[col * CS + (((r / 2) ^ key) * 2) + (r % 2)];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        buf = (buf + 1) % NBUF;
    }
    double sum = 0.0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * 512 + tid] = sum;
}

template <int RB, int NBUF, bool CHUNK = false>
void run(const double *X, i64 N, double *out, int per_cu, i64 ldx = 0) {
    if (ldx == 0) ldx = N;
    const size_t ldsb = (size_t)NBUF * 2 * TB * RB * 8;
    auto k = &stream_kernel<RB, NBUF, CHUNK>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int blocks = (256 * per_cu) / 6 * 6, nsplit = blocks / 6;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), ldsb, 0, X, ldx, N, nsplit, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), ldsb, 0, X, ldx, N, nsplit, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 6.0 * 2.0 * (double)N * TB * TB;  // six full blocks
    const double tf = flops / best / 1e9;
    printf("%s ld = N + %4lld  slabs of %2d rows, %d buffers (%3zu KB of LDS, %d workgroups per CU, %d ahead)  %8.3f ms  %6.2f TFLOP/s = %.3f of 78.6   (%.2f TB/s of panel reads)\n",
           CHUNK ? "contiguous slabs per split " : "every nsplit-th slab       ", (long long)(ldx - N), RB, NBUF, ldsb / 1024, per_cu, NBUF - 1, best, tf, tf / 78.6, 6.0 * 2.0 * N * TB * 8.0 / best / 1e9);
}

__global__ void fill(double *X, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        X[i] = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;
    }
}

int main() {
    const i64 N = 1 << 20;
    const int K = 512;
    double *X, *out;
    hipMalloc(&X, (size_t)(N + 4096) * K * 8);
    hipMalloc(&out, (size_t)512 * 512 * 8);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, X, (size_t)(N + 4096) * K);
    hipDeviceSynchronize();
    run<16, 2>(X, N, out, 2);   // the shipped shape
    for (i64 pad : {16, 32, 64, 128, 512, 2048, 4096 - 16}) run<16, 2>(X, N, out, 2, N + pad);   // columns NOT a power of two apart
    run<16, 4>(X, N, out, 1, N + 32);
    run<16, 2, true>(X, N, out, 2);
    run<16, 2, true>(X, N, out, 2, N + 128);
    run<16, 4, true>(X, N, out, 1);
    run<8, 4, true>(X, N, out, 2);
    run<16, 2>(X, N, out, 1);
    run<16, 3>(X, N, out, 1);
    run<16, 4>(X, N, out, 1);
    run<8, 2>(X, N, out, 2);
    run<8, 3>(X, N, out, 2);
    run<8, 4>(X, N, out, 2);
    run<8, 4>(X, N, out, 1);
    return 0;
}
