// mfma_f64_probe2.hip -- the SYRK's off-diagonal inner loop (syrk_kernels.hpp, syrk8_full_body) WITHOUT its LDS-DMA and WITHOUT its
// slab barrier, in variations, to find what keeps the bare loop at 0.857 of the fp64 matrix pipe when the instruction itself
// sustains 0.936 with the same operand traffic (mfma_f64_probe.hip).  512-thread workgroups, two per CU unless noted.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe2 mfma_f64_probe2.hip && ./mfma_f64_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int TB = 128, CS = 16, PANEL = TB * CS;  // a panel: 128 columns x 16 rows of doubles, column stride 16

// MODE bit 0: linear (unswizzled) operand addresses;  bit 1: ONE k-step per loop iteration (the loop overhead 4 x as often);
// bit 2: no buffer toggling;  bit 3: operands from a fixed address (every step the same 6 reads);  bit 4: random operand
// values (full-width mantissas) instead of 1 + i 1e-7;  bit 5: a workgroup barrier per slab
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 4) void probe2(double *out, int nslabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    double *lds = reinterpret_cast<double *>(raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int a0 = ((wv >> 1) & 3) * 32, b0 = (wv & 1) * 64;
    const int fl = (MODE & 1) ? 0 : (li >> 1);
    for (int i = tid; i < 2 * 2 * PANEL; i += NT) {
        if constexpr (MODE & 16) {
            unsigned long long h = (unsigned long long)(i + 1) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            lds[i] = ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;  // (small: 4 x 10^5 accumulations stay finite)
        } else {
            lds[i] = 1.0 + i * 1e-7;
        }
    }
    __syncthreads();
    f64x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
    int buf = 0;
    for (int s = 0; s < nslabs; ++s, buf ^= ((MODE & 4) ? 0 : 1)) {
        const double *As = lds + (size_t)buf * 2 * PANEL, *Bs = As + PANEL;
        if constexpr (MODE & 32) __syncthreads();
        if constexpr (MODE & 2) {
            const int kk = (s & 3) * 4;
            const int r = ((MODE & 8) ? 0 : kk) + lq;
            const int off = (((r / 2) ^ fl) * 2) + (r % 2);
            double a[2], b[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = As[(a0 + 16 * m + li) * CS + off];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = Bs[(b0 + 16 * n + li) * CS + off];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                const int r = ((MODE & 8) ? 0 : kk) + lq;
                const int off = (((r / 2) ^ fl) * 2) + (r % 2);
                double a[2], b[4];
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = As[(a0 + 16 * m + li) * CS + off];
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = Bs[(b0 + 16 * n + li) * CS + off];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[(size_t)blockIdx.x * NT + tid] = sum;
}

template <int MODE, int NT>
void run(const char *name, int per_cu) {
    const int blocks = 256 * per_cu;
    double *out;
    hipMalloc(&out, (size_t)blocks * NT * 8);
    const size_t ldsb = 2 * 2 * PANEL * 8;  // 64 KB, as the kernel
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe2<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int nslabs = (MODE & 2) ? 40000 : 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe2<MODE, NT>), dim3(blocks), dim3(NT), ldsb, 0, out, 100);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe2<MODE, NT>), dim3(blocks), dim3(NT), ldsb, 0, out, nslabs);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mfmas = (double)blocks * (NT / 64) * nslabs * ((MODE & 2) ? 8 : 32);
    const double tf = mfmas * 2048.0 / best / 1e9;
    printf("%-86s %d x %4d threads per CU  %8.3f ms  %6.2f TFLOP/s = %.3f of 78.6\n", name, per_cu, NT, best, tf, tf / 78.6);
    hipFree(out);
}

int main() {
    run<0, 512>("the kernel's loop: swizzled panels, 4 steps per slab, two buffers", 2);
    run<2, 512>("one step per loop iteration", 2);
    run<4, 512>("one buffer", 2);
    run<8, 512>("every step the same six reads", 2);
    run<16, 512>("random operand values", 2);
    run<32, 512>("a workgroup barrier per slab", 2);
    run<48, 512>("random operand values + a barrier per slab", 2);
    run<32, 512>("a barrier per slab, ONE workgroup per CU", 1);
    run<32, 1024>("a barrier per slab, one 1024-thread workgroup", 1);
    run<0, 512>("the kernel's loop, ONE workgroup per CU (two waves per SIMD)", 1);
    run<0, 256>("the kernel's loop in 256-thread workgroups", 4);
    run<0, 1024>("the kernel's loop in one 1024-thread workgroup", 1);
    return 0;
}
