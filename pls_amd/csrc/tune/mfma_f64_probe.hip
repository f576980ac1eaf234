// mfma_f64_probe.hip -- what rate does v_mfma_f64_16x16x4_f64 sustain on MI355X with NOTHING else going on?
// Every wave runs a loop of NACC independent accumulations (operands in registers, no memory, no LDS, no barrier); grid =
// 256 CUs x W waves per SIMD.  Prints TFLOP/s against the 78.6 TF datasheet figure.  A second form adds the SYRK's operand
// traffic: 6 ds_read_b64 per 8 MFMAs from LDS (conflict-free), still no barrier and no global memory.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256, 4) void probe(double *out, int iters) {
    __shared__ double sm[8 * 64 * 4];
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int lane = threadIdx.x & 63;
    double a[2] = {1.0 + lane * 1e-3, 2.0 - lane * 1e-3}, b[4] = {0.5, 0.25 + lane * 1e-4, 0.125, 1.0};
    if (LDS) {
        for (int i = threadIdx.x; i < 8 * 64 * 4; i += 256) sm[i] = 1.0 + i * 1e-6;
        __syncthreads();
    }
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            const int base = ((it & 3) * 6) * 64 + lane;
#pragma unroll
            for (int q = 0; q < 2; ++q) a[q] = sm[base + q * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q) b[q] = sm[base + (2 + q) * 64];
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 1], b[i & 3], acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int wps) {
    const int blocks = 256 * wps;  // 256-thread blocks = 4 waves = one wave per SIMD each: wps blocks per CU
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 200);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("%-44s waves/SIMD %d  %8.3f ms  %6.2f TFLOP/s = %.3f of 78.6\n", name, wps, best, flops / best / 1e9, flops / best / 1e9 / 78.6);
    hipFree(out);
}

int main() {
    for (int wps : {1, 2, 4}) run<8, false>("8 independent MFMAs per iteration, registers", wps);
    for (int wps : {1, 2, 4}) run<8, true>("+ 6 ds_read_b64 per 8 MFMAs (the SYRK's ratio)", wps);
    for (int wps : {2, 4}) run<4, true>("+ 6 ds_read_b64 per 4 MFMAs", wps);
    return 0;
}
