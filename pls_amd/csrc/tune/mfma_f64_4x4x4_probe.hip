// mfma_f64_4x4x4_probe.hip -- lane layout and rate of v_mfma_f64_4x4x4_4b_f64 beside v_mfma_f64_16x16x4_f64 on gfx950
// (measurement tool: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_4x4x4_probe mfma_f64_4x4x4_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double *a, const double *b, double *d) {
    const int l = threadIdx.x;
    double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
    d[l] = r;
}

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(double *out, int iters, double s) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-9, b = s;
    if (KIND == 0) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u], 0, 0, 0);
        }
        double t = 0;
        for (int u = 0; u < 8; ++u) t += acc[u];
        out[blockIdx.x * 256 + l] = t;
    } else if (KIND == 1) {
        d4 acc[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
        double t = 0;
        for (int u = 0; u < 4; ++u) t += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
        out[blockIdx.x * 256 + l] = t;
    } else {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = fma(a, b, acc[u]);
            asm volatile("" : "+v"(a));
        }
        double t = 0;
        for (int u = 0; u < 8; ++u) t += acc[u];
        out[blockIdx.x * 256 + l] = t;
    }
}

int main() {
    double ha[64], hb[64], hd[64];
    srand(7);
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + rand() % 97; hb[i] = 1 + rand() % 89; }
    double *a, *b, *d;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&d, 512);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
    // lane = 16 x + 4 y + z; try every assignment of (block, row/col, k) to the digits
    const int perm[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
    auto lane = [&](const int *p, int u, int v, int w) { int dg[3]; dg[p[0]] = u; dg[p[1]] = v; dg[p[2]] = w; return 16 * dg[0] + 4 * dg[1] + dg[2]; };
    int found = 0;
    for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += ha[lane(perm[pa], blk, i, k)] * hb[lane(perm[pb], blk, k, j)];
            // D digits: (block, i, j)
            if (s != hd[lane(perm[pd], blk, i, j)]) ok = false;
        }
        if (ok) {
            ++found;
            printf("layout: A(block,i,k)->digits %d%d%d  B(block,k,j)->digits %d%d%d  D(block,i,j)->digits %d%d%d   (digit 0 = lane/16, 1 = (lane/4)%%4, 2 = lane%%4)\n",
                   perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2]);
        }
    }
    if (!found) { printf("no digit layout fits; D:"); for (int i = 0; i < 64; ++i) printf(" %g", hd[i]); printf("\n"); }
    // rate
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus * 8, iters = 20000;
    double *out; hipMalloc(&out, (size_t)grid * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (kind == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-3);
            if (kind == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-3);
            if (kind == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-3);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double per = kind == 0 ? 8.0 * 256 : kind == 1 ? 4.0 * 1024 : 8.0 * 64;   // FMAs per wave per iteration
            const double fma = (double)grid * 4 * iters * per;
            if (rep == 2) printf("%s: %.3f ms, %.2f TFLOP/s fp64\n", kind == 0 ? "mfma 4x4x4_4b" : kind == 1 ? "mfma 16x16x4" : "v_fma_f64", ms, 2 * fma / ms * 1e-9);
        }
    }
    return 0;
}
