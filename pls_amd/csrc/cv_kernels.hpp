// Batched cross-validation folds (Model::cv_LOO / cv_LSO, src/pls.cpp:469-549): every fold refits the
// model on the rows that are NOT in its test set and records the residuals of the test rows for
// 1..A components.  Upstream that is one full refit per fold (N of them for leave-one-out, 10*N for
// main's leave-some-out).  Here all folds run in ONE launch, one workgroup per fold, on K-sized data
// only: with XX = X^T X and XY = X^T Y of the whole matrix formed once,
//     XX_train = XX - X_test^T X_test        XY_train = XY - X_test^T Y_test
// so a fold needs its few test rows and the shared XX, never the N x K matrix.  The component loop is
// the reference's KERNEL_TYPE2 recurrence (tt = r^T XX r, p = XX r / tt, src/pls.cpp:422-425) with the
// downdate applied on the fly:  XX_train r = XX r - X_test^T (X_test r).
// Residuals use the scores of the test rows, u_i = x_i . r_j:  yhat_c = sum_{j<=c} u_i^(j) q_j.
#pragma once
#include "small_kernels.hpp"

namespace plsk {

// Xt[f][i*K + k] = X[idx[f*ts + i] + k*ldx],  Yt[f][i*M + m] likewise.  grid = folds*ts workgroups.
template <typename T>
__global__ __launch_bounds__(WG) void cv_gather_kernel(const T *__restrict__ X, i64 ldx,
                                                       const T *__restrict__ Y, i64 ldy, int K, int M,
                                                       const i64 *__restrict__ idx,
                                                       double *__restrict__ Xt, double *__restrict__ Yt) {
    const i64 row = idx[blockIdx.x];
    for (int k = threadIdx.x; k < K; k += WG) Xt[(i64)blockIdx.x * K + k] = (double)X[row + (i64)k * ldx];
    for (int m = threadIdx.x; m < M; m += WG) Yt[(i64)blockIdx.x * M + m] = (double)Y[row + (i64)m * ldy];
}

// per-fold workspace layout (doubles)
struct CvLayout {
    i64 xy, w, p, r, q, red, v, u, yh, total;
    __host__ __device__ CvLayout(int K, int M, int A, int ts) {
        i64 o = 0;
        xy = o; o += (i64)K * M;
        w = o; o += (i64)K * A;
        p = o; o += (i64)K * A;
        r = o; o += (i64)K * A;
        q = o; o += (i64)M * A;
        red = o; o += K + 1;
        v = o; o += K;
        u = o; o += ts;
        yh = o; o += (i64)ts * M;
        total = (o + 1) & ~(i64)1;
    }
};

// E[m][obs + c*nobs], obs = fold*ts + i, nobs = folds*ts.  Dynamic LDS: A doubles.
__global__ __launch_bounds__(UPD_THREADS) void cv_folds_kernel(
    const double *__restrict__ XX, const double *__restrict__ XY, const double *__restrict__ Xt,
    const double *__restrict__ Yt, int K, int M, int A, int ts, double *__restrict__ ws,
    double *__restrict__ E, int power_iters) {
    extern __shared__ double cs[];
    __shared__ UpdShared sh;
    __shared__ double ttred[UPD_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    const i64 nobs = (i64)gridDim.x * ts;
    const CvLayout L(K, M, A, ts);
    double *base = ws + (i64)f * L.total;
    double *XYf = base + L.xy, *Wf = base + L.w, *Pf = base + L.p, *Rf = base + L.r, *Qf = base + L.q;
    double *red1 = base + L.red, *v = base + L.v, *u = base + L.u, *yh = base + L.yh;
    const double *xt = Xt + (i64)f * ts * K, *yt = Yt + (i64)f * ts * M;

    // covariance of the training rows: XY - X_test^T Y_test
    for (int j = tid; j < K * M; j += UPD_THREADS) {
        const int k = j % K, m = j / K;
        double s = XY[j];
        for (int i = 0; i < ts; ++i) s -= xt[(i64)i * K + k] * yt[(i64)i * M + m];
        XYf[j] = s;
    }
    for (int j = tid; j < ts * M; j += UPD_THREADS) yh[j] = 0.0;
    __syncthreads();
    component_update_call(nullptr, 1, XYf, Wf, Pf, Qf, Rf, v, K, M, A, -1, 0, power_iters, 0, cs, sh);

    for (int a = 0; a < A; ++a) {
        __syncthreads();  // r_a (in v) is complete
        for (int i = wv; i < ts; i += UPD_WAVES) {  // u_i = x_i . r  (score of test row i)
            double s = 0.0;
            for (int k = lane; k < K; k += WAVE) s = fma(xt[(i64)i * K + k], v[k], s);
            s = wave_sum(s);
            if (lane == 0) u[i] = s;
        }
        __syncthreads();
        double part = 0.0;
        for (int k = tid; k < K; k += UPD_THREADS) {  // praw = XX r - X_test^T u   (XX symmetric: row k)
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = 0;
            for (; j + 4 <= K; j += 4) {
                s0 = fma(XX[k + (i64)j * K], v[j], s0);
                s1 = fma(XX[k + (i64)(j + 1) * K], v[j + 1], s1);
                s2 = fma(XX[k + (i64)(j + 2) * K], v[j + 2], s2);
                s3 = fma(XX[k + (i64)(j + 3) * K], v[j + 3], s3);
            }
            for (; j < K; ++j) s0 = fma(XX[k + (i64)j * K], v[j], s0);
            double s = (s0 + s1) + (s2 + s3);
            for (int i = 0; i < ts; ++i) s -= xt[(i64)i * K + k] * u[i];
            red1[k] = s;
            part = fma(v[k], s, part);
        }
        const double tt = block_sum<UPD_WAVES>(part, ttred);  // tt = r^T XX_train r
        if (tid == 0) red1[K] = tt;
        __syncthreads();
        component_update_call(red1, 1, XYf, Wf, Pf, Qf, Rf, v, K, M, A, a, 0, power_iters, 0, cs, sh);
        __syncthreads();
        for (int j = tid; j < ts * M; j += UPD_THREADS) {  // residuals of the test rows with a+1 components
            const int i = j / M, m = j % M;
            const double fit = fma(u[i], Qf[m + (i64)a * M], yh[j]);
            yh[j] = fit;
            E[(i64)m * nobs * A + ((i64)f * ts + i) + (i64)a * nobs] = yt[(i64)i * M + m] - fit;
        }
    }
}

// ---- the general form: one refit per fold (pls_hip.hip cv_folds_refit), for shapes the batched kernel declines ----
// out[i + c*ldo] = X[idx[i] + c*ldx]: the training rows of a fold as a matrix of their own.   grid = (ceil(n/256), <= 1024)
template <typename T>
__global__ __launch_bounds__(WG) void gather_rows_kernel(const T *__restrict__ X, i64 ldx, const i64 *__restrict__ idx, i64 n,
                                                         int cols, T *__restrict__ out, i64 ldo) {
    const i64 i = (i64)blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const i64 row = idx[i];
    for (int c = blockIdx.y; c < cols; c += gridDim.y) out[i + (i64)c * ldo] = X[row + (i64)c * ldx];
}

// Residuals of one fold's test rows under that fold's model (R: K x A, Q: M x A): u_a = x_i . r_a, then the running fit
// yhat_c = sum_{a<=c} u_a q_a in the order of cv_folds_kernel.   grid = ts workgroups; us = ts*A doubles of scratch.
__global__ __launch_bounds__(WG) void cv_refit_residuals_kernel(const double *__restrict__ xt, const double *__restrict__ yt,
                                                                const double *__restrict__ R, const double *__restrict__ Q,
                                                                int K, int M, int A, int ts, i64 fold, i64 nobs,
                                                                double *__restrict__ us, double *__restrict__ E) {
    const int i = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double *x = xt + (i64)i * K;
    double *u = us + (i64)i * A;
    for (int a = wv; a < A; a += WG / WAVE) {
        double s = 0.0;
        for (int k = lane; k < K; k += WAVE) s = fma(x[k], R[k + (i64)a * K], s);
        s = wave_sum(s);
        if (lane == 0) u[a] = s;
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += WG) {
        double fit = 0.0;
        const double y = yt[(i64)i * M + m];
        for (int a = 0; a < A; ++a) {
            fit = fma(u[a], Q[m + (i64)a * M], fit);
            E[(i64)m * nobs * A + (fold * ts + i) + (i64)a * nobs] = y - fit;
        }
    }
}

}  // namespace plsk
