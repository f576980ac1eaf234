/*
 * pls_oracle.c -- CPU restatement of the tjhladish/PLS fit/predict hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (pls_amd/, include/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or
 * numeric assertions for this path (reference tests/ only checks packaging, see
 * SURVEY.md section 4), and the reference itself cannot be built here because its
 * Eigen dependency (gitlab libeigen/eigen, required >= 3.4.90, no pinned commit;
 * reference .gitmodules:1-3, CMakeLists.txt:11-21) is an empty un-vendored
 * submodule that is absent from this image.  This file therefore restates the
 * published algorithm (Dayal & MacGregor 1997, "modified kernel algorithm" #1/#2)
 * following the reference's own call sites line by line, and is cross-checked in
 * tests/ against two independent implementations available in the image
 * (oracle/pls_oracle.py in numpy, and scikit-learn's NIPALS PLSRegression).
 *
 * Conventions: all matrices column-major with explicit leading dimension
 * (the reference's Eigen::MatrixXd default storage order, include/PLS/pls.h:22-23),
 * IEEE fp64 (include/PLS/pls.h:22).  The reference carries w,p,q,r,t in
 * std::complex<double> with zero imaginary parts (src/pls.cpp:401-402); this
 * restatement is purely real.
 *
 * Build: see oracle/Makefile (liboracle.so: -O3 single thread, mirrors the
 * reference's Release build which has no OpenMP and no -march;
 * liboracle_omp.so: -O3 -march=x86-64-v3 -fopenmp, the generous all-core baseline).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

typedef int64_t i64;

/* ------------------------------------------------------------------------- */
/* Compensated arithmetic (optional)                                           */
/* ------------------------------------------------------------------------- */
/*
 * The restatement's sums run in index order like the reference's expressions.  Over N = 2^20 rows that order
 * alone carries an error of up to N*eps/2 ~ 1e-10 per product, the size of the north star's tolerance on B --
 * and the reference's own Eigen kernels sum in yet another (SIMD-blocked) order.  To tell the summation error of
 * a CPU route from a disagreement of the algorithm, every sum can be switched to Ogita-Rump-Oishi compensated
 * form (TwoProduct by fma, TwoSum): the result is what twice the working precision would give, rounded once.
 * Same operation sequence, same fp64 storage of every vector; only the rounding inside the sums goes away.
 * Off by default (the timed CPU baseline and the fixtures use the plain sums).
 */
static int g_compensated = 0;
ORACLE_API void oracle_set_compensated(int on) { g_compensated = on; }

static inline void acc2(double a, double b, double *s, double *c) { /* (s, c) += a*b, error-free */
    const double p = a * b;
    const double pe = fma(a, b, -p);
    const double t = *s + p;
    const double bb = t - *s;
    const double e = (*s - (t - bb)) + (p - bb);
    *s = t;
    *c += e + pe;
}
static double dot2(const double *a, const double *b, i64 n) {
    double s = 0.0, c = 0.0;
    for (i64 i = 0; i < n; ++i) acc2(a[i], b[i], &s, &c);
    return s + c;
}

/* ------------------------------------------------------------------------- */
/* Streaming products on a column-major N x K matrix                          */
/* ------------------------------------------------------------------------- */

/* The OpenMP build (liboracle_omp.so: the all-core CPU baseline, and the full-size checker of the GPU tests) partitions
 * EVERY sweep of X by ROW BLOCKS -- thread id owns rows [N id / nt, N (id + 1) / nt), the block boundaries rounded to 512
 * rows = one 4 KB page of a column -- the generator included: the thread that first touches a page of X (oracle_synth_x)
 * is the one that reads it in every later product, so on a multi-socket host the pages are local to their readers
 * (first-touch placement; the same partition as the GPU path's row shards).  Column sums then meet as per-thread partials
 * added in thread order (compensated mode: (sum, error) pairs combined error-free).  One thread = one block = the
 * reference's index-order sums; the single-threaded liboracle.so has none of this. */
#ifdef _OPENMP
static inline void row_block(i64 N, int nt, int id, i64 *a, i64 *b) {
    const i64 pages = (N + 511) / 512;
    *a = (pages * id / nt) * 512;
    *b = (pages * (id + 1) / nt) * 512;
    if (*a > N) *a = N;
    if (*b > N) *b = N;
}
/* out[j] = sum over the rows of col_j(A)[i] * col_m(B)[i] for the K x M pairs (xty) -- B = Y (M columns) or a vector */
static void omp_colsums(const double *X, i64 ldx, const double *Y, i64 ldy, i64 N, i64 K, i64 M, double *XY) {
    const int nt = omp_get_max_threads();
    double *ps = (double *)calloc((size_t)nt * (size_t)(K * M) * 2, sizeof(double)); /* [thread][sum K*M | err K*M] */
    if (!ps) { /* no room for the per-thread partials (~1 GB at K = 6000, M = 600, 16 threads): the serial loop */
        for (i64 k = 0; k < K; ++k)
            for (i64 m = 0; m < M; ++m) {
                double ss = 0.0, cc = 0.0;
                if (g_compensated) {
                    for (i64 i = 0; i < N; ++i) acc2(X[k * ldx + i], Y[m * ldy + i], &ss, &cc);
                } else {
                    for (i64 i = 0; i < N; ++i) ss += X[k * ldx + i] * Y[m * ldy + i];
                }
                XY[k + m * K] = ss + cc;
            }
        return;
    }
#pragma omp parallel num_threads(nt)
    {
        const int id = omp_get_thread_num();
        i64 a, b;
        row_block(N, omp_get_num_threads(), id, &a, &b);
        double *s = ps + (size_t)id * (size_t)(K * M) * 2, *c = s + K * M;
        for (i64 k = 0; k < K; ++k) {
            const double *xk = X + k * ldx;
            for (i64 m = 0; m < M; ++m) {
                const double *ym = Y + m * ldy;
                double ss = 0.0, cc = 0.0;
                if (g_compensated) {
                    for (i64 i = a; i < b; ++i) acc2(xk[i], ym[i], &ss, &cc);
                } else {
                    for (i64 i = a; i < b; ++i) ss += xk[i] * ym[i];
                }
                s[k + m * K] = ss;
                c[k + m * K] = cc;
            }
        }
    }
    for (i64 j = 0; j < K * M; ++j) { /* thread order; compensated: two-sum of the partial sums, errors added */
        double ss = 0.0, cc = 0.0;
        for (int t = 0; t < nt; ++t) {
            const double v = ps[(size_t)t * (size_t)(K * M) * 2 + j], e = ps[(size_t)t * (size_t)(K * M) * 2 + K * M + j];
            if (g_compensated) {
                const double tt = ss + v, bb = tt - ss;
                cc += ((ss - (tt - bb)) + (v - bb)) + e;
                ss = tt;
            } else {
                ss += v;
            }
        }
        XY[j] = ss + cc;
    }
    free(ps);
}
#endif

/* out(K x M) = X^T Y.  Reference: `Mat2D XY = X.transpose() * Y;` src/pls.cpp:396 */
ORACLE_API void oracle_xty(const double *X, i64 ldx, const double *Y, i64 ldy,
                           i64 N, i64 K, i64 M, double *XY /* K x M, ld K */) {
#ifdef _OPENMP
    if (omp_get_max_threads() > 1) {
        omp_colsums(X, ldx, Y, ldy, N, K, M, XY);
        return;
    }
#endif
    for (i64 k = 0; k < K; ++k) {
        const double *xk = X + k * ldx;
        for (i64 m = 0; m < M; ++m) {
            const double *ym = Y + m * ldy;
            if (g_compensated) {
                XY[k + m * K] = dot2(xk, ym, N);
                continue;
            }
            double s = 0.0;
            for (i64 i = 0; i < N; ++i) s += xk[i] * ym[i];
            XY[k + m * K] = s;
        }
    }
}

/* t = X v.  Reference: `t = X*r;` src/pls.cpp:419 (column-major axpy sweep). */
ORACLE_API void oracle_xv(const double *X, i64 ldx, i64 N, i64 K, const double *v, double *t) {
    if (g_compensated) { /* per-row compensated accumulation over the K columns */
        double *lo = (double *)calloc((size_t)(N > 0 ? N : 1), sizeof(double));
#pragma omp parallel
        {
#ifdef _OPENMP
            const int nt = omp_get_num_threads(), id = omp_get_thread_num();
            i64 a, b;
            row_block(N, nt, id, &a, &b);
#else
            const i64 a = 0, b = N;
#endif
            for (i64 i = a; i < b; ++i) t[i] = 0.0;
            for (i64 k = 0; k < K; ++k) {
                const double *xk = X + k * ldx;
                const double vk = v[k];
                for (i64 i = a; i < b; ++i) acc2(xk[i], vk, &t[i], &lo[i]);
            }
            for (i64 i = a; i < b; ++i) t[i] += lo[i];
        }
        free(lo);
        return;
    }
#ifdef _OPENMP
#pragma omp parallel
    {
        int nt = omp_get_num_threads(), id = omp_get_thread_num();
        i64 lo, hi;
        row_block(N, nt, id, &lo, &hi);
        for (i64 i = lo; i < hi; ++i) t[i] = 0.0;
        for (i64 k = 0; k < K; ++k) {
            const double *xk = X + k * ldx;
            const double vk = v[k];
            for (i64 i = lo; i < hi; ++i) t[i] += xk[i] * vk;
        }
    }
#else
    for (i64 i = 0; i < N; ++i) t[i] = 0.0;
    for (i64 k = 0; k < K; ++k) {
        const double *xk = X + k * ldx;
        const double vk = v[k];
        for (i64 i = 0; i < N; ++i) t[i] += xk[i] * vk;
    }
#endif
}

/* p = X^T t.  Reference: `p.noalias() = (X.transpose()*t);` src/pls.cpp:421 */
ORACLE_API void oracle_xtv(const double *X, i64 ldx, i64 N, i64 K, const double *t, double *p) {
#ifdef _OPENMP
    if (omp_get_max_threads() > 1) {
        omp_colsums(X, ldx, t, N, N, K, 1, p);
        return;
    }
#endif
    for (i64 k = 0; k < K; ++k) {
        const double *xk = X + k * ldx;
        if (g_compensated) {
            p[k] = dot2(xk, t, N);
            continue;
        }
        double s = 0.0;
        for (i64 i = 0; i < N; ++i) s += xk[i] * t[i];
        p[k] = s;
    }
}

static double dot(const double *a, const double *b, i64 n) {
    if (g_compensated) return dot2(a, b, n);
    double s = 0.0;
    for (i64 i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* ------------------------------------------------------------------------- */
/* Dominant eigenvector of the symmetric M x M matrix S^T S                   */
/* ------------------------------------------------------------------------- */

/*
 * Reference: src/pls.cpp:406-408 builds Eigen::EigenSolver on XY^T XY, and
 * find_dominant_ev (src/pls.cpp:113-129) picks the eigenvalue of largest |lambda|
 * whose imaginary part is exactly 0; dominant_eigenvector (:138-141) returns that
 * (unit-norm, arbitrary-sign) eigenvector.  XY^T XY is symmetric positive
 * semi-definite, so every eigenvalue is real and any accurate symmetric solver
 * gives the same vector up to sign.  Here: cyclic Jacobi.  Sign convention (the
 * reference leaves it implementation-defined): the entry of largest magnitude is
 * made positive (lowest index wins ties).
 */
ORACLE_API void oracle_dominant_eigvec_sts(const double *S, i64 K, i64 M, double *q /* M */) {
    double *G = (double *)malloc(sizeof(double) * M * M);
    double *V = (double *)malloc(sizeof(double) * M * M);
    for (i64 a = 0; a < M; ++a)
        for (i64 b = 0; b < M; ++b) {
            G[a + b * M] = dot(S + a * K, S + b * K, K);
            V[a + b * M] = (a == b) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (i64 a = 0; a < M; ++a)
            for (i64 b = 0; b < M; ++b) {
                if (a != b) off += G[a + b * M] * G[a + b * M];
                else diag += G[a + b * M] * G[a + b * M];
            }
        if (off == 0.0 || off <= 1e-32 * diag) break;
        for (i64 pp = 0; pp < M - 1; ++pp)
            for (i64 qq = pp + 1; qq < M; ++qq) {
                double apq = G[pp + qq * M];
                if (apq == 0.0) continue;
                double app = G[pp + pp * M], aqq = G[qq + qq * M];
                double theta = (aqq - app) / (2.0 * apq);
                double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (i64 k = 0; k < M; ++k) { /* G <- G J */
                    double gkp = G[k + pp * M], gkq = G[k + qq * M];
                    G[k + pp * M] = c * gkp - s * gkq;
                    G[k + qq * M] = s * gkp + c * gkq;
                }
                for (i64 k = 0; k < M; ++k) { /* G <- J^T G */
                    double gpk = G[pp + k * M], gqk = G[qq + k * M];
                    G[pp + k * M] = c * gpk - s * gqk;
                    G[qq + k * M] = s * gpk + c * gqk;
                }
                for (i64 k = 0; k < M; ++k) {
                    double vkp = V[k + pp * M], vkq = V[k + qq * M];
                    V[k + pp * M] = c * vkp - s * vkq;
                    V[k + qq * M] = s * vkp + c * vkq;
                }
            }
    }
    i64 best = 0;
    double bm = -1.0;
    for (i64 a = 0; a < M; ++a)
        if (fabs(G[a + a * M]) > bm) { bm = fabs(G[a + a * M]); best = a; }
    double nrm = sqrt(dot(V + best * M, V + best * M, M));
    i64 big = 0;
    for (i64 a = 1; a < M; ++a)
        if (fabs(V[a + best * M]) > fabs(V[big + best * M])) big = a;
    double sgn = (V[big + best * M] < 0.0) ? -1.0 : 1.0;
    for (i64 a = 0; a < M; ++a) q[a] = sgn * V[a + best * M] / nrm;
    free(G);
    free(V);
}

/* ------------------------------------------------------------------------- */
/* The fit: Model::plsr, src/pls.cpp:390-437                                  */
/* ------------------------------------------------------------------------- */

/* weight vector of one component from the (deflated) K x M matrix XY:
 * src/pls.cpp:403-411.  M==1: w = XY (:404); else w = XY q (:406-408); w /= sqrt(w^T w) (:411). */
static void direction(const double *XY, i64 K, i64 M, double *w, double *qtmp) {
    if (M == 1) {
        memcpy(w, XY, sizeof(double) * K);
    } else {
        oracle_dominant_eigvec_sts(XY, K, M, qtmp);
        for (i64 k = 0; k < K; ++k) {
            double s = 0.0;
            for (i64 m = 0; m < M; ++m) s += XY[k + m * K] * qtmp[m];
            w[k] = s;
        }
    }
    double nrm = sqrt(dot(w, w, K));
    for (i64 k = 0; k < K; ++k) w[k] /= nrm;
}

/* r = w - sum_{j<i} (P[:,j]^T w) R[:,j]   src/pls.cpp:412-416.
 * NB the inner products are against the ORIGINAL w, not the running r. */
static void rotate(const double *W_i, const double *P, const double *R, i64 K, i64 i, double *r) {
    memcpy(r, W_i, sizeof(double) * K);
    for (i64 j = 0; j < i; ++j) {
        double c = dot(P + j * K, W_i, K);
        const double *rj = R + j * K;
        for (i64 k = 0; k < K; ++k) r[k] -= c * rj[k];
    }
}

/* Y-loading and the reference's own deflation: src/pls.cpp:428-429
 * q = (r^T XY)^T / tt ;  XY -= p q^T tt */
static void yload_and_deflate_xy(double *XY, i64 K, i64 M, const double *r, const double *p,
                                 double tt, double *q) {
    for (i64 m = 0; m < M; ++m) q[m] = dot(r, XY + m * K, K) / tt;
    for (i64 m = 0; m < M; ++m)
        for (i64 k = 0; k < K; ++k) XY[k + m * K] -= (p[k] * q[m]) * tt;
}

/*
 * method 0 = KERNEL_TYPE1 (t = X r on the undeflated X, p = X^T t / tt; T filled),
 * method 1 = KERNEL_TYPE2 (XX = X^T X once, tt = r^T XX r, p = XX r / tt; T untouched).
 * Outputs (column-major): W,P,R K x A; Q M x A; T N x A (ld = ldt, may be NULL for method 1).
 * Returns 0, or 1 on bad arguments.  A > rank(X) gives inf/NaN columns exactly as the
 * reference does (division by tt ~ 0, src/pls.cpp:427-428).
 */

/* Storage emulation for BASELINE config 4 (fp32 storage of X, Y, T with fp64 arithmetic -- the reference has
 * no such mode, float_type being double, include/PLS/pls.h:22): when switched on, every score vector is rounded
 * to fp32 as soon as it is formed (it is STORED in fp32 and re-read by the loading product) and the NIPALS form
 * rounds the deflated matrix to fp32 after every rank-1 update.  All sums stay fp64.  With fp32-representable
 * inputs this is the arithmetic of a correct fp32-storage implementation, so two such routes measure how far the
 * storage rounding alone moves each component (the per-component conditioning the GPU tests scale by). */
static int g_f32_storage = 0;
ORACLE_API void oracle_set_f32_storage(int on) { g_f32_storage = on; }
static void round_f32(double *v, i64 n) {
    for (i64 i = 0; i < n; ++i) v[i] = (double)(float)v[i];
}

ORACLE_API int oracle_plsr(const double *X, i64 ldx, const double *Y, i64 ldy, i64 N, i64 K,
                           i64 M, i64 A, int method, double *W, double *P, double *Q, double *R,
                           double *T, i64 ldt) {
    if (N <= 0 || K <= 0 || M <= 0 || A <= 0 || A > K || ldx < N || ldy < N) return 1;
    if (method == 0 && (!T || ldt < N)) return 1;
    double *XY = (double *)malloc(sizeof(double) * K * M);
    double *XX = NULL;
    double *p = (double *)malloc(sizeof(double) * K);
    double *q = (double *)malloc(sizeof(double) * M);
    double *qe = (double *)malloc(sizeof(double) * M);
    double *t = (double *)malloc(sizeof(double) * N);

    oracle_xty(X, ldx, Y, ldy, N, K, M, XY); /* :396 */
    if (method == 1) {                         /* :398 */
        XX = (double *)malloc(sizeof(double) * K * K);
        oracle_xty(X, ldx, X, ldx, N, K, K, XX);
    }
    for (i64 i = 0; i < A; ++i) { /* :400 */
        double *w = W + i * K, *r = R + i * K;
        direction(XY, K, M, w, qe); /* :403-411 */
        rotate(w, P, R, K, i, r);   /* :412-416 */
        double tt;
        if (method == 0) {
            oracle_xv(X, ldx, N, K, r, t);  /* :419 */
            if (g_f32_storage) round_f32(t, N);
            tt = dot(t, t, N);              /* :420 */
            oracle_xtv(X, ldx, N, K, t, p); /* :421 */
            memcpy(T + i * ldt, t, sizeof(double) * N); /* :434 */
        } else {
            for (i64 k = 0; k < K; ++k) p[k] = dot(XX + k * K, r, K); /* :424, XX symmetric */
            tt = dot(r, p, K);                                        /* :423 */
        }
        for (i64 k = 0; k < K; ++k) p[k] /= tt;      /* :427 */
        yload_and_deflate_xy(XY, K, M, r, p, tt, q); /* :428-429 */
        memcpy(P + i * K, p, sizeof(double) * K);    /* :431 */
        memcpy(Q + i * M, q, sizeof(double) * M);    /* :432 */
    }
    free(XY); free(XX); free(p); free(q); free(qe); free(t);
    return 0;
}

/*
 * The north-star formulation of the same fit: classical NIPALS bookkeeping with an
 * explicit rank-1 deflation X <- X - t p^T after every component (the reference has no
 * such line; it deflates only the K x M matrix XY, src/pls.cpp:429).  Mathematically
 * identical in W,P,Q,R,T,B: t = X_a w_a = X r_a, X_a^T t = X^T t,
 * X_a^T Y = XY - sum_j p_j q_j^T tt_j.  Written independently of oracle_plsr so that the
 * two cross-check each other.  X is copied; the caller's matrix is not modified.
 */
ORACLE_API int oracle_plsr_nipals(const double *X, i64 ldx, const double *Y, i64 ldy, i64 N,
                                  i64 K, i64 M, i64 A, double *W, double *P, double *Q,
                                  double *R, double *T, i64 ldt) {
    if (N <= 0 || K <= 0 || M <= 0 || A <= 0 || A > K || ldx < N || ldy < N || !T || ldt < N)
        return 1;
    double *Xd = (double *)malloc(sizeof(double) * N * K);
    double *S = (double *)malloc(sizeof(double) * K * M);
    double *qe = (double *)malloc(sizeof(double) * M);
    for (i64 k = 0; k < K; ++k) memcpy(Xd + k * N, X + k * ldx, sizeof(double) * N);
    for (i64 a = 0; a < A; ++a) {
        double *w = W + a * K, *p = P + a * K, *q = Q + a * M, *r = R + a * K, *t = T + a * ldt;
        oracle_xty(Xd, N, Y, ldy, N, K, M, S); /* covariance of the DEFLATED X with Y */
        direction(S, K, M, w, qe);
        oracle_xv(Xd, N, N, K, w, t);  /* score from the deflated X and the raw weight */
        if (g_f32_storage) round_f32(t, N);
        double tt = dot(t, t, N);
        oracle_xtv(Xd, N, N, K, t, p); /* loading */
        for (i64 k = 0; k < K; ++k) p[k] /= tt;
        for (i64 m = 0; m < M; ++m) q[m] = dot(Y + m * ldy, t, N) / tt; /* q = Y^T t / tt */
        for (i64 k = 0; k < K; ++k) { /* X <- X - t p^T */
            double *xk = Xd + k * N;
            const double pk = p[k];
            if (g_f32_storage)
                for (i64 i = 0; i < N; ++i) xk[i] = (double)(float)fma(-t[i], pk, xk[i]);
            else
                for (i64 i = 0; i < N; ++i) xk[i] -= t[i] * pk;
        }
        rotate(w, P, R, K, a, r);
    }
    free(Xd); free(S); free(qe);
    return 0;
}

/* B = R[:, :c] Q[:, :c]^T   (K x M).  Reference: Model::coefficients, src/pls.cpp:444-447 */
ORACLE_API void oracle_coefficients(const double *R, const double *Q, i64 K, i64 M, i64 c,
                                    double *B) {
    for (i64 m = 0; m < M; ++m)
        for (i64 k = 0; k < K; ++k) {
            double s = 0.0;
            for (i64 j = 0; j < c; ++j) s += R[k + j * K] * Q[m + j * M];
            B[k + m * K] = s;
        }
}

/* out(N x C) = X Bm(K x C).  Reference: Model::fitted_values `X_new*coefficients(comp).real()`
 * src/pls.cpp:449-451 and Model::scores `X_new * R.leftCols(comp)` :439-442. */
ORACLE_API void oracle_xb(const double *X, i64 ldx, i64 N, i64 K, const double *Bm, i64 ldb,
                          i64 C, double *out, i64 ldo) {
    for (i64 c = 0; c < C; ++c) oracle_xv(X, ldx, N, K, Bm + c * ldb, out + c * ldo);
}

/* ------------------------------------------------------------------------- */
/* Pre-processing used by the CSV configs: src/pls.cpp:69-111, src/main.cpp:24-25 */
/* ------------------------------------------------------------------------- */

/* column mean; SST = sum (x-mean)^2 (:69-73, zero if N<2); sd = sqrt(SST/(N-1)) (:79-83);
 * z = (x - mean)/sd (:93-105).  The reference builds a zero-guarded local_sd (:94,100) but
 * divides by the UNGUARDED stdev (:103): a constant column yields NaN.  Restated faithfully. */
ORACLE_API void oracle_colwise_z_scores(const double *X, i64 ldx, i64 N, i64 K, double *Z,
                                        i64 ldz, double *mean_out, double *sd_out) {
    for (i64 k = 0; k < K; ++k) {
        const double *xk = X + k * ldx;
        double s = 0.0;
        for (i64 i = 0; i < N; ++i) s += xk[i];
        double mean = s / (double)N, sst = 0.0;
        if (N >= 2)
            for (i64 i = 0; i < N; ++i) sst += (xk[i] - mean) * (xk[i] - mean);
        double sd = sqrt(sst / (double)(N - 1));
        for (i64 i = 0; i < N; ++i) Z[i + k * ldz] = (xk[i] - mean) / sd;
        if (mean_out) mean_out[k] = mean;
        if (sd_out) sd_out[k] = sd;
    }
}

/* ------------------------------------------------------------------------- */
/* Row-sharded fit (KERNEL_TYPE1) with an injected all-reduce: the N>1 algorithm   */
/* of SURVEY.md section 8(e), used by the world_size-2 gloo tests to check the   */
/* product's partitioning + reducer host logic without a GPU.                    */
/* ------------------------------------------------------------------------- */

typedef int (*oracle_allreduce_fn)(void *user, double *buf, i64 count);

ORACLE_API int oracle_plsr_sharded(const double *X, i64 ldx, const double *Y, i64 ldy, i64 Nloc,
                                   i64 K, i64 M, i64 A, oracle_allreduce_fn allreduce,
                                   void *user, double *W, double *P, double *Q, double *R,
                                   double *T, i64 ldt) {
    if (Nloc < 0 || K <= 0 || M <= 0 || A <= 0 || A > K) return 1;
    double *XY = (double *)malloc(sizeof(double) * K * M);
    double *pk = (double *)malloc(sizeof(double) * (K + 1)); /* packed [p(K), tt] */
    double *q = (double *)malloc(sizeof(double) * M);
    double *qe = (double *)malloc(sizeof(double) * M);
    int rc = 0;
    oracle_xty(X, ldx, Y, ldy, Nloc, K, M, XY);
    if (allreduce && (rc = allreduce(user, XY, K * M))) goto done;
    for (i64 i = 0; i < A; ++i) {
        double *w = W + i * K, *r = R + i * K, *t = T + i * ldt;
        direction(XY, K, M, w, qe); /* replicated: identical inputs on every rank */
        rotate(w, P, R, K, i, r);
        oracle_xv(X, ldx, Nloc, K, r, t); /* t stays sharded like X */
        oracle_xtv(X, ldx, Nloc, K, t, pk);
        pk[K] = dot(t, t, Nloc);
        if (allreduce && (rc = allreduce(user, pk, K + 1))) goto done;
        double tt = pk[K];
        for (i64 k = 0; k < K; ++k) pk[k] /= tt;
        yload_and_deflate_xy(XY, K, M, r, pk, tt, q);
        memcpy(P + i * K, pk, sizeof(double) * K);
        memcpy(Q + i * M, q, sizeof(double) * M);
    }
done:
    free(XY); free(pk); free(q); free(qe);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* Synthetic inputs: counter-based, libm-free, exact in   */
/* fp64, so host and device generators are bit-identical.  Spec in DESIGN.md.      */
/* ------------------------------------------------------------------------- */

static inline uint64_t mix64(uint64_t z) { /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u24(uint64_t stream, uint64_t idx) { /* dyadic uniform in [-1,1) */
    uint64_t h = mix64(stream ^ idx);
    return (double)((int64_t)(h >> 40) - 8388608) * (1.0 / 8388608.0);
}
#define SYN_F 8
static const double SYN_LTAB[5] = {-1.0, -0.5, 0.0, 0.5, 1.0};
/* Noise amplitude of column k: 2^-(h mod 4) * (8 + (h/4) mod 8)/32 in [1/32, 15/32] (a 4-bit dyadic: every product
 * stays exact).  Column-dependent ON PURPOSE: with one amplitude for all columns the noise covariance of a tall matrix
 * is isotropic up to O(sqrt(K/N)), the Krylov sequence PLS builds after the 8 latent factors collapses, and at
 * N = 2^20 components beyond the 14th are rounding noise in every fp64 implementation (measured: two compensated CPU
 * formulations disagree by O(1) on them).  With a 15:1 spread of amplitudes all 20 (50) components of the BASELINE
 * configs are determined to ~1e-11. */
static inline double synth_noise_amp(uint64_t sA, uint64_t k) {
    const uint64_t h = mix64(sA ^ k);
    return ldexp((double)(8 + ((h >> 2) & 7)) / 32.0, -(int)(h & 3));
}

/* rows [row0, row0+nrows) of the global N x K matrix into X (ld = ldx) */
ORACLE_API void oracle_synth_x(double *X, i64 ldx, i64 row0, i64 nrows, i64 K, uint64_t seed) {
    const uint64_t sE = mix64(seed), sZ = mix64(seed + 1), sL = mix64(seed + 2), sA = mix64(seed + 5);
    /* (row blocks per thread in the OpenMP build: the pages of a block are first touched by the thread that reads them
     * in every later product -- see omp_colsums) */
#pragma omp parallel
    {
        i64 a = 0, b = nrows;
#ifdef _OPENMP
        row_block(nrows, omp_get_num_threads(), omp_get_thread_num(), &a, &b);
#endif
        for (i64 k = 0; k < K; ++k) {
            double L[SYN_F];
            for (int f = 0; f < SYN_F; ++f) L[f] = SYN_LTAB[mix64(sL ^ (uint64_t)(k * SYN_F + f)) % 5];
            const double amp = synth_noise_amp(sA, (uint64_t)k);
            for (i64 ii = a; ii < b; ++ii) {
                const uint64_t i = (uint64_t)(row0 + ii);
                double s = amp * u24(sE, i * (uint64_t)K + (uint64_t)k);
                for (int f = 0; f < SYN_F; ++f) s += u24(sZ, i * SYN_F + f) * L[f];
                X[ii + k * ldx] = s;
            }
        }
    }
}
ORACLE_API void oracle_synth_y(double *Y, i64 ldy, i64 row0, i64 nrows, i64 M, uint64_t seed) {
    const uint64_t sZ = mix64(seed + 1), sC = mix64(seed + 3), sN = mix64(seed + 4);
    for (i64 j = 0; j < M; ++j) {
        double C[SYN_F];
        for (int f = 0; f < SYN_F; ++f)
            C[f] = (double)((int)(mix64(sC ^ (uint64_t)(j * SYN_F + f)) % 3) - 1);
        const double scale = ldexp(1.0, -(int)(j % 16));
        for (i64 ii = 0; ii < nrows; ++ii) {
            const uint64_t i = (uint64_t)(row0 + ii);
            double s = 0.0;
            for (int f = 0; f < SYN_F; ++f) s += u24(sZ, i * SYN_F + f) * C[f];
            Y[ii + j * ldy] = scale * s + 0.125 * u24(sN, i * (uint64_t)M + (uint64_t)j);
        }
    }
}

/* (the all-core baseline sizes its team to the CPU share of the box: a cgroup quota is invisible to omp_get_max_threads) */
ORACLE_API void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

ORACLE_API int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
