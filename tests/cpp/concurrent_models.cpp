// Two host threads fit their own PLS::Models at the same time (the reference's Models share nothing, include/PLS/pls.h
// 184-266 upstream: neither do these -- every thread works on a device context of its own).  Each thread builds `rounds`
// Models on its own data set, predicts with them and runs cv_LOO; everything is compared bit for bit with the same work
// done by ONE thread before.  Then PLS::set_devices switches new Models to three virtual members; results agree to rounding.
// Usage: concurrent_models X.csv Y.csv   -> prints "concurrent ok"
#include <PLS/pls.h>

#include <cmath>
#include <iostream>
#include <thread>

static Mat2D slice_rows(const Mat2D &m, long r0, long n) {
    Mat2D o(n, m.cols());
    for (long j = 0; j < m.cols(); ++j)
        for (long i = 0; i < n; ++i) o(i, j) = m(r0 + i, j);
    return o;
}

struct Result {
    Mat2Dc B;
    Mat2D fitted;
    Mat2D loo;
};

static Result work(const Mat2D &X, const Mat2D &Y, size_t A) {
    PLS::Model m(X, Y, PLS::KERNEL_TYPE1, A);
    Result r;
    r.B = m.coefficients();
    r.fitted = m.fitted_values(X);
    r.loo = m.cv_LOO().errors()[0];
    return r;
}

template <typename M>
static bool same(const M &a, const M &b) {
    if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
    for (long j = 0; j < a.cols(); ++j)
        for (long i = 0; i < a.rows(); ++i)
            if (!(a(i, j) == b(i, j))) return false;
    return true;
}

int main(int argc, char **argv) {
    if (argc != 3) return 100;
    const Mat2D X = PLS::colwise_z_scores(PLS::read_matrix_file(argv[1]));
    const Mat2D Y = PLS::colwise_z_scores(PLS::read_matrix_file(argv[2]));
    const long N = X.rows(), half = N / 2;
    const Mat2D Xa = slice_rows(X, 0, half), Ya = slice_rows(Y, 0, half), Xb = slice_rows(X, half, N - half),
                Yb = slice_rows(Y, half, N - half);
    const size_t A = 3;
    const int rounds = 6;
    const Result ra = work(Xa, Ya, A), rb = work(Xb, Yb, A);  // one thread, one after the other
    bool ok_a = true, ok_b = true;
    std::thread ta([&] {
        for (int i = 0; i < rounds; ++i) {
            const Result r = work(Xa, Ya, A);
            ok_a = ok_a && same(r.B, ra.B) && same(r.fitted, ra.fitted) && same(r.loo, ra.loo);
        }
    });
    std::thread tb([&] {
        for (int i = 0; i < rounds; ++i) {
            const Result r = work(Xb, Yb, A);
            ok_b = ok_b && same(r.B, rb.B) && same(r.fitted, rb.fitted) && same(r.loo, rb.loo);
        }
    });
    ta.join();
    tb.join();
    if (!ok_a || !ok_b) {
        std::cout << "concurrent fits differ from the serial ones: thread a " << ok_a << ", thread b " << ok_b << "\n";
        return 1;
    }
    // a Model built on one thread and used from another keeps working on ITS context
    PLS::Model shared(Xa, Ya, PLS::KERNEL_TYPE1, A);
    Mat2D fitted_elsewhere;
    std::thread tc([&] { fitted_elsewhere = shared.fitted_values(Xa); });
    tc.join();
    if (!same(fitted_elsewhere, ra.fitted)) {
        std::cout << "a Model used from a second thread gave different fitted values\n";
        return 2;
    }
    // new Models on three virtual members (rows spread over three handles of device 0): same numbers to rounding
    PLS::set_devices(std::vector<int>{0, 0, 0});
    const Result r3 = work(Xa, Ya, A);
    PLS::set_devices(std::vector<int>());
    double err = 0.0, nrm = 0.0;
    for (long j = 0; j < r3.B.cols(); ++j)
        for (long i = 0; i < r3.B.rows(); ++i) {
            err += std::norm(r3.B(i, j) - ra.B(i, j));
            nrm += std::norm(ra.B(i, j));
        }
    if (!(std::sqrt(err / nrm) < 1e-10)) {
        std::cout << "three-member context: coefficients off by " << std::sqrt(err / nrm) << "\n";
        return 3;
    }
    std::cout << "concurrent ok\n";
    return 0;
}
