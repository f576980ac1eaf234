// Exercises the C++ PLS::Model API of include/PLS/pls.h (the reference's public surface) on the
// GPU-backed library and dumps every result with 17 significant digits for tests/test_gpu_cli.py
// to compare with the oracle.  Usage: model_api X.csv Y.csv ncomp
#include <PLS/pls.h>

#include <iomanip>
#include <iostream>

template <typename M>
static void dump(const char *name, const M &m) {
    std::cout << "@" << name << " " << m.rows() << " " << m.cols() << "\n";
    for (long i = 0; i < m.rows(); ++i) {
        for (long j = 0; j < m.cols(); ++j) std::cout << (j ? " " : "") << std::real(m(i, j));
        std::cout << "\n";
    }
}

int main(int argc, char **argv) {
    if (argc != 4) return 100;
    std::cout << std::setprecision(17);
    const Mat2D X = PLS::colwise_z_scores(PLS::read_matrix_file(argv[1]));
    const Mat2D Y = PLS::colwise_z_scores(PLS::read_matrix_file(argv[2]));
    const size_t A = static_cast<size_t>(std::atoi(argv[3]));
    dump("X", X);
    dump("Y", Y);
    PLS::Model m(X, Y, PLS::KERNEL_TYPE1, A);
    dump("coefficients", m.coefficients());
    dump("coefficients1", m.coefficients(1));
    dump("scores", m.scores(X));
    dump("loadingsX", m.loadingsX());
    dump("loadingsY", m.loadingsY());
    dump("fitted", m.fitted_values(X));
    dump("residuals", m.residuals(X, Y));
    dump("SSE", m.SSE(X, Y));
    dump("EV", m.explained_variance(X, Y));
    PLS::Residual nd = m.cv_NEW_DATA(X, Y);
    dump("newdata0", nd.errors()[0]);
    PLS::Residual loo = m.cv_LOO();
    dump("loo0", loo.errors()[0]);
    dump("loo_mse", PLS::validation(loo, PLS::MSE));
    dump("loo_opt", PLS::optimal_num_components(loo));
    std::mt19937 rng;
    PLS::Residual lso = m.cv_LSO(0.3, 20, rng);
    dump("lso_mse", PLS::validation(lso, PLS::MSE));
    // refit through the public plsr() on the same object (the reference's CV loops do this)
    m.plsr(X, Y, PLS::KERNEL_TYPE1);
    dump("coefficients_refit", m.coefficients());
    bool threw = false;
    try { m.coefficients(A + 1); } catch (const std::exception &) { threw = true; }
    std::cout << "@threw " << (threw ? 1 : 0) << "\n";
    std::cout << "@wilcoxon " << PLS::wilcoxon(loo.errors()[0].col(0), loo.errors()[0].col(static_cast<long>(A) - 1)) << "\n";
    std::cout << "@normalcdf " << PLS::normalcdf(0.5) << " " << PLS::normalcdf(-1.25) << "\n";
    return 0;
}
