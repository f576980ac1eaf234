// CPU-only exerciser of the host utilities of the C++ drop-in (no device call): split, read_matrix_file, column
// statistics and z-scores, normalcdf, wilcoxon, rand_nchoosek, validation, optimal_num_components, the dense matrix
// family and its Eigen-style printing.  Built with -fsanitize=address,undefined by tests/test_host_sanitized.py
// (sanitizers are host-only on this pool) and compared with numpy there.
#include <PLS/pls.h>

#include <fstream>
#include <iomanip>
#include <iostream>

template <typename M>
static void dump(const char *name, const M &m) {
    std::cout << "@" << name << " " << m.rows() << " " << m.cols() << "\n";
    for (long i = 0; i < m.rows(); ++i) {
        for (long j = 0; j < m.cols(); ++j) std::cout << (j ? " " : "") << m(i, j);
        std::cout << "\n";
    }
}

int main(int argc, char **argv) {
    if (argc != 2) return 100;
    std::cout << std::setprecision(17);
    const std::vector<std::string> f = PLS::split("1.5,,x,", ',');
    std::cout << "@split\n";
    for (const std::string &s : f) std::cout << "[" << s << "]\n";
    const Mat2D X = PLS::read_matrix_file(argv[1]);
    dump("X", X);
    dump("SST", PLS::SST(X));
    dump("stdev", PLS::colwise_stdev(X));
    dump("Z", PLS::colwise_z_scores(X));
    Row obs(X.cols());
    for (long j = 0; j < X.cols(); ++j) obs[j] = X(0, j);
    Row mean = Row::Zero(X.cols());
    for (long j = 0; j < X.cols(); ++j) {
        for (long i = 0; i < X.rows(); ++i) mean[j] += X(i, j);
        mean[j] /= static_cast<double>(X.rows());
    }
    dump("zrow", PLS::z_scores(obs, mean, PLS::colwise_stdev(X)));
    std::cout << "@normalcdf " << PLS::normalcdf(0.0) << " " << PLS::normalcdf(1.96) << " " << PLS::normalcdf(-3.0) << "\n";
    Col a(X.rows()), b(X.rows());
    for (long i = 0; i < X.rows(); ++i) {
        a[i] = X(i, 0);
        b[i] = X(i, 1) * 0.5;
    }
    std::cout << "@wilcoxon " << PLS::wilcoxon(a, b) << " " << PLS::wilcoxon(b, a) << "\n";
    std::mt19937 rng(7);
    std::vector<Eigen::Index> full(10), sample(7), comp(3);
    std::iota(full.begin(), full.end(), Eigen::Index(0));
    PLS::rand_nchoosek(rng, full, sample, comp);
    std::cout << "@choose";
    for (auto v : sample) std::cout << " " << v;
    std::cout << " |";
    for (auto v : comp) std::cout << " " << v;
    std::cout << "\n";
    const std::vector<size_t> ord = PLS::ordered(PLS::to_cvector(a));
    std::cout << "@ordered";
    for (size_t v : ord) std::cout << " " << v;
    std::cout << "\n";
    const Col back = PLS::to_evector<Col>(PLS::to_cvector(a));
    std::cout << "@roundtrip " << (back.size() == a.size() && back[0] == a[0]) << "\n";
    // Eigen-style printing of the dense family (default IOFormat): right-aligned to the widest coefficient
    Mat2D P(2, 2);
    P(0, 0) = 1; P(0, 1) = -22.5; P(1, 0) = 333; P(1, 1) = 4;
    std::cout << std::setprecision(6) << "@print\n" << P << "\n@end\n";
    Mat2Dc C(1, 2);
    C(0, 0) = std::complex<double>(1.5, 0); C(0, 1) = std::complex<double>(-2, 0);
    std::cout << "@cprint\n" << C << "\n@end\n";
    return 0;
}
