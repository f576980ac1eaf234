// ref_dump.cpp -- test infrastructure: a driver for a build of the REFERENCE itself (oracle/_ref), used to pin the
// CPU restatement (pls_oracle.c) against the reference's own arithmetic the day a real <Eigen/Dense> is on the include path.
// It is compiled only by `make -C oracle _ref` TOGETHER with /root/reference/src/pls.cpp, where that file lies (nothing of
// the reference is copied into this repository), and only when the reference compiles against the Eigen it finds.
//
//   pls_ref_dump X.csv Y.csv ncomp method(1|2) zscore(0|1)
//
// prints, with 17 significant digits, what PLS::Model::print_state prints (P, W, R, Q, T, coefficients: complex entries
// "(re,im)", src/pls.cpp:564-580) followed by "fitted:" = fitted_values(X) (src/pls.cpp:449-451) on stdout.
// Everything here goes through the reference's PUBLIC interface (include/PLS/pls.h:88-93,107-111,187-248).
#include <PLS/pls.h>

#include <cstdlib>
#include <iomanip>
#include <iostream>

int main(int argc, char **argv) {
    if (argc != 6) {
        std::cerr << "usage: pls_ref_dump X.csv Y.csv ncomp method(1|2) zscore(0|1)\n";
        return 100;
    }
    PLS::Mat2D X = PLS::read_matrix_file(argv[1]);
    PLS::Mat2D Y = PLS::read_matrix_file(argv[2]);
    const size_t ncomp = (size_t)std::atoi(argv[3]);
    const PLS::METHOD method = std::atoi(argv[4]) == 2 ? PLS::KERNEL_TYPE2 : PLS::KERNEL_TYPE1;
    if (std::atoi(argv[5])) {  // what the reference's main does before the fit (src/main.cpp:24-25)
        X = PLS::colwise_z_scores(X);
        Y = PLS::colwise_z_scores(Y);
    }
    PLS::Model model(X, Y, method, ncomp);
    std::cout << std::setprecision(17);
    model.print_state(std::cout);
    std::cout << "fitted:" << std::endl << model.fitted_values(X) << std::endl;
    return 0;
}
