// PLS -- the CSV-driven command line of the reference (tjhladish/PLS src/main.cpp:10-44), on the
// GPU-backed library:   PLS X.csv Y.csv num_components
// Reads both matrices, z-scores every column, fits num_components components (KERNEL_TYPE1),
// prints the model state and explained variance, then leave-one-out and leave-some-out
// (30 % held out, 10*N trials, default-seeded mt19937) validation tables -- all on std::cerr, like
// the reference; nothing is written to stdout.
#include <cstdlib>
#include <exception>
#include <iostream>
#include <random>
#include <string>

#include <PLS/pls.h>

int main(int argc, char *argv[]) {
    if (argc != 4) {
        std::cerr << "Usage: ./pls X_data.csv Y_data.csv num_components" << std::endl;
        std::cerr << "NB: X and Y csvs must be comma delimited, square numerical data, with no headers." << std::endl;
        return 100;
    }
    try {
        const Mat2D X = PLS::colwise_z_scores(PLS::read_matrix_file(argv[1]));
        const Mat2D Y = PLS::colwise_z_scores(PLS::read_matrix_file(argv[2]));
        const size_t ncomp = static_cast<size_t>(std::atoi(argv[3]));

        PLS::Model model(X, Y, PLS::KERNEL_TYPE1, ncomp);
        model.print_state();
        model.print_explained_variance(X, Y);

        PLS::print_validation(model.cv_LOO(), PLS::MSE);

        std::mt19937 rng;  // default seed, as in the reference
        PLS::print_validation(model.cv_LSO(0.3, 10 * static_cast<size_t>(X.rows()), rng), PLS::MSE);
    } catch (const std::exception &e) {
        std::cerr << "PLS: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
