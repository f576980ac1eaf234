// PLS/pls.h -- public API of the MI355X-native PLS library.
//
// Source-compatible with the reference's header (tjhladish/PLS include/PLS/pls.h:1-270): same
// global typedefs, same namespace, same free functions, same PLS::Model members with the same
// signatures and defaults, so the reference's CSV-driven main (src/main.cpp:1-44) builds against
// it unchanged.  What differs is underneath: Model::plsr / scores / coefficients / fitted_values
// run on the GPU through the C-ABI of pls_hip.h (pls_amd/host/pls.cpp), and when Eigen -- the
// reference's un-vendored dependency -- is not installed, the matrix typedefs resolve to the
// small column-major family in PLS/dense.h instead of Eigen's.
//
// The reference's optional MPFR switch (MPREAL_SUPPORT, include/PLS/pls.h:11-21 upstream) has no
// device analogue and is not provided: float_type is double.
#ifndef PLS_H
#define PLS_H

#include <algorithm>  // sort
#include <complex>
#include <iostream>  // std::cerr default streams
#include <memory>    // shared_ptr (device-resident data of a Model)
#include <numeric>   // iota
#include <random>    // std::mt19937
#include <string>
#include <vector>

typedef double float_type;

#if defined(PLS_USE_EIGEN) || (!defined(PLS_NO_EIGEN) && defined(__has_include) && __has_include(<Eigen/Core>))
#include <Eigen/Core>
typedef Eigen::MatrixXd Mat2D;
typedef Eigen::VectorXd Col;
typedef Eigen::RowVectorXd Row;
typedef Eigen::MatrixXcd Mat2Dc;
typedef Eigen::VectorXcd Colc;
typedef Eigen::VectorXi Coli;
typedef Eigen::Matrix<size_t, Eigen::Dynamic, 1> Colsz;
typedef Eigen::RowVectorXi Rowi;
typedef Eigen::Matrix<size_t, 1, Eigen::Dynamic> Rowsz;
#define PLS_HAVE_EIGEN 1
#else
#include "dense.h"
typedef PLS::dense::Matrix<float_type> Mat2D;
typedef PLS::dense::Matrix<float_type, PLS::dense::COLUMN> Col;
typedef PLS::dense::Matrix<float_type, PLS::dense::ROW> Row;
typedef PLS::dense::Matrix<std::complex<float_type> > Mat2Dc;
typedef PLS::dense::Matrix<std::complex<float_type>, PLS::dense::COLUMN> Colc;
typedef PLS::dense::Matrix<int, PLS::dense::COLUMN> Coli;
typedef PLS::dense::Matrix<size_t, PLS::dense::COLUMN> Colsz;
typedef PLS::dense::Matrix<int, PLS::dense::ROW> Rowi;
typedef PLS::dense::Matrix<size_t, PLS::dense::ROW> Rowsz;
#define PLS_HAVE_EIGEN 0
#endif

namespace PLS {

// ---- cross-validation residuals ---------------------------------------------------------
// errors()[y](obs, ncomp-1): residual of response y for observation obs with ncomp components.
typedef std::vector<Mat2D> ResidualData;
class Model;

class Residual {
    const std::vector<Mat2D> _residual;
    const std::string _method_label;
    Residual(const std::vector<Mat2D> &residual, const std::string &method)
        : _residual(residual), _method_label(method) {}
    friend class Model;

public:
    const std::vector<Mat2D> errors() const { return _residual; }
    const std::string method() const { return _method_label; }
};

// ---- small container helpers ------------------------------------------------------------
// indices that sort v ascending (argsort)
template <typename T>
std::vector<size_t> ordered(const T &v) {
    std::vector<size_t> idx(static_cast<size_t>(v.size()));
    std::iota(idx.begin(), idx.end(), size_t(0));
    std::sort(idx.begin(), idx.end(),
              [&v](size_t a, size_t b) { return *(v.begin() + a) < *(v.begin() + b); });
    return idx;
}

template <typename VECTYPE>
std::vector<float_type> to_cvector(const VECTYPE &data) {
    return std::vector<float_type>(data.begin(), data.end());
}

template <typename VECTYPE>
inline VECTYPE to_evector(const std::vector<float_type> &data) {
    VECTYPE v(static_cast<long>(data.size()));
    for (size_t i = 0; i < data.size(); ++i) v[static_cast<long>(i)] = data[i];
    return v;
}

// ---- I/O and column statistics (host code; reference src/pls.cpp:23-111) -------------------
std::vector<std::string> split(const std::string &s, const char separator = ',');
// no header, one row per line; exits with status 1 on ragged rows like the reference
Mat2D read_matrix_file(const std::string &filename, const char separator = ',');

Row SST(const Mat2D &mat, const Row &means);  // sum (x - mean)^2 per column
Row SST(const Mat2D &mat);
Row colwise_stdev(const Mat2D &mat, const Row &means);  // N-1 denominator
Row colwise_stdev(const Mat2D &mat);
Row z_scores(const Row &obs, const Row &mean, const Row &stdev);
Mat2D colwise_z_scores(const Mat2D &mat, const Row &mean, const Row &stdev);
Mat2D colwise_z_scores(const Mat2D &mat);

// ---- validation statistics (host code; reference src/pls.cpp:144-305) ----------------------
float_type normalcdf(const float_type z);
float_type wilcoxon(const Col &err_1, const Col &err_2);
void rand_nchoosek(std::mt19937 &rng, std::vector<Eigen::Index> &full,
                   std::vector<Eigen::Index> &sample, std::vector<Eigen::Index> &complement);

typedef enum { KERNEL_TYPE1, KERNEL_TYPE2 } METHOD;
typedef enum { RESS, MSE } VALIDATION_OUTPUT;

Mat2D validation(const Residual &residual, const PLS::VALIDATION_OUTPUT out_type);
Colsz optimal_num_components(const Residual &residual, const float_type ALPHA = 0.1);
void print_validation(const Residual &residual, const VALIDATION_OUTPUT out_type,
                      std::ostream &os = std::cerr);

// ---- where Models run (extension; the reference is a CPU library) --------------------------------
// The GPUs the Models created FROM NOW ON use: {0, 1, 2, 3} = those HIP devices with the rows of every matrix spread
// over them, an ordinal may repeat (virtual shards on one GPU), {} = back to the environment (PLS_HIP_DEVICES = "4" or
// "0,2,5"; default device 0).  Existing Models keep the context they were built on.  There is no
// process-wide device state: every host thread gets a context of its own (streams, workspace) on first use, a Model
// carries the context it was built on, so Models of different threads run at the same time -- as upstream, where a
// Model shares nothing with another (include/PLS/pls.h:184-266 there).
struct DeviceContext;
void set_devices(const std::vector<int> &devices);

// ---- the regression object -----------------------------------------------------------------
// X: N x K predictors, Y: N x M responses, A components.
// W (K x A) weights, P (K x A) X-loadings, Q (M x A) Y-loadings, R (K x A) weights that map the
// ORIGINAL X to scores, T (N x A) scores, B = R Q^T (K x M) regression coefficients.
struct Model {
    // fit immediately with at most max_components components
    Model(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm, const size_t &max_components);
    // ... or with X.cols() components
    Model(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm = KERNEL_TYPE1);

    // (re)fit on data with the same number of predictors; runs on the GPU (pls_hip_fit)
    void plsr(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm);

    const Mat2Dc scores(const Mat2D &X_new, const size_t comp) const;  // X_new * R[:, :comp]
    const Mat2Dc scores(const Mat2D &X_new) const { return scores(X_new, A); }

    const Mat2Dc loadingsX(const size_t comp) const;  // P[:, :comp]
    const Mat2Dc loadingsX() const { return loadingsX(A); }
    const Mat2Dc loadingsY(const size_t comp) const;  // Q[:, :comp]
    const Mat2Dc loadingsY() const { return loadingsY(A); }

    const Mat2Dc coefficients(const size_t comp) const;  // R[:, :comp] * Q[:, :comp]^T
    const Mat2Dc coefficients() const { return coefficients(A); }

    const Mat2D fitted_values(const Mat2D &X, const size_t comp) const;  // X * Re(coefficients)
    const Mat2D fitted_values(const Mat2D &X) const { return fitted_values(X, A); }

    const Mat2D residuals(const Mat2D &X, const Mat2D &Y, const size_t comp) const;
    const Mat2D residuals(const Mat2D &X, const Mat2D &Y) const { return residuals(X, Y, A); }

    const Row SSE(const Mat2D &X, const Mat2D &Y, const size_t comp) const;
    const Row SSE(const Mat2D &X, const Mat2D &Y) const { return SSE(X, Y, A); }

    const Row explained_variance(const Mat2D &X, const Mat2D &Y, const size_t comp) const;
    const Row explained_variance(const Mat2D &X, const Mat2D &Y) const {
        return explained_variance(X, Y, A);
    }

    Residual cv_LOO() const;
    Residual cv_NEW_DATA(const Mat2D &X, const Mat2D &Y) const;
    Residual cv_LSO(const float_type test_fraction, const size_t num_trials, std::mt19937 &rng) const;

    void print_explained_variance(const Mat2D &X, const Mat2D &Y, std::ostream &os = std::cerr) const;
    void print_state(std::ostream &os = std::cerr) const;

private:
    // The reference keeps host copies `const Mat2D _X, _Y` of the training data for its cross-validation methods
    // and a complex N x A score matrix T (include/PLS/pls.h:250-253 upstream).  Here the training data and the
    // scores of the last fit stay RESIDENT ON THE DEVICE(S) instead (row-sharded over the GPUs PLS_HIP_DEVICES
    // names): no second host copy of X, no N x A read-back per fit; print_state() fetches T when it is asked for.
    struct Resident;
    std::shared_ptr<DeviceContext> _ctx;      // the devices, streams and workspace this Model runs on
    std::shared_ptr<const Resident> _data;    // X, Y of the constructor
    std::shared_ptr<const Resident> _scores;  // T of the last plsr() (KERNEL_TYPE1 only, as upstream)
    size_t A;
    Mat2Dc P, W, R, Q;
    PLS::METHOD method;
    void fit_resident(const Resident &d, const METHOD &algorithm);

    // shape-only models used by cv_LSO: no data yet, plsr() is called per trial
    Model(const size_t &num_predictors, const size_t &num_responses, const METHOD &algorithm = KERNEL_TYPE1);
    Model(const size_t &num_predictors, const size_t &num_responses, const METHOD &algorithm,
          const size_t &max_components);
};

}  // namespace PLS

#endif  // PLS_H
