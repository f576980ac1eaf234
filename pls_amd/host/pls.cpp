// pls.cpp -- host side of the MI355X-native PLS library: the drop-in for the reference's
// src/pls.cpp.  Everything that touches N-sized data in the fit / predict path goes through
// the C-ABI of pls_hip.h to the HIP kernels (there is no CPU implementation of that path here:
// if the device or the library is missing, Model::plsr throws).  The remaining functions are the
// reference's host utilities -- CSV reader, column statistics, validation statistics and the
// cross-validation drivers that call the fit -- rewritten on plain loops over the column-major
// matrix interface shared by Eigen and PLS/dense.h.
//
// Reference behaviour is cited as (ref :line) = tjhladish/PLS src/pls.cpp.
#include <PLS/pls.h>

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>

#include "pls_hip.h"

namespace {

typedef Eigen::Index Index;

std::vector<int> device_list() {
    std::vector<int> devs;
    if (const char *e = std::getenv("PLS_HIP_DEVICES")) {
        const std::string spec(e);
        if (spec.find(',') == std::string::npos) {
            const int n = std::atoi(spec.c_str());
            for (int i = 0; i < n; ++i) devs.push_back(i);
        } else {
            size_t start = 0;
            while (start <= spec.size()) {
                const size_t pos = spec.find(',', start);
                const std::string tok = spec.substr(start, pos == std::string::npos ? std::string::npos : pos - start);
                if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
                if (pos == std::string::npos) break;
                start = pos + 1;
            }
        }
    }
    if (devs.empty()) devs.push_back(0);
    return devs;
}

// what PLS::set_devices last asked for (empty: the environment decides); every change starts a new generation of contexts
std::mutex g_cfg_mu;
std::vector<int> g_cfg_devs;
std::atomic<unsigned long> g_cfg_gen{1};

}  // namespace

// A device context: a pls_hip_group (include/pls_hip.h) over the GPUs PLS_HIP_DEVICES names -- "4" = devices 0..3,
// "0,2,5" = that list (an ordinal may repeat: virtual shards on one GPU; "3," = device 3 alone); default: device 0 --
// or over the list given to PLS::set_devices.  The rows of every matrix are spread over the members, one host thread per
// member inside the library; the library keeps its workspace in the member handles, so repeated fits reuse the same device
// buffers.  The reference has no shared state between Models (include/PLS/pls.h:184-266 upstream): here a Model carries its
// context, every host thread gets a context of its own on first use, and nothing is process-global -- two threads fit two
// Models at the same time on their own streams and workspaces.  A context serves one call at a time (`mu`): a Model that is
// handed to another thread simply queues behind its owner's calls.
struct PLS::DeviceContext {
    pls_hip_group g = nullptr;
    std::mutex mu;
    pls_hip_handle plain = nullptr;  // folds on gathered data when the group has several members (run_folds)
    int plain_device = 0;
    ~DeviceContext();
};

namespace {

// Contexts that outlive main() (a Model with static storage duration) are not torn down: by then the HIP runtime may be
// unloading.  The flag is raised by an atexit handler registered with the first context, i.e. one that runs BEFORE the
// handlers the runtime registered when it was loaded; thread-local contexts of the main thread go before any of them.
std::atomic<bool> g_exiting{false};

}  // namespace

PLS::DeviceContext::~DeviceContext() {
    if (g_exiting.load()) return;
    if (plain) pls_hip_destroy(plain);
    if (g) pls_hip_group_destroy(g);
}

namespace {

typedef std::shared_ptr<PLS::DeviceContext> Ctx;

Ctx make_context() {
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lock(g_cfg_mu);
        devs = g_cfg_devs;
    }
    if (devs.empty()) devs = device_list();  // (the environment is read again for every new context)
    static std::once_flag once;
    std::call_once(once, [] { std::atexit([] { g_exiting.store(true); }); });
    Ctx c = std::make_shared<PLS::DeviceContext>();
    const int rc = pls_hip_group_create(&c->g, static_cast<int>(devs.size()), devs.data());
    if (rc != PLS_HIP_OK)
        throw std::runtime_error("PLS: no usable MI355X (gfx950) device (pls_hip_group_create status " +
                                 std::to_string(rc) + "); this library has no CPU path");
    c->plain_device = devs[0];
    // PLS_HIP_ALGO = auto (default) | kernel | nipals | gram.  auto: the Gram plan whenever X^T X came with the
    // upload (it is accumulated on the matrix cores while the rows cross PCIe) or the cost model favours it, the
    // reference's own operation sequence (kernel) otherwise; same results to rounding either way.
    const char *e = std::getenv("PLS_HIP_ALGO");
    const std::string a(e ? e : "auto");
    pls_hip_group_set_option(c->g, PLS_HIP_OPT_ALGO,
                             a == "nipals" ? PLS_HIP_ALGO_NIPALS
                             : a == "gram" ? PLS_HIP_ALGO_GRAM
                             : a == "kernel" ? PLS_HIP_ALGO_KERNEL
                                             : PLS_HIP_ALGO_AUTO);
    return c;
}

// the calling thread's default context (created on first use, replaced after PLS::set_devices)
Ctx current_context() {
    thread_local Ctx mine;
    thread_local unsigned long mine_gen = 0;
    const unsigned long gen = g_cfg_gen.load();
    if (!mine || mine_gen != gen) {
        mine = make_context();
        mine_gen = gen;
    }
    return mine;
}

pls_hip_handle device(const Ctx &c) {  // member 0's handle: the K-sized products that need no sharding
    pls_hip_handle h = nullptr;
    pls_hip_group_handle(c->g, 0, &h);
    return h;
}

void check(const Ctx &c, int rc, const char *what) {
    if (rc != PLS_HIP_OK) {
        std::string msg = pls_hip_group_last_error(c->g);
        if (msg.empty()) msg = pls_hip_last_error(device(c));
        throw std::runtime_error(std::string("PLS: ") + what + " failed: " + msg);
    }
}

// a host matrix placed on the device(s) of a context; freed with the last owner.  Its calls do NOT take the context's
// lock (the caller holds it); the destructor does -- so a ResidentMatrix must outlive the lock_guard of the scope that
// fills it (declare it first).
struct ResidentMatrix {
    Ctx ctx;
    pls_hip_matrix m = nullptr;
    ResidentMatrix() {}
    explicit ResidentMatrix(const Ctx &c) : ctx(c) {}
    ResidentMatrix(const ResidentMatrix &) = delete;
    ResidentMatrix &operator=(const ResidentMatrix &) = delete;
    ~ResidentMatrix() {
        if (m && ctx) {
            std::lock_guard<std::mutex> lock(ctx->mu);
            pls_hip_group_free(ctx->g, m);
        }
    }
    void upload(const Mat2D &src) {
        check(ctx, pls_hip_group_upload(ctx->g, src.data(), src.rows(), src.rows(), src.cols(), PLS_HIP_F64, &m),
              "pls_hip_group_upload");
    }
    void alloc(Index rows, Index cols) {
        check(ctx, pls_hip_group_alloc(ctx->g, rows, cols, PLS_HIP_F64, &m), "pls_hip_group_alloc");
    }
    Mat2D download(Index col0, Index ncols) const {
        int64_t N = 0, K = 0;
        pls_hip_matrix_shape(m, &N, &K, nullptr);
        Mat2D out(static_cast<Index>(N), ncols);
        check(ctx, pls_hip_group_download(ctx->g, m, col0, ncols, out.data(), N), "pls_hip_group_download");
        return out;
    }
};

// X and Y of one data set: X^T X and X^T Y are formed while X streams in and stay with the pair (pls_hip.h)
void upload_pair(const Mat2D &X, const Mat2D &Y, ResidentMatrix &dX, ResidentMatrix &dY) {
    check(dX.ctx, pls_hip_group_upload_xy(dX.ctx->g, X.data(), X.rows(), Y.data(), Y.rows(), X.rows(), X.cols(), Y.cols(),
                                          PLS_HIP_F64, &dX.m, &dY.m),
          "pls_hip_group_upload_xy");
}

Row column_means(const Mat2D &mat) {
    Row m(mat.cols());
    for (Index j = 0; j < mat.cols(); ++j) {
        float_type s = 0;
        for (Index i = 0; i < mat.rows(); ++i) s += mat(i, j);
        m[j] = s / static_cast<float_type>(mat.rows());
    }
    return m;
}

// real K x c (or N x c) block -> the API's complex container (imaginary parts are zero: the
// reference only uses complex types because Eigen::EigenSolver returns them, ref :401-402)
Mat2Dc to_complex(const std::vector<float_type> &re, Index rows, Index cols) {
    Mat2Dc out(rows, cols);
    for (Index j = 0; j < cols; ++j)
        for (Index i = 0; i < rows; ++i) out(i, j) = std::complex<float_type>(re[static_cast<size_t>(i + j * rows)], 0);
    return out;
}

std::vector<float_type> real_part(const Mat2Dc &m, Index cols) {
    std::vector<float_type> re(static_cast<size_t>(m.rows() * cols));
    for (Index j = 0; j < cols; ++j)
        for (Index i = 0; i < m.rows(); ++i) re[static_cast<size_t>(i + j * m.rows())] = m(i, j).real();
    return re;
}

}  // namespace

// what a Model keeps on the device(s): the training data of its constructor, or the scores of its last fit
struct PLS::Model::Resident {
    ResidentMatrix X, Y, T;
    Index N = 0, K = 0, M = 0;
    explicit Resident(const Ctx &c) : X(c), Y(c), T(c) {}
};

namespace PLS {

// Extension (no upstream counterpart): which GPUs the Models created from now on use -- {0, 1, 2, 3} = those devices, an
// ordinal may repeat (virtual shards), {} = back to the PLS_HIP_DEVICES environment.  Existing Models keep
// the context they were built on.
void set_devices(const std::vector<int> &devices) {
    {
        std::lock_guard<std::mutex> lock(g_cfg_mu);
        g_cfg_devs = devices;
    }
    ++g_cfg_gen;
}

// ---------------------------------------------------------------------------------------------
// text input (ref :23-67)
// ---------------------------------------------------------------------------------------------
std::vector<std::string> split(const std::string &s, const char separator) {
    std::vector<std::string> fields;
    size_t start = 0;
    for (size_t pos = s.find(separator); pos != std::string::npos; pos = s.find(separator, start)) {
        fields.push_back(s.substr(start, pos - start));
        start = pos + 1;
    }
    fields.push_back(s.substr(start));  // the tail after the last separator (possibly empty)
    return fields;
}

Mat2D read_matrix_file(const std::string &filename, const char separator) {
    std::ifstream in(filename);
    std::vector<std::vector<float_type> > rows;
    std::string line;
    while (in.is_open() && std::getline(in, line)) {
        const std::vector<std::string> fields = split(line, separator);
        std::vector<float_type> r(fields.size());
        for (size_t i = 0; i < fields.size(); ++i) r[i] = std::stod(fields[i]);  // throws like the reference (ref :53)
        if (!rows.empty() && rows[0].size() != r.size()) {
            std::cerr << "Error: row " << rows.size() << " has " << r.size()
                      << " columns, but previous row(s) have " << rows[0].size() << " columns." << std::endl;
            std::exit(1);  // ref :54-58
        }
        rows.push_back(r);
    }
    if (rows.empty()) {  // the reference dereferences M[0] here (UB, ref :64); fail cleanly instead
        std::cerr << "Error: could not read any rows from " << filename << std::endl;
        std::exit(1);
    }
    Mat2D X(static_cast<Index>(rows.size()), static_cast<Index>(rows[0].size()));
    for (Index i = 0; i < X.rows(); ++i)
        for (Index j = 0; j < X.cols(); ++j) X(i, j) = rows[static_cast<size_t>(i)][static_cast<size_t>(j)];
    return X;
}

// ---------------------------------------------------------------------------------------------
// column statistics (ref :69-111)
// ---------------------------------------------------------------------------------------------
Row SST(const Mat2D &mat, const Row &means) {
    Row out = Row::Zero(mat.cols());
    if (mat.rows() < 2) return out;  // ref :71
    for (Index j = 0; j < mat.cols(); ++j) {
        float_type s = 0;
        for (Index i = 0; i < mat.rows(); ++i) {
            const float_type d = mat(i, j) - means[j];
            s += d * d;
        }
        out[j] = s;
    }
    return out;
}

Row SST(const Mat2D &mat) { return SST(mat, column_means(mat)); }

Row colwise_stdev(const Mat2D &mat, const Row &means) {
    Row out = SST(mat, means);
    const float_type n1 = static_cast<float_type>(mat.rows()) - 1;  // unbiased: N-1 (ref :82)
    for (Index j = 0; j < out.size(); ++j) out[j] = std::sqrt(out[j] / n1);
    return out;
}

Row colwise_stdev(const Mat2D &mat) { return colwise_stdev(mat, column_means(mat)); }

Row z_scores(const Row &obs, const Row &mean, const Row &stdev) {
    Row out(obs.size());
    for (Index j = 0; j < obs.size(); ++j) out[j] = (obs[j] - mean[j]) / stdev[j];
    return out;
}

Mat2D colwise_z_scores(const Mat2D &mat, const Row &mean, const Row &stdev) {
    // The reference prepares a zero-guarded copy of stdev but then divides by the unguarded
    // one (ref :94-103), so a constant column comes out as NaN; kept, so that results match.
    Mat2D zs(mat.rows(), mat.cols());
    for (Index j = 0; j < mat.cols(); ++j)
        for (Index i = 0; i < mat.rows(); ++i) zs(i, j) = (mat(i, j) - mean[j]) / stdev[j];
    return zs;
}

Mat2D colwise_z_scores(const Mat2D &mat) {
    const Row means = column_means(mat);
    return colwise_z_scores(mat, means, colwise_stdev(mat, means));
}

// ---------------------------------------------------------------------------------------------
// validation statistics (ref :144-305)
// ---------------------------------------------------------------------------------------------
// Abramowitz & Stegun 26.2.19 polynomial approximation of the normal CDF (ref :152-160)
float_type normalcdf(const float_type z) {
    const float_type a = std::fabs(z);
    const float_type poly = 1 + 0.196854 * a + 0.115194 * a * a + 0.000344 * a * a * a + 0.019527 * a * a * a * a;
    const float_type p = 0.5 / std::pow(poly, 4);
    return z < 0 ? p : 1.0 - p;
}

// Wilcoxon signed-rank statistic on |err_1| - |err_2|, normal approximation (ref :190-211)
float_type wilcoxon(const Col &err_1, const Col &err_2) {
    const size_t n = static_cast<size_t>(err_1.size());
    std::vector<float_type> mag(n);
    std::vector<int> sign(n);
    for (size_t i = 0; i < n; ++i) {
        const float_type d = std::fabs(err_1[static_cast<Index>(i)]) - std::fabs(err_2[static_cast<Index>(i)]);
        sign[i] = (0 < d) - (d < 0);
        mag[i] = std::fabs(d);
    }
    const std::vector<size_t> order = ordered(mag);
    float_type d = 0;
    for (size_t rank = 0; rank < n; ++rank) d += static_cast<float_type>(rank + 1) * sign[order[rank]];
    const float_type t = static_cast<float_type>(n * (n + 1)) / 2.0;
    const float_type v = (t - d) / 2.0;
    const float_type ev = t / 2.0;
    const float_type sv = std::sqrt(static_cast<float_type>(n * (n + 1) * (2 * n + 1)) / 24.0);
    return 1.0 - normalcdf((v - ev) / sv);
}

// shuffle `full`, then split it into the leading sample and the trailing complement (ref :218-227)
void rand_nchoosek(std::mt19937 &rng, std::vector<Eigen::Index> &full, std::vector<Eigen::Index> &sample,
                   std::vector<Eigen::Index> &complement) {
    std::shuffle(full.begin(), full.end(), rng);
    std::copy(full.begin(), full.begin() + static_cast<std::ptrdiff_t>(sample.size()), sample.begin());
    std::copy(full.begin() + static_cast<std::ptrdiff_t>(sample.size()), full.end(), complement.begin());
}

// rows = Y variable, cols = number of components; RESS = sum of squared residuals, MSE = RESS/N
Mat2D validation(const Residual &residual, const VALIDATION_OUTPUT out_type) {
    const std::vector<Mat2D> errors = residual.errors();
    if (errors.empty()) return Mat2D::Zero(0, 0);
    Mat2D out = Mat2D::Zero(static_cast<Index>(errors.size()), errors[0].cols());
    for (size_t y = 0; y < errors.size(); ++y)
        for (Index c = 0; c < errors[y].cols(); ++c) {
            float_type s = 0;
            for (Index i = 0; i < errors[y].rows(); ++i) s += errors[y](i, c) * errors[y](i, c);
            out(static_cast<Index>(y), c) = s;
        }
    if (out_type == MSE) {
        const float_type n = static_cast<float_type>(errors[0].rows());
        for (Index j = 0; j < out.cols(); ++j)
            for (Index i = 0; i < out.rows(); ++i) out(i, j) /= n;
    }
    return out;
}

// per Y variable: the smallest number of components whose errors are not significantly worse
// (Wilcoxon, ALPHA) than those at the PRESS minimum (ref :265-289)
Colsz optimal_num_components(const Residual &residual, const float_type ALPHA) {
    const std::vector<Mat2D> errors = residual.errors();
    const Mat2D press = validation(residual, RESS);
    Colsz best(press.rows());
    for (size_t y = 0; y < errors.size(); ++y) {
        size_t ref_min = 0;
        for (Index c = 1; c < press.cols(); ++c)
            if (press(static_cast<Index>(y), c) < press(static_cast<Index>(y), static_cast<Index>(ref_min)))
                ref_min = static_cast<size_t>(c);
        size_t pick = ref_min;
        const Col err_ref = errors[y].col(static_cast<Index>(ref_min));
        for (size_t alt = 0; alt < ref_min; ++alt) {
            const Col err_alt = errors[y].col(static_cast<Index>(alt));
            if (wilcoxon(err_ref, err_alt) > ALPHA) {
                pick = alt;
                break;
            }
        }
        best[static_cast<Index>(y)] = pick + 1;  // component counts start at 1
    }
    return best;
}

void print_validation(const Residual &residual, const VALIDATION_OUTPUT out_type, std::ostream &os) {
    os << residual.method() << " Validation:" << std::endl;
    Mat2D em = validation(residual, out_type);
    if (out_type == MSE) {
        os << "RMSE ";
        for (Index j = 0; j < em.cols(); ++j)
            for (Index i = 0; i < em.rows(); ++i) em(i, j) = std::sqrt(em(i, j));
    } else if (out_type == RESS) {
        os << "PRESS ";
    } else {
        os << "UNKNOWN ";
    }
    os << " Matrix (rows = Y variable; cols = # of components):" << std::endl << em << std::endl;
    os << "Optimal number of components (by Y variable):\t" << optimal_num_components(residual) << std::endl;
}

}  // namespace PLS

using namespace PLS;

// ---------------------------------------------------------------------------------------------
// Model: constructors (ref :323-359)
// ---------------------------------------------------------------------------------------------
Model::Model(const size_t &num_predictors, const size_t &num_responses, const METHOD &algorithm,
             const size_t &max_components)
    : A(max_components), method(algorithm) {
    if (max_components > num_predictors) throw std::invalid_argument("PLS::Model: max_components > predictors");
    const Index K = static_cast<Index>(num_predictors), M = static_cast<Index>(num_responses);
    P.setZero(K, static_cast<Index>(A));
    W.setZero(K, static_cast<Index>(A));
    R.setZero(K, static_cast<Index>(A));
    Q.setZero(M, static_cast<Index>(A));
}

Model::Model(const size_t &num_predictors, const size_t &num_responses, const METHOD &algorithm)
    : Model(num_predictors, num_responses, algorithm, num_predictors) {}

// The reference deep-copies X and Y into host members (ref :344).  Here the one copy that is made anyway -- the
// transfer to the device(s) -- IS the model's copy: the caller may destroy X and Y afterwards, exactly as upstream.
Model::Model(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm, const size_t &max_components)
    : A(max_components), method(algorithm) {
    // the reference only assert()s these (ref :345-347); a Release build would run into UB
    if (max_components > static_cast<size_t>(X.cols()) || X.rows() == 0 || X.rows() != Y.rows())
        throw std::invalid_argument("PLS::Model: need max_components <= X.cols(), X.rows() > 0, X.rows() == Y.rows()");
    const Index K = X.cols(), M = Y.cols();
    P.setZero(K, static_cast<Index>(A));
    W.setZero(K, static_cast<Index>(A));
    R.setZero(K, static_cast<Index>(A));
    Q.setZero(M, static_cast<Index>(A));
    _ctx = current_context();
    std::shared_ptr<Resident> d = std::make_shared<Resident>(_ctx);
    d->N = X.rows(); d->K = K; d->M = M;
    {
        std::lock_guard<std::mutex> lock(_ctx->mu);
        upload_pair(X, Y, d->X, d->Y);
    }
    _data = d;
    fit_resident(*d, algorithm);
}

Model::Model(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm)
    : Model(X, Y, algorithm, static_cast<size_t>(X.cols())) {}

// ---------------------------------------------------------------------------------------------
// the fit (ref :390-437): one call into the device library
// ---------------------------------------------------------------------------------------------
void Model::plsr(const Mat2D &X, const Mat2D &Y, const METHOD &algorithm) {
    if (X.rows() == 0 || X.rows() != Y.rows() || static_cast<size_t>(X.cols()) < A)
        throw std::invalid_argument("PLS::Model::plsr: need X.rows() > 0, X.rows() == Y.rows(), X.cols() >= A");
    if (!_ctx) _ctx = current_context();
    Resident d(_ctx);  // data of this call only: the constructor's X, Y stay what the cross-validation methods use (ref :487)
    d.N = X.rows(); d.K = X.cols(); d.M = Y.cols();
    {
        std::lock_guard<std::mutex> lock(_ctx->mu);
        upload_pair(X, Y, d.X, d.Y);
    }
    fit_resident(d, algorithm);
}

void Model::fit_resident(const Resident &d, const METHOD &algorithm) {
    method = algorithm;
    const Index K = d.K, M = d.M, Ai = static_cast<Index>(A);
    std::vector<float_type> w(static_cast<size_t>(K * Ai)), p(w.size()), r(w.size());
    std::vector<float_type> q(static_cast<size_t>(M * Ai));
    const Ctx &ctx = d.X.ctx;
    std::shared_ptr<Resident> sc;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (algorithm == KERNEL_TYPE1) {  // T exists for KERNEL_TYPE1 only (ref :394,434); it stays on the device(s)
            sc = std::make_shared<Resident>(ctx);
            sc->N = d.N; sc->K = Ai;
            sc->T.alloc(d.N, Ai);
        }
        check(ctx, pls_hip_group_fit(ctx->g, d.X.m, d.Y.m, Ai,
                                algorithm == KERNEL_TYPE1 ? PLS_HIP_KERNEL_TYPE1 : PLS_HIP_KERNEL_TYPE2, w.data(), p.data(),
                                q.data(), r.data(), sc ? sc->T.m : nullptr, nullptr),
              "pls_hip_group_fit");
    }
    _scores = sc;
    W = to_complex(w, K, Ai);
    P = to_complex(p, K, Ai);
    R = to_complex(r, K, Ai);
    Q = to_complex(q, M, Ai);
}

// ---------------------------------------------------------------------------------------------
// predict and metrics (ref :439-467)
// ---------------------------------------------------------------------------------------------
namespace {

// X_new (host) * Bm (K x C, host) -> N x C on the host: X_new goes to the device(s) through the staging pipeline,
// the product runs row-sharded, the result comes back
Mat2D product_on_device(const Ctx &ctx, const Mat2D &X_new, const std::vector<float_type> &bm, Index C) {
    const Index N = X_new.rows(), K = X_new.cols();
    Mat2D out(N, C);
    if (N == 0 || C == 0) return out;
    ResidentMatrix X(ctx), O(ctx);  // (released after the lock below)
    std::lock_guard<std::mutex> lock(ctx->mu);
    X.upload(X_new);
    O.alloc(N, C);
    check(ctx, pls_hip_group_xb(ctx->g, X.m, bm.data(), K, C, O.m), "pls_hip_group_xb");
    check(ctx, pls_hip_group_download(ctx->g, O.m, 0, C, out.data(), N), "pls_hip_group_download");
    return out;
}

}  // namespace

const Mat2Dc Model::scores(const Mat2D &X_new, const size_t comp) const {
    if (comp > A) throw std::invalid_argument("PLS::Model::scores: comp > A");  // assert in the reference (ref :440)
    const Index N = X_new.rows(), c = static_cast<Index>(comp);
    const Mat2D s = product_on_device(_ctx ? _ctx : current_context(), X_new, real_part(R, c), c);
    Mat2Dc out(N, c);
    for (Index j = 0; j < c; ++j)
        for (Index i = 0; i < N; ++i) out(i, j) = std::complex<float_type>(s(i, j), 0);
    return out;
}

const Mat2Dc Model::loadingsX(const size_t comp) const { return P.leftCols(static_cast<Index>(comp)); }
const Mat2Dc Model::loadingsY(const size_t comp) const { return Q.leftCols(static_cast<Index>(comp)); }

const Mat2Dc Model::coefficients(const size_t comp) const {
    if (comp > A) throw std::invalid_argument("PLS::Model::coefficients: comp > A");  // ref :445
    const Index K = R.rows(), M = Q.rows(), Ai = static_cast<Index>(A);
    const std::vector<float_type> r = real_part(R, Ai), q = real_part(Q, Ai);
    std::vector<float_type> b(static_cast<size_t>(K * M));
    {
        const Ctx ctx = _ctx ? _ctx : current_context();
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (pls_hip_coefficients(device(ctx), r.data(), q.data(), K, M, Ai, static_cast<Index>(comp), PLS_HIP_MEM_HOST,
                                 b.data()) != PLS_HIP_OK)
            throw std::runtime_error(std::string("PLS: pls_hip_coefficients failed: ") + pls_hip_last_error(device(ctx)));
    }
    return to_complex(b, K, M);
}

const Mat2D Model::fitted_values(const Mat2D &X_new, const size_t comp) const {
    const Mat2Dc Bc = coefficients(comp);
    return product_on_device(_ctx ? _ctx : current_context(), X_new, real_part(Bc, Bc.cols()), Bc.cols());
}

const Mat2D Model::residuals(const Mat2D &X_new, const Mat2D &Y_new, const size_t comp) const {
    Mat2D res = fitted_values(X_new, comp);
    for (Index j = 0; j < res.cols(); ++j)
        for (Index i = 0; i < res.rows(); ++i) res(i, j) = Y_new(i, j) - res(i, j);
    return res;
}

const Row Model::SSE(const Mat2D &X_new, const Mat2D &Y_new, const size_t comp) const {
    const Mat2D res = residuals(X_new, Y_new, comp);
    Row out = Row::Zero(res.cols());
    for (Index j = 0; j < res.cols(); ++j)
        for (Index i = 0; i < res.rows(); ++i) out[j] += res(i, j) * res(i, j);
    return out;
}

const Row Model::explained_variance(const Mat2D &X_new, const Mat2D &Y_new, const size_t comp) const {
    const Row sse = SSE(X_new, Y_new, comp), sst = SST(Y_new);
    Row out(sse.size());
    for (Index j = 0; j < sse.size(); ++j) out[j] = 1.0 - sse[j] / sst[j];
    return out;
}

// ---------------------------------------------------------------------------------------------
// cross-validation drivers (ref :469-549)
//
// Upstream these loops refit the model once per fold (N refits for leave-one-out, num_trials for
// leave-some-out, each through plsr on a freshly assembled matrix).  Here the folds of one call are
// handed to the device together (pls_hip_cv_folds): XX = X^T X and XY = X^T Y are formed once, every
// fold works on their downdates by its held-out rows, and the residuals come back in the layout of
// Residual::errors().  Same numbers up to fp64 rounding (tests compare with one refit per fold).
// The folds read the model's RESIDENT training data; with several devices the fold kernel still needs the
// whole matrix on one of them, so the data is gathered to the host once and handed to member 0.
//
// Deliberate difference kept from the first version: the reference builds its inner models with
// A = X.cols() components (the 3-argument constructors, ref :334-337, :356-359, used at :477 and :531)
// although only the outer model's A are ever read (:479, :540); components are computed strictly in
// sequence, so the first A are identical either way, and only those are computed here.
// ---------------------------------------------------------------------------------------------
namespace {

// residuals of `folds` (each `test_size` held-out rows, row-major index list) -> M matrices nobs x A
std::vector<Mat2D> run_folds(const ResidentMatrix &X, const ResidentMatrix &Y, Index N, Index K, Index M, size_t A,
                             const std::vector<int64_t> &test_idx, size_t test_size, size_t num_folds) {
    const Index nobs = static_cast<Index>(num_folds * test_size), Ai = static_cast<Index>(A);
    std::vector<float_type> e(static_cast<size_t>(nobs * Ai * M));
    {
        const Ctx &ctx = X.ctx;
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (pls_hip_group_size(ctx->g) == 1) {
            check(ctx, pls_hip_group_cv_folds(ctx->g, X.m, Y.m, Ai, test_idx.data(), static_cast<int64_t>(test_size),
                                              static_cast<int64_t>(num_folds), e.data()),
                  "pls_hip_group_cv_folds");
        } else {  // several devices: gather the rows once, member 0's device runs the folds
            const Mat2D Xh = X.download(0, K), Yh = Y.download(0, M);
            // (a member handle carries the group's reducer; folds need none: a plain handle of the context's own)
            if (!ctx->plain && pls_hip_create(&ctx->plain, ctx->plain_device, nullptr) != PLS_HIP_OK)
                throw std::runtime_error("PLS: pls_hip_create failed for the cross-validation handle");
            if (pls_hip_cv_folds(ctx->plain, Xh.data(), N, Yh.data(), N, N, K, M, Ai, test_idx.data(),
                                 static_cast<int64_t>(test_size), static_cast<int64_t>(num_folds), PLS_HIP_F64,
                                 PLS_HIP_MEM_HOST, e.data()) != PLS_HIP_OK)
                throw std::runtime_error(std::string("PLS: pls_hip_cv_folds failed: ") + pls_hip_last_error(ctx->plain));
        }
    }
    std::vector<Mat2D> Ev(static_cast<size_t>(M), Mat2D::Zero(nobs, Ai));
    for (Index m = 0; m < M; ++m)
        for (Index c = 0; c < Ai; ++c)
            for (Index i = 0; i < nobs; ++i)
                Ev[static_cast<size_t>(m)](i, c) = e[static_cast<size_t>(m * nobs * Ai + i + c * nobs)];
    return Ev;
}

}  // namespace

Residual Model::cv_LOO() const {
    if (!_data) throw std::invalid_argument("PLS::Model::cv_LOO: the model holds no training data");
    const size_t N = static_cast<size_t>(_data->N);
    if (N < 2) throw std::invalid_argument("PLS::Model::cv_LOO: need at least two observations");
    std::vector<int64_t> idx(N);
    std::iota(idx.begin(), idx.end(), int64_t(0));  // fold i leaves out row i (ref :478-488)
    return Residual(run_folds(_data->X, _data->Y, _data->N, _data->K, _data->M, A, idx, 1, N), "LOO");
}

Residual Model::cv_NEW_DATA(const Mat2D &X_new, const Mat2D &Y_new) const {
    if (X_new.cols() != R.rows() || Y_new.cols() != Q.rows())
        throw std::invalid_argument("PLS::Model::cv_NEW_DATA: column counts differ from the training data");
    std::vector<Mat2D> Ev(static_cast<size_t>(Y_new.cols()), Mat2D::Zero(X_new.rows(), static_cast<Index>(A)));
    for (size_t nc = 1; nc <= A; ++nc) {
        const Mat2D res = residuals(X_new, Y_new, nc);
        for (Index m = 0; m < res.cols(); ++m)
            for (Index i = 0; i < res.rows(); ++i) Ev[static_cast<size_t>(m)](i, static_cast<Index>(nc - 1)) = res(i, m);
    }
    return Residual(Ev, "NEW DATA");
}

Residual Model::cv_LSO(const float_type test_fraction, const size_t num_trials, std::mt19937 &rng) const {
    if (!_data) throw std::invalid_argument("PLS::Model::cv_LSO: the model holds no training data");
    const size_t N = static_cast<size_t>(_data->N);
    const size_t test_size = static_cast<size_t>(test_fraction * N + 0.5);
    const size_t train_size = N - test_size;
    if (test_size == 0 || train_size == 0) throw std::invalid_argument("PLS::Model::cv_LSO: empty train or test split");

    // the same shuffle stream as the reference (rand_nchoosek on one persistent index vector, ref :524-534):
    // trial rep trains on `sample` and tests on `complement`
    std::vector<Eigen::Index> sample(train_size), complement(test_size), full(N);
    std::iota(full.begin(), full.end(), Eigen::Index(0));
    std::vector<int64_t> idx(num_trials * test_size);
    for (size_t rep = 0; rep < num_trials; ++rep) {
        rand_nchoosek(rng, full, sample, complement);
        for (size_t i = 0; i < test_size; ++i) idx[rep * test_size + i] = static_cast<int64_t>(complement[i]);
    }
    return Residual(run_folds(_data->X, _data->Y, _data->N, _data->K, _data->M, A, idx, test_size, num_trials), "LSO");
}

// ---------------------------------------------------------------------------------------------
// text output (ref :551-580)
// ---------------------------------------------------------------------------------------------
// Same lines as the reference prints (ref :551-562), but the A calls of explained_variance -- A
// full X*B passes upstream -- are replaced by one X*R pass and one sweep over the scores
// (pls_hip_model_sse): SSE_c follows from Yhat_c = S[:, :c] Q[:, :c]^T.
void Model::print_explained_variance(const Mat2D &X, const Mat2D &Y, std::ostream &os) const {
    const int wd = static_cast<int>(std::ceil(std::log10(static_cast<double>(A))));
    const Index M = Y.cols(), Ai = static_cast<Index>(A);
    const std::vector<float_type> r = real_part(R, Ai), q = real_part(Q, Ai);
    std::vector<float_type> sse(static_cast<size_t>(M * Ai));
    {
        const Ctx ctx = _ctx ? _ctx : current_context();
        ResidentMatrix dX(ctx), dY(ctx);  // (released after the lock)
        std::lock_guard<std::mutex> lock(ctx->mu);
        dX.upload(X);
        dY.upload(Y);
        check(ctx, pls_hip_group_model_sse(ctx->g, dX.m, dY.m, Ai, r.data(), q.data(), sse.data()), "pls_hip_group_model_sse");
    }
    const Row sst = SST(Y);
    for (size_t nc = 1; nc <= A; ++nc) {
        Row ev(M), se(M);
        for (Index m = 0; m < M; ++m) {
            se[m] = sse[static_cast<size_t>(m + static_cast<Index>(nc - 1) * M)];
            ev[m] = 1.0 - se[m] / sst[m];
        }
        os << std::setw(wd) << nc << " components explained variance: " << ev;
        os << "  - SSE: " << se << std::endl;
    }
}

void Model::print_state(std::ostream &os) const {
    os << "P:" << std::endl << P << std::endl;
    os << "W:" << std::endl << W << std::endl;
    os << "R:" << std::endl << R << std::endl;
    os << "Q:" << std::endl << Q << std::endl;
    Mat2Dc T;  // fetched from the device(s) only here; empty for KERNEL_TYPE2, as upstream (ref :394,434)
    if (_scores) {
        std::lock_guard<std::mutex> lock(_scores->T.ctx->mu);
        const Mat2D t = _scores->T.download(0, static_cast<Index>(A));
        T = Mat2Dc(t.rows(), t.cols());
        for (Index j = 0; j < t.cols(); ++j)
            for (Index i = 0; i < t.rows(); ++i) T(i, j) = std::complex<float_type>(t(i, j), 0);
    }
    os << "T:" << std::endl << T << std::endl;
    os << "coefficients:" << std::endl << coefficients() << std::endl;
}
