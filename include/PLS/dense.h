// PLS/dense.h -- the small column-major dense matrix family used by PLS/pls.h when Eigen is not
// installed.  The reference declares its API on Eigen types (include/PLS/pls.h:22-33 upstream);
// Eigen is an un-vendored dependency, so this header supplies the subset of that interface which
// the public API and the CSV-driven main actually touch: rows(), cols(), size(), data(),
// operator()(i,j), operator()(i) / [i], resize, setZero, Zero, Constant, begin()/end(), and
// stream output in Eigen's default IOFormat (stream precision, one space between coefficients,
// every coefficient right-aligned to the widest one of the matrix, rows separated by newlines).
// Storage is contiguous, column-major, ld == rows -- exactly Eigen::MatrixXd's default -- which is
// what the C-ABI in pls_hip.h expects.
#ifndef PLS_DENSE_H
#define PLS_DENSE_H

#include <algorithm>
#include <complex>
#include <cstddef>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

namespace Eigen {  // only the one name the reference API spells out (rand_nchoosek's index vectors)
typedef std::ptrdiff_t Index;
}

namespace PLS {
namespace dense {

typedef Eigen::Index Index;

// Shape tags: a generic matrix, a column vector (cols fixed to 1), a row vector (rows fixed to 1)
enum Shape { GENERAL = 0, COLUMN = 1, ROW = 2 };

template <typename T, Shape S = GENERAL>
class Matrix {
    Index r_, c_;
    std::vector<T> d_;

public:
    typedef T Scalar;

    Matrix() : r_(S == ROW ? 1 : 0), c_(S == COLUMN ? 1 : 0) {}
    Matrix(Index rows, Index cols) : r_(rows), c_(cols), d_(static_cast<size_t>(rows * cols)) {}
    explicit Matrix(Index n) : r_(S == ROW ? 1 : n), c_(S == ROW ? n : 1), d_(static_cast<size_t>(n)) {
        static_assert(S != GENERAL, "one-argument constructor is for vectors");
    }
    // shape conversion between the three tags (e.g. a 1 x K Mat2D assigned to a Row)
    template <Shape S2>
    Matrix(const Matrix<T, S2> &o) : r_(o.rows()), c_(o.cols()), d_(o.begin(), o.end()) {}

    static Matrix Zero(Index rows, Index cols) { return Constant(rows, cols, T()); }
    static Matrix Zero(Index n) { return Constant(n, T()); }
    static Matrix Constant(Index rows, Index cols, const T &v) {
        Matrix m(rows, cols);
        std::fill(m.d_.begin(), m.d_.end(), v);
        return m;
    }
    static Matrix Constant(Index n, const T &v) {
        Matrix m(n);
        std::fill(m.d_.begin(), m.d_.end(), v);
        return m;
    }

    Index rows() const { return r_; }
    Index cols() const { return c_; }
    Index size() const { return r_ * c_; }
    T *data() { return d_.data(); }
    const T *data() const { return d_.data(); }

    T &operator()(Index i, Index j) { return d_[static_cast<size_t>(i + j * r_)]; }
    const T &operator()(Index i, Index j) const { return d_[static_cast<size_t>(i + j * r_)]; }
    T &operator()(Index i) { return d_[static_cast<size_t>(i)]; }
    const T &operator()(Index i) const { return d_[static_cast<size_t>(i)]; }
    T &operator[](Index i) { return d_[static_cast<size_t>(i)]; }
    const T &operator[](Index i) const { return d_[static_cast<size_t>(i)]; }

    typename std::vector<T>::iterator begin() { return d_.begin(); }
    typename std::vector<T>::iterator end() { return d_.end(); }
    typename std::vector<T>::const_iterator begin() const { return d_.begin(); }
    typename std::vector<T>::const_iterator end() const { return d_.end(); }

    void resize(Index rows, Index cols) {
        r_ = rows;
        c_ = cols;
        d_.assign(static_cast<size_t>(rows * cols), T());
    }
    void resize(Index n) { resize(S == ROW ? 1 : n, S == ROW ? n : 1); }
    Matrix &setZero() {
        std::fill(d_.begin(), d_.end(), T());
        return *this;
    }
    Matrix &setZero(Index rows, Index cols) {
        resize(rows, cols);
        return *this;
    }

    // copies of one row / one column (the reference uses X.row(i) / X.col(j) as values)
    Matrix<T, ROW> row(Index i) const {
        Matrix<T, ROW> out(c_);
        for (Index j = 0; j < c_; ++j) out[j] = (*this)(i, j);
        return out;
    }
    Matrix<T, COLUMN> col(Index j) const {
        Matrix<T, COLUMN> out(r_);
        for (Index i = 0; i < r_; ++i) out[i] = (*this)(i, j);
        return out;
    }
    Matrix leftCols(Index n) const {
        Matrix out(r_, n);
        std::copy(d_.begin(), d_.begin() + static_cast<size_t>(r_ * n), out.d_.begin());
        return out;
    }
};

// Eigen's default IOFormat: precision taken from the stream, coefficients separated by " ",
// rows by "\n", all coefficients padded on the left to the width of the widest one.
template <typename T, Shape S>
std::ostream &operator<<(std::ostream &os, const Matrix<T, S> &m) {
    if (m.size() == 0) return os;
    std::vector<std::string> txt(static_cast<size_t>(m.size()));
    size_t width = 0;
    for (Index j = 0; j < m.cols(); ++j)
        for (Index i = 0; i < m.rows(); ++i) {
            std::ostringstream ss;
            ss.copyfmt(os);
            ss.width(0);
            ss << m(i, j);
            txt[static_cast<size_t>(i + j * m.rows())] = ss.str();
            width = std::max(width, ss.str().size());
        }
    for (Index i = 0; i < m.rows(); ++i) {
        if (i) os << "\n";
        for (Index j = 0; j < m.cols(); ++j) {
            if (j) os << " ";
            const std::string &s = txt[static_cast<size_t>(i + j * m.rows())];
            os << std::string(width - s.size(), ' ') << s;
        }
    }
    return os;
}

}  // namespace dense
}  // namespace PLS

#endif  // PLS_DENSE_H
