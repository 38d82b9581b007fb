// ORACLE — TEST INFRASTRUCTURE ONLY.  Decompositions for the Eigen shim (see ../Core).  Algorithms restated from
// the published descriptions of Eigen's FullPivHouseholderQR / PartialPivLU / ColPivHouseholderQR / JacobiSVD;
// same arithmetic as oracle/src/solvers_rel.cc and refine.cc.
#ifndef ORACLE_EIGEN_SHIM_DECOMP
#define ORACLE_EIGEN_SHIM_DECOMP
#include "JacobiSVD3x3.h"

namespace Eigen {

template <typename T, int R, int C> class ShimFullPivQR {
  public:
    template <typename D> explicit ShimFullPivQR(const Dense<D, T, R, C> &A) {
        rows_ = A.rows();
        cols_ = A.cols();
        qr_.resize(rows_, cols_);
        for (Index j = 0; j < cols_; ++j)
            for (Index i = 0; i < rows_; ++i)
                qr_(i, j) = A(i, j);
        const Index size = std::min(rows_, cols_);
        tau_.assign(static_cast<size_t>(size), T(0));
        rowswap_.assign(static_cast<size_t>(size), 0);
        T biggest = T(0);
        const T precision = std::numeric_limits<T>::epsilon() * T(size);
        for (Index k = 0; k < size; ++k) {
            Index pr = k, pc = k;
            T best = std::abs(qr_(k, k));
            for (Index c = k; c < cols_; ++c)
                for (Index r = k; r < rows_; ++r)
                    if (std::abs(qr_(r, c)) > best) {
                        best = std::abs(qr_(r, c));
                        pr = r;
                        pc = c;
                    }
            if (k == 0)
                biggest = best;
            if (best <= biggest * precision) {
                for (Index i = k; i < size; ++i)
                    rowswap_[static_cast<size_t>(i)] = i;
                break;
            }
            rowswap_[static_cast<size_t>(k)] = pr;
            if (pr != k)
                for (Index c = k; c < cols_; ++c)
                    std::swap(qr_(k, c), qr_(pr, c));
            if (pc != k)
                for (Index r = 0; r < rows_; ++r)
                    std::swap(qr_(r, k), qr_(r, pc));
            T tail_sq = T(0);
            for (Index r = k + 1; r < rows_; ++r)
                tail_sq += qr_(r, k) * qr_(r, k);
            const T c0 = qr_(k, k);
            T beta;
            if (tail_sq <= std::numeric_limits<T>::min()) {
                tau_[static_cast<size_t>(k)] = T(0);
                beta = c0;
                for (Index r = k + 1; r < rows_; ++r)
                    qr_(r, k) = T(0);
            } else {
                beta = std::sqrt(c0 * c0 + tail_sq);
                if (c0 >= T(0))
                    beta = -beta;
                for (Index r = k + 1; r < rows_; ++r)
                    qr_(r, k) = qr_(r, k) / (c0 - beta);
                tau_[static_cast<size_t>(k)] = (beta - c0) / beta;
            }
            qr_(k, k) = beta;
            const T tk = tau_[static_cast<size_t>(k)];
            if (tk != T(0))
                for (Index c = k + 1; c < cols_; ++c) {
                    T t = T(0);
                    for (Index r = k + 1; r < rows_; ++r)
                        t += qr_(r, k) * qr_(r, c);
                    t += qr_(k, c);
                    qr_(k, c) -= tk * t;
                    for (Index r = k + 1; r < rows_; ++r)
                        qr_(r, c) -= tk * qr_(r, k) * t;
                }
        }
    }
    Matrix<T, R, R> matrixQ() const {
        Matrix<T, R, R> Q;
        Q.resize(rows_, rows_);
        Q.setIdentity();
        const Index size = std::min(rows_, cols_);
        for (Index k = size - 1; k >= 0; --k) {
            const T tk = tau_[static_cast<size_t>(k)];
            if (tk != T(0))
                for (Index c = k; c < rows_; ++c) {
                    T t = T(0);
                    for (Index r = k + 1; r < rows_; ++r)
                        t += qr_(r, k) * Q(r, c);
                    t += Q(k, c);
                    Q(k, c) -= tk * t;
                    for (Index r = k + 1; r < rows_; ++r)
                        Q(r, c) -= tk * qr_(r, k) * t;
                }
            const Index sw = rowswap_[static_cast<size_t>(k)];
            if (sw != k)
                for (Index c = 0; c < rows_; ++c)
                    std::swap(Q(k, c), Q(sw, c));
        }
        return Q;
    }

  private:
    Index rows_, cols_;
    Matrix<T, Dynamic, Dynamic> qr_;
    std::vector<T> tau_;
    std::vector<Index> rowswap_;
};

template <typename T, int N> class ShimPartialPivLU {
  public:
    template <typename D> explicit ShimPartialPivLU(const Dense<D, T, N, N> &A) {
        n_ = A.rows();
        lu_.resize(n_, n_);
        for (Index j = 0; j < n_; ++j)
            for (Index i = 0; i < n_; ++i)
                lu_(i, j) = A(i, j);
        perm_.resize(static_cast<size_t>(n_));
        for (Index i = 0; i < n_; ++i)
            perm_[static_cast<size_t>(i)] = i;
        for (Index k = 0; k < n_; ++k) {
            Index piv = k;
            T best = std::abs(lu_(k, k));
            for (Index i = k + 1; i < n_; ++i)
                if (std::abs(lu_(i, k)) > best) {
                    best = std::abs(lu_(i, k));
                    piv = i;
                }
            if (piv != k) {
                for (Index j = 0; j < n_; ++j)
                    std::swap(lu_(k, j), lu_(piv, j));
                std::swap(perm_[static_cast<size_t>(k)], perm_[static_cast<size_t>(piv)]);
            }
            if (lu_(k, k) != T(0))
                for (Index i = k + 1; i < n_; ++i)
                    lu_(i, k) /= lu_(k, k);
            for (Index i = k + 1; i < n_; ++i) {
                const T f = lu_(i, k);
                for (Index j = k + 1; j < n_; ++j)
                    lu_(i, j) -= f * lu_(k, j);
            }
        }
    }
    template <typename D, int C> Matrix<T, N, C> solve(const Dense<D, T, N, C> &B) const {
        Matrix<T, N, C> X;
        X.resize(n_, B.cols());
        for (Index c = 0; c < B.cols(); ++c) {
            for (Index i = 0; i < n_; ++i)
                X(i, c) = B(perm_[static_cast<size_t>(i)], c);
            for (Index i = 1; i < n_; ++i) {
                T s = X(i, c);
                for (Index j = 0; j < i; ++j)
                    s -= lu_(i, j) * X(j, c);
                X(i, c) = s;
            }
            for (Index i = n_ - 1; i >= 0; --i) {
                T s = X(i, c);
                for (Index j = i + 1; j < n_; ++j)
                    s -= lu_(i, j) * X(j, c);
                X(i, c) = s / lu_(i, i);
            }
        }
        return X;
    }

  private:
    Index n_;
    Matrix<T, Dynamic, Dynamic> lu_;
    std::vector<Index> perm_;
};

// least squares through Householder QR with column pivoting (3x2 use in relpose_5pt.cc:381)
template <typename T, int R, int C> class ShimColPivQR {
  public:
    template <typename D> explicit ShimColPivQR(const Dense<D, T, R, C> &A) : A_(A) {}
    template <typename D, int C2> Matrix<T, C, C2> solve(const Dense<D, T, R, C2> &b) const {
        const Index rows = A_.rows(), cols = A_.cols();
        Matrix<T, Dynamic, Dynamic> Q(rows, cols);
        std::vector<T> rhs(static_cast<size_t>(rows));
        for (Index i = 0; i < rows; ++i) {
            rhs[static_cast<size_t>(i)] = b(i, 0);
            for (Index j = 0; j < cols; ++j)
                Q(i, j) = A_(i, j);
        }
        std::vector<Index> colperm(static_cast<size_t>(cols));
        for (Index j = 0; j < cols; ++j)
            colperm[static_cast<size_t>(j)] = j;
        Matrix<T, Dynamic, Dynamic> Rm(cols, cols);
        for (Index k = 0; k < cols; ++k) {
            Index best = k;
            T bn = T(-1);
            for (Index j = k; j < cols; ++j) {
                T n = T(0);
                for (Index i = k; i < rows; ++i)
                    n += Q(i, j) * Q(i, j);
                if (n > bn) {
                    bn = n;
                    best = j;
                }
            }
            if (best != k) {
                for (Index i = 0; i < rows; ++i)
                    std::swap(Q(i, k), Q(i, best));
                for (Index i = 0; i < k; ++i)
                    std::swap(Rm(i, k), Rm(i, best));
                std::swap(colperm[static_cast<size_t>(k)], colperm[static_cast<size_t>(best)]);
            }
            T tail = T(0);
            for (Index i = k + 1; i < rows; ++i)
                tail += Q(i, k) * Q(i, k);
            const T c0 = Q(k, k);
            T beta = std::sqrt(c0 * c0 + tail);
            if (c0 >= T(0))
                beta = -beta;
            std::vector<T> v(static_cast<size_t>(rows), T(0));
            T tau = T(0);
            if (tail > std::numeric_limits<T>::min()) {
                v[static_cast<size_t>(k)] = T(1);
                for (Index i = k + 1; i < rows; ++i)
                    v[static_cast<size_t>(i)] = Q(i, k) / (c0 - beta);
                tau = (beta - c0) / beta;
            } else {
                beta = c0;
            }
            Rm(k, k) = beta;
            for (Index cc = k + 1; cc < cols; ++cc) {
                T t = T(0);
                for (Index i = k; i < rows; ++i)
                    t += v[static_cast<size_t>(i)] * Q(i, cc);
                for (Index i = k; i < rows; ++i)
                    Q(i, cc) -= tau * v[static_cast<size_t>(i)] * t;
                Rm(k, cc) = Q(k, cc);
            }
            T t = T(0);
            for (Index i = k; i < rows; ++i)
                t += v[static_cast<size_t>(i)] * rhs[static_cast<size_t>(i)];
            for (Index i = k; i < rows; ++i)
                rhs[static_cast<size_t>(i)] -= tau * v[static_cast<size_t>(i)] * t;
        }
        std::vector<T> w(static_cast<size_t>(cols));
        for (Index i = cols - 1; i >= 0; --i) {
            T s = rhs[static_cast<size_t>(i)];
            for (Index j = i + 1; j < cols; ++j)
                s -= Rm(i, j) * w[static_cast<size_t>(j)];
            w[static_cast<size_t>(i)] = s / Rm(i, i);
        }
        Matrix<T, C, C2> x;
        x.resize(cols, 1);
        for (Index j = 0; j < cols; ++j)
            x(colperm[static_cast<size_t>(j)], 0) = w[static_cast<size_t>(j)];
        return x;
    }

  private:
    Matrix<T, R, C> A_;
};

// A.selfadjointView<Lower>().llt().solve(b): Cholesky of the lower triangle.  Follows the structure of Eigen's
// unblocked LLT (per column k: x = A_kk - |A10|^2, stop if x <= 0, A21 = (A21 - A20 A10^T) / x) and of its
// triangular solves (forward: column-oriented updates; backward with L^T: row dot product, then subtract),
// with plain ascending sums.
template <typename T> class ShimLLT {
  public:
    explicit ShimLLT(const Matrix<T, Dynamic, Dynamic> &A) : L_(A) {
        const Index n = L_.rows();
        for (Index k = 0; k < n; ++k) {
            T x = L_(k, k);
            if (k > 0) {
                T sq = L_(k, 0) * L_(k, 0);
                for (Index m = 1; m < k; ++m)
                    sq += L_(k, m) * L_(k, m);
                x -= sq;
            }
            if (x <= T(0))
                break;
            x = std::sqrt(x);
            L_(k, k) = x;
            for (Index r = k + 1; r < n; ++r) {
                if (k > 0) {
                    T d = L_(r, 0) * L_(k, 0);
                    for (Index m = 1; m < k; ++m)
                        d += L_(r, m) * L_(k, m);
                    L_(r, k) -= d;
                }
                L_(r, k) /= x;
            }
        }
    }
    template <typename D, int R, int C> Matrix<T, Dynamic, 1> solve(const Dense<D, T, R, C> &b) const {
        const Index n = L_.rows();
        Matrix<T, Dynamic, 1> x(n);
        for (Index i = 0; i < n; ++i)
            x(i) = b(i);
        for (Index i = 0; i < n; ++i) {
            x(i) /= L_(i, i);
            for (Index r = i + 1; r < n; ++r)
                x(r) -= x(i) * L_(r, i);
        }
        for (Index i = n - 1; i >= 0; --i) {
            if (i + 1 < n) {
                T d = L_(i + 1, i) * x(i + 1);
                for (Index j = i + 2; j < n; ++j)
                    d += L_(j, i) * x(j);
                x(i) -= d;
            }
            x(i) /= L_(i, i);
        }
        return x;
    }

  private:
    Matrix<T, Dynamic, Dynamic> L_;
};
template <typename T> class ShimSelfAdjoint {
  public:
    template <typename D, int R, int C> explicit ShimSelfAdjoint(const Dense<D, T, R, C> &A) : A_(A) {}
    ShimLLT<T> llt() const { return ShimLLT<T>(A_); }

  private:
    Matrix<T, Dynamic, Dynamic> A_;
};
template <typename Derived, typename T, int RT, int CT>
template <int UpLo>
auto Dense<Derived, T, RT, CT>::selfadjointView() const {
    static_assert(UpLo == Lower, "only the lower view is used by the reference");
    return ShimSelfAdjoint<T>(*this);
}
template <typename Derived, typename T, int RT, int CT> auto Dense<Derived, T, RT, CT>::fullPivHouseholderQr() const {
    return ShimFullPivQR<T, RT, CT>(*this);
}
template <typename Derived, typename T, int RT, int CT> auto Dense<Derived, T, RT, CT>::partialPivLu() const {
    return ShimPartialPivLU<T, RT>(*this);
}
template <typename Derived, typename T, int RT, int CT> auto Dense<Derived, T, RT, CT>::colPivHouseholderQr() const {
    return ShimColPivQR<T, RT, CT>(*this);
}


// HouseholderQR without pivoting (householderQr().householderQ(): the full R x R orthogonal factor) - the Householder
// vectors and signs of Eigen's make_householder, applied column by column.
template <typename T, int R, int C> class ShimHouseholderQR {
  public:
    template <typename D> explicit ShimHouseholderQR(const Dense<D, T, R, C> &A) {
        rows_ = A.rows();
        cols_ = A.cols();
        qr_.resize(rows_, cols_);
        for (Index j = 0; j < cols_; ++j)
            for (Index i = 0; i < rows_; ++i)
                qr_(i, j) = A(i, j);
        const Index size = std::min(rows_, cols_);
        tau_.assign(static_cast<size_t>(size), T(0));
        for (Index k = 0; k < size; ++k) {
            T tail_sq = T(0);
            for (Index r = k + 1; r < rows_; ++r)
                tail_sq += qr_(r, k) * qr_(r, k);
            const T c0 = qr_(k, k);
            T beta;
            if (tail_sq <= std::numeric_limits<T>::min()) {
                tau_[static_cast<size_t>(k)] = T(0);
                beta = c0;
                for (Index r = k + 1; r < rows_; ++r)
                    qr_(r, k) = T(0);
            } else {
                beta = std::sqrt(c0 * c0 + tail_sq);
                if (c0 >= T(0))
                    beta = -beta;
                for (Index r = k + 1; r < rows_; ++r)
                    qr_(r, k) = qr_(r, k) / (c0 - beta);
                tau_[static_cast<size_t>(k)] = (beta - c0) / beta;
            }
            qr_(k, k) = beta;
            const T tk = tau_[static_cast<size_t>(k)];
            if (tk != T(0))
                for (Index c = k + 1; c < cols_; ++c) {
                    T t = T(0);
                    for (Index r = k + 1; r < rows_; ++r)
                        t += qr_(r, k) * qr_(r, c);
                    t += qr_(k, c);
                    qr_(k, c) -= tk * t;
                    for (Index r = k + 1; r < rows_; ++r)
                        qr_(r, c) -= tk * qr_(r, k) * t;
                }
        }
    }
    Matrix<T, R, R> householderQ() const {
        Matrix<T, R, R> Q;
        Q.resize(rows_, rows_);
        Q.setIdentity();
        const Index size = std::min(rows_, cols_);
        for (Index k = size - 1; k >= 0; --k) {
            const T tk = tau_[static_cast<size_t>(k)];
            if (tk != T(0))
                for (Index c = k; c < rows_; ++c) {
                    T t = T(0);
                    for (Index r = k + 1; r < rows_; ++r)
                        t += qr_(r, k) * Q(r, c);
                    t += Q(k, c);
                    Q(k, c) -= tk * t;
                    for (Index r = k + 1; r < rows_; ++r)
                        Q(r, c) -= tk * qr_(r, k) * t;
                }
        }
        return Q;
    }
    // x = R^-1 (Q^T b) for a square (or tall, least-squares) system: the reflectors applied to b in order, back substitution
    template <typename D, int BR, int BC> Matrix<T, C, BC> solve(const Dense<D, T, BR, BC> &b) const {
        Matrix<T, Dynamic, Dynamic> c;
        c.resize(b.rows(), b.cols());
        for (Index j = 0; j < b.cols(); ++j)
            for (Index i = 0; i < b.rows(); ++i)
                c(i, j) = b(i, j);
        const Index size = std::min(rows_, cols_);
        for (Index k = 0; k < size; ++k) {
            const T tk = tau_[static_cast<size_t>(k)];
            if (tk == T(0))
                continue;
            for (Index j = 0; j < c.cols(); ++j) {
                T t = T(0);
                for (Index r = k + 1; r < rows_; ++r)
                    t += qr_(r, k) * c(r, j);
                t += c(k, j);
                c(k, j) -= tk * t;
                for (Index r = k + 1; r < rows_; ++r)
                    c(r, j) -= tk * qr_(r, k) * t;
            }
        }
        Matrix<T, C, BC> x;
        x.resize(cols_, b.cols());
        for (Index j = 0; j < b.cols(); ++j)
            for (Index i = size - 1; i >= 0; --i) {
                T s = c(i, j);
                for (Index k = i + 1; k < size; ++k)
                    s -= qr_(i, k) * x(k, j);
                x(i, j) = s / qr_(i, i);
            }
        return x;
    }

  private:
    Index rows_, cols_;
    Matrix<T, Dynamic, Dynamic> qr_;
    std::vector<T> tau_;
};
template <typename Derived, typename T, int RT, int CT> auto Dense<Derived, T, RT, CT>::householderQr() const {
    return ShimHouseholderQR<T, RT, CT>(*this);
}

// Eigenvalues of a real general matrix: Householder reduction to upper Hessenberg form, then the implicitly shifted
// (Francis double-shift) QR iteration - the algorithm Eigen documents for EigenSolver / RealSchur (EISPACK orthes + hqr),
// without eigenvectors (the reference's template solvers ask for the eigenvalues only and recover the solutions themselves).
// Restated from the textbook form; Eigen's own shift strategy and deflation tests differ in details, so agreement with an
// Eigen-built PoseLib is at rounding level here as everywhere in this shim.
template <typename MatT> class EigenSolver {
  public:
    typedef typename MatT::Scalar T;
    explicit EigenSolver(const MatT &A, bool compute_eigenvectors = true) {
        (void)compute_eigenvectors; // never used by the units compiled against this shim
        const int n = static_cast<int>(A.rows());
        std::vector<std::vector<T>> a(static_cast<size_t>(n), std::vector<T>(static_cast<size_t>(n)));
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                a[i][j] = A(i, j);
        hessenberg(a, n);
        values_.resize(n, 1);
        hqr(a, n);
    }
    const Matrix<std::complex<T>, MatT::RowsAtCompileTime, 1> &eigenvalues() const { return values_; }

  private:
    Matrix<std::complex<T>, MatT::RowsAtCompileTime, 1> values_;

    static void hessenberg(std::vector<std::vector<T>> &a, int n) { // Householder similarity transformations
        for (int k = 0; k + 2 < n; ++k) {
            T tail_sq = T(0);
            for (int r = k + 2; r < n; ++r)
                tail_sq += a[r][k] * a[r][k];
            if (tail_sq <= std::numeric_limits<T>::min())
                continue;
            const T c0 = a[k + 1][k];
            T beta = std::sqrt(c0 * c0 + tail_sq);
            if (c0 >= T(0))
                beta = -beta;
            std::vector<T> v(static_cast<size_t>(n), T(0));
            v[k + 1] = T(1);
            for (int r = k + 2; r < n; ++r)
                v[r] = a[r][k] / (c0 - beta);
            const T tau = (beta - c0) / beta;
            for (int c = 0; c < n; ++c) { // H A   (H = I - tau v v^T)
                T t = T(0);
                for (int r = k + 1; r < n; ++r)
                    t += v[r] * a[r][c];
                for (int r = k + 1; r < n; ++r)
                    a[r][c] -= tau * v[r] * t;
            }
            for (int r = 0; r < n; ++r) { // (H A) H
                T t = T(0);
                for (int c = k + 1; c < n; ++c)
                    t += a[r][c] * v[c];
                for (int c = k + 1; c < n; ++c)
                    a[r][c] -= tau * t * v[c];
            }
            a[k + 1][k] = beta;
            for (int r = k + 2; r < n; ++r)
                a[r][k] = T(0);
        }
    }
    void hqr(std::vector<std::vector<T>> &a, int n) { // eigenvalues of an upper Hessenberg matrix
        T anorm = T(0);
        for (int i = 0; i < n; ++i)
            for (int j = std::max(i - 1, 0); j < n; ++j)
                anorm += std::abs(a[i][j]);
        int nn = n - 1;
        T t = T(0);
        T p = T(0), q = T(0), r = T(0), s = T(0), w = T(0), x = T(0), y = T(0), z = T(0);
        while (nn >= 0) {
            int its = 0, l;
            do {
                for (l = nn; l >= 1; --l) { // look for a small sub-diagonal element
                    s = std::abs(a[l - 1][l - 1]) + std::abs(a[l][l]);
                    if (s == T(0))
                        s = anorm;
                    if (std::abs(a[l][l - 1]) <= std::numeric_limits<T>::epsilon() * s) {
                        a[l][l - 1] = T(0);
                        break;
                    }
                }
                x = a[nn][nn];
                if (l == nn) { // one root
                    values_(nn, 0) = std::complex<T>(x + t, T(0));
                    --nn;
                } else {
                    y = a[nn - 1][nn - 1];
                    w = a[nn][nn - 1] * a[nn - 1][nn];
                    if (l == nn - 1) { // two roots
                        p = T(0.5) * (y - x);
                        q = p * p + w;
                        z = std::sqrt(std::abs(q));
                        x += t;
                        if (q >= T(0)) { // real pair
                            z = p + (p >= T(0) ? std::abs(z) : -std::abs(z));
                            values_(nn - 1, 0) = std::complex<T>(x + z, T(0));
                            values_(nn, 0) = std::complex<T>(z != T(0) ? x - w / z : x + z, T(0));
                        } else { // complex pair
                            values_(nn - 1, 0) = std::complex<T>(x + p, z);
                            values_(nn, 0) = std::complex<T>(x + p, -z);
                        }
                        nn -= 2;
                    } else { // no root yet: a QR step
                        if (its == 60) { // (no convergence: report what is there, as NaN so that the callers drop it)
                            for (int i = 0; i <= nn; ++i)
                                values_(i, 0) = std::complex<T>(std::numeric_limits<T>::quiet_NaN(), T(0));
                            return;
                        }
                        if (its == 10 || its == 20) { // exceptional shift
                            t += x;
                            for (int i = 0; i <= nn; ++i)
                                a[i][i] -= x;
                            s = std::abs(a[nn][nn - 1]) + std::abs(a[nn - 1][nn - 2]);
                            y = x = T(0.75) * s;
                            w = T(-0.4375) * s * s;
                        }
                        ++its;
                        int m;
                        for (m = nn - 2; m >= l; --m) { // two consecutive small sub-diagonal elements
                            z = a[m][m];
                            r = x - z;
                            s = y - z;
                            p = (r * s - w) / a[m + 1][m] + a[m][m + 1];
                            q = a[m + 1][m + 1] - z - r - s;
                            r = a[m + 2][m + 1];
                            s = std::abs(p) + std::abs(q) + std::abs(r);
                            p /= s;
                            q /= s;
                            r /= s;
                            if (m == l)
                                break;
                            const T u = std::abs(a[m][m - 1]) * (std::abs(q) + std::abs(r));
                            const T v = std::abs(p) * (std::abs(a[m - 1][m - 1]) + std::abs(z) + std::abs(a[m + 1][m + 1]));
                            if (u <= std::numeric_limits<T>::epsilon() * v)
                                break;
                        }
                        for (int i = m + 2; i <= nn; ++i) {
                            a[i][i - 2] = T(0);
                            if (i != m + 2)
                                a[i][i - 3] = T(0);
                        }
                        for (int k = m; k <= nn - 1; ++k) { // double QR step on rows l..nn and columns m..nn
                            if (k != m) {
                                p = a[k][k - 1];
                                q = a[k + 1][k - 1];
                                r = T(0);
                                if (k != nn - 1)
                                    r = a[k + 2][k - 1];
                                if ((x = std::abs(p) + std::abs(q) + std::abs(r)) != T(0)) {
                                    p /= x;
                                    q /= x;
                                    r /= x;
                                }
                            }
                            const T sq = std::sqrt(p * p + q * q + r * r);
                            if ((s = (p >= T(0) ? sq : -sq)) != T(0)) {
                                if (k == m) {
                                    if (l != m)
                                        a[k][k - 1] = -a[k][k - 1];
                                } else {
                                    a[k][k - 1] = -s * x;
                                }
                                p += s;
                                x = p / s;
                                y = q / s;
                                z = r / s;
                                q /= p;
                                r /= p;
                                for (int j = k; j <= nn; ++j) { // row modification
                                    p = a[k][j] + q * a[k + 1][j];
                                    if (k != nn - 1) {
                                        p += r * a[k + 2][j];
                                        a[k + 2][j] -= p * z;
                                    }
                                    a[k + 1][j] -= p * y;
                                    a[k][j] -= p * x;
                                }
                                const int mmin = nn < k + 3 ? nn : k + 3;
                                for (int i = l; i <= mmin; ++i) { // column modification
                                    p = x * a[i][k] + y * a[i][k + 1];
                                    if (k != nn - 1) {
                                        p += z * a[i][k + 2];
                                        a[i][k + 2] -= p * r;
                                    }
                                    a[i][k + 1] -= p * q;
                                    a[i][k] -= p;
                                }
                            }
                        }
                    }
                }
            } while (l < nn - 1);
        }
    }
};

// 3x3 SVD: Eigen's two-sided Jacobi iteration in Eigen's operation order (src/JacobiSVD3x3.h says why the order
// matters: the sign of every LO-refined fundamental matrix hangs on it).  Only the shape the reference needs.
template <typename MatT> class JacobiSVD {
  public:
    typedef typename MatT::Scalar T;
    JacobiSVD(const MatT &A, unsigned int = 0) {
        T a[3][3], u[3][3], v[3][3], s[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                a[i][j] = A(i, j);
        eigen_shim_detail::jacobi_svd3(a, u, s, v);
        for (int i = 0; i < 3; ++i) {
            s_(i) = s[i];
            for (int j = 0; j < 3; ++j)
                U_(i, j) = u[i][j], V_(i, j) = v[i][j];
        }
    }
    const MatT &matrixU() const { return U_; }
    const MatT &matrixV() const { return V_; }
    const Matrix<T, 3, 1> &singularValues() const { return s_; }

  private:
    MatT U_, V_;
    Matrix<T, 3, 1> s_;
};

template <typename T, int N> class DiagonalMatrix {
  public:
    DiagonalMatrix() {}
    DiagonalMatrix(T a, T b, T c) {
        d_(0) = a, d_(1) = b, d_(2) = c;
    }
    Matrix<T, N, 1> &diagonal() { return d_; }
    const Matrix<T, N, 1> &diagonal() const { return d_; }
    operator Matrix<T, N, N>() const {
        Matrix<T, N, N> m;
        for (int i = 0; i < N; ++i)
            m(i, i) = d_(i);
        return m;
    }

  private:
    Matrix<T, N, 1> d_;
};
template <typename T, int N, typename A, int R>
Matrix<T, R, N> operator*(const Dense<A, T, R, N> &a, const DiagonalMatrix<T, N> &d) {
    Matrix<T, R, N> r;
    r.resize(a.rows(), a.cols());
    for (Index j = 0; j < a.cols(); ++j)
        for (Index i = 0; i < a.rows(); ++i)
            r(i, j) = a(i, j) * d.diagonal()(j);
    return r;
}
template <typename T, int N, typename B, int C>
Matrix<T, N, C> operator*(const DiagonalMatrix<T, N> &d, const Dense<B, T, N, C> &b) {
    Matrix<T, N, C> r;
    r.resize(b.rows(), b.cols());
    for (Index j = 0; j < b.cols(); ++j)
        for (Index i = 0; i < b.rows(); ++i)
            r(i, j) = d.diagonal()(i) * b(i, j);
    return r;
}


} // namespace Eigen
#endif
