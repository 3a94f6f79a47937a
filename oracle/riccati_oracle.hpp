// ORACLE - TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the product path
// (trajopt_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// PARITY UNPINNED: the reference (hanyas/trajopt) ships no tests, golden vectors or fixtures, its
// native core needs Armadillo + OpenBLAS (absent here, unbuildable) and its Python needs autograd +
// gym (absent).  This file is therefore a line-by-line CPU restatement of the reference algorithm,
// pinned only by analytic known-answer tests (tests/test_oracle_*.py) and by an independent NumPy
// mirror (oracle/numpy_mirror.py).
//
// Restates:
//   * backward_pass          /root/reference/trajopt/ilqr/src/util.cpp:83-182   (row A1 of SURVEY 8a)
//   * Riccati.backward_pass  /root/reference/trajopt/riccati/riccati.py:100-126 (row A9)
//
// Third-party arithmetic the reference calls and that is not under /root/reference:
//   Armadillo (header-only C++, version unpinned, README.md:24-30): operator*, .t(), is_sympd(),
//   inv(), eye(), join_vert().  Restated here from Armadillo's published algorithms:
//     - 3-term products A*B*C are evaluated as (A*B)*C when rows(A)*cols(B) <= rows(B)*cols(C),
//       else A*(B*C)   (glue_times 3-operand ordering heuristic);
//     - X.is_sympd(): tol = 100*eps*||X||_F; false unless ||X - X^T||_inf / ||X||_inf <= tol;
//       then X.diag() -= tol and plain Cholesky must succeed (every pivot > 0);
//     - inv(X) of a matrix that passed is_sympd(): solved here through the Cholesky factor of X
//       (Armadillo >= 9.9 takes the same sympd route inside inv(); older versions use LU - the two
//       agree to round-off * cond(X)).
//
// All matrices are column-major, time is the slowest axis:  M(i, j, t) = M[i + rows*j + rows*cols*t]
// exactly as the reference's f_style arrays (util.cpp:10,21).
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace oracle {

// C(ra x cb) = op(A) * B with A given as (ra x ka) or transposed; tiny dense helpers, scalar type S.
template <class S>
struct M {
    int r = 0, c = 0;
    std::vector<S> a;
    M() {}
    M(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, S(0)) {}
    S& operator()(int i, int j) { return a[(size_t)i + (size_t)r * j]; }
    const S& operator()(int i, int j) const { return a[(size_t)i + (size_t)r * j]; }
};

template <class S>
static M<S> load(const double* p, int r, int c) {
    M<S> m(r, c);
    for (int k = 0; k < r * c; ++k) m.a[k] = (S)p[k];
    return m;
}
template <class S>
static void store(double* p, const M<S>& m) {
    for (int k = 0; k < m.r * m.c; ++k) p[k] = (double)m.a[k];
}
template <class S>
static M<S> mul(const M<S>& A, const M<S>& B) {
    M<S> C(A.r, B.c);
    for (int j = 0; j < B.c; ++j)
        for (int i = 0; i < A.r; ++i) {
            S acc = 0;
            for (int k = 0; k < A.c; ++k) acc += A(i, k) * B(k, j);
            C(i, j) = acc;
        }
    return C;
}
template <class S>
static M<S> tr(const M<S>& A) {
    M<S> C(A.c, A.r);
    for (int j = 0; j < A.c; ++j)
        for (int i = 0; i < A.r; ++i) C(j, i) = A(i, j);
    return C;
}
template <class S>
static M<S> add(const M<S>& A, const M<S>& B) {
    M<S> C(A.r, A.c);
    for (size_t k = 0; k < C.a.size(); ++k) C.a[k] = A.a[k] + B.a[k];
    return C;
}
template <class S>
static M<S> scale(const M<S>& A, S s) {
    M<S> C(A.r, A.c);
    for (size_t k = 0; k < C.a.size(); ++k) C.a[k] = s * A.a[k];
    return C;
}
// Armadillo's ordering for a 3-operand product (glue_times::apply for 3 matrices).
template <class S>
static M<S> mul3(const M<S>& A, const M<S>& B, const M<S>& C) {
    if ((long)A.r * B.c <= (long)B.r * C.c) return mul(mul(A, B), C);
    return mul(A, mul(B, C));
}

template <class S>
static M<S> eye_m(int n) {
    M<S> I(n, n);
    for (int i = 0; i < n; ++i) I(i, i) = S(1);
    return I;
}

// Plain (unpivoted) lower Cholesky in place; false if a pivot is not > 0 (NaN included).
template <class S>
static bool chol_lower(M<S>& X) {
    const int n = X.r;
    for (int j = 0; j < n; ++j) {
        S d = X(j, j);
        for (int k = 0; k < j; ++k) d -= X(j, k) * X(j, k);
        if (!(d > S(0))) return false;
        d = std::sqrt(d);
        X(j, j) = d;
        for (int i = j + 1; i < n; ++i) {
            S s = X(i, j);
            for (int k = 0; k < j; ++k) s -= X(i, k) * X(j, k);
            X(i, j) = s / d;
        }
    }
    return true;
}

// Armadillo Base::is_sympd() with its default tolerance (see header comment).
template <class S>
static bool is_sympd(const M<S>& X) {
    const int n = X.r;
    if (n == 0 || X.c != n) return false;
    S fro = 0;
    for (S e : X.a) fro += e * e;
    fro = std::sqrt(fro);
    const S tol = S(100) * (S)std::numeric_limits<double>::epsilon() * fro;
    S norm_a = 0, norm_d = 0;  // inf-norms of X and of X - X^T
    for (int i = 0; i < n; ++i) {
        S ra = 0, rd = 0;
        for (int j = 0; j < n; ++j) {
            ra += std::fabs(X(i, j));
            rd += std::fabs(X(i, j) - X(j, i));
        }
        if (ra > norm_a || ra != ra) norm_a = ra;
        if (rd > norm_d || rd != rd) norm_d = rd;
    }
    if (norm_a != S(0)) {
        if (!((norm_d / norm_a) <= tol)) return false;
    }
    M<S> Y = X;
    for (int i = 0; i < n; ++i) Y(i, i) -= tol;
    return chol_lower(Y);
}

// X^{-1} * R through the Cholesky factor of X (X has passed is_sympd()).
template <class S>
static M<S> sympd_solve(const M<S>& X, const M<S>& R) {
    M<S> L = X;
    chol_lower(L);
    const int n = X.r;
    M<S> Y = R;
    for (int c = 0; c < R.c; ++c) {
        for (int i = 0; i < n; ++i) {
            S s = Y(i, c);
            for (int k = 0; k < i; ++k) s -= L(i, k) * Y(k, c);
            Y(i, c) = s / L(i, i);
        }
        for (int i = n - 1; i >= 0; --i) {
            S s = Y(i, c);
            for (int k = i + 1; k < n; ++k) s -= L(k, i) * Y(k, c);
            Y(i, c) = s / L(i, i);
        }
    }
    return Y;
}

// General inverse by Gauss-Jordan with partial pivoting (np.linalg.inv stand-in for riccati.py:118).
template <class S>
static bool gj_solve(const M<S>& X, const M<S>& R, M<S>& out) {
    const int n = X.r;
    M<S> A = X;
    out = R;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A(i, k)) > std::fabs(A(piv, k))) piv = i;
        if (A(piv, k) == S(0)) return false;
        if (piv != k) {
            for (int j = 0; j < n; ++j) std::swap(A(k, j), A(piv, j));
            for (int j = 0; j < out.c; ++j) std::swap(out(k, j), out(piv, j));
        }
        for (int i = k + 1; i < n; ++i) {
            S f = A(i, k) / A(k, k);
            for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j);
            for (int j = 0; j < out.c; ++j) out(i, j) -= f * out(k, j);
        }
    }
    for (int c = 0; c < out.c; ++c)
        for (int i = n - 1; i >= 0; --i) {
            S s = out(i, c);
            for (int k = i + 1; k < n; ++k) s -= A(i, k) * out(k, c);
            out(i, c) = s / A(i, i);
        }
    return true;
}

// ---------------------------------------------------------------------------------------------
// A1: util.cpp:83-182.  Output buffers must be zero-filled by the caller (the reference leaves
// slices below a divergence uninitialised; SURVEY 8a quirks 1-2: dV is zeroed here).
// Returns the reference's `_diverge` int (time index of the failure, 0 when none) and writes the
// unambiguous flag (1 = a step failed the PD test) to *failed.
// ---------------------------------------------------------------------------------------------
template <class S>
int backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                  const double* Cxu, const double* A, const double* B, double lmbda, int reg, int n,
                  int m, int T, double* Qxx, double* Qux, double* Quu, double* qx, double* qu,
                  double* V, double* v, double* dV, double* K, double* kff, int* failed) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    int diverge = 0;  // util.cpp:120
    if (failed) *failed = 0;
    S dV0 = 0, dV1 = 0;
    // util.cpp:123-124
    M<S> Vn = load<S>(Cxx + nn * T, n, n);
    M<S> vn = load<S>(cx + (size_t)n * T, n, 1);
    if (V) store(V + nn * T, Vn);
    if (v) store(v + (size_t)n * T, vn);

    for (int i = T - 1; i >= 0; --i) {  // util.cpp:126
        M<S> Ai = load<S>(A + nn * i, n, n), Bi = load<S>(B + nm * i, n, m);
        M<S> At = tr(Ai), Bt = tr(Bi);
        M<S> Cxxi = load<S>(Cxx + nn * i, n, n), Cuui = load<S>(Cuu + mm * i, m, m);
        M<S> Cxui = load<S>(Cxu + nm * i, n, m);
        M<S> cxi = load<S>(cx + (size_t)n * i, n, 1), cui = load<S>(cu + (size_t)m * i, m, 1);

        M<S> Qxx_i = add(Cxxi, mul3(At, Vn, Ai));      // :128
        M<S> Quu_i = add(Cuui, mul3(Bt, Vn, Bi));      // :129
        M<S> Qux_i = tr(add(Cxui, mul3(At, Vn, Bi)));  // :130
        M<S> qu_i = add(cui, mul(Bt, vn));             // :132
        M<S> qx_i = add(cxi, mul(At, vn));             // :133
        if (Qxx) store(Qxx + nn * i, Qxx_i);
        if (Quu) store(Quu + mm * i, Quu_i);
        if (Qux) store(Qux + nm * i, Qux_i);
        if (qu) store(qu + (size_t)m * i, qu_i);
        if (qx) store(qx + (size_t)n * i, qx_i);

        M<S> Vreg = Vn;  // :135-137
        if (reg == 2)
            for (int d = 0; d < n; ++d) Vreg(d, d) += (S)lmbda;
        M<S> Qux_reg = tr(add(Cxui, mul3(At, Vreg, Bi)));  // :139
        M<S> Quu_reg = add(Cuui, mul3(Bt, Vreg, Bi));      // :141
        if (reg == 1)
            for (int d = 0; d < m; ++d) Quu_reg(d, d) += (S)lmbda;  // :142-143

        if (!is_sympd(Quu_reg)) {  // :145-148
            diverge = i;
            if (failed) *failed = 1;
            break;
        }
        M<S> K_i = scale(sympd_solve(Quu_reg, Qux_reg), S(-1));  // :150-151
        M<S> kff_i = scale(sympd_solve(Quu_reg, qu_i), S(-1));   // :152
        if (K) store(K + nm * i, K_i);
        if (kff) store(kff + (size_t)m * i, kff_i);

        M<S> Kt = tr(K_i), kt = tr(kff_i);
        dV0 += mul(kt, qu_i)(0, 0);                        // :154
        dV1 += S(0.5) * mul3(kt, Quu_i, kff_i)(0, 0);
        // :156-157
        M<S> v_i = add(add(add(qx_i, mul3(Kt, Quu_i, kff_i)), mul(Kt, qu_i)), mul(tr(Qux_i), kff_i));
        // :159-161
        M<S> V_i = add(add(add(Qxx_i, mul3(Kt, Quu_i, K_i)), mul(Kt, Qux_i)), mul(tr(Qux_i), K_i));
        V_i = scale(add(V_i, tr(V_i)), S(0.5));
        if (V) store(V + nn * i, V_i);
        if (v) store(v + (size_t)n * i, v_i);
        Vn = V_i;
        vn = v_i;
    }
    if (dV) {
        dV[0] = (double)dV0;
        dV[1] = (double)dV1;
    }
    return diverge;
}

// ---------------------------------------------------------------------------------------------
// A9: riccati.py:100-126 ("no-1/2, affine-c" time-varying LQR).  `literal_elementwise` = 1 keeps
// the reference's elementwise `Qux.T * K` at :123 (an outer product, only defined here for m == 1);
// 0 uses the intended matrix product Qux^T K (the form the GPS core uses, gps/src/util.cpp:344).
// ---------------------------------------------------------------------------------------------
template <class S>
int riccati_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu,
                          const double* Cxu, const double* A, const double* B, const double* c,
                          int n, int m, int T, int literal_elementwise, double* V, double* v,
                          double* K, double* kff) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    if (literal_elementwise && m != 1) return -1;
    M<S> Vn = load<S>(Cxx + nn * T, n, n);            // :104
    M<S> vn = load<S>(cx + (size_t)n * T, n, 1);      // :105
    store(V + nn * T, Vn);
    store(v + (size_t)n * T, vn);
    for (int t = T - 1; t >= 0; --t) {  // :107
        M<S> At = tr(load<S>(A + nn * t, n, n)), Bt = tr(load<S>(B + nm * t, n, m));
        M<S> Ai = tr(At), Bi = tr(Bt);
        M<S> ci = load<S>(c + (size_t)n * t, n, 1);
        // numpy evaluates a @ b @ c left to right
        M<S> Qxx = add(load<S>(Cxx + nn * t, n, n), mul(mul(At, Vn), Ai));        // :108
        M<S> Quu = add(load<S>(Cuu + mm * t, m, m), mul(mul(Bt, Vn), Bi));        // :109
        M<S> Qux = add(tr(load<S>(Cxu + nm * t, n, m)), mul(mul(Bt, Vn), Ai));    // :110
        M<S> qx = add(add(load<S>(cx + (size_t)n * t, n, 1), mul(mul(scale(At, S(2)), Vn), ci)),
                      mul(At, vn));                                               // :112-113
        M<S> qu = add(add(load<S>(cu + (size_t)m * t, m, 1), mul(mul(scale(Bt, S(2)), Vn), ci)),
                      mul(Bt, vn));                                               // :115-116
        M<S> Kt, kt;
        if (!gj_solve(Quu, Qux, Kt)) return t + 1;  // np.linalg.inv raises LinAlgError (:118)
        gj_solve(Quu, qu, kt);
        M<S> K_t = scale(Kt, S(-1));       // :120
        M<S> kff_t = scale(kt, S(-0.5));   // :121
        M<S> V_t = add(Qxx, mul(tr(Qux), K_t));                       // :123 (see header)
        M<S> v_t = add(qx, scale(mul(tr(Qux), kff_t), S(2)));         // :124  (2 kff^T Qux)^T
        store(K + nm * t, K_t);
        store(kff + (size_t)m * t, kff_t);
        store(V + nn * t, V_t);
        store(v + (size_t)n * t, v_t);
        Vn = V_t;
        vn = v_t;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// the max-entropy (GPS) backward pass, /root/reference/trajopt/gps/src/util.cpp:272-374.
// Same recursion skeleton as A1/A9 with the affine dynamics term c, the process noise sigma_dyn, the constant c0
// and a per-step temperature alpha:  Q.. = -(C.. + F'V'F)/alpha  (negative definite Quu expected; the reference
// flags a step whose Quu passes is_sympd(), :329-332), K = -Quu^-1 Qux, kff = -1/2 Quu^-1 qu,
// sigma_ctl = sym(-1/2 Quu^-1), V = sym(-alpha (Qxx + Qux'K)), v = -alpha (qx + 2 Qux'kff),
// v0 = -alpha (1/2 qu'kff + q0 + 1/2 (m log 2pi - log det(-2 Quu))).
// Output buffers must be zero-filled by the caller.  Returns the reference's `_diverge` (0 = none, else the failing
// time index); *failed gets the unambiguous flag; a singular Quu (Armadillo's inv() throws) returns -(t + 1).
// ---------------------------------------------------------------------------------------------
template <class S>
static S det_gj(const M<S>& X) {
    const int n = X.r;
    M<S> A = X;
    S d = 1;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A(i, k)) > std::fabs(A(piv, k))) piv = i;
        if (A(piv, k) == S(0)) return S(0);
        if (piv != k) {
            for (int j = 0; j < n; ++j) std::swap(A(k, j), A(piv, j));
            d = -d;
        }
        d *= A(k, k);
        for (int i = k + 1; i < n; ++i) {
            S f = A(i, k) / A(k, k);
            for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j);
        }
    }
    return d;
}

template <class S>
int gps_backward_pass(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                      const double* c0, const double* A, const double* B, const double* c, const double* sigma_dyn,
                      const double* alpha, int n, int m, int T, double* Qxx, double* Qux, double* Quu, double* qx,
                      double* qu, double* q0, double* V, double* v, double* v0, double* K, double* kff,
                      double* sigma_ctl, int* failed) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    *failed = 0;
    M<S> Vn = load<S>(Cxx + nn * T, n, n);          // :318
    M<S> vn = load<S>(cx + (size_t)n * T, n, 1);    // :319
    S v0n = (S)c0[T];                               // :320
    store(V + nn * T, Vn);
    store(v + (size_t)n * T, vn);
    v0[T] = (double)v0n;
    for (int i = T - 1; i >= 0; --i) {  // :322
        const S al = (S)alpha[i];
        M<S> Ai = load<S>(A + nn * i, n, n), Bi = load<S>(B + nm * i, n, m), ci = load<S>(c + (size_t)n * i, n, 1);
        M<S> At = tr(Ai), Bt = tr(Bi);
        M<S> Qxx_i = scale(add(load<S>(Cxx + nn * i, n, n), mul3(At, Vn, Ai)), S(-1) / al);          // :324
        M<S> Quu_i = scale(add(load<S>(Cuu + mm * i, m, m), mul3(Bt, Vn, Bi)), S(-1) / al);          // :325
        M<S> Qux_i = scale(tr(add(load<S>(Cxu + nm * i, n, m), mul3(At, Vn, Bi))), S(-1) / al);      // :326
        M<S> Vc = mul(Vn, ci);
        M<S> qu_i = scale(add(add(load<S>(cu + (size_t)m * i, m, 1), scale(mul(Bt, Vc), S(2))), mul(Bt, vn)),
                          S(-1) / al);                                                                 // :328
        M<S> qx_i = scale(add(add(load<S>(cx + (size_t)n * i, n, 1), scale(mul(At, Vc), S(2))), mul(At, vn)),
                          S(-1) / al);                                                                 // :329
        S trVS = 0;  // trace(V' sigma_dyn)
        {
            M<S> Sg = load<S>(sigma_dyn + nn * i, n, n);
            for (int r = 0; r < n; ++r)
                for (int k = 0; k < n; ++k) trVS += Vn(r, k) * Sg(k, r);
        }
        S cVc = 0, vc = 0;
        for (int r = 0; r < n; ++r) {
            cVc += ci(r, 0) * Vc(r, 0);
            vc += vn(r, 0) * ci(r, 0);
        }
        const S q0_i = -((S)c0[i] + v0n + cVc + trVS + vc) / al;                                       // :330-331
        store(Qxx + nn * i, Qxx_i);
        store(Quu + mm * i, Quu_i);
        store(Qux + nm * i, Qux_i);
        store(qu + (size_t)m * i, qu_i);
        store(qx + (size_t)n * i, qx_i);
        q0[i] = (double)q0_i;
        if (is_sympd(Quu_i)) {  // :333-336
            *failed = 1;
            return i;
        }
        M<S> Quu_inv, I(m, m);
        for (int k = 0; k < m; ++k) I(k, k) = S(1);
        if (!gj_solve(Quu_i, I, Quu_inv)) return -(i + 1);  // :338 inv() throws
        M<S> K_i = scale(mul(Quu_inv, Qux_i), S(-1));        // :339
        M<S> kff_i = scale(mul(Quu_inv, qu_i), S(-0.5));     // :340
        M<S> sg = scale(Quu_inv, S(-0.5));                   // :342
        sg = scale(add(tr(sg), sg), S(0.5));                 // :343
        M<S> V_i = scale(add(Qxx_i, mul(tr(Qux_i), K_i)), -al);          // :348
        V_i = scale(add(V_i, tr(V_i)), S(0.5));                          // :349
        M<S> v_i = scale(add(qx_i, scale(mul(tr(Qux_i), kff_i), S(2))), -al);   // :351
        S qk = 0;
        for (int k = 0; k < m; ++k) qk += qu_i(k, 0) * kff_i(k, 0);
        const S ldet = std::log(det_gj(scale(Quu_i, S(-2))));
        const S v0_i = -al * (S(0.5) * qk + q0_i + S(0.5) * (S(m) * std::log(S(2) * (S)M_PI) - ldet));   // :352-353
        store(K + nm * i, K_i);
        store(kff + (size_t)m * i, kff_i);
        store(sigma_ctl + mm * i, sg);
        store(V + nn * i, V_i);
        store(v + (size_t)n * i, v_i);
        v0[i] = (double)v0_i;
        Vn = V_i;
        vn = v_i;
        v0n = v0_i;
    }
    return 0;
}

}  // namespace oracle
