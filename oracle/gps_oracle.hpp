// TEST INFRASTRUCTURE ONLY (never used by the product path): CPU restatement of the remaining functions of
// the reference's GPS native module, /root/reference/trajopt/gps/src/util.cpp:83-269:
//   kl_divergence :83-121, quad_expectation :123-134, augment_cost :136-193, forward_pass :195-269.
// Parity unpinned: the reference holds no vectors for them and cannot be built here (Armadillo absent); pinned by
// closed-form identities and the NumPy mirror in tests/test_oracle_known_answers.py.
// inv_sympd(X) is restated as the Cholesky-based inverse (sympd_solve against the identity), det() by elimination.
#pragma once
#include "riccati_oracle.hpp"

namespace oracle {

template <class S>
static M<S> eye(int n) {
    M<S> I(n, n);
    for (int i = 0; i < n; ++i) I(i, i) = S(1);
    return I;
}
template <class S>
static S trace_of(const M<S>& A) {
    S t = 0;
    for (int i = 0; i < A.r; ++i) t += A(i, i);
    return t;
}
// inv_sympd: false when X is not positive definite (Armadillo throws)
template <class S>
static bool inv_sympd(const M<S>& X, M<S>& out) {
    M<S> L = X;
    if (!chol_lower(L)) return false;
    out = sympd_solve(X, eye<S>(X.r));
    return true;
}

// util.cpp:83-121.  Returns 0, or t + 1 when q_sigma_ctl[t] is not positive definite.
template <class S>
int gps_kl_divergence(const double* pK, const double* pk, const double* pS, const double* qK, const double* qk,
                      const double* qS, const double* mu_x, const double* sigma_x, int n, int m, int T, double* kl) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    for (int i = 0; i < T; ++i) {
        M<S> qSi = load<S>(qS + mm * i, m, m), pSi = load<S>(pS + mm * i, m, m), lam;
        if (!inv_sympd(qSi, lam)) return i + 1;                                           // :102
        M<S> dK = add(load<S>(qK + nm * i, m, n), scale(load<S>(pK + nm * i, m, n), S(-1)));
        M<S> dk = add(scale(load<S>(qk + (size_t)m * i, m, 1), S(-1)), load<S>(pk + (size_t)m * i, m, 1));
        M<S> diff_K = mul3(tr(dK), lam, dK);                                              // :104
        M<S> diff_crs = mul3(tr(dK), lam, dk);                                            // :105
        M<S> diff_kff = mul3(tr(dk), lam, dk);                                            // :106
        M<S> mu = load<S>(mu_x + (size_t)n * i, n, 1), sx = load<S>(sigma_x + nn * i, n, n);
        using std::log;
        S v = S(0.5) * log(det_gj(qSi) / det_gj(pSi)) + S(0.5) * trace_of(mul(lam, pSi)) - S(0.5) * S(m) +
              S(0.5) * trace_of(mul(diff_K, sx)) + S(0.5) * mul3(tr(mu), diff_K, mu)(0, 0) -
              mul(tr(mu), diff_crs)(0, 0) + S(0.5) * diff_kff(0, 0);                      // :108-114
        kl[i] = (double)v;
    }
    return 0;
}

// util.cpp:123-134
template <class S>
double gps_quad_expectation(const double* mu_, const double* sigma_, const double* Q_, const double* q_, double q0,
                            int n) {
    M<S> mu = load<S>(mu_, n, 1), sg = load<S>(sigma_, n, n), Q = load<S>(Q_, n, n), q = load<S>(q_, n, 1);
    S r = mul3(tr(mu), Q, mu)(0, 0) + mul(tr(mu), q)(0, 0) + S(q0) + trace_of(mul(Q, sg));
    return (double)r;
}

// util.cpp:136-193.  Returns 0, or t + 1 when sigma_ctl[t] is not positive definite.
template <class S>
int gps_augment_cost(const double* Cxx, const double* cx, const double* Cuu, const double* cu, const double* Cxu,
                     const double* c0, const double* K, const double* kff, const double* sigma_ctl,
                     const double* alpha, int n, int m, int T, double* agCxx, double* agcx, double* agCuu,
                     double* agcu, double* agCxu, double* agc0) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const S two_pi = S(2) * S(3.14159265358979323846264338327950288L);
    for (int i = 0; i < T; ++i) {
        M<S> sg = load<S>(sigma_ctl + mm * i, m, m), lam;
        if (!inv_sympd(sg, lam)) return i + 1;                                            // :164
        const S al = (S)alpha[i];
        M<S> Ki = load<S>(K + nm * i, m, n), ki = load<S>(kff + (size_t)m * i, m, 1), Kt = tr(Ki);
        // scalar * matrix products: Armadillo folds the scalars into the product (0.5 * alpha) * (K' lam K)
        store(agCxx + nn * i, add(load<S>(Cxx + nn * i, n, n), scale(mul3(Kt, lam, Ki), S(0.5) * al)));    // :166
        store(agCuu + mm * i, add(load<S>(Cuu + mm * i, m, m), scale(lam, S(0.5) * al)));                  // :167
        store(agCxu + nm * i, add(load<S>(Cxu + nm * i, n, m), scale(mul(Kt, lam), S(-0.5) * al)));        // :168
        store(agcx + (size_t)n * i, add(load<S>(cx + (size_t)n * i, n, 1), scale(mul3(Kt, lam, ki), al))); // :169
        store(agcu + (size_t)m * i, add(load<S>(cu + (size_t)m * i, m, 1), scale(mul(lam, ki), -al)));     // :170
        using std::log;
        S v = (S)c0[i] + S(0.5) * al * log(det_gj(scale(sg, two_pi))) +
              S(0.5) * al * mul3(tr(ki), lam, ki)(0, 0);                                  // :171-172
        agc0[i] = (double)v;
    }
    for (size_t k = 0; k < nn; ++k) agCxx[nn * T + k] = Cxx[nn * T + k];                   // :176-181
    for (int k = 0; k < n; ++k) agcx[(size_t)n * T + k] = cx[(size_t)n * T + k];
    for (size_t k = 0; k < mm; ++k) agCuu[mm * T + k] = Cuu[mm * T + k];
    for (int k = 0; k < m; ++k) agcu[(size_t)m * T + k] = cu[(size_t)m * T + k];
    for (size_t k = 0; k < nm; ++k) agCxu[nm * T + k] = Cxu[nm * T + k];
    agc0[T] = c0[T];
    return 0;
}

// util.cpp:195-269.  The last slice of sigma_xu outside its state block and the last action rows of mu_xu are
// uninitialised / zero in the reference (:253-256); zero here.
template <class S>
void gps_forward_pass(const double* mu_x0, const double* sigma_x0, const double* A, const double* B, const double* c,
                      const double* sigma_dyn, const double* K, const double* kff, const double* sigma_ctl, int n,
                      int m, int T, double* mu_x, double* sigma_x, double* mu_u, double* sigma_u, double* mu_xu,
                      double* sigma_xu) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    const int w = n + m;
    const size_t ww = (size_t)w * w;
    auto sym = [](const M<S>& X) { return scale(add(X, tr(X)), S(0.5)); };
    M<S> mx = load<S>(mu_x0, n, 1), sx = load<S>(sigma_x0, n, n);
    store(mu_x, mx);
    store(sigma_x, sx);
    for (int i = 0; i < T; ++i) {
        M<S> Ki = load<S>(K + nm * i, m, n), Ai = load<S>(A + nn * i, n, n), Bi = load<S>(B + nm * i, n, m);
        M<S> mu = add(mul(Ki, mx), load<S>(kff + (size_t)m * i, m, 1));                               // :229
        M<S> su = sym(add(load<S>(sigma_ctl + mm * i, m, m), mul3(Ki, sx, tr(Ki))));                  // :232-233
        M<S> sxk = mul(sx, tr(Ki)), ksx = mul(Ki, sx);
        M<S> J(w, w);                                                                                 // :237-238
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) J(a, b) = sx(a, b);
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < m; ++b) { J(a, n + b) = sxk(a, b); J(n + b, a) = ksx(b, a); }
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) J(n + a, n + b) = su(a, b);
        J = sym(J);                                                                                   // :239
        M<S> mxu(w, 1);
        for (int a = 0; a < n; ++a) mxu(a, 0) = mx(a, 0);
        for (int a = 0; a < m; ++a) mxu(n + a, 0) = mu(a, 0);
        M<S> F(n, w);
        for (int a = 0; a < n; ++a) {
            for (int b = 0; b < n; ++b) F(a, b) = Ai(a, b);
            for (int b = 0; b < m; ++b) F(a, n + b) = Bi(a, b);
        }
        M<S> sn = sym(add(load<S>(sigma_dyn + nn * i, n, n), mul3(F, J, tr(F))));                     // :246-248
        M<S> mn = add(mul(F, mxu), load<S>(c + (size_t)n * i, n, 1));                                 // :251
        store(mu_u + (size_t)m * i, mu);
        store(sigma_u + mm * i, su);
        store(mu_xu + (size_t)w * i, mxu);
        store(sigma_xu + ww * i, J);
        store(mu_x + (size_t)n * (i + 1), mn);
        store(sigma_x + nn * (i + 1), sn);
        mx = mn;
        sx = sn;
    }
    M<S> last(w, 1), Jl(w, w);                                                                        // :253-256
    for (int a = 0; a < n; ++a) {
        last(a, 0) = mx(a, 0);
        for (int b = 0; b < n; ++b) Jl(a, b) = sx(a, b);
    }
    store(mu_xu + (size_t)w * T, last);
    store(sigma_xu + ww * T, Jl);
}

}  // namespace oracle
