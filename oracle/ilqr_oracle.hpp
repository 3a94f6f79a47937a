// ORACLE - TEST INFRASTRUCTURE ONLY (see riccati_oracle.hpp for the rules).  PARITY UNPINNED.
//
// CPU restatement, for ONE problem, of
//   iLQR.forward_pass          /root/reference/trajopt/ilqr/ilqr.py:89-102       (row A3)
//   LinearControl.action       /root/reference/trajopt/ilqr/objects.py:137-139
//   dyn.taylor_expansion       /root/reference/trajopt/ilqr/objects.py:114-117   (row A4)
//   cost.taylor_expansion      /root/reference/trajopt/ilqr/objects.py:68-78     (row A5)
//   iLQR.run                   /root/reference/trajopt/ilqr/ilqr.py:136-237      (row A6)
//   Riccati taylor expansions  /root/reference/trajopt/riccati/objects.py:54-65,102-108 (row A10)
// Deviations from the reference, all deliberate:
//   * the initial feed-forward noise kff is an input (quirk 9: the reference draws it from the
//     global NumPy RNG, ilqr.py:70);
//   * "did the backward pass fail" is the unambiguous flag, not `np.any(diverge)` which misses a
//     failure at time index 0 (quirk 2).
#pragma once
#include <cmath>
#include <limits>
#include <vector>
#include "envs_oracle.hpp"
#include "riccati_oracle.hpp"

namespace oracle {

struct ILQROptions {
    double lmbda = 1., dlmbda = 1., min_lmbda = 1e-6, max_lmbda = 1e6, mult_lmbda = 1.6;
    double tolfun = 1e-6, tolgrad = 1e-4, min_imp = 0.;
    int reg = 1;
};

// ilqr.py:89-102.  state (n, T+1), action (m, T), cost (T+1), column-major.
inline void forward_pass(const trajopt_env_t& e, int T, const double* x0, const double* K, const double* kff,
                         double alpha, const double* xref, const double* uref, const double* weighting,
                         double* state, double* action, double* cost_out) {
    const int n = e.dm_state, m = e.dm_act;
    for (int i = 0; i < n; ++i) state[i] = x0[i];
    for (int t = 0; t < T; ++t) {
        const double* x = state + (size_t)n * t;
        double* u = action + (size_t)m * t;
        for (int j = 0; j < m; ++j) {
            // objects.py:137-139: uref + alpha * kff + K @ (x - xref)
            double kx = 0.;
            for (int i = 0; i < n; ++i) kx += K[j + m * i + (size_t)m * n * t] * (x[i] - xref[i + (size_t)n * t]);
            const double a = uref[j + (size_t)m * t] + alpha * kff[j + (size_t)m * t] + kx;
            u[j] = clip(a, e.umax[j]);  // ilqr.py:97 (ulim = action_space.high = umax)
        }
        cost_out[t] = cost(e, x, u, u, weighting[t]);  // ilqr.py:98: u_last = u_t (quirk 6)
        dynamics<double>(e, x, u, state + (size_t)n * (t + 1));
    }
    double zero[TRAJOPT_ENV_MAX_ACT] = {0, 0, 0, 0};
    cost_out[T] = cost(e, state + (size_t)n * T, zero, zero, weighting[T]);  // ilqr.py:101
}

// objects.py:114-117 and :68-78 in one sweep.  Cost arrays have T+1 slices, A/B have T.
inline int quadratize(const trajopt_env_t& e, int T, const double* xref, const double* uref,
                      const double* weighting, double* A, double* B, double* Cxx, double* cx, double* Cuu,
                      double* cu, double* Cxu) {
    const int n = e.dm_state, m = e.dm_act;
    double zero[TRAJOPT_ENV_MAX_ACT] = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t)
        if (dynamics_jacobians(e, xref + (size_t)n * t, uref + (size_t)m * t, A + (size_t)n * n * t,
                               B + (size_t)n * m * t, nullptr))
            return -1;
    for (int t = 0; t <= T; ++t) {
        // _u = hstack(u, 0): u_t is the zero pad at t == T; u_last = _u[t-1], and _u[-1] is the pad too
        const double* u = (t < T) ? uref + (size_t)m * t : zero;
        const double* ul = (t >= 1) ? uref + (size_t)m * (t - 1) : zero;
        cost_derivatives(e, xref + (size_t)n * t, u, ul, weighting[t], Cxx + (size_t)n * n * t, cx + (size_t)n * t,
                         Cuu + (size_t)m * m * t, cu + (size_t)m * t, Cxu + (size_t)n * m * t);
    }
    return 0;
}

// riccati/objects.py:54-65 (cost: 1/2-scaled Hessians, de-centred linear terms) and :102-108
// (dynamics with affine residual c).
inline int quadratize_riccati(const trajopt_env_t& e, int T, const double* xref, const double* uref,
                              const double* weighting, double* A, double* B, double* c, double* Cxx, double* cx,
                              double* Cuu, double* cu, double* Cxu) {
    const int n = e.dm_state, m = e.dm_act;
    if (quadratize(e, T, xref, uref, weighting, A, B, Cxx, cx, Cuu, cu, Cxu)) return -1;
    double zero[TRAJOPT_ENV_MAX_ACT] = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        const double* x = xref + (size_t)n * t;
        const double* u = uref + (size_t)m * t;
        double f[TRAJOPT_ENV_MAX_STATE];
        dynamics<double>(e, x, u, f);
        for (int i = 0; i < n; ++i) {
            double s = f[i];
            double ax = 0., bu = 0.;
            for (int j = 0; j < n; ++j) ax += A[i + n * j + (size_t)n * n * t] * x[j];
            for (int j = 0; j < m; ++j) bu += B[i + n * j + (size_t)n * m * t] * u[j];
            c[i + (size_t)n * t] = s - ax - bu;
        }
    }
    for (int t = 0; t <= T; ++t) {
        const double* x = xref + (size_t)n * t;
        const double* u = (t < T) ? uref + (size_t)m * t : zero;
        double* H = Cxx + (size_t)n * n * t; double* R = Cuu + (size_t)m * m * t; double* P = Cxu + (size_t)n * m * t;
        double* gx = cx + (size_t)n * t; double* gu = cu + (size_t)m * t;
        // cx = dcdx - dcdxx @ x - 2 dcdxu @ u ; cu = dcdu - dcduu @ u - 2 x^T dcdxu   (:64-65)
        for (int i = 0; i < n; ++i) {
            double s = gx[i];
            double hx = 0., pu = 0.;
            for (int j = 0; j < n; ++j) hx += H[i + n * j] * x[j];
            for (int j = 0; j < m; ++j) pu += P[i + n * j] * u[j];
            gx[i] = s - hx - 2. * pu;
        }
        for (int j = 0; j < m; ++j) {
            double s = gu[j];
            double ru = 0., xp = 0.;
            for (int k = 0; k < m; ++k) ru += R[j + m * k] * u[k];
            for (int i = 0; i < n; ++i) xp += x[i] * P[i + n * j];
            gu[j] = s - ru - 2. * xp;
        }
        for (int k = 0; k < n * n; ++k) H[k] *= 0.5;  // :60-62
        for (int k = 0; k < m * m; ++k) R[k] *= 0.5;
        for (int k = 0; k < n * m; ++k) P[k] *= 0.5;
    }
    return 0;
}

struct ILQRSolution {
    std::vector<double> xref, uref, K, kff, V, v, Qxx, Qux, Quu, qx, qu, trace;
    double last_return = -std::numeric_limits<double>::infinity();
    double lmbda = 1., dlmbda = 1., alpha = std::nan("");
    int iters = 0;
    int status = 0;  // why the loop ended: 0 nb_iter exhausted, 1 tolgrad, 2 tolfun, 3 lmbda > max
};

// ilqr.py:136-237.
inline ILQRSolution ilqr_run(const trajopt_env_t& e, int T, const double* x0, const double* kff_init,
                             const double* weighting, const double* alphas, int nb_alphas, const ILQROptions& o,
                             int nb_iter) {
    const int n = e.dm_state, m = e.dm_act;
    const size_t nn = (size_t)n * n, nm = (size_t)n * m, mm = (size_t)m * m;
    ILQRSolution s;
    s.lmbda = o.lmbda; s.dlmbda = o.dlmbda;
    s.xref.assign((size_t)n * (T + 1), 0.); s.uref.assign((size_t)m * T, 0.);
    for (int i = 0; i < n; ++i) s.xref[i] = x0[i];  // ilqr.py:59-60
    s.K.assign(nm * T, 0.); s.kff.assign(kff_init, kff_init + (size_t)m * T);  // :69-70
    s.V.assign(nn * (T + 1), 0.); s.v.assign((size_t)n * (T + 1), 0.);
    s.Qxx.assign(nn * T, 0.); s.Qux.assign(nm * T, 0.); s.Quu.assign(mm * T, 0.);
    s.qx.assign((size_t)n * T, 0.); s.qu.assign((size_t)m * T, 0.);

    std::vector<double> st((size_t)n * (T + 1)), ac((size_t)m * T), co(T + 1);
    // :139-147 initial trajectory
    for (int a = 0; a < nb_alphas; ++a) {
        forward_pass(e, T, x0, s.K.data(), s.kff.data(), alphas[a], s.xref.data(), s.uref.data(), weighting,
                     st.data(), ac.data(), co.data());
        bool ok = true;
        for (double xv : st) ok = ok && (xv < 1e8);
        if (ok) {
            s.xref = st; s.uref = ac;
            double r = 0.; for (double c : co) r += c;
            s.last_return = r;
            break;
        }
    }
    s.trace.push_back(s.last_return);  // :149

    std::vector<double> A(nn * T), B(nm * T), Cxx(nn * (T + 1)), cx((size_t)n * (T + 1)), Cuu(mm * (T + 1)),
        cu((size_t)m * (T + 1)), Cxu(nm * (T + 1));
    std::vector<double> lK(nm * T), lkff((size_t)m * T), lV(nn * (T + 1)), lv((size_t)n * (T + 1)), lQxx(nn * T),
        lQux(nm * T), lQuu(mm * T), lqx((size_t)n * T), lqu((size_t)m * T);
    double dV[2];
    for (int iter = 0; iter < nb_iter; ++iter) {  // :151
        s.iters = iter + 1;
        quadratize(e, T, s.xref.data(), s.uref.data(), weighting, A.data(), B.data(), Cxx.data(), cx.data(),
                   Cuu.data(), cu.data(), Cxu.data());  // :153-156
        bool backpass_done = false;
        while (!backpass_done) {  // :162-173
            std::fill(lK.begin(), lK.end(), 0.); std::fill(lkff.begin(), lkff.end(), 0.);
            std::fill(lV.begin(), lV.end(), 0.); std::fill(lv.begin(), lv.end(), 0.);
            std::fill(lQxx.begin(), lQxx.end(), 0.); std::fill(lQux.begin(), lQux.end(), 0.);
            std::fill(lQuu.begin(), lQuu.end(), 0.); std::fill(lqx.begin(), lqx.end(), 0.);
            std::fill(lqu.begin(), lqu.end(), 0.);
            int failed = 0;
            backward_pass<double>(Cxx.data(), cx.data(), Cuu.data(), cu.data(), Cxu.data(), A.data(), B.data(),
                                  s.lmbda, o.reg, n, m, T, lQxx.data(), lQux.data(), lQuu.data(), lqx.data(),
                                  lqu.data(), lV.data(), lv.data(), dV, lK.data(), lkff.data(), &failed);
            if (failed) {
                s.dlmbda = std::fmax(s.dlmbda * o.mult_lmbda, o.mult_lmbda);  // :166
                s.lmbda = std::fmax(s.lmbda * s.dlmbda, o.min_lmbda);         // :167
                if (s.lmbda > o.max_lmbda) break;
            } else {
                backpass_done = true;
            }
        }
        // :176  mean over action dims of max over time of |kff| / (|uref| + 1)
        double g_norm = 0.;
        for (int j = 0; j < m; ++j) {
            double mx = -std::numeric_limits<double>::infinity();
            for (int t = 0; t < T; ++t) {
                const double q = std::fabs(lkff[j + (size_t)m * t]) / (std::fabs(s.uref[j + (size_t)m * t]) + 1.);
                if (q > mx || q != q) mx = q;  // np.max propagates NaN
                if (mx != mx) break;
            }
            g_norm += mx;
        }
        g_norm /= m;
        if (g_norm < o.tolgrad && s.lmbda < 1e-5) {  // :177-180
            s.dlmbda = std::fmin(s.dlmbda / o.mult_lmbda, 1. / o.mult_lmbda);
            s.lmbda = s.lmbda * s.dlmbda * (s.lmbda > o.min_lmbda ? 1. : 0.);
            s.status = 1;
            break;
        }
        bool fwdpass_done = false;
        double ret = 0., dreturn = 0.;
        if (backpass_done) {  // :186-202
            for (int a = 0; a < nb_alphas; ++a) {
                const double alpha = alphas[a];
                s.alpha = alpha;
                forward_pass(e, T, x0, lK.data(), lkff.data(), alpha, s.xref.data(), s.uref.data(), weighting,
                             st.data(), ac.data(), co.data());
                ret = 0.; for (double c : co) ret += c;
                dreturn = s.last_return - ret;
                const double expected = -1. * alpha * (dV[0] + alpha * dV[1]);
                const double imp = dreturn / expected;
                if (imp >= o.min_imp) { fwdpass_done = true; break; }
            }
        }
        if (fwdpass_done) {  // :205-223
            s.dlmbda = std::fmin(s.dlmbda / o.mult_lmbda, 1. / o.mult_lmbda);
            s.lmbda = s.lmbda * s.dlmbda * (s.lmbda > o.min_lmbda ? 1. : 0.);
            s.xref = st; s.uref = ac; s.last_return = ret;
            s.V = lV; s.v = lv; s.Qxx = lQxx; s.Qux = lQux; s.Quu = lQuu; s.qx = lqx; s.qu = lqu;
            s.K = lK; s.kff = lkff;
            s.trace.push_back(s.last_return);
            if (dreturn < o.tolfun) { s.status = 2; break; }
        } else {  // :224-231
            s.dlmbda = std::fmax(s.dlmbda * o.mult_lmbda, o.mult_lmbda);
            s.lmbda = std::fmax(s.lmbda * s.dlmbda, o.min_lmbda);
            if (s.lmbda > o.max_lmbda) { s.status = 3; break; }
        }
    }
    return s;
}

}  // namespace oracle
