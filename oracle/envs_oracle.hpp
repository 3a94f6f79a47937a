// ORACLE - TEST INFRASTRUCTURE ONLY (see riccati_oracle.hpp for the rules).  PARITY UNPINNED.
//
// CPU restatement of the env layer the iLQR / Riccati hot path consumes (row A8 of SURVEY 8a):
//   pendulum         /root/reference/trajopt/envs/pendulum/pendulum.py:58-76 (dynamics), :111-125 (cost)
//   cartpole         /root/reference/trajopt/envs/cartpole/cartpole.py:62-97, :112-127
//   double cartpole  /root/reference/trajopt/envs/double_cartpole/double_cartpole.py:61-155, :170-187
//   double pendulum  /root/reference/trajopt/envs/double_pendulum/double_pendulum.py:61-125, :140-154
//   LQR (affine LTI) /root/reference/trajopt/envs/lqr/lqr_v0.py:50-66, :89-94
//   quadrotor        build-defined 12-state rigid body (NOT in the reference; SURVEY "Facts", C4)
// and of the autograd derivatives the reference takes of them (objects.py:58-63,108-109): forward-mode
// dual numbers here, which reproduce autograd's reverse-mode Jacobians to round-off.  autograd's
// np.clip VJP is g * (ans != a_min) * (ans != a_max): derivative 0 when the output sits on a bound.
#pragma once
#include <cmath>
#include "../include/trajopt_env.h"

namespace oracle {

template <int K>
struct Dual {
    double v;
    double d[K > 0 ? K : 1];
    Dual() : v(0) { for (int k = 0; k < K; ++k) d[k] = 0; }
    Dual(double x) : v(x) { for (int k = 0; k < K; ++k) d[k] = 0; }
};
template <int K> Dual<K> operator+(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v + b.v; for (int k = 0; k < K; ++k) r.d[k] = a.d[k] + b.d[k]; return r; }
template <int K> Dual<K> operator-(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v - b.v; for (int k = 0; k < K; ++k) r.d[k] = a.d[k] - b.d[k]; return r; }
template <int K> Dual<K> operator-(const Dual<K>& a) { Dual<K> r; r.v = -a.v; for (int k = 0; k < K; ++k) r.d[k] = -a.d[k]; return r; }
template <int K> Dual<K> operator*(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v * b.v; for (int k = 0; k < K; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
template <int K> Dual<K> operator/(const Dual<K>& a, const Dual<K>& b) { Dual<K> r; r.v = a.v / b.v; for (int k = 0; k < K; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v; return r; }
template <int K> Dual<K> operator+(const Dual<K>& a, double b) { Dual<K> r = a; r.v += b; return r; }
template <int K> Dual<K> operator+(double b, const Dual<K>& a) { Dual<K> r = a; r.v += b; return r; }
template <int K> Dual<K> operator-(const Dual<K>& a, double b) { Dual<K> r = a; r.v -= b; return r; }
template <int K> Dual<K> operator-(double b, const Dual<K>& a) { Dual<K> r = -a; r.v += b; return r; }
template <int K> Dual<K> operator*(const Dual<K>& a, double b) { Dual<K> r; r.v = a.v * b; for (int k = 0; k < K; ++k) r.d[k] = a.d[k] * b; return r; }
template <int K> Dual<K> operator*(double b, const Dual<K>& a) { return a * b; }
template <int K> Dual<K> operator/(const Dual<K>& a, double b) { Dual<K> r; r.v = a.v / b; for (int k = 0; k < K; ++k) r.d[k] = a.d[k] / b; return r; }
template <int K> Dual<K> operator/(double b, const Dual<K>& a) { return Dual<K>(b) / a; }
template <int K> Dual<K> sin(const Dual<K>& a) { Dual<K> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int k = 0; k < K; ++k) r.d[k] = c * a.d[k]; return r; }
template <int K> Dual<K> cos(const Dual<K>& a) { Dual<K> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int k = 0; k < K; ++k) r.d[k] = s * a.d[k]; return r; }
template <int K> Dual<K> tan(const Dual<K>& a) { Dual<K> r; r.v = std::tan(a.v); double s = 1.0 + r.v * r.v; for (int k = 0; k < K; ++k) r.d[k] = s * a.d[k]; return r; }
inline double val(double a) { return a; }
template <int K> double val(const Dual<K>& a) { return a.v; }
using std::sin; using std::cos; using std::tan;

// np.clip(x, -lim, lim) and its autograd derivative mask.
inline double clip(double x, double lim) { return std::fmin(std::fmax(x, -lim), lim); }
template <int K> Dual<K> clip(const Dual<K>& x, double lim) {
    Dual<K> r; r.v = clip(x.v, lim);
    const double mask = (r.v != -lim && r.v != lim) ? 1.0 : 0.0;
    for (int k = 0; k < K; ++k) r.d[k] = mask * x.d[k];
    return r;
}

// ---- continuous-time vector fields ----------------------------------------------------------
template <class X> void f_pendulum(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // pendulum.py:61-66
    const double g = e.phys[0], m = e.phys[1], l = e.phys[2], k = e.phys[3];
    X th = x[0], dth = x[1];
    dx[0] = dth;
    dx[1] = -3. * g / (2. * l) * sin(th + M_PI) + 3. / (m * l * l) * (u[0] - k * dth);
}
template <class X> void f_cartpole(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // cartpole.py:67-87
    const double g = e.phys[0], Mc = e.phys[1], Mp = e.phys[2], l = e.phys[3], fr = e.phys[4];
    const double Mt = Mc + Mp;
    X th = x[1], dq = x[2], dth = x[3];
    X sth = sin(th), cth = cos(th);
    X num = g * sth + cth * (-(u[0] - fr * dq) - Mp * l * dth * dth * sth) / Mt;
    X denom = l * ((4. / 3.) - Mp * cth * cth / Mt);
    X ddth = num / denom;
    X ddx = (u[0] + Mp * l * (dth * dth * sth - ddth * cth)) / Mt;
    dx[0] = dq; dx[1] = dth; dx[2] = ddx; dx[3] = ddth;
}
template <class X> void f_double_cartpole(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // double_cartpole.py:71-145
    const double g = e.phys[0], Mc = e.phys[1], Mp1 = e.phys[2], Mp2 = e.phys[3], L1 = e.phys[4], L2 = e.phys[5];
    const double Mt = Mc + Mp1 + Mp2, l1 = L1 / 2., l2 = L2 / 2., J1 = Mp1 * L1 / 12., J2 = Mp2 * L2 / 12.;
    X th1 = x[1], th2 = x[2], dq = x[3], dth1 = x[4], dth2 = x[5];
    X s1 = sin(th1), c1 = cos(th1), s2 = sin(th2), c2 = cos(th2);
    X sdth = sin(th1 - th2), cdth = cos(th1 - th2);
    const double l1_mp1_mp2 = Mp1 * l1 + Mp2 * L2;
    X l1_mp1_mp2_c1 = l1_mp1_mp2 * c1;
    const double Mp2_l2 = Mp2 * l2;
    X Mp2_l2_c2 = Mp2_l2 * c2;
    const double l1_l2_Mp2 = L1 * l2 * Mp2;
    X l1_l2_Mp2_cdth = l1_l2_Mp2 * cdth;
    X Mm[3][3];
    Mm[0][0] = X(Mt); Mm[0][1] = l1_mp1_mp2_c1; Mm[0][2] = Mp2_l2_c2;
    Mm[1][0] = l1_mp1_mp2_c1; Mm[1][1] = X((l1 * l1) * Mp1 + (L1 * L1) * Mp2 + J1); Mm[1][2] = l1_l2_Mp2_cdth;
    Mm[2][0] = Mp2_l2_c2; Mm[2][1] = l1_l2_Mp2_cdth; Mm[2][2] = X((l2 * l2) * Mp2 + J2);
    X C12 = -l1_mp1_mp2 * dth1 * s1, C13 = -Mp2_l2 * dth2 * s2;
    X C23 = l1_l2_Mp2 * dth2 * sdth, C32 = -l1_l2_Mp2 * dth1 * sdth;
    X G2 = -(Mp1 * l1 + Mp2 * L1) * g * s1, G3 = -Mp2 * l2 * g * s2;
    // rhs = action - C dx - G   (:139-143)
    X rhs[3];
    rhs[0] = u[0] - (C12 * dth1 + C13 * dth2);
    rhs[1] = X(0.) - (C23 * dth2) - G2;
    rhs[2] = X(0.) - (C32 * dth1) - G3;
    // ddx = inv(M) rhs : Gaussian elimination with partial pivoting on the values
    int piv[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int p = k;
        for (int i = k + 1; i < 3; ++i)
            if (std::fabs(val(Mm[piv[i]][k])) > std::fabs(val(Mm[piv[p]][k]))) p = i;
        int tmp = piv[k]; piv[k] = piv[p]; piv[p] = tmp;
        for (int i = k + 1; i < 3; ++i) {
            X fct = Mm[piv[i]][k] / Mm[piv[k]][k];
            for (int j = k; j < 3; ++j) Mm[piv[i]][j] = Mm[piv[i]][j] - fct * Mm[piv[k]][j];
            rhs[piv[i]] = rhs[piv[i]] - fct * rhs[piv[k]];
        }
    }
    X dd[3];
    for (int i = 2; i >= 0; --i) {
        X s = rhs[piv[i]];
        for (int j = i + 1; j < 3; ++j) s = s - Mm[piv[i]][j] * dd[j];
        dd[i] = s / Mm[piv[i]][i];
    }
    dx[0] = dq; dx[1] = dth1; dx[2] = dth2; dx[3] = dd[0]; dx[4] = dd[1]; dx[5] = dd[2];
}
template <class X> void f_double_pendulum(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // double_pendulum.py:66-113, term by term
    const double g = e.phys[0], m1 = e.phys[1], m2 = e.phys[2], l1 = e.phys[3], l2 = e.phys[4], k1 = e.phys[5],
                 k2 = e.phys[6];
    const double I1 = m1 * (l1 * l1 + 1e-4) / 3.0, I2 = m2 * (l2 * l2 + 1e-4) / 3.0;
    const double l1CM = l1 / 2., l2CM = l2 / 2.;
    X th1 = x[0] + M_PI, th2 = x[1], dth1 = x[2], dth2 = x[3];
    X s1 = sin(th1), c1 = cos(th1), s2 = sin(th2), c2 = cos(th2);
    X h11 = I1 + I2 + l1CM * l1CM * m1 + l1 * l1 * m2 + l2CM * l2CM * m2 + 2. * l1 * l2CM * m2 * c2;
    X h12 = I2 + l2CM * l2CM * m2 + l1 * l2CM * m2 * c2;
    X b1 = g * l1CM * m1 * s1 + g * l1 * m2 * s1 + g * l2CM * m2 * c2 * s1 - 2. * dth1 * dth2 * l1 * l2CM * m2 * s2 -
           dth2 * dth2 * l1 * l2CM * m2 * s2 + g * l2CM * m2 * c1 * s2;
    X h21 = I2 + l2CM * l2CM * m2 + l1 * l2CM * m2 * c2;
    X h22 = X(I2 + l2CM * l2CM * m2);
    X b2 = g * l2CM * m2 * c2 * s1 + dth1 * dth1 * l1 * l2CM * m2 * s2 + g * l2CM * m2 * c1 * s2;
    X u1 = u[0] - k1 * dth1, u2 = u[1] - k2 * dth2;
    X det = h11 * h22 - h12 * h21;
    dx[0] = dth1; dx[1] = dth2;
    dx[2] = (h22 * (u1 - b1) - h12 * (u2 - b2)) / det;
    dx[3] = (h11 * (u2 - b2) - h21 * (u1 - b1)) / det;
}
template <class X> void f_quadrotor(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // build-defined: x = [p(3), euler phi/theta/psi, v(3) world, body rates p q r], u = 4 rotor thrusts
    const double g = e.phys[0], mass = e.phys[1], L = e.phys[2], Ixx = e.phys[3], Iyy = e.phys[4],
                 Izz = e.phys[5], kM = e.phys[6];
    X ph = x[3], th = x[4], ps = x[5], p = x[9], q = x[10], r = x[11];
    X sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps), cps = cos(ps);
    X F = u[0] + u[1] + u[2] + u[3];
    X tx = L * (u[1] - u[3]), ty = L * (u[2] - u[0]), tz = kM * (u[0] - u[1] + u[2] - u[3]);
    X tth = sth / cth;
    dx[0] = x[6]; dx[1] = x[7]; dx[2] = x[8];
    dx[3] = p + sph * tth * q + cph * tth * r;
    dx[4] = cph * q - sph * r;
    dx[5] = (sph * q + cph * r) / cth;
    dx[6] = (cph * sth * cps + sph * sps) * F / mass;
    dx[7] = (cph * sth * sps - sph * cps) * F / mass;
    dx[8] = cph * cth * F / mass - g;
    dx[9] = (tx - (Izz - Iyy) * q * r) / Ixx;
    dx[10] = (ty - (Ixx - Izz) * p * r) / Iyy;
    dx[11] = (tz - (Iyy - Ixx) * p * q) / Izz;
}

template <class X> void f_quad_pendulum(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // quad_pendulum.py:62-228; the linear system M dd = r (the reference expands inv(M) r by cofactors, :188-222)
    // is solved by Gaussian elimination with partial pivoting on the values
    const double G = e.phys[0], m1 = e.phys[1], m2 = e.phys[2], m3 = e.phys[3], m4 = e.phys[4];
    const double l1 = e.phys[5], l2 = e.phys[6], l3 = e.phys[7], l4 = e.phys[8];
    const double fr1 = e.phys[9], fr2 = e.phys[10], fr3 = e.phys[11], fr4 = e.phys[12];
    const double I1 = m1 * (l1 * l1 + 1e-4) / 3.0, I2 = m2 * (l2 * l2 + 1e-4) / 3.0, I3 = m3 * (l3 * l3 + 1e-4) / 3.0,
                 I4 = m4 * (l4 * l4 + 1e-4) / 3.0;
    const double h1 = 0.5 * l1, h2 = 0.5 * l2, h3 = 0.5 * l3, h4 = 0.5 * l4;   // centres of mass
    // lumped constants: w4 = l4cm m4, w3 = l3cm m3 + l3 m4, w2 = l2cm m2 + l2 (m3 + m4), w1 = l1cm m1 + l1 (m2 + m3 + m4)
    const double w4 = h4 * m4, w3 = h3 * m3 + l3 * m4, w2 = h2 * m2 + l2 * (m3 + m4), e8 = l1 * (m2 + m3 + m4);
    const X w1 = x[4], wv2 = x[5], wv3 = x[6], wv4 = x[7];   // angular velocities
    const X t1 = x[0] + M_PI;   // downward position = pi (quad_pendulum.py:77)
    const X s1 = sin(t1), c1 = cos(t1), s2 = sin(x[1]), c2 = cos(x[1]), s3 = sin(x[2]), c3 = cos(x[2]),
            s4 = sin(x[3]), c4 = cos(x[3]);
    // sums of angles by the addition formulas, as the reference forms them (:92-103)
    const X s12 = s1 * c2 + c1 * s2, c12 = c1 * c2 - s1 * s2, s23 = s2 * c3 + c2 * s3, c23 = c2 * c3 - s2 * s3;
    const X s34 = s3 * c4 + c3 * s4, c34 = c3 * c4 - s3 * s4;
    const X s1234 = s12 * c34 + c12 * s34, s123 = s12 * c3 + c12 * s3, s234 = s2 * c34 + c2 * s34, c234 = c2 * c34 - s2 * s34;
    // inertia matrix (symmetric; the reference's a..p, :147-181)
    const X M44 = I4 + h4 * w4;
    const X M34 = M44 + (l3 * w4) * c4;
    const X M24 = M34 + (l2 * w4) * c34;
    const X M14 = M24 + (l1 * w4) * c234;
    const X M33 = I3 + M34 + h3 * (h3 * m3) + l3 * (l3 * m4) + (l3 * w4) * c4;
    const X M23 = M33 + (l2 * w3) * c3 + (l2 * w4) * c34;
    const X M13 = M23 + (l1 * w3) * c23 + (l1 * w4) * c234;
    const X M22 = M23 + I2 + h2 * (h2 * m2) + (l2 * l2) * (m3 + m4) + (l2 * w3) * c3 + (l2 * w4) * c34;
    const X M12 = M22 + M13 - M23 + (l1 * w2) * c2;
    const X M11 = 2.0 * M12 + I1 - M22 + (h1 * h1) * m1 + l1 * e8;
    // velocity products and generalised forces (:105-145, :152-186)
    const X v12 = w1 + wv2, v123 = v12 + wv3, q11 = w1 * w1, q12 = v12 * v12;
    const X e4b = 2.0 * w1 * wv3 + 2.0 * wv2 * wv3 + wv3 * wv3;
    const X e4a = 2.0 * w1 * wv4 + 2.0 * wv2 * wv4 + 2.0 * wv3 * wv4 + wv4 * wv4;
    const X e9a = 2.0 * w1 * wv2 + wv2 * wv2, e9 = e9a + e4b;
    const X e1 = G * (s123 * w3 + s1234 * w4);
    const X e2 = (2.0 * v123 + wv4) * wv4 * (l3 * w4) * s4;
    const X e3 = (G * w2) * s12;
    const X e4 = (e4b + e4a) * ((l2 * w4) * s34);
    const X e5a = q11 * l1 * s234, e5 = w4 * e5a;
    const X e6 = e4b * (l2 * w3) * s3;
    const X ls23 = (l1 * w3) * s23;
    const X r4 = u[3] - fr4 * wv4 - w4 * (v123 * v123 * l3 * s4 + q12 * l2 * s34 + e5a + G * s1234);
    const X r3 = u[2] - fr3 * wv3 - ((q12 * (l2 * w3) * s3 + q11 * ls23) - e2 + q12 * ((l2 * w4) * s34) + e5 + e1);
    const X r2 = u[1] - fr2 * wv2 - (q11 * (l1 * w2) * s2 - e6 + q11 * ls23 - e2 - e4 + e5 + e3 + e1);
    const X r1 = u[0] - fr1 * w1 - (-e9a * (l1 * w2) * s2 - e6 - e9 * ls23 - e2 - e4 - (e9 + e4a) * (l1 * w4) * s234 + e3 +
                                    G * (h1 * m1 + e8) * s1 + e1);
    X Mm[4][4] = {{M11, M12, M13, M14}, {M12, M22, M23, M24}, {M13, M23, M33, M34}, {M14, M24, M34, M44}};
    X rhs[4] = {r1, r2, r3, r4};
    int piv[4] = {0, 1, 2, 3};
    for (int k = 0; k < 4; ++k) {
        int p = k;
        for (int i = k + 1; i < 4; ++i)
            if (std::fabs(val(Mm[piv[i]][k])) > std::fabs(val(Mm[piv[p]][k]))) p = i;
        int tmp = piv[k]; piv[k] = piv[p]; piv[p] = tmp;
        for (int i = k + 1; i < 4; ++i) {
            X fct = Mm[piv[i]][k] / Mm[piv[k]][k];
            for (int j = k; j < 4; ++j) Mm[piv[i]][j] = Mm[piv[i]][j] - fct * Mm[piv[k]][j];
            rhs[piv[i]] = rhs[piv[i]] - fct * rhs[piv[k]];
        }
    }
    X dd[4];
    for (int i = 3; i >= 0; --i) {
        X sacc = rhs[piv[i]];
        for (int j = i + 1; j < 4; ++j) sacc = sacc - Mm[piv[i]][j] * dd[j];
        dd[i] = sacc / Mm[piv[i]][i];
    }
    dx[0] = w1; dx[1] = wv2; dx[2] = wv3; dx[3] = wv4;
    dx[4] = dd[0]; dx[5] = dd[1]; dx[6] = dd[2]; dx[7] = dd[3];
}

template <class X> void f_robot(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    // robot.py:53-61
    const double l = e.phys[0];
    dx[0] = x[3] * cos(x[2]);
    dx[1] = x[3] * sin(x[2]);
    dx[2] = x[3] * tan(u[1]) / l;
    dx[3] = u[0];
}

template <class X> void vector_field(const trajopt_env_t& e, const X* x, const X* u, X* dx) {
    switch (e.model) {
        case TRAJOPT_ENV_PENDULUM: f_pendulum(e, x, u, dx); break;
        case TRAJOPT_ENV_CARTPOLE: f_cartpole(e, x, u, dx); break;
        case TRAJOPT_ENV_DOUBLE_CARTPOLE: f_double_cartpole(e, x, u, dx); break;
        case TRAJOPT_ENV_QUADROTOR: f_quadrotor(e, x, u, dx); break;
        case TRAJOPT_ENV_DOUBLE_PENDULUM: f_double_pendulum(e, x, u, dx); break;
        case TRAJOPT_ENV_ROBOT: f_robot(e, x, u, dx); break;
        case TRAJOPT_ENV_QUAD_PENDULUM: f_quad_pendulum(e, x, u, dx); break;
        default: break;
    }
}

// env.dynamics(x, u): clip(u) -> RK4 -> clip(xn)   (cartpole.py:62-97); LQR: A x + B clip(u) + c, no
// state clip (lqr_v0.py:50-66).
template <class X> void dynamics(const trajopt_env_t& e, const X* x, const X* u, X* xn) {
    const int n = e.dm_state, m = e.dm_act;
    X uc[TRAJOPT_ENV_MAX_ACT];
    for (int j = 0; j < m; ++j) uc[j] = clip(u[j], e.umax[j]);
    if (e.model == TRAJOPT_ENV_LQR) {
        for (int i = 0; i < n; ++i) {
            X s(0.);
            for (int j = 0; j < n; ++j) s = s + e.lin_A[i + n * j] * x[j];
            X s2(0.);
            for (int j = 0; j < m; ++j) s2 = s2 + e.lin_B[i + n * j] * uc[j];
            xn[i] = s + s2 + e.lin_c[i];
        }
        return;
    }
    const double dt = e.model == TRAJOPT_ENV_ROBOT ? -e.dt : e.dt;  // robot.py:62-67 steps RK4 by -dt
    X k1[TRAJOPT_ENV_MAX_STATE], k2[TRAJOPT_ENV_MAX_STATE], k3[TRAJOPT_ENV_MAX_STATE], k4[TRAJOPT_ENV_MAX_STATE],
        y[TRAJOPT_ENV_MAX_STATE];
    vector_field(e, x, uc, k1);
    for (int i = 0; i < n; ++i) y[i] = x[i] + 0.5 * dt * k1[i];
    vector_field(e, y, uc, k2);
    for (int i = 0; i < n; ++i) y[i] = x[i] + 0.5 * dt * k2[i];
    vector_field(e, y, uc, k3);
    for (int i = 0; i < n; ++i) y[i] = x[i] + dt * k3[i];
    vector_field(e, y, uc, k4);
    for (int i = 0; i < n; ++i) {
        X s = x[i] + dt / 6. * (k1[i] + 2. * k2[i] + 2. * k3[i] + k4[i]);
        xn[i] = clip(s, e.xmax[i]);
    }
}

inline double wrap_angle(double x) {  // cartpole.py:10-12, Python's % (sign of the divisor)
    double r = std::fmod(x + M_PI, 2. * M_PI);
    if (r < 0) r += 2. * M_PI;
    return r - M_PI;
}

// Feature f of trajopt_env_t::features at x (after the periodic wrap) and its derivative w.r.t. its one state
// (e.g. cartpole.py:161-166: [x0, cos x1, sin x1, x2, x3]).
inline double feature(const trajopt_env_t& e, const double* x, int f, double* dz) {
    const int i = e.features[f] & 0xff, kind = e.features[f] >> 8;
    const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
    if (kind == TRAJOPT_FEAT_COS) { *dz = -std::sin(y); return std::cos(y); }
    if (kind == TRAJOPT_FEAT_SIN) { *dz = std::cos(y); return std::sin(y); }
    *dz = 1.0;
    return y;
}

// env.cost(x, u, u_last, a) (cartpole.py:99-105,112-127); identity features: J = I, j = 0.
inline double cost(const trajopt_env_t& e, const double* x, const double* u, const double* u_last, double a) {
    const int n = e.dm_state, m = e.dm_act;
    double c = 0.;
    for (int j = 0; j < m; ++j) {
        const double d = e.slew_rate ? (u[j] - u_last[j]) : u[j];
        c += d * e.uw[j] * d;
    }
    if (a != 0.) {  // Python `if a:` (cartpole.py:120): an exactly-zero weight drops the state cost
        double s = 0.;
        if (e.nb_features == 0) {
            for (int i = 0; i < n; ++i) {
                const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
                s += (y - e.goal[i]) * e.gw[i] * (y - e.goal[i]);
            }
        } else {  // z = J(y) y + j = phi(y) at the evaluation point (cartpole.py:122-125)
            for (int f = 0; f < e.nb_features; ++f) {
                double dz;
                const double z = feature(e, x, f, &dz);
                s += (z - e.goal[f]) * e.gw[f] * (z - e.goal[f]);
            }
        }
        c += a * s;
    }
    return e.cost_times_dt ? e.dt * c : c;
}

// autograd derivatives of cost() in the iLQR convention (ilqr/objects.py:72-78): no 1/2, no de-centring.
inline void cost_derivatives(const trajopt_env_t& e, const double* x, const double* u, const double* u_last,
                             double a, double* Cxx, double* cx, double* Cuu, double* cu, double* Cxu) {
    const int n = e.dm_state, m = e.dm_act;
    const double sc = e.cost_times_dt ? e.dt : 1.0;
    for (int k = 0; k < n * n; ++k) Cxx[k] = 0.;
    for (int k = 0; k < m * m; ++k) Cuu[k] = 0.;
    for (int k = 0; k < n * m; ++k) Cxu[k] = 0.;
    for (int j = 0; j < m; ++j) {
        const double d = e.slew_rate ? (u[j] - u_last[j]) : u[j];
        Cuu[j + m * j] = sc * 2. * e.uw[j];
        cu[j] = sc * 2. * e.uw[j] * d;
    }
    for (int i = 0; i < n; ++i) {
        if (a != 0. && e.nb_features == 0) {
            const double y = ((e.periodic >> i) & 1) ? wrap_angle(x[i]) : x[i];
            Cxx[i + n * i] = sc * 2. * a * e.gw[i];
            cx[i] = sc * 2. * a * e.gw[i] * (y - e.goal[i]);
        } else {
            cx[i] = 0.;
        }
    }
    if (a != 0. && e.nb_features != 0) {
        // autograd sees z = J y + j with J, j constants (getval, cartpole.py:122-124):
        // cx = 2a J'G(z - g), Cxx = 2a J'GJ, accumulated feature by feature (row f of J has one entry)
        for (int f = 0; f < e.nb_features; ++f) {
            const int i = e.features[f] & 0xff;
            double dz;
            const double z = feature(e, x, f, &dz);
            cx[i] += sc * 2. * a * e.gw[f] * (z - e.goal[f]) * dz;
            Cxx[i + n * i] += sc * 2. * a * e.gw[f] * dz * dz;
        }
    }
}

// A4: A = d dynamics / dx, B = d dynamics / du at (x, u)  (ilqr/objects.py:114-117), column-major.
template <int N, int Mm>
void dynamics_jacobians_T(const trajopt_env_t& e, const double* x, const double* u, double* A, double* B,
                          double* xn_out) {
    using D = Dual<N + Mm>;
    D xd[TRAJOPT_ENV_MAX_STATE], ud[TRAJOPT_ENV_MAX_ACT], xn[TRAJOPT_ENV_MAX_STATE];
    for (int i = 0; i < N; ++i) { xd[i] = D(x[i]); xd[i].d[i] = 1.0; }
    for (int j = 0; j < Mm; ++j) { ud[j] = D(u[j]); ud[j].d[N + j] = 1.0; }
    dynamics(e, xd, ud, xn);
    for (int i = 0; i < N; ++i) {
        for (int j = 0; j < N; ++j) A[i + N * j] = xn[i].d[j];
        for (int j = 0; j < Mm; ++j) B[i + N * j] = xn[i].d[N + j];
        if (xn_out) xn_out[i] = xn[i].v;
    }
}
inline int dynamics_jacobians(const trajopt_env_t& e, const double* x, const double* u, double* A, double* B,
                              double* xn_out) {
    const int n = e.dm_state, m = e.dm_act;
    if (n == 2 && m == 1) dynamics_jacobians_T<2, 1>(e, x, u, A, B, xn_out);
    else if (n == 4 && m == 1) dynamics_jacobians_T<4, 1>(e, x, u, A, B, xn_out);
    else if (n == 6 && m == 1) dynamics_jacobians_T<6, 1>(e, x, u, A, B, xn_out);
    else if (n == 12 && m == 4) dynamics_jacobians_T<12, 4>(e, x, u, A, B, xn_out);
    else if (n == 8 && m == 4) dynamics_jacobians_T<8, 4>(e, x, u, A, B, xn_out);
    else if (n == 2 && m == 2) dynamics_jacobians_T<2, 2>(e, x, u, A, B, xn_out);
    else if (n == 4 && m == 2) dynamics_jacobians_T<4, 2>(e, x, u, A, B, xn_out);
    else if (n == 3 && m == 1) dynamics_jacobians_T<3, 1>(e, x, u, A, B, xn_out);
    else return -1;
    return 0;
}

}  // namespace oracle
