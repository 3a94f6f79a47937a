// CPU restatement of the device sin / cos of trajopt_amd/csrc/env_models.h (sincos_g), the same operations in the same
// order with fma(), checked against long double libm on random arguments per decade and near multiples of pi / 4.
//   gcc -O2 -ffp-contract=off sincos_check.c -o sincos_check -lm && ./sincos_check [samples per decade]
// `old` is the form it replaced (reduction by pi / 2 without its big-argument guard), for comparison.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline uint64_t d2u(double x){uint64_t u;memcpy(&u,&x,8);return u;}
static inline double u2d(uint64_t u){double x;memcpy(&x,&u,8);return x;}
// old
static void sc_old(double x,double*s,double*c){
    const double fk = rint(x * 6.36619772367581382433e-01);
    double r = fma(-fk, 1.57079632673412561417e+00, x);
    r = fma(-fk, 6.07710050630396597660e-11, r);
    r = fma(-fk, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double sr = fma(r * z, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)fk;
    double s0 = (q&1)? cr:sr, c0=(q&1)? sr:cr;
    if (q&2) s0=-s0; if ((q+1)&2) c0=-c0; *s=s0;*c=c0;
}
// new: reduce by pi, half-angle polynomials, double-angle recombination
static void sc_new(double x,double*s,double*c){
    const double MAGIC = 6755399441055744.0;
    const double t = fma(x, 3.18309886183790691216e-01, MAGIC);
    const double fk = t - MAGIC;
    double r = fma(-fk, 2.0*1.57079632673412561417e+00, x);
    r = fma(-fk, 2.0*6.07710050630396597660e-11, r);
    r = fma(-fk, 2.0*2.02226624879595063154e-21, r);
    const double w = r * r;
    // S'_i = S_i / 4^i, C'_i = C_i / 4^(i+1)
    double ps = fma(w, 1.58969099521155010221e-10/4096.0, -2.50507602534068634195e-08/1024.0);
    double pc = fma(w, -1.13596475577881948265e-11/16384.0, 2.08757232129817482790e-09/4096.0);
    ps = fma(w, ps, 2.75573137070700676789e-06/256.0);
    pc = fma(w, pc, -2.75573143513906633035e-07/1024.0);
    ps = fma(w, ps, -1.98412698298579493134e-04/64.0);
    pc = fma(w, pc, 2.48015872894767294178e-05/256.0);
    ps = fma(w, ps, 8.33333333332248946124e-03/16.0);
    pc = fma(w, pc, -1.38888888888741095749e-03/64.0);
    ps = fma(w, ps, -1.66666666666666324348e-01/4.0);
    pc = fma(w, pc, 4.16666666666666019037e-02/16.0);
    const double sh2 = fma(r * w, ps, r);                       // 2 sin(r/2)
    const double ch = fma(w * w, pc, fma(-0.125, w, 1.0));      // cos(r/2)
    const double sa = sh2 * ch;
    const double ca = fma(sh2 * sh2, -0.5, 1.0);
    const uint64_t sg = (uint64_t)((uint32_t)d2u(t) << 31) << 32;
    *s = u2d(d2u(sa) ^ sg); *c = u2d(d2u(ca) ^ sg);
}
int main(int argc, char** argv){
    const int NS = argc > 1 ? atoi(argv[1]) : 2000000;
    double mo_s=0,mo_c=0,mn_s=0,mn_c=0; srand48(1);
    double scales[] = {1, 10, 1000, 1e5, 1e6, 1e9, 1e12, 1e15};
    for (int k=0;k<8;k++){
      mo_s=mo_c=mn_s=mn_c=0;
      for (int i=0;i<NS;i++){
        double x = (drand48()*2-1)*scales[k];
        if (i<1000) x = (i-500)*M_PI/4 * (1+ (drand48()-0.5)*1e-12) * (k<3?1:scales[k]/1000);   // near multiples of pi/4
        long double sl=sinl((long double)x), cl=cosl((long double)x);
        double s,c; sc_old(x,&s,&c);
        double e=fabs((double)(s-sl)); if(e>mo_s)mo_s=e; e=fabs((double)(c-cl)); if(e>mo_c)mo_c=e;
        sc_new(x,&s,&c);
        e=fabs((double)(s-sl)); if(e>mn_s)mn_s=e; e=fabs((double)(c-cl)); if(e>mn_c)mn_c=e;
      }
      printf("scale %g: old %.3g %.3g  new %.3g %.3g\n", scales[k], mo_s,mo_c,mn_s,mn_c);
    }
    double s,c; sc_new(INFINITY,&s,&c); printf("inf: %g %g\n",s,c); sc_new(NAN,&s,&c); printf("nan: %g %g\n",s,c);
    sc_new(1e300,&s,&c); printf("1e300: %g %g\n",s,c); sc_new(1e17,&s,&c); printf("1e17: %g %g\n",s,c);sc_new(-0.0,&s,&c); printf("-0: %g %g\n",s,c);
    return 0;
}
