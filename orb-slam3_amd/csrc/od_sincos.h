// od_sincos.h -- shared by the device code (orbx_kernels.hip.h) and by tests/sincos_exhaustive.cpp, which checks it on the host
// against (float)sin((double)x) / (float)cos((double)x) for EVERY float in [0, 2 pi] (the arithmetic is IEEE double fma / mul / add
// on both sides, so the host run speaks for the device).
#pragma once
#ifndef OD_FN
#define OD_FN static inline
#endif
// sin and cos of a float angle in [0, 2 pi], to be rounded once to float: the oracle's (float)cos((double)angle).  Double precision
// with a two-term Cody-Waite reduction by pi/2 (k <= 4, so k * PIO2_HI is within one fma of exact) and the fdlibm kernel polynomials
// on |r| <= pi/4: error < 1 ulp of double, i.e. the float rounding differs from the correctly rounded one only within ~2^-28 of a
// rounding boundary -- the same class as the library sincos(double) (range reduction for any argument, ~190 instructions) it replaces.
OD_FN void od_sincos(float x, double* sn, double* cs) {
    const double xd = (double)x;
    const double kd = __builtin_rint(xd * 0.63661977236758134308);
    const int k = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632679489655800e+00, xd);
    r = __builtin_fma(-kd, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double S = __builtin_fma(z * r, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w1 = 1.0 - hz;
    const double C = w1 + (((1.0 - w1) - hz) + z * z * pc);       // fdlibm __kernel_cos: 1 - z/2 with its rounding error carried
    const bool sw = (k & 1) != 0;
    const double s0 = sw ? C : S, c0 = sw ? S : C;
    *sn = (k & 2) ? -s0 : s0;
    *cs = ((k + 1) & 2) ? -c0 : c0;
}

