// phylo_math.h -- the arithmetic contract of the MI355X CSMC path (DESIGN.md "Arithmetic contract").
//
// Everything the sweep's results depend on is built from IEEE-754 binary64 +,-,*,/ and explicitly
// written fma(), compiled with -ffp-contract=off on BOTH the device (hipcc) and the host, plus integer
// arithmetic.  The same operation sequences, restated independently in oracle/csrc/ora_math.h, give
// bit-identical results on the CPU, which is what makes "resampling indices bit-exact under a fixed
// seed" a property of the construction instead of a statistical hope.
//
//   pm_exp / pm_log : classic argument-reduction + minimax-polynomial kernels (the published fdlibm /
//                     FreeBSD msun e_exp.c, e_log.c algorithms; < 1 ulp), plain mul/add/div only.
//   pm_philox4x32   : Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011), counter-based.
//   pm_expm4        : Higham 2005 (SIAM J. Matrix Anal. Appl. 26(4)) Pade scaling-and-squaring, the
//                     algorithm behind tf.linalg.expm (reference call sites vcsmc.py:183-184) and
//                     scipy.linalg.expm (csmc.py:304-305).
//
// pm_exp and pm_log restate the algorithms and constants of fdlibm's e_exp.c / e_log.c, whose licence asks that this
// notice be preserved:
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//   Developed at SunPro, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD static inline
#endif

// ------------------------------------------------------------------------------------------------
// bit casts
// ------------------------------------------------------------------------------------------------
PM_HD uint64_t pm_bits(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
PM_HD double pm_from_bits(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
PM_HD double pm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PM_HD double pm_inf() { return pm_from_bits(0x7ff0000000000000ull); }
PM_HD double pm_nan() { return pm_from_bits(0x7ff8000000000000ull); }
PM_HD bool pm_isnan(double x) { return (pm_bits(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }

// ------------------------------------------------------------------------------------------------
// exp
// ------------------------------------------------------------------------------------------------
PM_HD double pm_exp(double x) {
    const double o_threshold = 7.09782712893383973096e+02;
    const double u_threshold = -7.45133219101941108420e+02;
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    if (pm_isnan(x)) return x;
    if (x > o_threshold) return pm_inf();
    if (x < u_threshold) return 0.0;
    const double ax = x < 0.0 ? -x : x;
    double hi, lo, r;
    int k;
    if (ax > 0.34657359027997264) {            // 0.5 ln 2
        k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
        const double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
        r = hi - lo;
    } else if (ax < 3.7252902984619141e-09) {  // 2^-28
        return 1.0 + x;
    } else {
        k = 0;
        hi = x;
        lo = 0.0;
        r = x;
    }
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021) return pm_from_bits(pm_bits(y) + ((uint64_t)(int64_t)k << 52));
    // subnormal result: scale in two exact steps
    const double y2 = pm_from_bits(pm_bits(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
    return y2 * 9.33263618503218878990e-302;   // 2^-1000
}

// pm_exp restricted to x <= 0 (and not NaN) WITHOUT branches: the resampling scan exponentiates logw - max, eight values per
// thread, and straight-line code lets the eight evaluations overlap.  Bit-identical to pm_exp on its domain:
//   * k = 0 is the general reduction with t = 0 (x - 0*ln2HI = x, 0*ln2LO = 0, exactly);
//   * pm_exp's k == 0 result 1 - ((r c)/(c - 2) - r) equals the general 1 - ((lo - (r c)/(2 - c)) - hi) with lo = 0, hi = r:
//     (c - 2) = -(2 - c) exactly, so the quotient only changes sign, and 0 - q = -q bit for bit for q != 0 (q = 0 needs
//     r = 0, i.e. the |x| < 2^-28 case, which returns 1 + x);
//   * the result cases (tiny, underflow, normal and subnormal scaling) are selected at the end.
// tests/test_gpu_parity.py compares it with pm_exp on a million arguments and the edge cases.
PM_HD double pm_exp_nonpos(double x) {
    const double u_threshold = -7.45133219101941108420e+02;
    const double ln2HI = 6.93147180369123816490e-01;
    const double ln2LO = 1.90821492927058770002e-10;
    const double invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;
    const double ax = -x;
    const double xc = x < -800.0 ? -800.0 : x;                     // keep the int conversion in range (result is 0 anyway)
    const int k = ax > 0.34657359027997264 ? (int)(invln2 * xc + -0.5) : 0;
    const double t = (double)k;
    const double hi = xc - t * ln2HI;
    const double lo = t * ln2LO;
    const double r = hi - lo;
    const double tt = r * r;
    const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const double yn = pm_from_bits(pm_bits(y) + ((uint64_t)(int64_t)k << 52));
    const double ys = pm_from_bits(pm_bits(y) + ((uint64_t)(int64_t)(k + 1000) << 52)) * 9.33263618503218878990e-302;
    double res = k >= -1021 ? yn : ys;
    res = ax < 3.7252902984619141e-09 ? 1.0 + x : res;
    res = x < u_threshold ? 0.0 : res;
    return res;
}

// ------------------------------------------------------------------------------------------------
// log
// ------------------------------------------------------------------------------------------------
PM_HD double pm_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01;
    const double Lg2 = 3.999999999940941908e-01;
    const double Lg3 = 2.857142874366239149e-01;
    const double Lg4 = 2.222219843214978396e-01;
    const double Lg5 = 1.818357216161805012e-01;
    const double Lg6 = 1.531383769920937332e-01;
    const double Lg7 = 1.479819860511658591e-01;
    uint64_t u = pm_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -pm_inf();           // log(+-0) = -inf
        if (hx >> 31) return pm_nan();                 // log(-#) = NaN
        k -= 54;                                       // subnormal: scale up
        x = x * 18014398509481984.0;                   // 2^54
        u = pm_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;                                      // inf or NaN
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;                   // reduce x into [sqrt(2)/2, sqrt(2))
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    u = ((uint64_t)hx << 32) | (u & 0xffffffffull);
    x = pm_from_bits(u);
    const double f = x - 1.0;
    const double hfsq = 0.5 * f * f;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

// ------------------------------------------------------------------------------------------------
// Log of a running product: sum_s log(x_s) over the sites of one canonical column is accumulated as a
// mantissa product p in [1,2) and an integer exponent E (log prod = log sum; one log per column instead of
// one per site, and a smaller rounding error than adding up logs).  Factors that are not positive normal
// numbers (zero, subnormal, negative, inf, NaN) go through pm_log into `extra` so their special values
// propagate exactly as they would in a plain sum of logs.
//
// Contract v5: the S sites of a row are cut into TILES of T = site tile sites (a multiple of 64;
// pm_site_tile is the policy, a context can override it); inside a tile, site s belongs to column (s - tile start) mod 64
// -- ONE WAVE owns a (row, tile), lane = column -- ; the 64 finished columns are added by the adjacent-pair tree and the tile
// values left to right.
// ------------------------------------------------------------------------------------------------
PM_HD int pm_site_tile(int S) { (void)S; return 2048; }

struct pm_lp { double p; int E; double extra; };

PM_HD pm_lp pm_lp_init() { pm_lp a = {1.0, 0, 0.0}; return a; }

PM_HD void pm_lp_mul(pm_lp& a, double x) {
    const uint64_t bx = pm_bits(x);
    const int ex = (int)((bx >> 52) & 0x7ff);
    // negative, zero / subnormal, inf / nan: everything outside the positive normal range, in ONE unsigned compare
    if (bx - 0x0010000000000000ull >= 0x7fe0000000000000ull) {
        a.extra = a.extra + pm_log(x);
        return;
    }
    const double mx = pm_from_bits((bx & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    a.p = a.p * mx;                                    // [1, 4)
    const uint64_t bp = pm_bits(a.p);
    a.E += (ex - 1023) + ((int)((bp >> 52) & 0x7ff) - 1023);
    a.p = pm_from_bits((bp & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
}

// Two factors with ONE renormalisation -- bit for bit pm_lp_mul(a, x1); pm_lp_mul(a, x2), for every input:
//   x = m 2^e with m in [1,2): fl(p x) = fl(p m) 2^e as long as nothing leaves the normal range (rounding to 53 bits does not
//   depend on the scale), so q = fl(fl(p x1) x2) carries the mantissa of the two per-factor updates and the sum of their
//   exponents.  p >= 1 and x1 >= 2^-1022 keep fl(p x1) out of the subnormals; an overflow there gives q = inf; the second
//   product was rounded with full precision iff q >= 2^-1021 (a q that ROUNDS UP to 2^-1022 was rounded on the subnormal grid).
//   So: both factors positive normal and 2^-1021 <= q < inf -> take q; anything else -> the two per-factor updates.
PM_HD void pm_lp_mul2(pm_lp& a, double x1, double x2) {
    const double q = (a.p * x1) * x2;
    const uint64_t bq = pm_bits(q);
    const uint32_t h1 = (uint32_t)(pm_bits(x1) >> 32), h2 = (uint32_t)(pm_bits(x2) >> 32), hq = (uint32_t)(bq >> 32);
    const bool ok = (h1 - 0x00100000u < 0x7fe00000u) & (h2 - 0x00100000u < 0x7fe00000u) & (hq - 0x00200000u < 0x7fd00000u);
    if (ok) {
        a.E += (int)(hq >> 20) - 1023;
        a.p = pm_from_bits((bq & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    } else {
        pm_lp_mul(a, x1);
        pm_lp_mul(a, x2);
    }
}

PM_HD double pm_lp_finish(const pm_lp& a) {
    const double dE = (double)a.E;
    return ((pm_log(a.p) + dE * 1.90821492927058770002e-10) + dE * 6.93147180369123816490e-01) + a.extra;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10.  key = 64-bit seed; counter = (c0, c1, c2, c3).
// ------------------------------------------------------------------------------------------------
struct pm_u32x4 { uint32_t x, y, z, w; };

PM_HD pm_u32x4 pm_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    pm_u32x4 r = {c0, c1, c2, c3};
    return r;
}

enum { PM_STREAM_PAIR = 0, PM_STREAM_BRANCH = 1, PM_STREAM_RESAMPLE = 2, PM_STREAM_TWIST = 3 };
#define PM_CDF_SCALE 17592186044416.0 /* 2^44 */

// 64 random bits -> double in (0,1]
PM_HD double pm_unit_oc(uint32_t lo, uint32_t hi) {
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    return ((double)(b >> 11) + 1.0) * 1.1102230246251565404e-16;   // 2^-53
}

PM_HD uint64_t pm_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// integer resampling weight of a particle: floor(exp(logw - m) * 2^44); NaN counts as -inf
PM_HD uint64_t pm_weight_int(double logw, double m, bool all_bad) {
    if (all_bad) return 1ull;
    if (pm_isnan(logw)) return 0ull;
    return (uint64_t)(pm_exp(logw - m) * PM_CDF_SCALE);
}

// ------------------------------------------------------------------------------------------------
// 4x4 matrix exponential, Higham 2005 Algorithm 2.3 (orders 3,5,7,9,13 by ||A||_1).
// Row-major double[16].  Every product c_ij is the fma chain a_i0*b_0j, +a_i1*b_1j, ...
// ------------------------------------------------------------------------------------------------
PM_HD void pm_mm4(const double* a, const double* b, double* c) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = a[i * 4 + 0] * b[0 * 4 + j];
            acc = pm_fma(a[i * 4 + 1], b[1 * 4 + j], acc);
            acc = pm_fma(a[i * 4 + 2], b[2 * 4 + j], acc);
            acc = pm_fma(a[i * 4 + 3], b[3 * 4 + j], acc);
            c[i * 4 + j] = acc;
        }
}

// solve D X = Nm (4x4, 4 right-hand sides) by LU with partial pivoting; D and Nm are overwritten,
// the solution is left in Nm.
PM_HD void pm_solve4(double* D, double* Nm) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = D[c * 4 + c] < 0.0 ? -D[c * 4 + c] : D[c * 4 + c];
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double v = D[r * 4 + c] < 0.0 ? -D[r * 4 + c] : D[r * 4 + c];
            if (v > best) { best = v; p = r; }
        }
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {   // swap without dynamic register indexing
            if (p == r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double t = D[c * 4 + j]; D[c * 4 + j] = D[r * 4 + j]; D[r * 4 + j] = t;
                    t = Nm[c * 4 + j]; Nm[c * 4 + j] = Nm[r * 4 + j]; Nm[r * 4 + j] = t;
                }
            }
        }
        const double piv = D[c * 4 + c];
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            const double l = D[r * 4 + c] / piv;
#pragma unroll
            for (int j = c + 1; j < 4; ++j) D[r * 4 + j] = pm_fma(-l, D[c * 4 + j], D[r * 4 + j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) Nm[r * 4 + j] = pm_fma(-l, Nm[c * 4 + j], Nm[r * 4 + j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {           // back substitution, column j of the right-hand side
#pragma unroll
        for (int r = 3; r >= 0; --r) {
            double acc = Nm[r * 4 + j];
#pragma unroll
            for (int c = 3; c > r; --c) acc = pm_fma(-D[r * 4 + c], Nm[c * 4 + j], acc);
            Nm[r * 4 + j] = acc / D[r * 4 + r];
        }
    }
}

PM_HD void pm_expm4(const double* Q, double t, double* P) {
    double A[16], A2[16], U[16], V[16], W[16], T1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) A[i] = Q[i] * t;       // tf.tensordot(t, Q, 0), vcsmc.py:181
    double norm = 0.0;                                  // ||A||_1 = max column sum of |a_ij|
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double cs = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cs = cs + (A[i * 4 + j] < 0.0 ? -A[i * 4 + j] : A[i * 4 + j]);
        if (cs > norm) norm = cs;
    }
    int s = 0;
    if (!(norm <= 5.371920351148152)) {                 // theta_13
        double lim = 5.371920351148152;
        while (norm > lim && s < 1000) { lim = lim * 2.0; ++s; }
        const double sc = pm_from_bits((uint64_t)(1023 - s) << 52);   // 2^-s, exact
#pragma unroll
        for (int i = 0; i < 16; ++i) A[i] = A[i] * sc;
    }
    pm_mm4(A, A, A2);
    if (norm <= 1.495585217958292e-2) {                 // Pade 3: b = {120, 60, 12, 1}
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            W[i] = A2[i] + 60.0 * id;                   // b3 A2 + b1 I
            V[i] = 12.0 * A2[i] + 120.0 * id;           // b2 A2 + b0 I
        }
        pm_mm4(A, W, U);
    } else if (norm <= 2.539398330063230e-1) {          // Pade 5: {30240,15120,3360,420,30,1}
        double A4[16];
        pm_mm4(A2, A2, A4);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            W[i] = (A4[i] + 420.0 * A2[i]) + 15120.0 * id;
            V[i] = (30.0 * A4[i] + 3360.0 * A2[i]) + 30240.0 * id;
        }
        pm_mm4(A, W, U);
    } else if (norm <= 9.504178996162932e-1) {          // Pade 7
        double A4[16], A6[16];
        pm_mm4(A2, A2, A4);
        pm_mm4(A4, A2, A6);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            W[i] = ((A6[i] + 1512.0 * A4[i]) + 277200.0 * A2[i]) + 8648640.0 * id;
            V[i] = ((56.0 * A6[i] + 25200.0 * A4[i]) + 1995840.0 * A2[i]) + 17297280.0 * id;
        }
        pm_mm4(A, W, U);
    } else if (norm <= 2.097847961257068e0) {           // Pade 9
        double A4[16], A6[16], A8[16];
        pm_mm4(A2, A2, A4);
        pm_mm4(A4, A2, A6);
        pm_mm4(A6, A2, A8);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            W[i] = (((A8[i] + 3960.0 * A6[i]) + 2162160.0 * A4[i]) + 302702400.0 * A2[i]) + 8821612800.0 * id;
            V[i] = (((90.0 * A8[i] + 110880.0 * A6[i]) + 30270240.0 * A4[i]) + 2075673600.0 * A2[i]) +
                   17643225600.0 * id;
        }
        pm_mm4(A, W, U);
    } else {                                            // Pade 13
        double A4[16], A6[16];
        pm_mm4(A2, A2, A4);
        pm_mm4(A4, A2, A6);
        // U = A [ A6 (b13 A6 + b11 A4 + b9 A2) + b7 A6 + b5 A4 + b3 A2 + b1 I ]
#pragma unroll
        for (int i = 0; i < 16; ++i) T1[i] = (A6[i] + 16380.0 * A4[i]) + 40840800.0 * A2[i];
        pm_mm4(A6, T1, W);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            W[i] = (((W[i] + 33522128640.0 * A6[i]) + 10559470521600.0 * A4[i]) + 1187353796428800.0 * A2[i]) +
                   32382376266240000.0 * id;
        }
        pm_mm4(A, W, U);
        // V = A6 (b12 A6 + b10 A4 + b8 A2) + b6 A6 + b4 A4 + b2 A2 + b0 I
#pragma unroll
        for (int i = 0; i < 16; ++i) T1[i] = (182.0 * A6[i] + 960960.0 * A4[i]) + 1323241920.0 * A2[i];
        pm_mm4(A6, T1, V);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double id = (i % 5 == 0) ? 1.0 : 0.0;
            V[i] = (((V[i] + 670442572800.0 * A6[i]) + 129060195264000.0 * A4[i]) + 7771770303897600.0 * A2[i]) +
                   64764752532480000.0 * id;
        }
    }
    double D[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        D[i] = V[i] - U[i];
        P[i] = V[i] + U[i];
    }
    pm_solve4(D, P);
    for (int q = 0; q < s; ++q) {
        pm_mm4(P, P, T1);
#pragma unroll
        for (int i = 0; i < 16; ++i) P[i] = T1[i];
    }
}

// JC69 closed form (vcsmc.py:126-129 Q: off-diagonal 1/4, diagonal -3/4): P_ii = 1/4 + 3/4 e^-t
PM_HD void pm_jc69(double t, double* P) {
    const double e = pm_exp(-t);
    const double d = 0.25 + 0.75 * e;
    const double o = 0.25 - 0.25 * e;
#pragma unroll
    for (int i = 0; i < 16; ++i) P[i] = (i % 5 == 0) ? d : o;
}
