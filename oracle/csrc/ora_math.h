/* ora_math.h -- CPU oracle's own statement of the arithmetic contract (DESIGN.md "Arithmetic contract").
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/cpu_ref.py header).  Written independently of
 * phylo_amd/csrc/phylo_math.h from the same published algorithms; because both sides use only IEEE
 * binary64 + - * / and explicit fma with contraction disabled, equal operation sequences give equal
 * bits, and tests/test_gpu_parity.py checks exactly that.  Plain C11, gcc -ffp-contract=off -mfma.
 *
 *   exp, log  : FreeBSD msun / fdlibm e_exp.c, e_log.c (Sun Microsystems 1993, public algorithm)
 *   Philox    : Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11
 *   expm      : Higham, "The scaling and squaring method for the matrix exponential revisited",
 *               SIAM J. Matrix Anal. Appl. 26(4), 2005, Algorithm 2.3 -- what tf.linalg.expm
 *               (tensorflow 1.15, call sites vcsmc.py:183-184) and scipy.linalg.expm (csmc.py:304-305)
 *               implement.
 *
 * ora_exp and ora_log restate the algorithms and constants of fdlibm's e_exp.c / e_log.c, whose licence asks that
 * this notice be preserved:
 *   ====================================================
 *   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
 *   Developed at SunPro, a Sun Microsystems, Inc. business.
 *   Permission to use, copy, modify, and distribute this
 *   software is freely granted, provided that this notice
 *   is preserved.
 *   ====================================================
 */
#ifndef ORA_MATH_H
#define ORA_MATH_H
#include <stdint.h>
#include <string.h>

static inline uint64_t ora_bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double ora_dbl(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline double ora_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#define ORA_INF ora_dbl(0x7ff0000000000000ull)
#define ORA_NAN ora_dbl(0x7ff8000000000000ull)
static inline int ora_isnan(double x) { return (ora_bits(x) & 0x7fffffffffffffffull) > 0x7ff0000000000000ull; }

static inline double ora_exp(double x) {
    static const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00;
    static const double P[5] = {1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05,
                                -1.65339022054652515390e-06, 4.13813679705723846039e-08};
    if (ora_isnan(x)) return x;
    if (x > 7.09782712893383973096e+02) return ORA_INF;
    if (x < -7.45133219101941108420e+02) return 0.0;
    double ax = x < 0.0 ? -x : x;
    double hi = x, lo = 0.0, r = x;
    int k = 0;
    if (ax > 0.34657359027997264) {
        k = (int)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
        double t = (double)k;
        hi = x - t * ln2HI;
        lo = t * ln2LO;
        r = hi - lo;
    } else if (ax < 3.7252902984619141e-09) {
        return 1.0 + x;
    }
    double t = r * r;
    double c = r - t * (P[0] + t * (P[1] + t * (P[2] + t * (P[3] + t * P[4]))));
    if (k == 0) return 1.0 - ((r * c) / (c - 2.0) - r);
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    if (k >= -1021) return ora_dbl(ora_bits(y) + ((uint64_t)(int64_t)k << 52));
    y = ora_dbl(ora_bits(y) + ((uint64_t)(int64_t)(k + 1000) << 52));
    return y * 9.33263618503218878990e-302;
}

static inline double ora_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    static const double Lg[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
                                 2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
                                 1.479819860511658591e-01};
    uint64_t u = ora_bits(x);
    uint32_t hx = (uint32_t)(u >> 32);
    int k = 0;
    if (hx < 0x00100000u || (hx >> 31)) {
        if ((u << 1) == 0) return -ORA_INF;
        if (hx >> 31) return ORA_NAN;
        k -= 54;
        x = x * 18014398509481984.0;
        u = ora_bits(x);
        hx = (uint32_t)(u >> 32);
    } else if (hx >= 0x7ff00000u) {
        return x;
    } else if (hx == 0x3ff00000u && (u << 32) == 0) {
        return 0.0;
    }
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = ora_dbl(((uint64_t)hx << 32) | (u & 0xffffffffull));
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]));
    double t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])));
    double R = t2 + t1;
    double dk = (double)k;
    return s * (hfsq + R) + dk * ln2_lo - hfsq + f + dk * ln2_hi;
}

static inline void ora_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double ora_unit_oc(uint32_t lo, uint32_t hi) {
    uint64_t b = ((uint64_t)hi << 32) | lo;
    return ((double)(b >> 11) + 1.0) * 1.1102230246251565404e-16;
}

static inline uint64_t ora_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

/* ---- 4x4 helpers ------------------------------------------------------------------------------ */
static inline void ora_mm4(const double* a, const double* b, double* c) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = a[i * 4] * b[j];
            for (int k = 1; k < 4; ++k) acc = ora_fma(a[i * 4 + k], b[k * 4 + j], acc);
            c[i * 4 + j] = acc;
        }
}

static inline double ora_abs(double x) { return x < 0.0 ? -x : x; }

/* Gaussian elimination with partial pivoting (first maximal row wins), solution left in B. */
static inline void ora_solve4(double* D, double* B) {
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = ora_abs(D[c * 4 + c]);
        for (int r = c + 1; r < 4; ++r)
            if (ora_abs(D[r * 4 + c]) > best) { best = ora_abs(D[r * 4 + c]); p = r; }
        if (p != c)
            for (int j = 0; j < 4; ++j) {
                double t = D[c * 4 + j]; D[c * 4 + j] = D[p * 4 + j]; D[p * 4 + j] = t;
                t = B[c * 4 + j]; B[c * 4 + j] = B[p * 4 + j]; B[p * 4 + j] = t;
            }
        double piv = D[c * 4 + c];
        for (int r = c + 1; r < 4; ++r) {
            double l = D[r * 4 + c] / piv;
            for (int j = c + 1; j < 4; ++j) D[r * 4 + j] = ora_fma(-l, D[c * 4 + j], D[r * 4 + j]);
            for (int j = 0; j < 4; ++j) B[r * 4 + j] = ora_fma(-l, B[c * 4 + j], B[r * 4 + j]);
        }
    }
    for (int j = 0; j < 4; ++j)
        for (int r = 3; r >= 0; --r) {
            double acc = B[r * 4 + j];
            for (int c = 3; c > r; --c) acc = ora_fma(-D[r * 4 + c], B[c * 4 + j], acc);
            B[r * 4 + j] = acc / D[r * 4 + r];
        }
}

/* Pade numerator/denominator pieces: given the even powers, W = sum_odd b_{2i+1} A^{2i} (so U = A W)
 * and V = sum_even b_{2i} A^{2i}; coefficients of Higham 2005 Table / eq. (10.33). */
static inline void ora_expm4(const double* Q, double t, double* P) {
    static const double th3 = 1.495585217958292e-2, th5 = 2.539398330063230e-1, th7 = 9.504178996162932e-1,
                        th9 = 2.097847961257068e0, th13 = 5.371920351148152;
    double A[16], A2[16], A4[16], A6[16], A8[16], U[16], V[16], W[16], T[16], D[16];
    for (int i = 0; i < 16; ++i) A[i] = Q[i] * t;
    double norm = 0.0;
    for (int j = 0; j < 4; ++j) {
        double cs = 0.0;
        for (int i = 0; i < 4; ++i) cs = cs + ora_abs(A[i * 4 + j]);
        if (cs > norm) norm = cs;
    }
    int s = 0;
    if (!(norm <= th13)) {
        double lim = th13;
        while (norm > lim && s < 1000) { lim = lim * 2.0; ++s; }
        double sc = ora_dbl((uint64_t)(1023 - s) << 52);
        for (int i = 0; i < 16; ++i) A[i] = A[i] * sc;
    }
    ora_mm4(A, A, A2);
#define ID(i) (((i) % 5 == 0) ? 1.0 : 0.0)
    if (norm <= th3) {
        for (int i = 0; i < 16; ++i) { W[i] = A2[i] + 60.0 * ID(i); V[i] = 12.0 * A2[i] + 120.0 * ID(i); }
        ora_mm4(A, W, U);
    } else if (norm <= th5) {
        ora_mm4(A2, A2, A4);
        for (int i = 0; i < 16; ++i) {
            W[i] = (A4[i] + 420.0 * A2[i]) + 15120.0 * ID(i);
            V[i] = (30.0 * A4[i] + 3360.0 * A2[i]) + 30240.0 * ID(i);
        }
        ora_mm4(A, W, U);
    } else if (norm <= th7) {
        ora_mm4(A2, A2, A4);
        ora_mm4(A4, A2, A6);
        for (int i = 0; i < 16; ++i) {
            W[i] = ((A6[i] + 1512.0 * A4[i]) + 277200.0 * A2[i]) + 8648640.0 * ID(i);
            V[i] = ((56.0 * A6[i] + 25200.0 * A4[i]) + 1995840.0 * A2[i]) + 17297280.0 * ID(i);
        }
        ora_mm4(A, W, U);
    } else if (norm <= th9) {
        ora_mm4(A2, A2, A4);
        ora_mm4(A4, A2, A6);
        ora_mm4(A6, A2, A8);
        for (int i = 0; i < 16; ++i) {
            W[i] = (((A8[i] + 3960.0 * A6[i]) + 2162160.0 * A4[i]) + 302702400.0 * A2[i]) + 8821612800.0 * ID(i);
            V[i] = (((90.0 * A8[i] + 110880.0 * A6[i]) + 30270240.0 * A4[i]) + 2075673600.0 * A2[i]) +
                   17643225600.0 * ID(i);
        }
        ora_mm4(A, W, U);
    } else {
        ora_mm4(A2, A2, A4);
        ora_mm4(A4, A2, A6);
        for (int i = 0; i < 16; ++i) T[i] = (A6[i] + 16380.0 * A4[i]) + 40840800.0 * A2[i];
        ora_mm4(A6, T, W);
        for (int i = 0; i < 16; ++i)
            W[i] = (((W[i] + 33522128640.0 * A6[i]) + 10559470521600.0 * A4[i]) + 1187353796428800.0 * A2[i]) +
                   32382376266240000.0 * ID(i);
        ora_mm4(A, W, U);
        for (int i = 0; i < 16; ++i) T[i] = (182.0 * A6[i] + 960960.0 * A4[i]) + 1323241920.0 * A2[i];
        ora_mm4(A6, T, V);
        for (int i = 0; i < 16; ++i)
            V[i] = (((V[i] + 670442572800.0 * A6[i]) + 129060195264000.0 * A4[i]) + 7771770303897600.0 * A2[i]) +
                   64764752532480000.0 * ID(i);
    }
#undef ID
    for (int i = 0; i < 16; ++i) { D[i] = V[i] - U[i]; P[i] = V[i] + U[i]; }
    ora_solve4(D, P);
    for (int q = 0; q < s; ++q) {
        ora_mm4(P, P, T);
        for (int i = 0; i < 16; ++i) P[i] = T[i];
    }
}

static inline void ora_jc69(double t, double* P) {
    double e = ora_exp(-t);
    double d = 0.25 + 0.75 * e, o = 0.25 - 0.25 * e;
    for (int i = 0; i < 16; ++i) P[i] = (i % 5 == 0) ? d : o;
}

/* log of a running product (see DESIGN.md "Arithmetic contract"): mantissa product in [1,2) + integer
 * exponent; factors that are not positive normal numbers are routed through ora_log into `extra`. */
typedef struct { double p; int E; double extra; } ora_lp;
static inline void ora_lp_init(ora_lp* a) { a->p = 1.0; a->E = 0; a->extra = 0.0; }
static inline void ora_lp_mul(ora_lp* a, double x) {
    uint64_t bx = ora_bits(x);
    int ex = (int)((bx >> 52) & 0x7ff);
    if ((bx >> 63) || ex == 0 || ex == 0x7ff) { a->extra = a->extra + ora_log(x); return; }
    double mx = ora_dbl((bx & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    a->p = a->p * mx;
    uint64_t bp = ora_bits(a->p);
    a->E += (ex - 1023) + ((int)((bp >> 52) & 0x7ff) - 1023);
    a->p = ora_dbl((bp & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
}
static inline double ora_lp_finish(const ora_lp* a) {
    double dE = (double)a->E;
    return ((ora_log(a->p) + dE * 1.90821492927058770002e-10) + dE * 6.93147180369123816490e-01) + a->extra;
}

/* canonical sum of logs over sites (contract v5): the S sites are cut into TILES of T sites
 * (T = ora_site_tile(S), a multiple of 64); inside a tile site s multiplies into log-product column (s - tile start) mod 64,
 * in increasing s; each column is finished to a double, the 64 column values are added by the adjacent-pair tree, and the
 * tile values are added left to right.  (Until contract v4: one tile, 256 columns.) */
static int ora_tile_override = 0;                       /* tests / experiments: ora_set_site_tile */
static inline int ora_site_tile(long S) { (void)S; return ora_tile_override > 0 ? ora_tile_override : 2048; }
static inline double ora_tree64(double* v) {
    for (int st = 1; st < 64; st <<= 1)
        for (int i = 0; i < 64; i += 2 * st) v[i] = v[i] + v[i + st];
    return v[0];
}
typedef struct { ora_lp col[64]; long T, tile; int ntiles; double total; } ora_canon_lp;
static inline void ora_canon_lp_init(ora_canon_lp* c, long S) {
    for (int i = 0; i < 64; ++i) ora_lp_init(&c->col[i]);
    c->T = ora_site_tile(S); c->tile = 0; c->ntiles = 0; c->total = 0.0;
}
static inline void ora_canon_lp_close_tile(ora_canon_lp* c) {
    double v[64];
    for (int i = 0; i < 64; ++i) { v[i] = ora_lp_finish(&c->col[i]); ora_lp_init(&c->col[i]); }
    const double t = ora_tree64(v);
    c->total = c->ntiles ? c->total + t : t;
    ++c->ntiles;
}
/* sites must arrive in increasing s */
static inline void ora_canon_lp_mul(ora_canon_lp* c, long s, double x) {
    while (s >= (c->tile + 1) * c->T) { ora_canon_lp_close_tile(c); ++c->tile; }
    ora_lp_mul(&c->col[(s - c->tile * c->T) & 63], x);
}
static inline double ora_canon_lp_total(ora_canon_lp* c) { ora_canon_lp_close_tile(c); return c->total; }

/* a sum of at most 64 terms in the site-sum tree (contract v3: the 25 code-pair terms of a leaf-leaf look-ahead row):
 * term i sits in column i of ONE tile, every other column is 0 */
typedef struct { double col[64]; } ora_canon64;
static inline void ora_canon64_init(ora_canon64* c) { for (int i = 0; i < 64; ++i) c->col[i] = 0.0; }
static inline void ora_canon64_add(ora_canon64* c, int i, double v) { c->col[i] = c->col[i] + v; }
static inline double ora_canon64_total(ora_canon64* c) { return ora_tree64(c->col); }

/* canonical sum: 256 columns (element s goes to column s mod 256, added in increasing s), then an
 * adjacent-pair tree inside each group of 64 columns, then the four groups left to right. */
typedef struct { double col[256]; } ora_canon;
static inline void ora_canon_init(ora_canon* c) { for (int i = 0; i < 256; ++i) c->col[i] = 0.0; }
static inline void ora_canon_add(ora_canon* c, long s, double v) { c->col[s & 255] = c->col[s & 255] + v; }
static inline double ora_canon_total(ora_canon* c) {
    for (int g = 0; g < 4; ++g)
        for (int st = 1; st < 64; st <<= 1)
            for (int i = 0; i < 64; i += 2 * st) c->col[g * 64 + i] = c->col[g * 64 + i] + c->col[g * 64 + i + st];
    return ((c->col[0] + c->col[64]) + c->col[128]) + c->col[192];
}
#endif
