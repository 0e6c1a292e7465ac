/*
 * ig_detmath.h -- the arithmetic contract of the instaGRAAL likelihood path.
 *
 * Everything the per-contact Rippe/Poisson term needs (powf, expf, log10, the
 * fixed-point quantiser and the exact limb accumulator) written ONLY with
 * operations that IEEE-754 defines bit-for-bit: + - * / fma, integer ops and
 * int<->float conversions.  The same header is compiled by gcc (CPU oracle,
 * -mfma -ffp-contract=off) and by hipcc for gfx950 (-ffp-contract=off), so a
 * term evaluated on the host and on an MI355X is the same 64-bit pattern and
 * sums of quantised terms are order-independent integers.  That is what makes
 * "identical argmax / identical fragment order" a provable property instead of
 * a hope (SURVEY.md section 7 "Hard parts": libm differences, atomics order).
 *
 * Reference arithmetic being restated (file:line under /root/reference/src/instagraal):
 *   kernels/kernel_sparse_adapt.cu:111-124  factorial()
 *   kernels/kernel_sparse_adapt.cu:153-163  rippe_contacts()
 *   kernels/kernel_sparse_adapt.cu:200-225  rippe_contacts_circ()
 *   kernels/kernel_sparse_adapt.cu:251-270  evaluate_likelihood_pxl_double()
 *   kernels/kernel_sparse_adapt.cu:4172-4208, 4315-4353, 4426-4462  per-contact term
 *   kernels/kernel_sparse_adapt.cu:3882-3899, 3955-3972           zero-pixel term
 * The reference calls CUDA's powf/expf/log10; those are replaced here by
 * double-precision table-driven range reduction + short polynomials whose result,
 * rounded to the reference's precision class (float for P(s), double for the
 * Poisson term), is within 1 ulp of the correctly rounded value.
 */
#ifndef IG_DETMATH_H
#define IG_DETMATH_H

#include <stdint.h>

#include "ig_detmath_tables.h"

#if defined(__HIPCC__)
#define IG_HD __host__ __device__ __forceinline__
#else
#define IG_HD static inline
#endif

#ifdef __cplusplus
#define IG_BITCAST_FN 1
#endif

/* ---- bit casts ------------------------------------------------------- */
IG_HD uint64_t ig_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
IG_HD double ig_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
IG_HD uint32_t ig_f2u(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); return u; }
IG_HD float ig_u2f(uint32_t u) { float x; __builtin_memcpy(&x, &u, 4); return x; }

IG_HD double ig_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* fma whose addend is a compile-time constant (polynomial coefficients).  On gfx950 the three-operand form with
 * the constant in a scalar register pair is spelled out: left alone, the compiler keeps every coefficient in a
 * vector register pair and copies it before each two-operand v_fmac (twice the instructions, +40 VGPRs).
 * Same single-rounding fma, same bits. */
#if defined(__HIP_DEVICE_COMPILE__)
IG_HD double ig_fma_k(double a, double b, double k)
{
    double r;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
#else
IG_HD double ig_fma_k(double a, double b, double k) { return __builtin_fma(a, b, k); }
#endif

#define IG_INF (ig_u2d(0x7ff0000000000000ULL))
#define IG_NAN (ig_u2d(0x7ff8000000000000ULL))
#define IG_INFF (ig_u2f(0x7f800000u))
#define IG_NANF (ig_u2f(0x7fc00000u))

IG_HD int ig_isnan(double x) { return (ig_d2u(x) & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL; }
IG_HD int ig_isinf(double x) { return (ig_d2u(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
IG_HD int ig_isnanf(float x) { return (ig_f2u(x) & 0x7fffffffu) > 0x7f800000u; }
IG_HD int ig_isinff(float x) { return (ig_f2u(x) & 0x7fffffffu) == 0x7f800000u; }

/* fmaxf semantics of CUDA's max(float,float): a NaN operand loses. */
IG_HD float ig_fmaxf(float a, float b)
{
    if (ig_isnanf(a)) return b;
    if (ig_isnanf(b)) return a;
    return a > b ? a : b;
}

/* One table for both functions: [0,256) pairs (r_j, -log2 r_j) of the 128 log intervals, [256,384) 2^(j/128).
 * Every function below takes the table pointer explicitly so that a GPU kernel can pass its LDS copy
 * (ds_read instead of a global gather); ig_tab() is the constant-memory / static copy. */
#define IG_TAB_SIZE 384
#define IG_TAB_EXP 256
IG_HD const double* ig_tab(void)
{
    static const double tab[IG_TAB_SIZE] = IG_TAB_INIT;
    return tab;
}

/* frexp of a positive, finite, NORMAL double: x = m * 2^e, m in [1/2, 1).  (One instruction each on gfx950; for
 * other inputs the two targets differ, and every caller discards the result then.) */
#if defined(__HIP_DEVICE_COMPILE__)
IG_HD double ig_frexp_mant(double x) { return __builtin_amdgcn_frexp_mant(x); }
IG_HD int ig_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
IG_HD double ig_scale2(double v, int k) { return __builtin_ldexp(v, k); }
#else
IG_HD double ig_frexp_mant(double x) { return ig_u2d((ig_d2u(x) & 0x000fffffffffffffULL) | 0x3fe0000000000000ULL); }
IG_HD int ig_frexp_exp(double x) { return (int)((ig_d2u(x) >> 52) & 0x7ffu) - 1022; }
/* v * 2^k, exact: the callers keep the result normal */
IG_HD double ig_scale2(double v, int k) { return v * ig_u2d((uint64_t)(k + 1023) << 52); }
#endif

/* log2 of a positive, finite, NORMAL double.  m in [1/2,1) falls in interval j (its 7 leading fraction bits); with the
 * tabulated reciprocal r_j of the interval centre u = m r_j - 1 is tiny (|u| <= 2^-8, the fma makes it exact up to
 * its own rounding) and log2(x) = e - log2(r_j) + log2(1 + u), a degree-5 polynomial: absolute error below 3e-16
 * (what the callers need: they exponentiate it or multiply it by a count). */
IG_HD double ig_log2_pos(double x, const double* T)
{
    const int j = (int)((ig_d2u(x) >> 45) & 127u);
    const double m = ig_frexp_mant(x);
    const int e = ig_frexp_exp(x);
    const double u = ig_fma(m, T[2 * j], -1.0);
    double p = IG_LOG_C5;
    p = ig_fma_k(p, u, IG_LOG_C4);
    p = ig_fma_k(p, u, IG_LOG_C3);
    p = ig_fma_k(p, u, IG_LOG_C2);
    p = ig_fma_k(p, u, IG_LOG_C1);
    return ig_fma(p, u, T[2 * j + 1]) + (double)e;
}

/* 2^y for |y| <= 1000 (finite): y = (128 k + i)/128 + t with |t| <= 2^-8 (all steps exact),
 * 2^y = 2^k * 2^(i/128) * (1 + P(t)), P of degree 4.  128 y is rounded to an integer (half-even) by adding 1.5 * 2^52:
 * the integer is then the low word of the sum. */
#define IG_RINT_MAGIC 6755399441055744.0
IG_HD double ig_exp2_core(double y, const double* T)
{
    const double z = ig_fma(y, 128.0, IG_RINT_MAGIC);
    const int ji = (int)(uint32_t)ig_d2u(z);
    const double jd = z - IG_RINT_MAGIC;        /* exact */
    const double t = ig_fma(jd, -0.0078125, y); /* exact */
    double p = IG_EXP_C4;
    p = ig_fma_k(p, t, IG_EXP_C3);
    p = ig_fma_k(p, t, IG_EXP_C2);
    p = ig_fma_k(p, t, IG_EXP_C1);
    p = p * t;
    const double tj = T[IG_TAB_EXP + (ji & 127)];
    const double v = ig_fma(tj, p, tj);
    return ig_scale2(v, ji >> 7); /* never sub-normal for |y| <= 1000 */
}

IG_HD double ig_exp2(double y, const double* T)
{
    if (ig_isnan(y)) return IG_NAN; /* canonical NaN: payloads differ between targets */
    if (y > 1000.0) return IG_INF;
    if (y < -1000.0) return 0.0;
    return ig_exp2_core(y, T);
}

/* powf(x, y) as used by KA:159, 217 (x = distance or n, y = slope / 2.0f / -3.0f / n). */
IG_HD float ig_powf(float x, float y, const double* T)
{
    if (ig_isnanf(x) || ig_isnanf(y)) return IG_NANF;
    if (y == 0.0f) return 1.0f;
    if (y == 2.0f) return x * x;
    if (x < 0.0f) return IG_NANF; /* non-integer exponents only on this path */
    if (x == 0.0f) return y < 0.0f ? IG_INFF : 0.0f;
    if (ig_isinff(x)) return y < 0.0f ? 0.0f : IG_INFF;
    return (float)ig_exp2((double)y * ig_log2_pos((double)x, T), T);
}

/* expf(x) as used by KA:121, 159, 217. */
IG_HD float ig_expf(float x, const double* T)
{
    if (ig_isnanf(x)) return IG_NANF;
    return (float)ig_exp2((double)x * IG_LOG2_E, T);
}

/* log10(x), double, as used by KA:259-262. */
IG_HD double ig_log10(double x, const double* T)
{
    if (ig_isnan(x)) return IG_NAN;
    if (x < 0.0) return IG_NAN;
    if (x == 0.0) return -IG_INF;
    if (ig_isinf(x)) return x;
    return ig_log2_pos(x, T) * IG_LOG2_10_INV;
}

/* ---- model ------------------------------------------------------------- */
typedef struct ig_params {
    float kuhn, lm, c1, slope, d, d_max, fact, v_inter; /* KA:91-100, same order */
} ig_params;

/* KA:153-163 */
IG_HD float ig_rippe(float s, const ig_params p, const double* T)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max)) {
        float pw = ig_powf(s, p.slope, T);
        float e = 1.0f; /* d == 2 (always, optim_rippe_curve_update.py:8): expf(0 / (t^2 + 2)) is exactly 1 */
        if (p.d != 2.0f) {
            float t = s * p.lm / p.kuhn;
            e = ig_expf((p.d - 2.0f) / (ig_powf(t, 2.0f, T) + p.d), T);
        }
        result = (p.c1 * pw * e) * p.fact;
    }
    return ig_fmaxf(result, p.v_inter);
}

/* KA:200-225 (note the clamp with d_max, quirk Q6) */
IG_HD float ig_rippe_circ(float s, float s_tot, const ig_params p, const double* T)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max)) {
        float K = p.lm / p.kuhn;
        float n = K * s * (s_tot - s) / s_tot;
        result = (ig_powf(p.kuhn, -3.0f, T) * ig_powf(n, p.slope, T) *
                  ig_expf((p.d - 2.0f) / (ig_powf(n, 2.0f, T) + p.d), T)) *
                 p.fact;
    }
    return ig_fmaxf(result, p.d_max);
}

/* log10(ob!) part of KA:251-270; ob >= 1. `lgf_small` = table of
 * log10((double)factorial_f32(ob)) for ob = 0..14 (KA:111-124), built once on
 * the host by ig_build_lgf_table(). */
IG_HD double ig_lgfact(int ob, const double* lgf_small, const double* T)
{
    if (ob < 15) return lgf_small[ob];
    double o = (double)ob;
    /* ob*log10(ob) - ob + log10(sqrt(2*pi*ob)); the sqrt is folded into the log */
    return (o * ig_log10(o, T) - o) + 0.5 * ig_log10(o * 2.0 * 3.14159265358979323846, T);
}

/* The float literal 0.43429448190325182f of KA:4019, 4208, 4353, 4462. */
#define IG_LOG_E_F ((double)0.43429448190325182f)

/* One non-zero pixel: KA:4207-4208 / 4352-4353 / 4462.
 * ex, ex_z : expected contacts (float, already clamped), ob : observed count,
 * lgf = ig_lgfact(ob).  Evaluation order follows the reference expression. */
IG_HD double ig_pixel_term(float ex, float ex_z, int ob, double lgf, const double* T)
{
    double e = (double)ex;
    double res = 0.0;
    if (e != 0.0) {
        double o = (double)ob;
        if (ob > 0) res = o * ig_log10(e, T) - e - lgf;
        else res = -e;
    }
    return res + (double)ex_z * IG_LOG_E_F;
}

/* ---- the per-contact term --------------------------------------------------------------------------------
 * Linear contig, d == 2 (the only value the reference ever uses, optim_rippe_curve_update.py:8), sane parameters:
 *     P(s) = amp * s^slope,  amp = c1 * fact,  clamped below by v_inter and replaced by it outside (0, d_max);
 *     term = ob * log10 P(s) - P(s) - log10(ob!) + P_z * log10(e)               (KA:4426-4462, 251-270)
 * Evaluated with ONE log2 and ONE exp2 in double: y = slope * log2(s) + log2(amp), clamped below by log2(v_inter)
 * (the max of KA:153-163 taken on the exponents), P = 2^y, log10 P = y * log10(2)
 * (the reference rounds P(s) to float and takes log10 of that float again: a second log for 1e-7 of relative
 * difference per term, inside its own float noise).  ig_hot holds what does not depend on the contact.
 * Everything else (circular contigs, d != 2, degenerate parameters, ob == 0) composes ig_rippe* and ig_pixel_term. */
typedef struct ig_hot {
    int fast;         /* the parameters are in the domain of the one-log evaluation */
    float d_max, v_inter;
    double slope, log2_amp, log2_v_inter;
} ig_hot;

IG_HD ig_hot ig_hot_make(const ig_params p, const double* T)
{
    ig_hot h;
    h.d_max = p.d_max;
    h.v_inter = p.v_inter;
    h.slope = (double)p.slope;
    const double amp = (double)p.c1 * (double)p.fact; /* exact: two 24-bit significands */
    /* |slope * log2(s) + log2(amp)| < 6.5 * 149 + 30 < 1000 for every positive float s (log2 in [-149, 128]):
     * ig_exp2_core's range is respected; all comparisons false for NaNs */
    h.fast = (p.d == 2.0f) && (p.slope != 0.0f) && (p.slope != 2.0f) && (p.slope > -6.5f) && (p.slope < 6.5f) &&
             (p.v_inter > 0.0f) && (p.v_inter < IG_INFF) && (amp > 0x1p-30) && (amp < 0x1p30);
    h.log2_amp = h.fast ? ig_log2_pos(amp, T) : 0.0;
    h.log2_v_inter = h.fast ? ig_log2_pos((double)p.v_inter, T) : 0.0;
    return h;
}

/* h->fast, ob > 0.  inter: trans pair (P = P_z = v_inter, the caller passes ex_z = v_inter).  ig_log2_pos of an s
 * outside (0, inf) is finite garbage or NaN and is discarded by the select. */
IG_HD double ig_term_hot(float s, int inter, int ob, double lgf, float ex_z, const ig_hot* h, const double* T)
{
    const int in = (s > 0.0f) && (s < h->d_max) && !inter;
    const double y = ig_fma(h->slope, ig_log2_pos((double)s, T), h->log2_amp);
    const double yc = (y > h->log2_v_inter) ? y : h->log2_v_inter;
    const double yy = in ? yc : h->log2_v_inter; /* else: the trans level */
    const double ex = ig_exp2_core(yy, T);
    const double lg = yy * IG_LOG2_10_INV;
    const double t = ig_fma((double)ob, lg, -ex) - lgf;
    return t + (double)ex_z * IG_LOG_E_F;
}

/* one (i, j) contact: KA:4430-4462 (same text at 4182-4208, 4327-4353) */
IG_HD double ig_pair_term(const ig_params p, const ig_hot* h, int cis, float s, float s_z, float s_tot, float s_tot_z, int ob,
                          double lgf, const double* T)
{
    float ex, ex_z;
    if (cis) {
        if (s_tot == 0) {
            ex_z = (s_z < p.d_max) ? ig_rippe(s_z, p, T) : p.v_inter;
            if (h->fast && ob > 0) return ig_term_hot(s, 0, ob, lgf, ex_z, h, T);
            ex = ig_rippe(s, p, T);
        } else {
            ex = ig_rippe_circ(s, s_tot, p, T);
            ex_z = (s_z < p.d_max) ? ig_rippe_circ(s_z, s_tot_z, p, T) : p.v_inter;
        }
    } else {
        ex = p.v_inter;
        ex_z = p.v_inter;
        if (h->fast && ob > 0) return ig_term_hot(0.0f, 1, ob, lgf, ex_z, h, T);
    }
    return ig_pixel_term(ex, ex_z, ob, lgf, T);
}

/* ---- exact accumulation -------------------------------------------------
 * A term is quantised to a multiple of 2^-32 (round-half-even) and added as
 * a 64-bit integer; integer addition is associative, so any thread/wave/
 * block/GPU decomposition gives the same sum.  Terms are clamped to
 * |t| < 2^20 and NaN -> 0 so the conversion is defined on every target. */
#define IG_QSCALE 4294967296.0
#define IG_QCLAMP 1048576.0

IG_HD int64_t ig_quantize(double t)
{
    if (ig_isnan(t)) return 0;
    if (t > IG_QCLAMP) t = IG_QCLAMP;
    if (t < -IG_QCLAMP) t = -IG_QCLAMP;
    return (int64_t)__builtin_rint(t * IG_QSCALE);
}

/* A sum is carried as two int64 limbs: value = hi * 2^32 + lo (lo >= 0). */
typedef struct ig_acc {
    int64_t hi, lo;
} ig_acc;

IG_HD void ig_acc_add(ig_acc* a, int64_t q)
{
    a->hi += q >> 32; /* arithmetic shift: floor(q / 2^32) */
    a->lo += (int64_t)(uint32_t)q;
}

/* limbs -> double, one rounding: value / 2^32 */
IG_HD double ig_acc_to_double(int64_t hi, int64_t lo)
{
    int64_t H = hi + (lo >> 32);
    int64_t L = lo & 0xffffffffLL;
    return (double)H + (double)L * (1.0 / 4294967296.0);
}

/* normalise limbs so that 0 <= lo < 2^32 (unique representation) */
IG_HD void ig_acc_normalize(int64_t* hi, int64_t* lo)
{
    *hi += (*lo >> 32);
    *lo &= 0xffffffffLL;
}

#endif /* IG_DETMATH_H */
