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

/* log2 of a positive, finite, NORMAL double.  m in [1,2) falls in interval j (7 bits); with the tabulated
 * reciprocal r_j of the interval centre u = m r_j - 1 is tiny (|u| <= 2^-8, the fma makes it exact up to
 * its own rounding) and log2(x) = e - log2(r_j) + log2(1 + u), a degree-5 polynomial.  Intervals above
 * sqrt(2) use m/2 (folded into r_j) and e + 1 so that log2 near 1 from below does not cancel. */
IG_HD double ig_log2_pos(double x, const double* T)
{
    const uint64_t b = ig_d2u(x);
    const int j = (int)((b >> 45) & 127u);
    const int e = (int)((b >> 52) & 0x7ffu) - 1023 + (j >= IG_LOG_SPLIT);
    const double m = ig_u2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
    const double u = ig_fma(m, T[2 * j], -1.0);
    double p = IG_LOG_C5;
    p = ig_fma_k(p, u, IG_LOG_C4);
    p = ig_fma_k(p, u, IG_LOG_C3);
    p = ig_fma_k(p, u, IG_LOG_C2);
    p = ig_fma_k(p, u, IG_LOG_C1);
    return ig_fma(p, u, T[2 * j + 1]) + (double)e;
}

/* 2^y for |y| <= 1000 (finite): y = (128 k + i)/128 + t with |t| <= 2^-8 (all steps exact),
 * 2^y = 2^k * 2^(i/128) * (1 + P(t)), P of degree 4. */
IG_HD double ig_exp2_core(double y, const double* T)
{
    const double jd = __builtin_rint(y * 128.0); /* round-half-even of an exact product */
    const double t = ig_fma(jd, -0.0078125, y);  /* exact */
    const int ji = (int)jd;
    double p = IG_EXP_C4;
    p = ig_fma_k(p, t, IG_EXP_C3);
    p = ig_fma_k(p, t, IG_EXP_C2);
    p = ig_fma_k(p, t, IG_EXP_C1);
    p = p * t;
    const double tj = T[IG_TAB_EXP + (ji & 127)];
    const double v = ig_fma(tj, p, tj);
    return v * ig_u2d((uint64_t)((ji >> 7) + 1023) << 52); /* exact scaling, never sub-normal for |y| <= 1000 */
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
 * Evaluated with ONE log2 and ONE exp2 in double: y = slope * log2(s), P = amp * 2^y, log10 P = (y + log2 amp) * log10(2)
 * (the reference rounds P(s) to float and takes log10 of that float again: a second log for 1e-7 of relative
 * difference per term, inside its own float noise).  ig_hot holds what does not depend on the contact.
 * Everything else (circular contigs, d != 2, degenerate parameters, ob == 0) composes ig_rippe* and ig_pixel_term. */
typedef struct ig_hot {
    int fast;         /* the parameters are in the domain of the one-log evaluation */
    float d_max, v_inter;
    double slope, amp, log2_amp, v_inter_d, lg_v_inter;
} ig_hot;

IG_HD ig_hot ig_hot_make(const ig_params p, const double* T)
{
    ig_hot h;
    h.d_max = p.d_max;
    h.v_inter = p.v_inter;
    h.slope = (double)p.slope;
    h.amp = (double)p.c1 * (double)p.fact; /* exact: two 24-bit significands */
    h.v_inter_d = (double)p.v_inter;
    /* |slope * log2(s)| <= 1000 for every positive float s (log2 in [-149, 128]): ig_exp2's range checks cannot fire;
     * amp a normal positive double; all comparisons false for NaNs */
    h.fast = (p.d == 2.0f) && (p.slope != 0.0f) && (p.slope != 2.0f) && (p.slope > -6.5f) && (p.slope < 6.5f) &&
             (p.v_inter > 0.0f) && (p.v_inter < IG_INFF) && (h.amp > 1e-300) && (h.amp < 1e300);
    h.log2_amp = h.fast ? ig_log2_pos(h.amp, T) : 0.0;
    h.lg_v_inter = h.fast ? ig_log2_pos(h.v_inter_d, T) * IG_LOG2_10_INV : 0.0; /* == ig_log10(v_inter) */
    return h;
}

/* h->fast, ob > 0.  inter: trans pair (P = P_z = v_inter, the caller passes ex_z = v_inter) */
IG_HD double ig_term_hot(float s, int inter, int ob, double lgf, float ex_z, const ig_hot* h, const double* T)
{
    const int in = (s > 0.0f) && (s < h->d_max);
    const double L = ig_log2_pos((double)(in ? s : 1.0f), T);
    const double y = h->slope * L;
    const double res = h->amp * ig_exp2_core(y, T);
    const int cis_model = in && !inter && (res > h->v_inter_d); /* else: the trans level (KA:153-163 max(., v_inter)) */
    const double ex = cis_model ? res : h->v_inter_d;
    const double lg = cis_model ? (y + h->log2_amp) * IG_LOG2_10_INV : h->lg_v_inter;
    const double t = (((double)ob * lg) - ex) - lgf;
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
