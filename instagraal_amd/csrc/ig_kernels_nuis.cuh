/* ig_kernels_nuis.cuh -- the nuisance step's Metropolis test (CL:2961-3051, 1296-1344) decided from a SCREENED pass.
 *
 * step_nuisance_parameters evaluates the full likelihood under its test parameters on the state before the last move and
 * accepts iff exp((L_test - L_move) / T) >= u.  The exact pass (k_full_nz_tiled) costs ~90 us at the headline shape; once
 * the chain has settled most steps are rejected, and |L_test - L_move - T ln u| is in the tens to hundreds, so a float
 * evaluation of L_test itself (bound ~1000: sum(ob), sum(P) ~ 1e8 at u = 6e-8) decides only half of them
 * (tools/nuis_margins.py).  What is screened here is the DIFFERENCE
 *
 *     D = sum over the contacts read this pass of [ t(theta_test) - t(theta_cur) ],
 *
 * whose float error scales with the size of the proposal (1 % steps), not with the terms: L_test = L_cur(state before the
 * move; exact, the maintained sum) + (exact integer difference of the all-trans tiles' histogram sums) + D +- B, B ~ 10..50.
 * A step whose interval lies entirely below T ln u is rejected without the exact pass; anything else (accepted, undecided,
 * bound void) runs the exact pass as before -- an accepted step needs the exact sum for the promotion anyway.  The returned
 * 8-tuple is unchanged (CL:3038-3051 returns likelihood_t, which a rejected step leaves alone).
 *
 * The term (ig_term_hot, linear contigs, one-log domain), per contact with count ob, distance s, rank distance d:
 *     y = slope log2 s + la,   yy = in ? max(y, lv) : lv,   t = ob yy log10(2) - 2^yy - log10(ob!) + pzc[d]
 * and its difference between two parameter sets (c = current, t = test; the log-factorial cancels):
 *     dt = ob log10(2) dyy - 2^yy_c (2^dyy - 1) + (pzc_t[d] - pzc_c[d]),   dyy = yy_t - yy_c.
 * Three classes of contacts.  The model's slope is negative (else the pass is void), so y is a decreasing function of s and
 * the classes are intervals of s, told apart by two thresholds computed once per step in double (rounded to the safe side):
 *     A  in range under both sets and y_c >= max(lv_c, lv_t) + DY (s < thrA):  neither clamps, dyy = dslope L + dla  (small, accurate)
 *     B  out of range under both, or y_c <= min(lv_c, lv_t) - DY (s >= thrB):   both at their level: dyy = lv_t - lv_c and
 *        2^yy_t - 2^yy_c = v_inter_t - v_inter_c are constants of the step (any size: a d_max proposal moves the level by tens of percent)
 *     C  anything else (a thin shell around the clamp levels / between the two d_max): both terms evaluated in full;
 *        full-size error, accounted term by term.
 * DY = |dslope| LB + |dla| bounds |dy| over |log2 s| <= LB = 24 (checked at the end: a larger |L| voids the pass).
 * 2^x - 1 for |x| <= 1/4 is a degree-6 polynomial in x ln 2 (truncation 5e-9 relative).
 * A proposal that leaves slope and amplitude alone (the d_max and trans-level proposals, CL:2996-3017: half of the steps) has
 * dyy = 0 EXACTLY in class A: its term there is the table difference dpzc[d] -- no logarithm, no exponential (template ZDY).
 *
 * Error budget, u = 2^-24, v_log_f32 / v_exp_f32 within K = 2 units of 2u (|result| + 1) resp. 2u result (measured over
 * their whole domain: 0.98 / 0.71, tests/test_hip_screen.py fails above 2):
 *     e(L) <= 4u (|L| + 1);  e(y_c) <= 5.25 u (|y_c| + Cy),  Cy = |la| + |slope| + |lv|;
 *     e(dyy): A  u [ |dslope| (5 |L| + 4) + |dla| + |dy| ],  B  2 u |dlv|            =: u G
 *     ob log10(2) dyy:  u log10(2) ob (G + 3 DX),  DX = max(DY, |dlv|)                 (rounding of ob log10(2), fma, pair add)
 *     2^yy_c (2^dyy - 1) =: dex:  |dex| (4u + ln2 5.25 u (|yy| + Cy))  [v_exp_f32 and the error of its argument]
 *                                 + |dex| 9 u [polynomial, products]  +  2^yy_c 0.84 u G  [sensitivity to e(dyy)]
 *     final adds (DIFF_BATCH terms are added in float before they join the double sum), table conversion:
 *                                 u ((2 + DIFF_BATCH) (log10(2) ob DX + |dex|) + (3 + DIFF_BATCH) |dpzc|)
 *     the contract itself: two quantisations (2^-32 each) and its double roundings (4e-15 of the magnitudes)
 * accumulated per workgroup from sum(ob), sum|dex|, sum|dex||yy|, sum 2^yy_c, max|L|, max|yy| and, for class C, a direct
 * per-term bound.  IG_NUIS_SCREEN_VERIFY=1 runs the exact pass on every step as well and checks |screened - exact| <= B on
 * the host (tests/test_hip_nuis_screen.py). */
#pragma once

#ifndef DIFF_THREADS
#define DIFF_THREADS 512 /* two workgroups per CU (their LDS), four waves per SIMD: room for DIFF_BATCH x 2 contacts in flight per thread */
#endif
#ifndef DIFF_BATCH
#define DIFF_BATCH 8
#endif
#define DIFF_X0 0.25f
#define DIFF_LB 24.0f
/* (DIFF_FIX = 2^20: ig_common.cuh -- sums and bounds are published as integers: deterministic totals) */

struct alignas(16) DiffConst {
    float dpzc[LDS_PZ + 2]; /* (pzc_t - pzc_c)[d], the last entries: the trans level's */
    float slope_c, la_c, lv_c, dslope, dla, dlv;
    float slope_t, la_t, lv_t, dmax_c, dmax_t;
    float dmin, dmax2;    /* min / max of the two d_max */
    float thr_a, thr_b;   /* class A: 0 < s < thr_a (cis); class B: s >= thr_b, or not cis, or s == 0 */
    float s17;            /* s >= s17  <=>  y <= 17 (class A terms of the ZDY kernel: the contract's clamp is out of reach) */
    int zdy;              /* dslope == 0 and dla == 0 exactly */
    float cy;             /* max over the two sets of |la| + |slope| + |lv| (rounded up) */
    float a_dslope, a_dla, a_dlv, dy_max; /* magnitudes (rounded up); DY */
    float dex_b;          /* v_inter_t - v_inter_c: class B's 2^yy_t - 2^yy_c */
    float pzc_abs_max;    /* largest |pzc| of either set */
    unsigned cut;         /* rank distances from here on are not in the tables' LDS copies (a longer table: the pass is void if one occurs) */
    int ok;               /* both sets in the one-log domain, DY and |dlv| within DIFF_X0 */
    int ok0;              /* ... what the histogram tier needs of that: the one-log domain, negative slopes, tables */
};

/* built by the blocks of k_nuis_prepare that build the test set's tables: thread i of that range */
__device__ __forceinline__ void build_diff_const(int i, const Glob* g, const ig_params pt, float mean_kb, float pzv_t, int pz_n_t,
                                                 const ScoreConst* __restrict__ sc0, int pz_n_c, DiffConst* out, const ScreenConst* __restrict__ scr0)
{
    const ig_params pc = g->par[0];
    if (i < LDS_PZ + 2) {
        const double t = (double)(i < min(pz_n_t, LDS_PZ) ? pzv_t : pt.v_inter) * IG_LOG_E_F;
        out->dpzc[i] = (float)(t - sc0->tab.pzc[i]);
    }
    if (i == 0) {
        const ig_hot hc = ig_hot_make(pc, ig_tab()), ht = ig_hot_make(pt, ig_tab());
        const double dslope = ht.slope - hc.slope, dla = ht.log2_amp - hc.log2_amp, dlv = ht.log2_v_inter - hc.log2_v_inter;
        out->slope_c = (float)hc.slope;
        out->la_c = (float)hc.log2_amp;
        out->lv_c = (float)hc.log2_v_inter;
        out->slope_t = (float)ht.slope;
        out->la_t = (float)ht.log2_amp;
        out->lv_t = (float)ht.log2_v_inter;
        out->dslope = (float)dslope;
        out->dla = (float)dla;
        out->dlv = (float)dlv;
        out->dmax_c = hc.d_max;
        out->dmax_t = ht.d_max;
        out->dmin = fminf(hc.d_max, ht.d_max);
        out->dmax2 = fmaxf(hc.d_max, ht.d_max);
        const double up = 1.0 + 0x1p-20;
        const double cyc = __builtin_fabs(hc.log2_amp) + __builtin_fabs(hc.slope) + __builtin_fabs(hc.log2_v_inter);
        const double cyt = __builtin_fabs(ht.log2_amp) + __builtin_fabs(ht.slope) + __builtin_fabs(ht.log2_v_inter);
        const double cy = __builtin_fmax(cyc, cyt) * up + 1e-6;
        out->cy = (float)(cy * up);
        const double a_ds = __builtin_fabs(dslope) * up, a_dl = __builtin_fabs(dla) * up, a_dv = __builtin_fabs(dlv) * up;
        const double dy_max = (a_ds * (double)DIFF_LB + a_dl) * up;
        out->a_dslope = (float)(a_ds * up);
        out->a_dla = (float)(a_dl * up);
        out->a_dlv = (float)(a_dv * up);
        out->dy_max = (float)(dy_max * up);
        out->dex_b = (float)((double)pt.v_inter - (double)pc.v_inter);
        /* y_c(s) = slope_c log2 s + la_c decreases with s (slope_c < 0, see ok): y_c >= HI <=> s <= 2^((HI - la_c) / slope_c).  HI / LO
         * leave room for the test set's y = y_c + dy (|dy| <= DY) and for the last bits of the contract's own evaluation */
        const double hi = __builtin_fmax(hc.log2_v_inter, ht.log2_v_inter) + dy_max + 1e-6 * (1.0 + cy);
        const double lo = __builtin_fmin(hc.log2_v_inter, ht.log2_v_inter) - dy_max - 1e-6 * (1.0 + cy);
        auto s_of = [&](double level) { /* the s at which y_c reaches `level` */
            const double e = (level - hc.log2_amp) / (hc.slope < 0.0 ? hc.slope : -1.0);
            return ig_exp2(__builtin_fmin(__builtin_fmax(e, -140.0), 120.0), ig_tab());
        };
        const float s_a = (float)(s_of(hi) * (1.0 - 1e-6)), s_b = (float)(s_of(lo) * (1.0 + 1e-6));
        out->thr_a = fminf(fminf(hc.d_max, ht.d_max), s_a);
        out->thr_b = fminf(fmaxf(hc.d_max, ht.d_max), __uint_as_float(__float_as_uint(s_b) + 1u)); /* s > s_b: the next float up */
        out->s17 = (float)(s_of(17.0) * (1.0 + 1e-6));
        out->zdy = (dslope == 0.0 && dla == 0.0) ? 1 : 0;
        out->cut = (pz_n_c > LDS_PZ || pz_n_t > LDS_PZ) ? (unsigned)LDS_PZ : 0xffffffffu;
        const bool ok0 = hc.fast && ht.fast && pc.slope < 0.0f && pt.slope < 0.0f && cy < 200.0 && pz_n_c > 0 && pz_n_t > 0 && scr0->pzc_max < 1e5f;
        out->ok0 = ok0 ? 1 : 0;
        out->ok = (ok0 && dy_max <= (double)DIFF_X0 && a_dv <= 64.0) ? 1 : 0;
    }
    if (i == 1) { /* the largest |pzc| of either set (only a 4e-15 slack on the contract's own double roundings hangs on it):
                   * the model's from its screening constants, the test set's from its largest entry -- P_z decreases with the
                   * rank distance for slope < 0 (other slopes: the pass is void, see ok above), entry 0 is the trans level */
        const float p1 = (mean_kb < pt.d_max) ? ig_rippe(mean_kb, pt, ig_tab()) : pt.v_inter;
        const double mx = __builtin_fmax((double)scr0->pzc_max, (double)fmaxf(p1, pt.v_inter) * IG_LOG_E_F);
        out->pzc_abs_max = (float)(mx * 1.001 + 1e-30);
    }
}

struct DiffLds {
    float dpzc[LDS_PZ + 2];
    uint2 rrec[FULL_TB], crec[FULL_TB];
    int rctg[FULL_TB], cctg[FULL_TB];
    double red_s[DIFF_THREADS / 64];
    float red_f[6][DIFF_THREADS / 64];
    unsigned red_bad[DIFF_THREADS / 64];
};

/* out8: [0],[1] exact integer limbs (hist sums under the test set minus under the current set), [2] s_fix, [3] b_fix,
 * [4] void flags, [5] contacts read */
/* The pass is a stream of 8-byte contacts (160 MB at the headline shape) against two staged blocks; a thread keeps DIFF_BATCH
 * contacts being summed and DIFF_BATCH more on their way, a workgroup takes an even share of the pass as a few contiguous runs
 * (TileRuns, ig_kernels_score.cuh).  ZDY: the proposal leaves slope and amplitude alone (DiffConst.zdy). */
template <bool ZDY>
__global__ void __launch_bounds__(DIFF_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) /* two workgroups per CU: at most 128 VGPRs */
    k_full_diff_tiled(const TileWork* __restrict__ work, const uint2* __restrict__ tc, const int4* __restrict__ rec, const DiffConst* __restrict__ dc,
                      int M, long long* out8, int n_static, TileDyn* dyn, const int* __restrict__ dyn_list, NuisHost* hn, int hn_seq,
                      const long long* __restrict__ partial_t, const long long* __restrict__ partial_c, int n_partial,
                      const long long* __restrict__ zero_sums, long long* trace)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    DiffLds& L = *(DiffLds*)lds_raw;
    /* trace (ig_debug_diff_trace): per workgroup {start, end (100 MHz clock), XCC_ID << 32 | HW_ID, runs << 32 | contacts, ticks
     * between a run's start and its blocks being staged, ticks in the contact loops} */
    long long t_start = 0, t_stage = 0, t_loop = 0, t_mark = 0;
    int n_items = 0;
    if (trace && threadIdx.x == 0) t_start = (long long)wall_clock64();
    const int dyn_count = dyn->count; /* (requested with the constants below: one round trip less in front of the first contacts) */
    __shared__ float s_pmx[DIFF_THREADS / 64];
    {
        float mx = 0.0f; /* the largest |dpzc| (the bound's table-conversion term) while the table goes to LDS */
        for (int i = threadIdx.x; i < LDS_PZ + 2; i += blockDim.x) {
            const float v = dc->dpzc[i];
            L.dpzc[i] = v;
            mx = fmaxf(mx, fabsf(v));
        }
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
        if ((threadIdx.x & 63) == 0) s_pmx[threadIdx.x >> 6] = mx;
    }
    const float slope_c = dc->slope_c, la_c = dc->la_c, lv_c = dc->lv_c, dslope = dc->dslope, dla = dc->dla, dlv = dc->dlv;
    const float thr_a = dc->thr_a, thr_b = dc->thr_b, dex_b = dc->dex_b;
    /* (class C's constants as well: its branch is taken by many waves -- one lane between the thresholds is enough -- and must not
     * start with a round trip to memory) */
    const float slope_t = dc->slope_t, la_t = dc->la_t, lv_t = dc->lv_t, dmax_c = dc->dmax_c, dmax_t = dc->dmax_t, cy_f = dc->cy;
    const unsigned cut = dc->cut;
    const int ok = dc->ok && (dc->zdy != 0) == ZDY;
    const float c10 = (float)IG_LOG2_10_INV, ln2 = 0.69314718f;
    double acc = 0.0;
    /* sum(ob) where a term depends on it; sum |dex|; sum |dex| |yy|; sum 2^yy_c (ZDY: contacts in class B); max |L| (ZDY: min s in
     * class A, negated); max |yy|; class C's direct bound; sum(ob) over everything */
    float s_ob = 0.0f, s_dex = 0.0f, s_dexy = 0.0f, s_ex = 0.0f, l_max = ZDY ? -3.0e38f : 0.0f, y_max = 0.0f, b_c = 0.0f, s_oball = 0.0f;
    unsigned bad = ok ? 0u : 1u;
    long long n_read = 0;
    const int nth = blockDim.x;
    __syncthreads();
    TileRuns runs(work, dyn_list, n_static, ok ? dyn_count : -n_static); /* this workgroup's share of the pass; nothing when the pass is void */
    TileRun wk;
    int st_bi = -1, st_bj = -1;
    while (runs.next(wk)) {
        const bool diag = wk.bi == wk.bj;
        const uint2* src = tc + wk.off;
        const int n = wk.n;
        n_read += (threadIdx.x == 0) ? n : 0;
        n_items++;
        if (trace && threadIdx.x == 0) t_mark = (long long)wall_clock64();
        uint2 nx[DIFF_BATCH];
#pragma unroll
        for (int q = 0; q < DIFF_BATCH; q++) nx[q] = src[min((int)threadIdx.x + q * nth, n - 1)];
        if (st_bi != wk.bi || st_bj != wk.bj) {
            __syncthreads(); /* everybody is through with the blocks of the run before */
#if !(defined(DIFF_ABLATE) && (DIFF_ABLATE & 4)) /* tuning builds: no staging */
            for (int i = threadIdx.x; i < FULL_TB; i += nth) {
                const int gi = wk.bi * FULL_TB + i, gj = wk.bj * FULL_TB + i;
                const int4 a = gi < M ? rec[gi] : make_int4(0, 0, -1, 0);
                L.rrec[i] = make_uint2((unsigned)a.x, (unsigned)a.w | (__int_as_float(a.y) != 0.0f ? 0x80000000u : 0u));
                L.rctg[i] = a.z;
                if (!diag) {
                    const int4 b = gj < M ? rec[gj] : make_int4(0, 0, -2, 0);
                    L.crec[i] = make_uint2((unsigned)b.x, (unsigned)b.w | (__int_as_float(b.y) != 0.0f ? 0x80000000u : 0u));
                    L.cctg[i] = b.z;
                }
            }
#endif
            __syncthreads();
            st_bi = wk.bi;
            st_bj = wk.bj;
        }
        if (trace && threadIdx.x == 0) {
            const long long now = (long long)wall_clock64();
            t_stage += now - t_mark;
            t_mark = now;
        }
        const uint2* cre = diag ? L.rrec : L.crec;
        const int* cct = diag ? L.rctg : L.cctg;
        for (int e0 = threadIdx.x; e0 < n; e0 += DIFF_BATCH * nth) {
            uint2 vv[DIFF_BATCH];
#pragma unroll
            for (int q = 0; q < DIFF_BATCH; q++) vv[q] = nx[q];
#pragma unroll
            for (int q = 0; q < DIFF_BATCH; q++) nx[q] = src[min(e0 + (DIFF_BATCH + q) * nth, n - 1)];
            float dt[DIFF_BATCH];
#pragma unroll
            for (int q = 0; q < DIFF_BATCH; q++) {
                const uint2 v = vv[q];
                const bool live = e0 + q * nth < n;
                const unsigned li = v.x & (FULL_TB - 1), lj = (v.x >> 11) & (FULL_TB - 1);
#if defined(DIFF_ABLATE) && (DIFF_ABLATE & 2) /* tuning builds: no LDS gathers */
                const uint2 ri = make_uint2(__float_as_uint((float)li), li), rj = make_uint2(__float_as_uint((float)lj * 1.5f), lj);
                const bool cis = (li ^ lj) & 1;
#else
                const uint2 ri = L.rrec[li], rj = cre[lj];
                const bool cis = L.rctg[li] == cct[lj];
#endif
                const unsigned d = abs_diff_u32(ri.y & 0x7fffffffu, rj.y & 0x7fffffffu);
                const float sv = fabsf(__uint_as_float(ri.x) - __uint_as_float(rj.x));
                const bool pos = cis && (sv > 0.0f);
                const bool A = pos && (sv < thr_a);
                const bool B = !pos || (sv >= thr_b);
                const float obf = (float)v.y;
                const float m = obf * c10;
                const float pz = L.dpzc[cis ? min(d, (unsigned)LDS_PZ) : (unsigned)LDS_PZ];
                bad |= (live && ((cis && (((ri.y | rj.y) >> 31) || d >= cut)) || (v.y - 1u >= 16383u))) ? 2u : 0u;
                float t, obacc, dexa, exa, yya, lacc;
                if (ZDY) { /* class A: the table difference alone; class B: constants of the step */
                    t = A ? pz : __builtin_fmaf(m, dlv, -dex_b) + pz;
                    obacc = A ? 0.0f : obf;
                    dexa = 0.0f;
                    exa = A ? 0.0f : 1.0f; /* counts class B */
                    yya = 0.0f;
                    lacc = A ? -sv : -3.0e38f; /* max of -s = -(min s) over class A: y <= 17 there iff min s >= s17 */
                } else {
#if defined(DIFF_ABLATE) && (DIFF_ABLATE & 1) /* tuning builds: no transcendental functions */
                    const float lg = sv * 0.001f;
#else
                    const float lg = __builtin_amdgcn_logf(sv);
#endif
                    const float y = __builtin_fmaf(slope_c, lg, la_c);
                    const float dy = __builtin_fmaf(dslope, lg, dla);
                    const float yy = A ? y : lv_c;
                    const float dyy = A ? dy : dlv;
                    const float ex = __builtin_amdgcn_exp2f(yy);
                    const float z = dyy * ln2;
                    float e = __builtin_fmaf(z, 1.0f / 720.0f, 1.0f / 120.0f);
                    e = __builtin_fmaf(z, e, 1.0f / 24.0f);
                    e = __builtin_fmaf(z, e, 1.0f / 6.0f);
                    e = __builtin_fmaf(z, e, 0.5f);
                    e = __builtin_fmaf(z, e, 1.0f);
                    const float dex = A ? ex * (z * e) : dex_b;
                    t = __builtin_fmaf(m, dyy, -dex) + pz;
                    obacc = obf;
                    dexa = fabsf(dex);
                    exa = ex;
                    yya = fabsf(yy);
                    lacc = A ? fabsf(lg) : 0.0f;
                }
                if (__any(live && !(A || B))) {
                    if (!(A || B)) { /* class C: both terms in full */
                        const float lg = __builtin_amdgcn_logf(sv);
                        const float y = __builtin_fmaf(slope_c, lg, la_c);
                        const bool in_c = pos && (sv < dmax_c), in_t = pos && (sv < dmax_t);
                        const float yt0 = __builtin_fmaf(slope_t, lg, la_t);
                        const float yc = in_c ? fmaxf(y, lv_c) : lv_c, yt = in_t ? fmaxf(yt0, lv_t) : lv_t;
                        const float exc = __builtin_amdgcn_exp2f(yc), ext = __builtin_amdgcn_exp2f(yt);
                        t = __builtin_fmaf(m, yt - yc, -(ext - exc)) + pz;
                        const float ya = fabsf(yc) + fabsf(yt), ym = fmaxf(fabsf(yc), fabsf(yt));
                        const float bc = m * (11.0f * ya + 11.0f * cy_f) + (exc + ext) * (8.0f + 3.7f * (ym + cy_f)) + 4.0f * fabsf(pz);
                        b_c += live ? bc : 0.0f;
                        y_max = fmaxf(y_max, live ? ym : 0.0f);
                        obacc = 0.0f;
                        dexa = 0.0f;
                        exa = 0.0f;
                        yya = 0.0f;
                        lacc = ZDY ? -3.0e38f : 0.0f;
                    }
                }
                if (!live) {
                    t = 0.0f;
                    obacc = 0.0f;
                    dexa = 0.0f;
                    exa = 0.0f;
                    yya = 0.0f;
                    lacc = ZDY ? -3.0e38f : 0.0f;
                }
                dt[q] = t;
                s_ob += obacc;
                s_oball += live ? obf : 0.0f;
                s_ex += exa;
                l_max = fmaxf(l_max, lacc);
                if (!ZDY) {
                    s_dex += dexa;
                    s_dexy = __builtin_fmaf(dexa, yya, s_dexy);
                    y_max = fmaxf(y_max, yya);
                }
            }
            float ts = 0.0f; /* a few float additions before the double one: u sum|dt| of them, in the bound's "final adds" */
#pragma unroll
            for (int q = 0; q < DIFF_BATCH; q++) ts += dt[q];
            acc += (double)ts;
        }
        if (trace && threadIdx.x == 0) t_loop += (long long)wall_clock64() - t_mark;
    }
    /* ---- the workgroup's sum and bound */
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        s_ob += __shfl_down(s_ob, o, 64);
        s_oball += __shfl_down(s_oball, o, 64);
        s_dex += __shfl_down(s_dex, o, 64);
        s_dexy += __shfl_down(s_dexy, o, 64);
        s_ex += __shfl_down(s_ex, o, 64);
        b_c += __shfl_down(b_c, o, 64);
        l_max = fmaxf(l_max, __shfl_down(l_max, o, 64));
        y_max = fmaxf(y_max, __shfl_down(y_max, o, 64));
        bad |= __shfl_down(bad, o, 64);
        n_read += __shfl_down(n_read, o, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ float s_ymax[DIFF_THREADS / 64], s_oba[DIFF_THREADS / 64];
    __shared__ long long s_nread[DIFF_THREADS / 64];
    __syncthreads(); /* (the staged blocks' last readers are through: red_* do not overlap them, but keep the phases apart) */
    if (lane == 0) {
        L.red_s[wv] = acc;
        L.red_f[0][wv] = s_ob;
        L.red_f[1][wv] = s_dex;
        L.red_f[2][wv] = s_dexy;
        L.red_f[3][wv] = s_ex;
        L.red_f[4][wv] = b_c;
        L.red_f[5][wv] = l_max;
        L.red_bad[wv] = bad;
        s_ymax[wv] = y_max;
        s_oba[wv] = s_oball;
        s_nread[wv] = n_read;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, Sob = 0.0, Sdex = 0.0, Sdexy = 0.0, Sex = 0.0, Bc = 0.0, Lm = -3.0e38, Ym = 0.0, Soa = 0.0;
        unsigned Bd = 0;
        long long n = 0;
        for (int v = 0; v < (int)(blockDim.x >> 6); v++) {
            S += L.red_s[v];
            Sob += (double)L.red_f[0][v];
            Sdex += (double)L.red_f[1][v];
            Sdexy += (double)L.red_f[2][v];
            Sex += (double)L.red_f[3][v];
            Bc += (double)L.red_f[4][v];
            Lm = __builtin_fmax(Lm, (double)L.red_f[5][v]);
            Ym = __builtin_fmax(Ym, (double)s_ymax[v]);
            Soa += (double)s_oba[v];
            Bd |= L.red_bad[v];
            n += s_nread[v];
        }
        if (ok && n > 0) {
            const double u = 0x1p-24, c10d = IG_LOG2_10_INV;
            const double cy = dc->cy, a_ds = dc->a_dslope, a_dl = dc->a_dla, a_dv = dc->a_dlv, DY = dc->dy_max;
            const double nn = (double)n;
            const double fa = 1.0 + 2.0 * u * (nn / (double)blockDim.x + 32.0) * 2.0; /* the float accumulators' own roundings */
            double pmx = 0.0; /* largest |dpzc| */
            for (int q = 0; q < (int)(blockDim.x >> 6); q++) pmx = __builtin_fmax(pmx, (double)s_pmx[q]);
            if (ZDY) { /* class A: dt = dpzc[d] (a table conversion); class B: constants; what was counted in s_ex is class B's size */
                const double dexb = __builtin_fabs((double)dc->dex_b);
                Sdex = Sex * dexb * (1.0 + 0x1p-20);
                Sdexy = Sdex * __builtin_fabs((double)dc->lv_c);
                /* the contract's own double roundings need sum 2^yy_c: below 2^17 per contact (checked: min s of class A against s17) */
                Sex = nn * 131072.0;
                if (!(-Lm >= (double)dc->s17)) Bd |= 4u; /* (no class A contact: Lm = -3e38, fine) */
                Ym = __builtin_fmax(Ym, 17.0); /* (class C's own maximum stands) */
                Lm = 0.0;
            }
            const double DX = __builtin_fmax(DY, a_dv);
            const double G = 1.05 * __builtin_fmax(a_ds * (5.0 * Lm + 4.0) + a_dl + DY, 2.0 * a_dv);
            const double T1 = u * c10d * Sob * (G + 3.0 * DX);
            const double T2 = ZDY ? u * 13.0 * Sdex : u * (13.0 * Sdex + 3.68 * (Sdexy + cy * Sdex) + 0.84 * G * Sex);
            const double T3 = u * ((2.0 + DIFF_BATCH) * c10d * Sob * DX + (2.0 + DIFF_BATCH) * Sdex + (3.0 + DIFF_BATCH) * nn * pmx);
            const double T4 = nn * 0x1p-31 + 4e-15 * (Soa * (Ym + 5.0) + 2.6 * Sex + nn * (double)dc->pzc_abs_max);
            const double bound = 1.01 * fa * (T1 + T2 + T3 + u * Bc) + T4 + 2.0 / DIFF_FIX;
            /* outside the screening term's domain: |L| beyond the bound DY was derived for, |yy| beyond the contract's clamp
             * (|t| < 2^20 needs P < 2^18 and counts < 2^14), anything not a number */
            if (!(Lm <= (double)DIFF_LB) || !(Ym <= 17.5) || !(__builtin_fabs(S) < 1e15) || !(bound < 1e12)) Bd |= 4u;
            atomic_add_ll(&out8[2], (long long)__builtin_rint(S * DIFF_FIX));
            atomic_add_ll(&out8[3], (long long)__builtin_ceil(bound * DIFF_FIX));
            atomic_add_ll(&out8[5], n);
        }
        if (Bd) atomicOr((unsigned long long*)&out8[4], (unsigned long long)Bd);
    }
    /* the all-trans tiles: exact integer histogram sums under the test set minus under the current set (k_tile_trans) */
    if (blockIdx.x == 0 && ok) {
        long long hi = 0, lo = 0;
        for (int i = threadIdx.x; i < n_partial; i += blockDim.x) {
            hi += partial_t[2 * (size_t)i] - partial_c[2 * (size_t)i];
            lo += partial_t[2 * (size_t)i + 1] - partial_c[2 * (size_t)i + 1];
        }
        hi = wave_sum_ll(hi);
        lo = wave_sum_ll(lo);
        if (lane == 0 && (hi | lo)) {
            atomic_add_ll(&out8[0], hi);
            atomic_add_ll(&out8[1], lo);
        }
    }
    /* the last workgroup through: the counter back for the exact pass that may follow over the same list, the sums to the
     * (mapped) host memory, then the flag */
    __syncthreads(); /* (block 0: every wave's share of the histogram sums is out) */
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&dyn->done, 1) == (int)gridDim.x - 1) {
            __threadfence();
            dyn->done = 0;
            if (hn) {
                for (int q = 0; q < 8; q++) hn->diff[q] = __hip_atomic_load(&out8[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int q = 0; q < 8; q++) hn->sums[q] = __hip_atomic_load(&zero_sums[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                hn->diff_seq = hn_seq;
            }
        }
        if (trace) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            long long* tr = trace + 8 * (size_t)blockIdx.x;
            tr[0] = t_start;
            tr[1] = (long long)wall_clock64();
            tr[2] = (long long)(((unsigned long long)xcc << 32) | hw);
            tr[3] = ((long long)n_items << 32) | (n_read & 0xffffffffLL);
            tr[4] = t_stage;
            tr[5] = t_loop;
        }
    }
}

/* ================================================================================================================================
 * Tier 0: the same difference from a HISTOGRAM of the cis contacts' distances, without reading a contact.
 *
 * In x = log2 s the exponent of the term is piecewise LINEAR: yy(x) = x < log2 d_max ? max(slope x + la, lv) : lv, with at most
 * two kinks per parameter set (where the power law meets the trans level; d_max).  Between the kinks of both sets
 *     dt = ob log10(2) (dA x + dB) - [2^(A_t x + B_t) - 2^(A_c x + B_c)] + dpzc[d]
 * with (A, B) = (slope, la) or (0, lv) per set: the first part needs sum(ob) and sum(ob x) of the contacts of a bin EXACTLY (it
 * is linear), the second a Taylor expansion around the bin's centre with sum(1), sum(x) and a remainder of second order in the bin
 * width (2^-NH_OCT_BITS octaves: ~1e-6 of the bin's sum of P), the third a histogram over the rank distance.  The histogram
 * (NuisHist: integers, maintained with atomics) is built once from the tables (k_hist_build), follows every move that changes
 * the genome (k_hist_walk: the contacts inside the move's contigs whose bin changes, before and after from the winner's columns)
 * and is evaluated per step by k_hist_eval: ~50 k bins instead of the 160 MB of contacts of the pass above.  A bin that holds a
 * kink of either set is bounded piece by piece (interval arithmetic on its sums).  Trans contacts are not in the histogram: their
 * number and the sum of their counts are the totals minus the cis ones, and their term depends on the count alone.
 * x is the hardware's log2 (v_log_f32) of the float distance, rounded to 2^-NH_FRAC_BITS: within e = 2^-22 (|x| + 2) + 2^-20 of
 * the true log2 s; every bound below is taken over the bin widened by e.
 * The interval it yields is used exactly like the pass's (nuis_end_impl); where it does not decide, the pass over the contacts
 * runs as before. */
struct NhKey {
    int kind; /* 0 trans (not in the histogram), 1 binned, 2 cis at distance 0, 3 cis on a ring, 4 cis outside the binned range */
    int bin, off, d;
};
__device__ __forceinline__ NhKey nh_key(bool cis, bool ring, float sv, unsigned d, int dh_n)
{
    NhKey k;
    k.kind = 0;
    k.bin = 0;
    k.off = 0;
    k.d = (int)min(d, (unsigned)dh_n);
    if (!cis) return k;
    if (ring) {
        k.kind = 3;
        return k;
    }
    if (!(sv > 0.0f)) {
        k.kind = 2;
        return k;
    }
    const float xf = __builtin_rintf(__builtin_amdgcn_logf(sv) * (float)(1 << NH_FRAC_BITS)); /* (a power of two: the product is exact) */
    const float lim = (float)(NH_LMAX << NH_FRAC_BITS);
    if (!(xf >= -lim) || !(xf < lim)) {
        k.kind = 4;
        return k;
    }
    const int q = (int)xf + (NH_LMAX << NH_FRAC_BITS);
    k.kind = 1;
    k.bin = q >> (NH_FRAC_BITS - NH_OCT_BITS);
    k.off = q & (NH_SUB - 1);
    return k;
}
__device__ __forceinline__ bool nh_same(const NhKey& a, const NhKey& b)
{
    return a.kind == b.kind && (a.kind == 0 || (a.bin == b.bin && a.off == b.off && a.d == b.d));
}
__device__ __forceinline__ void nh_apply(const NuisHist& h, const NhKey& k, int ob, long long sign)
{
    if (k.kind == 0) return;
    if (k.kind == 1) {
        long long* b = h.bins + 4 * (size_t)k.bin;
        atomic_add_ll(&b[0], sign);
        atomic_add_ll(&b[1], sign * k.off);
        atomic_add_ll(&b[2], sign * ob);
        atomic_add_ll(&b[3], sign * (long long)ob * k.off);
        atomic_add_ll(&h.dh[k.d], sign);
        if (sign > 0 && k.d > LDS_PZ && (long long)k.d > __hip_atomic_load(&h.misc[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax((unsigned long long*)&h.misc[7], (unsigned long long)k.d);
    } else if (k.kind == 2) {
        atomic_add_ll(&h.misc[0], sign);
        atomic_add_ll(&h.misc[1], sign * ob);
        atomic_add_ll(&h.dh[k.d], sign);
        if (sign > 0 && k.d > LDS_PZ && (long long)k.d > __hip_atomic_load(&h.misc[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax((unsigned long long*)&h.misc[7], (unsigned long long)k.d);
    } else {
        atomic_add_ll(&h.misc[k.kind == 3 ? 2 : 3], sign);
    }
}

/* the histogram of the state in `t`, from scratch (h zeroed by the caller): a wave per row of the contact matrix */
__global__ void __launch_bounds__(256) k_hist_build(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables t, int M, NuisHist h)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long n_all = 0, ob_all = 0, n_big = 0;
    for (int i = blockIdx.x * (blockDim.x >> 6) + wv; i < M; i += gridDim.x * (blockDim.x >> 6)) {
        const long long b = rowptr[i], e = rowptr[i + 1];
        if (b == e) continue;
        const int2 cp1 = t.cp[i];
        const float d1 = t.dist[i];
        const bool ring1 = t.stot[i] != 0.0f;
        for (long long q = b + lane; q < e; q += 64) {
            const int2 v = cc[q];
            const int2 cp2 = t.cp[v.x];
            n_all++;
            ob_all += v.y;
            if ((unsigned)v.y - 1u >= 16383u) n_big++; /* (the pass is void for good: such a contact is in no bin) */
            if (cp2.x != cp1.x || v.y <= 0) continue;
            const bool ring = ring1 || t.stot[v.x] != 0.0f;
            const NhKey k = nh_key(true, ring, fabsf(d1 - t.dist[v.x]), abs_diff_u32((unsigned)cp1.y, (unsigned)cp2.y), h.dh_n);
            nh_apply(h, k, v.y, 1);
        }
    }
    n_all = wave_sum_ll(n_all);
    ob_all = wave_sum_ll(ob_all);
    n_big = wave_sum_ll(n_big);
    if (lane == 0) {
        if (n_all) atomic_add_ll(&h.misc[4], n_all);
        if (ob_all) atomic_add_ll(&h.misc[5], ob_all);
        if (n_big) atomic_add_ll(&h.misc[6], n_big);
    }
}

/* the histogram follows the move of slot w (already applied; `tab`: the tables of the state BEFORE it, as k_delta's third mode):
 * every contact inside the move's contigs whose key differs between the current genome's column and the winner's */
__global__ void __launch_bounds__(256) k_hist_walk(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, const Glob* g,
                                                   MoveBuf mb, int w, NuisHist h)
{
    const MoveCtl& mc = mb.ctl[PS(w)];
    if (g->error || !mc.n_dirty) return;
    const int cw = CW(w, mc.ch_c), k = mc.ch_k;
    const CandMeta& m = mb.meta[cw];
    const int M = mb.sM, m_loc = m.m_loc;
    const uint2* col0 = mb.coords + (size_t)(cw * NSLOT) * M;
    const uint2* colk = mb.coords + (size_t)(cw * NSLOT + k) * M;
    const ColMeta* cm0 = mb.cmeta + (size_t)(cw * NSLOT) * NCODE;
    const ColMeta* cmk = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    const int* subs = mb.subs + (size_t)cw * M;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dh_n = h.dh_n;
    auto key_of = [dh_n](uint2 a, uint2 b, const ColMeta* cm) {
        const int ci = (int)(a.y >> 28), cj = (int)(b.y >> 28);
        const bool cis = ci == cj;
        return nh_key(cis, cis && cm[ci].stot != 0.0f, fabsf(__uint_as_float(a.x) - __uint_as_float(b.x)),
                      abs_diff_u32(a.y & 0x0fffffffu, b.y & 0x0fffffffu), dh_n);
    };
    for (int r = blockIdx.x * (blockDim.x >> 6) + wv; r < m_loc; r += gridDim.x * (blockDim.x >> 6)) {
        const int i = subs[r];
        const long long b = rowptr[i], e = rowptr[i + 1];
        if (b == e) continue;
        const int2 cp1 = tab.cp[i];
        const uint2 a0 = col0[r], a1 = colk[r];
        for (long long q = b + lane; q < e; q += 64) {
            const int2 v = cc[q];
            const int2 cp2 = tab.cp[v.x];
            if (!slice_keep(m, cp1.x, cp2.x, cp1.y, cp2.y, v.y, true)) continue;
            const int lj = ((m.same || cp2.x == m.ctgA) ? 0 : m.SLA) + cp2.y;
            const NhKey k0 = key_of(a0, col0[lj], cm0), k1 = key_of(a1, colk[lj], cmk);
            if (nh_same(k0, k1)) continue;
            nh_apply(h, k0, v.y, -1);
            nh_apply(h, k1, v.y, 1);
        }
    }
}

/* out16: [2] sum, [3] bound (2^-20 units), [4] void flags (as the pass: 1 parameters, 2 contacts, 4 sums), [5] binned cis contacts,
 * [6] their counts, [7] an upper bound of their sum of P under both sets, [8] workgroups through; zero between two launches (the last
 * workgroup publishes and clears).  Grid: n_zero blocks of the zero-pixel sum under the test set (-> zero_out), NH_NB / 256 blocks
 * of bins, one block for the rank-distance histogram. */
/* CHAIN (k_chain_hist_eval): one of the CHAIN_SEG sets of a segment -- the test set's parameters come with its constants (never
 * Glob.par[1]), the interval goes to the decide wave (ChainTest) instead of the host */
template <bool CHAIN>
__device__ __forceinline__ void hist_eval_body(NuisHist h, const Glob* g, const ScoreConst* __restrict__ sc_t, const ScoreConst* __restrict__ sc_c,
                                               const DiffConst* __restrict__ dc, long long* out16, NuisHost* hn, int hn_seq, Tables zt, int M,
                                               long long* zero_out, int n_zero_blocks, const long long* __restrict__ zero_sums, PzTab pz_t,
                                               PzTab pz_c, const int block, const int n_blocks, ChainTest* ct)
{
    __shared__ double red[5][4];
    __shared__ unsigned red_bad[4];
    __shared__ long long red_n[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (block < n_zero_blocks) {
        if (CHAIN) full_zero_block_p(zt, sc_t->par, sc_t->mean_kb, g->n_tot_pxl, M, zero_out, block, n_zero_blocks, pz_t.v, pz_t.n);
        else full_zero_block(zt, g, 1, M, zero_out, block, n_zero_blocks, pz_t.v, pz_t.n);
    } else {
        const int bb = block - n_zero_blocks;
        const ig_hot hc = sc_c->hot, ht = sc_t->hot;
        const double c10 = IG_LOG2_10_INV, ln2 = 0.69314718055994530942;
        const double a_c = hc.slope, b_c = hc.log2_amp, lv_c = hc.log2_v_inter, a_t = ht.slope, b_t = ht.log2_amp, lv_t = ht.log2_v_inter;
        const bool ok = dc->ok0 != 0;
        double S = 0.0, E = 0.0, SP = 0.0, mag = 0.0, yhi = -1e300, ylo = 1e300; /* (yy decreases with x: its extremes sit at the bin's ends) */
        unsigned bad = ok ? 0u : 1u;
        long long n_cis = 0, ob_cis = 0;
        if (ok && bb < NH_NB / 256) {
            const int bin = bb * 256 + tid;
            const long long n = h.bins[4 * (size_t)bin], sd = h.bins[4 * (size_t)bin + 1], sob = h.bins[4 * (size_t)bin + 2],
                            sobd = h.bins[4 * (size_t)bin + 3];
            if (n < 0 || sd < 0 || sob < n || sobd < 0) bad |= 4u | 16u;
            if (n > 0 && !(bad & 4u)) {
                n_cis = n;
                ob_cis = sob;
                const double wbin = 1.0 / (double)(1 << NH_OCT_BITS), unit = 1.0 / (double)(1 << NH_FRAC_BITS);
                const double x0 = (double)bin * wbin - (double)NH_LMAX;
                const double e = 0x1p-22 * (__builtin_fmax(__builtin_fabs(x0), __builtin_fabs(x0 + wbin)) + 2.0) + 0x1p-20;
                const double lo = x0 - e, hi = x0 + wbin + e;
                const double nd = (double)n, sobf = (double)sob;
                /* the kinks of the two sets (slopes < 0, see ok0): where the power law meets the trans level, and d_max */
                const double xd_c = ig_log2_pos((double)hc.d_max, ig_tab()), xd_t = ig_log2_pos((double)ht.d_max, ig_tab());
                double kk[4] = {(lv_c - b_c) / a_c, (lv_t - b_t) / a_t, xd_c, xd_t};
                int nk = 0;
                double cut[6];
                cut[0] = lo;
                for (int q = 0; q < 4; q++)
                    if (kk[q] > lo - 1e-9 && kk[q] < hi + 1e-9) {
                        const double v = __builtin_fmin(__builtin_fmax(kk[q], lo), hi);
                        int at = ++nk;
                        while (at > 1 && cut[at - 1] > v) {
                            cut[at] = cut[at - 1];
                            at--;
                        }
                        cut[at] = v;
                    }
                cut[nk + 1] = hi;
                /* (A, B) of a set at a point that is not a kink */
                auto piece = [&](double x, double a, double b, double lv, double xd, double& A, double& B) {
                    const bool pw = (x < xd) && (a * x + b > lv);
                    A = pw ? a : 0.0;
                    B = pw ? b : lv;
                };
                if (nk == 0) { /* smooth bin: exact first part, Taylor for the second */
                    const double xm = x0 + 0.5 * wbin;
                    double Ac, Bc, At, Bt;
                    piece(xm, a_c, b_c, lv_c, xd_c, Ac, Bc);
                    piece(xm, a_t, b_t, lv_t, xd_t, At, Bt);
                    const double dA = At - Ac, dB = Bt - Bc;
                    const double sobx = x0 * sobf + (double)sobd * unit; /* sum ob x_i */
                    const double t1 = c10 * (dA * sobx + dB * sobf);
                    const double e1 = c10 * __builtin_fabs(dA) * e * sobf;
                    const double Pt = exp2(At * xm + Bt), Pc = exp2(Ac * xm + Bc);
                    const double dP = Pt - Pc, dP1 = ln2 * (At * Pt - Ac * Pc);
                    const double sx = (x0 - xm) * nd + (double)sd * unit; /* sum (x_i - xm) */
                    const double t2 = dP * nd + dP1 * sx;
                    const double Pt_lo = exp2(At * lo + Bt) * (1.0 + 1e-12), Pc_lo = exp2(Ac * lo + Bc) * (1.0 + 1e-12);
                    const double dyy = __builtin_fmax(__builtin_fabs(dA * lo + dB), __builtin_fabs(dA * hi + dB));
                    const double dPmax = Pc_lo * (exp2(dyy) - 1.0) * (1.0 + 1e-9) + 1e-300;
                    const double M1 = ln2 * (__builtin_fabs(dA) * Pt_lo + __builtin_fabs(Ac) * dPmax);
                    const double M2 = ln2 * ln2 * (__builtin_fabs(At * At - Ac * Ac) * Pt_lo + Ac * Ac * dPmax);
                    const double hw = 0.5 * wbin + e;
                    const double e2 = M1 * e * nd + 0.5 * M2 * nd * hw * hw;
                    S = t1 - t2;
                    E = e1 + e2;
                    SP = nd * (Pt_lo + Pc_lo);
                    mag = __builtin_fabs(t1) + __builtin_fabs(t2) + c10 * (__builtin_fabs(dA) * ((double)NH_LMAX + 1.0) + __builtin_fabs(dB)) * sobf + (Pt_lo + Pc_lo) * nd;
                    yhi = __builtin_fmax(Ac * lo + Bc, At * lo + Bt);
                    ylo = __builtin_fmin(Ac * hi + Bc, At * hi + Bt);
                } else { /* a kink inside: the hull over the pieces of g(x) = log10(2) dyy(x) and of dP(x) */
                    double g_lo = 1e300, g_hi = -1e300, p_lo = 1e300, p_hi = -1e300;
                    for (int q = 0; q <= nk; q++) {
                        const double u = cut[q], v = cut[q + 1];
                        if (!(v > u)) continue;
                        const double xm = 0.5 * (u + v);
                        double Ac, Bc, At, Bt;
                        piece(xm, a_c, b_c, lv_c, xd_c, Ac, Bc);
                        piece(xm, a_t, b_t, lv_t, xd_t, At, Bt);
                        const double dA = At - Ac, dB = Bt - Bc;
                        const double gu = c10 * (dA * u + dB), gv = c10 * (dA * v + dB);
                        g_lo = __builtin_fmin(g_lo, __builtin_fmin(gu, gv));
                        g_hi = __builtin_fmax(g_hi, __builtin_fmax(gu, gv));
                        const double Ptu = exp2(At * u + Bt), Pcu = exp2(Ac * u + Bc), Ptv = exp2(At * v + Bt), Pcv = exp2(Ac * v + Bc);
                        const double dyy = __builtin_fmax(__builtin_fabs(dA * u + dB), __builtin_fabs(dA * v + dB));
                        const double dPmax = Pcu * (exp2(dyy) - 1.0) * (1.0 + 1e-9);
                        const double M2 = ln2 * ln2 * (__builtin_fabs(At * At - Ac * Ac) * Ptu + Ac * Ac * dPmax) * (1.0 + 1e-9);
                        const double cv = M2 * (v - u) * (v - u) * 0.125 + 1e-12 * (Ptu + Pcu);
                        p_lo = __builtin_fmin(p_lo, __builtin_fmin(Ptu - Pcu, Ptv - Pcv) - cv);
                        p_hi = __builtin_fmax(p_hi, __builtin_fmax(Ptu - Pcu, Ptv - Pcv) + cv);
                        SP = __builtin_fmax(SP, nd * (Ptu + Pcu) * (1.0 + 1e-12));
                        yhi = __builtin_fmax(yhi, __builtin_fmax(Ac * u + Bc, At * u + Bt));
                        ylo = __builtin_fmin(ylo, __builtin_fmin(Ac * v + Bc, At * v + Bt));
                    }
                    const double d_lo = g_lo * sobf - p_hi * nd, d_hi = g_hi * sobf - p_lo * nd;
                    S = 0.5 * (d_lo + d_hi);
                    E = 0.5 * (d_hi - d_lo);
                    mag = __builtin_fabs(d_lo) + __builtin_fabs(d_hi) + __builtin_fabs(g_lo * sobf) + __builtin_fabs(p_hi * nd);
                    if (!(g_hi >= g_lo) || !(p_hi >= p_lo)) bad |= 4u | 128u;
                }
                /* the contract clamps a term to |t| < 2^20, the expansion does not: P < 2^17.5 and, with counts below 2^14,
                 * |ob log10 P| + log10(ob!) + P_z < 16383 (100 log10 2 + 3.8) + 1e5 < 2^20 - 2^17.5 */
                if (!(yhi <= 17.5) || !(ylo >= -100.0)) bad |= 4u | 32u;
                if (!(__builtin_fabs(S) < 1e15) || !(E < 1e12)) bad |= 4u | 64u; /* not a number */
                E += 1e-13 * mag;
            }
        } else if (ok) { /* the last blocks: P_z by rank distance -- the staged entries where the tables end inside the staged part (the
                          * trans level from there on), table / formula as the contract's own rare path reads them (pz_lookup) beyond */
            const int j = bb - NH_NB / 256, nj = n_blocks - n_zero_blocks - NH_NB / 256;
            const int d_end = min(h.dh_n, max(LDS_PZ, (int)h.misc[7]));
            const bool staged_all = dc->cut == 0xffffffffu;
            for (int d = j * 256 + tid; d <= d_end; d += nj * 256) {
                const long long n = h.dh[d];
                if (n < 0) bad |= 4u;
                if (n > 0) {
                    double pt, pc;
                    if (d < LDS_PZ || staged_all) {
                        pt = sc_t->tab.pzc[min(d, LDS_PZ)];
                        pc = sc_c->tab.pzc[min(d, LDS_PZ)];
                    } else {
                        pt = (double)pz_lookup(pz_t, sc_t->par, sc_t->mean_kb, d) * IG_LOG_E_F;
                        pc = (double)pz_lookup(pz_c, sc_c->par, sc_c->mean_kb, d) * IG_LOG_E_F;
                    }
                    S += (pt - pc) * (double)n;
                    E += 1e-13 * (__builtin_fabs(pt) + __builtin_fabs(pc)) * (double)n;
                }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            S += __shfl_down(S, o, 64);
            E += __shfl_down(E, o, 64);
            SP += __shfl_down(SP, o, 64);
            bad |= __shfl_down(bad, o, 64);
            n_cis += __shfl_down(n_cis, o, 64);
            ob_cis += __shfl_down(ob_cis, o, 64);
        }
        if (lane == 0) {
            red[0][wv] = S;
            red[1][wv] = E;
            red[2][wv] = SP;
            red_bad[wv] = bad;
            red_n[0][wv] = n_cis;
            red_n[1][wv] = ob_cis;
        }
        __syncthreads();
        if (tid == 0) {
            S = red[0][0] + red[0][1] + red[0][2] + red[0][3];
            E = red[1][0] + red[1][1] + red[1][2] + red[1][3];
            SP = red[2][0] + red[2][1] + red[2][2] + red[2][3];
            bad = red_bad[0] | red_bad[1] | red_bad[2] | red_bad[3];
            n_cis = red_n[0][0] + red_n[0][1] + red_n[0][2] + red_n[0][3];
            ob_cis = red_n[1][0] + red_n[1][1] + red_n[1][2] + red_n[1][3];
            if (S != 0.0) atomic_add_ll(&out16[2], (long long)__builtin_rint(S * DIFF_FIX));
            if (E != 0.0) atomic_add_ll(&out16[3], (long long)__builtin_ceil(1.01 * E * DIFF_FIX) + 1);
            if (n_cis) atomic_add_ll(&out16[5], n_cis);
            if (ob_cis) atomic_add_ll(&out16[6], ob_cis);
            if (SP != 0.0) atomic_add_ll(&out16[7], (long long)__builtin_ceil(SP) + 1);
            if (bad) atomicOr((unsigned long long*)&out16[4], (unsigned long long)bad);
        }
    }
    /* the last workgroup through: what does not belong to a bin (trans contacts, cis ones at distance 0, the contract's own
     * roundings), then the sums to the (mapped) host memory and the flag; the words are cleared for the next launch */
    /* (everything a workgroup contributes is an atomic, acknowledged by the time the barrier lets thread 0 through: no release
     * fence in front of the ticket -- on this part an agent-scope release is a write-back of the XCD's L2, once per workgroup) */
    __syncthreads();
    if (tid == 0) {
#ifdef HIST_TICKET_FENCE
        __threadfence();
#endif
        if (atomicAdd((unsigned long long*)&out16[8], 1ull) == (unsigned long long)n_blocks - 1ull) {
            __threadfence();
            long long o[8];
            for (int q = 0; q < 8; q++) o[q] = __hip_atomic_load(&out16[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const ig_hot hc = sc_c->hot, ht = sc_t->hot;
            const long long n_zero = h.misc[0], ob_zero = h.misc[1], n_ring = h.misc[2], n_out = h.misc[3], n_all = h.misc[4], ob_all = h.misc[5],
                            n_big = h.misc[6];
            unsigned bad = (unsigned)o[4];
            if (!dc->ok0) bad |= 1u;
            if (n_ring != 0) bad |= 2u | 1024u; /* (bits from 16 up: which test it was, for the trace) */
            if (n_out != 0) bad |= 2u | 2048u;
            if (n_big != 0) bad |= 2u | 4096u;
            const long long n_tr = n_all - (o[5] + n_zero + n_ring + n_out), ob_tr = ob_all - (o[6] + ob_zero);
            if (n_tr < 0 || ob_tr < n_tr || n_zero < 0 || ob_zero < n_zero || n_all <= 0) bad |= 4u | 256u;
            if (!(__builtin_fmax(hc.log2_v_inter, ht.log2_v_inter) <= 17.5) || !(__builtin_fmin(hc.log2_v_inter, ht.log2_v_inter) >= -100.0)) bad |= 4u | 32u;
            const double c10 = IG_LOG2_10_INV;
            const double dlv = ht.log2_v_inter - hc.log2_v_inter, dexb = (double)ht.v_inter - (double)hc.v_inter;
            const double dpz_tr = sc_t->tab.pzc[LDS_PZ] - sc_c->tab.pzc[LDS_PZ];
            const double Dg = c10 * dlv * (double)(ob_tr + ob_zero) - dexb * (double)(n_tr + n_zero) + dpz_tr * (double)n_tr;
            const double vv = (double)ht.v_inter + (double)hc.v_inter;
            const double sp = (double)o[7] + vv * (double)(n_tr + n_zero);
            const double nn = (double)n_all;
            /* the contract itself (two quantisations per contact, its double roundings), as the pass above (T4) */
            const double T4 = nn * 0x1p-31 + 4e-15 * ((double)ob_all * 105.0 + 2.6 * sp + nn * (double)dc->pzc_abs_max);
            const double Eg = T4 + 2.0 / DIFF_FIX +
                              1e-13 * (c10 * __builtin_fabs(dlv) * (double)ob_all + __builtin_fabs(dexb) * nn + __builtin_fabs(dpz_tr) * nn);
            if (!(__builtin_fabs(Dg) < 1e15) || !(Eg < 1e12)) bad |= 4u | 512u;
            o[2] += (long long)__builtin_rint(Dg * DIFF_FIX);
            o[3] += (long long)__builtin_ceil(Eg * DIFF_FIX) + 1;
            o[4] = (long long)bad;
            o[0] = o[1] = 0;
            o[5] = n_all;
            for (int q = 0; q < 9; q++) out16[q] = 0;
            if (CHAIN) { /* the interval and the zero-pixel likelihood of the set (as nuis_z_from_sums forms it on the host) */
                long long zh = __hip_atomic_load(&zero_sums[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                          zl = __hip_atomic_load(&zero_sums[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const long long ni = __hip_atomic_load(&zero_sums[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ig_acc_normalize((int64_t*)&zh, (int64_t*)&zl);
                const double log_e = 0.43429448190325182;
                ct->s_fix = o[2];
                ct->b_fix = o[3];
                ct->flags = o[4];
                ct->z = ig_acc_to_double(zh, zl) * log_e + log_e * (g->n_tot_pxl - (double)ni) * -1.0 * (double)sc_t->par.v_inter;
                for (int q = 0; q < 8; q++) ((long long*)zero_sums)[q] = 0; /* (the next segment adds to them again) */
            } else if (hn) {
                for (int q = 0; q < 8; q++) hn->diff[q] = o[q];
                for (int q = 0; q < 8; q++) hn->sums[q] = __hip_atomic_load(&zero_sums[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                hn->diff_seq = hn_seq;
            }
            out16[9] = o[2]; /* (device copies for a host that could not map its memory, and for the tests) */
            out16[10] = o[3];
            out16[11] = o[4];
        }
    }
}
__global__ void __launch_bounds__(256) k_hist_eval(NuisHist h, const Glob* g, const ScoreConst* __restrict__ sc_t, const ScoreConst* __restrict__ sc_c,
                                                   const DiffConst* __restrict__ dc, long long* out16, NuisHost* hn, int hn_seq, Tables zt, int M,
                                                   long long* zero_out, int n_zero_blocks, const long long* __restrict__ zero_sums, PzTab pz_t,
                                                   PzTab pz_c)
{
    hist_eval_body<false>(h, g, sc_t, sc_c, dc, out16, hn, hn_seq, zt, M, zero_out, n_zero_blocks, zero_sums, pz_t, pz_c, (int)blockIdx.x, (int)gridDim.x,
                          nullptr);
}

/* ---- chains: CHAIN_SEG test sets per launch (ig_common.cuh, ChainIn) ------------------------------------------------------------ */
struct ChainSet { /* what k_hist_eval needs of one test set: built by k_chain_prepare exactly as k_nuis_prepare builds the single step's */
    ScoreConst sc;
    DiffConst dc;
    float pz[PZ_MAX];
    int pz_n;
};
/* the P_z table's length for a parameter set, as ig_set_params / enqueue_nuis_pass size it on the host */
__device__ __forceinline__ int chain_pz_n(float d_max, float mean_kb)
{
    const double need = (mean_kb > 0) ? (double)d_max / (double)mean_kb + 2.0 : 0.0;
    return (need > 0 && need < (double)PZ_MAX) ? (int)need : ((need >= (double)PZ_MAX) ? PZ_MAX : 0);
}
/* grid (blocks of 256 over the longest table, sets of the call): set y = the uploaded set in[y] */
__global__ void __launch_bounds__(256) k_chain_prepare(const Glob* g, const ChainIn* __restrict__ in, float mean_kb, ChainSet* sets,
                                                       const double* __restrict__ lgf_tab, const ScoreConst* __restrict__ sc0, int pz_n0,
                                                       const ScreenConst* __restrict__ scr0, long long* zs, int n_sets)
{
    const int set0 = 0;
    const int k = (int)blockIdx.y;
    ChainSet& S = sets[k];
    const ChainIn& ci = in[set0 + k];
    const ig_params p = {ci.p[0], ci.p[1], ci.p[2], ci.p[3], ci.p[4], ci.p[5], ci.p[6], ci.p[7]};
    const int pz_n = chain_pz_n(p.d_max, mean_kb);
    const int i = (int)blockIdx.x * blockDim.x + threadIdx.x;
    const float s_z = (float)i * mean_kb;
    const float pzv = (i < pz_n && s_z < p.d_max) ? ig_rippe(s_z, p, ig_tab()) : p.v_inter;
    if (i < pz_n) S.pz[i] = pzv;
    if (i < IG_TAB_SIZE) S.sc.tab.mt[i] = ig_tab()[i];
    if (i < LDS_PZ + 2) S.sc.tab.pzc[i] = (double)(i < min(pz_n, LDS_PZ) ? pzv : p.v_inter) * IG_LOG_E_F;
    if (i < LDS_LGF) S.sc.tab.lgf[i] = lgf_tab[i];
    if (i == 0) {
        S.sc.hot = ig_hot_make(p, ig_tab());
        S.sc.par = p;
        S.sc.mean_kb = mean_kb;
        S.pz_n = pz_n;
        if (k < CHAIN_SEG) /* (the segments' scratch words: zero between two launches -- k_chain_hist_eval's last workgroups clear them) */
            for (int q = 0; q < 8; q++) zs[8 * k + q] = 0;
    }
    build_diff_const(i, g, p, mean_kb, pzv, pz_n, sc0, pz_n0, &S.dc, scr0);
}
/* grid (blocks of one evaluation, sets) */
__global__ void __launch_bounds__(256) k_chain_hist_eval(NuisHist h, const Glob* g, const ChainSet* __restrict__ sets, const ScoreConst* __restrict__ sc_c,
                                                         long long* out16, Tables zt, int M, long long* zs, int n_zero_blocks, PzTab pz_c, ChainTest* tests)
{
    const int k = (int)blockIdx.y;
    const ChainSet& S = sets[k];
    hist_eval_body<true>(h, g, &S.sc, sc_c, &S.dc, out16 + 16 * k, nullptr, 0, zt, M, zs + 8 * k + 2, n_zero_blocks, zs + 8 * k, PzTab{S.pz, S.pz_n}, pz_c,
                         (int)blockIdx.x, (int)gridDim.x, tests + k);
}
