/* ig_kernels_screen.cuh -- two-tier scoring of a speculative batch.
 *
 * A batch needs the WINNER of every move and the winner's exact sums, not 105 exact scores per move: on the headline
 * workload 5 of the 104 scored columns of a move lie within 8 of the best score and 8 within 128 (tools/score_gaps.py), the
 * rest lose by thousands.  So every (contact, column) gets a cheap float evaluation of the same term with a RIGOROUS bound
 * on its distance from the contract's exact term (k_screen); the columns that can still win -- upper bound of the score
 * not below the best lower bound among the columns that are certainly scored -- are the contenders (k_contend), and only
 * those (plus column 0, the current genome, of every candidate that has one) go through the exact k_score_list.  The decide
 * step then sees the non-contenders as not scored.  The outcome -- winners, exact sums, genomes -- is the same as scoring
 * everything exactly, provided the bound holds; IG_SCREEN_VERIFY=1 scores everything exactly as well and checks the bound
 * column by column (tests), and the two hardware functions the bound leans on are checked over their whole domain
 * (ig_debug_transcendental_error, tests/test_hip_screen.py).
 *
 * The screening term (linear contigs, one-log domain; anything else makes the column a contender unconditionally), f32,
 * round to nearest, u = 2^-24; K_L, K_E = assumed error of v_log_f32 / v_exp_f32 in units of 2 u (|result|, 1) resp. 2 u result:
 *     L  = v_log_f32(s)                     |L - log2 s|  <= K_L 2u (|L| + 1)
 *     y  = fma(slope, L, la)                la = (float) log2(amp)
 *     yy = in ? max(y, lv) : lv             lv = (float) log2(v_inter)
 *     ex = v_exp_f32(yy)                    |ex - 2^yy|   <= K_E 2u 2^yy
 *     m  = (float) ob * yy
 *     t  = fma(m, (float) log10(2), -ex) + (float) pzc[d]
 * against the contract's  ob * yy * log10(2) - ex + pzc  (- log10(ob!), which is the same in every column of a candidate
 * and drops out of the score differences).  Propagating the roundings with K_L = K_E = 4:
 *     |t - exact| <= u { ob (4 |yy| + 2.72 Cy) + ex (6.3 |yy| + 6.3 Cy + 10.1) + 3.1 pzc } + 2^-32,
 *     Cy = |log2 amp| + |slope| + |log2 v_inter|;
 * two terms are added in float before they join the double sum (one more rounding, u (|t0| + |t1|)): the coefficients used
 * are 4.4, 11.2 and 4.2.
 * A workgroup accumulates sum(t) in double, sum(ob), sum(ex) and max |yy| and publishes its partial sum and bound as
 * integers (units of 2^-20: integer atomics, deterministic totals). */
#pragma once

#define SCR_KL 4.0
#define SCR_KE 4.0
#define SCR_FIX 1048576.0 /* 2^20 */
#ifndef EXACT_CHUNK
#define EXACT_CHUNK 4096 /* entries per work item of the exact kernel in a two-tier batch */
#endif

struct alignas(16) ScreenConst {
    float pzc[LDS_PZ + 2]; /* (float)(P_z log10 e) per rank distance; from the table's end on and for trans pairs: the trans level */
    float slope, la, lv, d_max;
    float cy;      /* |log2 amp| + |slope| + |log2 v_inter| (rounded up) */
    float pzc_max; /* largest table entry */
    float zc_ub;   /* upper bound of P_z log10(e) for a pair on a circular contig (< 0: none, the bound of such a column is void) */
    int fast;      /* parameters in the one-log domain */
    int pz_n;      /* length of the P_z table the exact path uses */
    const float* pz_full; /* that table (rank distances from LDS_PZ on are read from it: windows beyond the staged size only) */
    ig_params par;        /* ... and what the exact path evaluates behind its end (pz_lookup) */
    float mean_kb;
    int pz_far_ok; /* P_z decreases with the rank distance (slope < 0): pzc_max bounds the entries behind the LDS copy as well */
};

__device__ __forceinline__ void build_screen_const_block(const Glob* g, PzTab pz, ScreenConst* out, int which)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const ig_params p = g->par[which];
    const int pzn = min(pz.n, LDS_PZ);
    if (i < LDS_PZ + 2) out->pzc[i] = (float)((double)(i < pzn ? pz.v[i] : p.v_inter) * IG_LOG_E_F);
    if (i == 0) {
        const ig_hot h = ig_hot_make(p, ig_tab());
        out->slope = p.slope;
        out->la = (float)h.log2_amp;
        out->lv = (float)h.log2_v_inter;
        out->d_max = h.d_max;
        out->cy = (float)((__builtin_fabs(h.log2_amp) + __builtin_fabs(h.slope) + __builtin_fabs(h.log2_v_inter)) * 1.0001 + 1e-6);
        out->fast = h.fast;
        out->pz_n = pz.n;
        out->pz_full = pz.v;
        out->par = p;
        out->mean_kb = g->mean_kb;
        out->pz_far_ok = (h.fast && p.slope < 0.0f) ? 1 : 0;
        /* Pairs on a CIRCULAR contig (a ring made by a candidate: KA:3588-3649) are evaluated with rippe_contacts_circ, which
         * clamps BELOW with d_max (quirk Q6): P >= d_max, hundreds of contacts expected per pair -- such a column loses by
         * orders of magnitude and only needs an UPPER bound to be ruled out:
         *     ob log10 P - P <= max over P >= d_max            (concave: at P = max(d_max, ob / ln 10))
         *     P_z = rippe_contacts_circ(d mean, len mean) <= max(kuhn^-3 fact n0^slope, d_max),  n0 = (lm / kuhn) mean / 2
         * the second for slope < 0 (P decreasing in n = K s (S - s) / S, which is smallest, K mean (len - 1) / len >= n0, at
         * rank distance 1); other slopes: no bound, the column is scored exactly. */
        float zc = -1.0f;
        if (h.fast && p.slope < 0.0f && p.kuhn > 0.0f && p.lm > 0.0f && g->mean_kb > 0.0f && p.d_max > 0.0f && p.d_max < 1e30f) { /* (any d_max: screen_term takes the contract's clamp at -2^20 into the bound) */
            const double n0 = ((double)p.lm / (double)p.kuhn) * (double)g->mean_kb * 0.5;
            const double pmax = ig_exp2(h.slope * ig_log2_pos(n0, ig_tab()), ig_tab()) * (double)p.fact / ((double)p.kuhn * p.kuhn * p.kuhn);
            const double top = __builtin_fmax(__builtin_fmax(pmax, (double)p.d_max), (double)p.v_inter);
            if (top < 1e30) zc = (float)(top * IG_LOG_E_F * 1.001);
        }
        out->zc_ub = zc;
    }
    if (blockIdx.x == 0) { /* the largest table entry: one wave */
        if (threadIdx.x < 64) {
            float mx = (float)((double)p.v_inter * IG_LOG_E_F);
            for (int q = threadIdx.x; q < pzn; q += 64) mx = fmaxf(mx, (float)((double)pz.v[q] * IG_LOG_E_F));
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
            if (threadIdx.x == 0) out->pzc_max = mx * 1.0001f;
        }
    }
}
__global__ void k_build_screen_const(const Glob* g, PzTab pz, ScreenConst* out) { build_screen_const_block(g, pz, out, 0); }

/* per (candidate, column): approximate slice sum and its bound (2^-20 units), and the columns whose bound is void */
struct ScreenSum {
    long long s_fix, b_fix;
};

#ifndef SCREEN_BATCH
#define SCREEN_BATCH 2 /* entries a lane has in flight per step (one pair): 2 / 4 / 6 / 8 -> 208 / 212 / 218 / 232 us per launch at the headline shape */
#endif
struct alignas(16) ScreenLds {
    float pzc[LDS_PZ + 2]; /* first member: copied with 16-byte vectors */
    ColMeta cm[NCODE];
    double red_s[SCORE_THREADS / 64];
    float red_e[SCORE_THREADS / 64], red_y[SCORE_THREADS / 64], red_o[SCORE_THREADS / 64];
    unsigned red_bad[SCORE_THREADS / 64];
    unsigned pad_[2];
    /* LAST, and flush with the end of the block: inside a ScreenLds2 the second column's array follows it without a gap, and the
     * one-column routine stages windows of up to 2 x LDS_COL_SMALL sub-fragments across the two (screen_column) */
    uint2 col[LDS_COL_SMALL];
};
static_assert(sizeof(ScreenLds) == offsetof(ScreenLds, col) + sizeof(uint2) * LDS_COL_SMALL, "col must end the block");

/* one screened term; MASKED: the lane's entry may lie past the end of the list (last, partly filled step) */
template <bool STAGED, bool HAS_CUT, bool MASKED, bool CIRC = false, bool QUAD = false>
__device__ __forceinline__ void screen_term(unsigned long long pk, bool live, const uint2* gcol, const ScreenLds& L, float slope, float la,
                                            float lv, float d_max, float c10, unsigned cut, double& acc, float& exs, float& obs, float& ymax,
                                            unsigned& bad, unsigned circ_mask = 0, float zc_ub = 0.0f, const ScreenConst* __restrict__ far_sc = nullptr)
{
    const unsigned lo = (unsigned)pk, hi = (unsigned)(pk >> 32);
    bad |= MASKED ? (live ? hi : 0u) : hi; /* bits 8.. = the count: the largest count's leading bit survives the OR (checked at the end) */
    const unsigned li = lo & 0xfffffu, lj = __builtin_amdgcn_alignbit(hi, lo, 20) & 0xfffffu, ob = hi >> 8;
    const uint2 ai = STAGED ? L.col[li] : gcol[li];
    const uint2 bj = STAGED ? L.col[lj] : gcol[lj];
    /* packed lists: ranks below 2^20, the contig code in bits 28..30 -- for two different codes the difference of the
     * words is at least 2^28 - 2^20: "same contig" is d < 2^27, and min(d, LDS_PZ) lands on the trans level by itself */
    const unsigned dq = abs_diff_u32(ai.y, bj.y);
    const bool cis = dq < (1u << 27);
    const unsigned d = QUAD ? dq >> 2 : dq; /* QUAD: the staged column holds 4 x rank (screen_pair); across contigs d stays >= LDS_PZ */
    const float sv = fabsf(__uint_as_float(ai.x) - __uint_as_float(bj.x));
    const bool in = cis && (sv > 0.0f) && (sv < d_max);
    float pzc = L.pzc[min(d, (unsigned)LDS_PZ)];
    if (HAS_CUT) { /* a P_z table longer than its LDS copy and a window longer than that copy: the far pairs' entries from the table
                    * itself (same conversion as the copy's; pz_full == nullptr: no bound for them, the column is scored exactly) */
        const bool far = cis && d >= cut && (!MASKED || live);
        if (far && far_sc) pzc = (float)((double)pz_lookup(PzTab{far_sc->pz_full, far_sc->pz_n}, far_sc->par, far_sc->mean_kb, (int)d) * IG_LOG_E_F);
        bad |= (far && !far_sc) ? 0x80000000u : 0u; /* counts are below 2^24: bit 31 is free */
    }
    const float lg2 = __builtin_amdgcn_logf(sv);
    const float y = __builtin_fmaf(slope, lg2, la);
    float ymx;
    __asm__("v_max_f32 %0, %1, %2" : "=v"(ymx) : "v"(y), "v"(lv)); /* y is a number wherever it is used */
    const float yy = in ? ymx : lv;
    const float ex = __builtin_amdgcn_exp2f(yy);
    const float obf = (float)ob; /* exact: packed lists hold counts below 2^24 */
    const float m = obf * yy;
    float t = __builtin_fmaf(m, c10, -ex) + pzc;
    float exa = ex, oba = obf, yya = yy;
    if (CIRC) { /* a pair on a circular contig: an upper bound of its term (k_build_screen_const), nothing towards the error sums */
        const bool ring = cis && ((circ_mask >> (ai.y >> 28)) & 1u);
        const float em = fmaxf(d_max, 0.43429448f * obf);
        const float lem = __builtin_amdgcn_logf(em) * c10;
        /* ... of the term as the contract evaluates it: CLAMPED at -2^20 (ig_quantize) -- with a far d_max (a settled nuisance chain:
         * 3e6 kb, i.e. 3e6 contacts expected per ring pair) every such term sits at the clamp, above the unclamped bound.  The
         * screened sums leave log10(ob!) out (it is the same in every column): clamp(t - lgf) + lgf <= max(ub, -2^20 + lgf), and
         * log10(ob!) <= ob log10(ob) */
        const float ub0 = __builtin_fmaf(obf, lem, -em) + zc_ub + 1e-5f * (obf * fabsf(lem) + em);
        const float lgf_ub = obf * __builtin_amdgcn_logf(fmaxf(obf, 1.0f)) * c10 * 1.00001f + 1e-3f;
        const float ub = fmaxf(ub0, -1048576.0f + lgf_ub);
        t = ring ? ub : t;
        exa = ring ? 0.0f : exa;
        oba = ring ? 0.0f : oba;
        yya = ring ? 0.0f : yya;
    }
    if (MASKED) {
        t = live ? t : 0.0f;
        exa = live ? ex : 0.0f;
        oba = live ? obf : 0.0f;
        yya = live ? yy : 0.0f;
    }
    acc += (double)t;
    exs += exa;
    obs += oba;
    __asm__("v_max_f32 %0, %0, |%1|" : "+v"(ymax) : "v"(yya));
}

/* two screened terms at once, for the common case (column staged in LDS, no ring on the window, P_z table inside its LDS
 * copy, full step): the same term as screen_term, with the arithmetic on PAIRS (v_pk_add / v_pk_fma / v_pk_mul_f32: one
 * instruction for both lanes' two terms).  The staged column holds the rank pre-multiplied by 4 (k_screen), so the byte
 * offset into the P_z table is min(|4 rank_i - 4 rank_j|, 4 LDS_PZ) without a shift; the two terms are added in float before
 * they join the double sum (one more rounding of the pair: in the bound's coefficients). */
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void screen_pair(unsigned long long pk0, unsigned long long pk1, const ScreenLds& L, float slope, float la, float lv,
                                            float d_max, float c10, double& acc, f32x2& exs2, f32x2& obs2, float& ymax, unsigned& bad)
{
    const unsigned lo0 = (unsigned)pk0, hi0 = (unsigned)(pk0 >> 32), lo1 = (unsigned)pk1, hi1 = (unsigned)(pk1 >> 32);
    bad |= hi0 | hi1;
    const char* colb = (const char*)L.col;
    const uint2 a0 = *(const uint2*)(colb + ((lo0 << 3) & 0x7ffff8u)), b0 = *(const uint2*)(colb + (__builtin_amdgcn_alignbit(hi0, lo0, 17) & 0x7ffff8u));
    const uint2 a1 = *(const uint2*)(colb + ((lo1 << 3) & 0x7ffff8u)), b1 = *(const uint2*)(colb + (__builtin_amdgcn_alignbit(hi1, lo1, 17) & 0x7ffff8u));
    const unsigned d0 = abs_diff_u32(a0.y, b0.y), d1 = abs_diff_u32(a1.y, b1.y); /* 4 x rank distance, or >= 2^28 - 2^22 across contigs */
    /* two scalar subtractions: a packed one needs {a0.x, a1.x} in a register pair, i.e. two v_mov per v_pk_add */
    f32x2 sv;
    __asm__("v_sub_f32 %0, %1, %2" : "=v"(sv.x) : "v"(__uint_as_float(a0.x)), "v"(__uint_as_float(b0.x)));
    __asm__("v_sub_f32 %0, %1, %2" : "=v"(sv.y) : "v"(__uint_as_float(a1.x)), "v"(__uint_as_float(b1.x)));
    const bool in0 = (d0 < (1u << 27)) && (sv.x != 0.0f) && (fabsf(sv.x) < d_max);
    const bool in1 = (d1 < (1u << 27)) && (sv.y != 0.0f) && (fabsf(sv.y) < d_max);
    const char* pzb = (const char*)L.pzc;
    const f32x2 pzc = {*(const float*)(pzb + min(d0, 4u * LDS_PZ)), *(const float*)(pzb + min(d1, 4u * LDS_PZ))};
    const f32x2 lg2 = {__builtin_amdgcn_logf(fabsf(sv.x)), __builtin_amdgcn_logf(fabsf(sv.y))};
    const f32x2 y = __builtin_elementwise_fma(f32x2{slope, slope}, lg2, f32x2{la, la});
    float m0, m1;
    __asm__("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(y.x), "v"(lv)); /* y is a number wherever it is used */
    __asm__("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(y.y), "v"(lv));
    const f32x2 yy = {in0 ? m0 : lv, in1 ? m1 : lv};
    const f32x2 ex = {__builtin_amdgcn_exp2f(yy.x), __builtin_amdgcn_exp2f(yy.y)};
    const f32x2 obf = {(float)(hi0 >> 8), (float)(hi1 >> 8)};
    const f32x2 m = obf * yy;
    const f32x2 t = __builtin_elementwise_fma(m, f32x2{c10, c10}, -ex) + pzc;
    float tsum;
    __asm__("v_add_f32 %0, %1, %2" : "=v"(tsum) : "v"(t.x), "v"(t.y));
    acc += (double)tsum;
    exs2 += ex;
    obs2 += obf;
    __asm__("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ymax) : "v"(yy.x), "v"(yy.y));
}

/* a wave streams steps of 64 x SCREEN_BATCH consecutive entries (steps wave, wave + 4, ...); the entries of the next step
 * are loaded before this step's terms.  The lists live in a pool with slack behind its last entry (ensure_move_buffers):
 * the look-ahead loads need no clamping; only the last, partly filled step of a wave masks its lanes. */
template <bool STAGED, bool HAS_CUT, bool CIRC = false, bool PAIRS = false, bool QUAD = PAIRS>
__device__ __forceinline__ void screen_loop(const unsigned long long* __restrict__ slp, unsigned n, const uint2* gcol, const ScreenLds& L,
                                            float slope, float la_s, float lv_s, float d_max, unsigned cut, double& acc, float& exs, float& obs,
                                            float& ymax, unsigned& bad, unsigned circ_mask = 0, float zc_ub = 0.0f,
                                            const ScreenConst* __restrict__ far_sc = nullptr)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned step = 64 * SCREEN_BATCH, stride = step * (SCORE_THREADS / 64);
    const float c10 = (float)IG_LOG2_10_INV;
    float la, lv; /* in vector registers once, not copied from scalar ones before every use */
    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(la) : "s"(la_s));
    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(lv) : "s"(lv_s));
    const unsigned long long* ptr = slp + wave * step + lane;
    unsigned long long nx[SCREEN_BATCH];
#pragma unroll
    for (int u = 0; u < SCREEN_BATCH; u++) nx[u] = ptr[u * 64];
    f32x2 exs2 = {0.0f, 0.0f}, obs2 = {0.0f, 0.0f};
    unsigned s0 = wave * step; /* first entry of the wave's current step */
    for (; s0 + step <= n; s0 += stride) { /* full steps */
        unsigned long long pk[SCREEN_BATCH];
#pragma unroll
        for (int u = 0; u < SCREEN_BATCH; u++) pk[u] = nx[u];
        ptr += stride;
#pragma unroll
        for (int u = 0; u < SCREEN_BATCH; u++) nx[u] = ptr[u * 64];
        if (PAIRS) {
#pragma unroll
            for (int u = 0; u < SCREEN_BATCH; u += 2) screen_pair(pk[u], pk[u + 1], L, slope, la, lv, d_max, c10, acc, exs2, obs2, ymax, bad);
        } else {
#pragma unroll
            for (int u = 0; u < SCREEN_BATCH; u++)
                screen_term<STAGED, HAS_CUT, false, CIRC, QUAD>(pk[u], true, gcol, L, slope, la, lv, d_max, c10, cut, acc, exs, obs, ymax, bad, circ_mask,
                                                                  zc_ub, far_sc);
        }
    }
    if (PAIRS) {
        exs += exs2.x + exs2.y;
        obs += obs2.x + obs2.y;
    }
    if (s0 < n) { /* the partly filled step */
        const unsigned long long safe = slp[0];
#pragma unroll
        for (int u = 0; u < SCREEN_BATCH; u++) {
            const bool live = s0 + u * 64 + lane < n;
            screen_term<STAGED, HAS_CUT, true, CIRC, QUAD>(live ? nx[u] : safe, live, gcol, L, slope, la, lv, d_max, c10, cut, acc, exs, obs, ymax, bad,
                                                            circ_mask, zc_ub, far_sc);
        }
    }
}

/* a workgroup's share of a segment of n entries when Q workgroups split it (narrow batches with long lists: few (segment, pair,
 * candidate) triples, each long): part q of Q equal parts, cut at multiples of a workgroup's step; the partial sums of the parts
 * meet in the same atomics as the segments' */
__device__ __forceinline__ void screen_chunk(long long& n, long long& off, int q, int Q)
{
    if (Q <= 1) return;
    const long long unit = 64 * SCREEN_BATCH * (SCORE_THREADS / 64);
    const long long per = ((n + Q - 1) / Q + unit - 1) / unit * unit;
    const long long b = (long long)q * per, e = b + per < n ? b + per : n;
    off += b < n ? b : n;
    n = e > b ? e - b : 0;
}
/* one column k of candidate (w, c) over segment `seg` of its slice list: stage, stream, publish sum and bound.  Called by
 * every thread of the workgroup (barriers inside). */
__device__ __forceinline__ void screen_column(ScreenLds& L, const ScreenConst* __restrict__ sc, const MoveBuf& mb, ScreenSum* __restrict__ scr,
                                              unsigned* __restrict__ scr_void, unsigned* __restrict__ scr_ub, int w, int c, int k, int seg, int q = 0,
                                              int Q = 1)
{
    const int cw = CW(w, c);
    const int C = mb.ctl[PS(w)].C;
    const int n_uniq = mb.meta[cw].n_uniq, m_loc = mb.meta[cw].m_loc;
    long long n = mb.part[(size_t)cw * P_STRIDE + P_CNT + seg];
    long long off = mb.sloff[(size_t)cw * SLICE_SEG + seg];
    if (off >= 0) screen_chunk(n, off, q, Q);
    const float slope = sc->slope, la = sc->la, lv = sc->lv, d_max = sc->d_max, cy = sc->cy, pzc_max = sc->pzc_max;
    const int fast = sc->fast, pz_n = sc->pz_n;
    /* a column whose genome is the current genome on the window (k_mutate: nothing changed) has column 0's sums exactly */
    if (c >= C || k > n_uniq || n == 0 || off < 0) return;
    if (k > 0 && mb.sinfo[cw * NSLOT + mb.meta[cw].uniq[k - 1]].x == 0) return;
    const uint2* gcol = mb.coords + (size_t)(cw * NSLOT + k) * mb.sM;
    /* staged: in LDS -- up to LDS_COL_SMALL sub-fragments with the ranks pre-multiplied for the two-terms-at-a-time loop; up to twice
     * that across the second column's array (the block is always a ScreenLds2's first member) for the one-term loop: windows of
     * 1 025 .. 2 048 sub-fragments (two grown contigs) were gathered from global memory, a fifth of the launch's time for a few
     * workgroups at the headline shape (tools/screen_probe.py) */
    const bool staged = m_loc <= 2 * LDS_COL_SMALL;
    {
        const float4* src = (const float4*)sc->pzc;
        float4* dst = (float4*)L.pzc;
        for (int i = threadIdx.x; i < (LDS_PZ + 2) / 4; i += SCORE_THREADS) dst[i] = src[i];
        if (threadIdx.x < (LDS_PZ + 2) % 4) L.pzc[(LDS_PZ + 2) / 4 * 4 + threadIdx.x] = sc->pzc[(LDS_PZ + 2) / 4 * 4 + threadIdx.x];
    }
    if (threadIdx.x < NCODE) L.cm[threadIdx.x] = mb.cmeta[(size_t)(cw * NSLOT + k) * NCODE + threadIdx.x];
    /* the common case -- staged column, P_z table inside its LDS copy -- takes the two-terms-at-a-time loop (screen_pair),
     * which wants the ranks of the staged column pre-multiplied by 4 (ranks are below 2^20 in packed lists: no overlap with
     * the contig code in bits 28..30) */
    /* (a staged window holds at most LDS_COL_SMALL = LDS_PZ sub-fragments: every rank distance inside it is in the LDS copy,
     * however long the table) */
    static_assert(LDS_COL_SMALL <= LDS_PZ, "a staged window's rank distances must lie inside the staged P_z table");
    const bool pairs = m_loc <= LDS_COL_SMALL;
    if (staged)
        for (int i = threadIdx.x; i < m_loc; i += SCORE_THREADS) {
            uint2 v = gcol[i];
            if (pairs) v.y = (v.y & 0xf0000000u) | ((v.y & 0x0fffffffu) << 2);
            L.col[i] = v;
        }
    __syncthreads();
    unsigned circ_mask = 0;
#pragma unroll
    for (int q = 0; q < NCODE; q++) circ_mask |= (L.cm[q].stot != 0) ? (1u << q) : 0u;
    const float zc_ub = sc->zc_ub;
    /* outside the screening term's domain: the column goes through the exact kernel.  A ring on the window (a candidate
     * that closes a contig) is inside it as far as an UPPER bound goes -- all it takes to rule the column out; the current
     * genome's own column (k = 0) needs both bounds */
    if (!fast || (circ_mask && (zc_ub < 0.0f || k == 0))) {
        if (threadIdx.x == 0) atomicOr(&scr_void[cw], 1u << k);
        return;
    }
    if (circ_mask && threadIdx.x == 0) atomicOr(&scr_ub[cw], 1u << k); /* every segment's workgroup: an empty one leaves before */
    double acc = 0.0;
    float exs = 0.0f, ymax = 0.0f, obs = 0.0f;
    unsigned bad = 0;
    const unsigned cut = pz_n > LDS_PZ ? (unsigned)LDS_PZ : 0xffffffffu;
    const ScreenConst* far_sc = sc->pz_far_ok ? sc : nullptr;
    const unsigned long long* slp = mb.sl_pk + off;
    if (circ_mask) {
        if (pairs) screen_loop<true, false, true, false, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad, circ_mask, zc_ub);
        else if (staged) screen_loop<true, true, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad, circ_mask, zc_ub, far_sc);
        else screen_loop<false, true, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad, circ_mask, zc_ub, far_sc);
    } else if (pairs) {
        screen_loop<true, false, false, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad);
    } else if (pz_n > LDS_PZ) { /* a P_z table longer than its LDS copy, a window longer than that copy: far pairs from the table itself */
        if (staged) screen_loop<true, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad, 0, 0.0f, far_sc);
        else screen_loop<false, true>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad, 0, 0.0f, far_sc);
    } else {
        if (staged) screen_loop<true, false>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad);
        else screen_loop<false, false>(slp, (unsigned)n, gcol, L, slope, la, lv, d_max, cut, acc, exs, obs, ymax, bad);
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        exs += __shfl_down(exs, o, 64);
        obs += __shfl_down(obs, o, 64);
        ymax = fmaxf(ymax, __shfl_down(ymax, o, 64));
        bad |= __shfl_down(bad, o, 64);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        L.red_s[wv] = acc;
        L.red_e[wv] = exs;
        L.red_o[wv] = obs;
        L.red_y[wv] = ymax;
        L.red_bad[wv] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0.0, E = 0.0, O = 0.0, Y = 0.0;
        unsigned B = 0;
        for (int v = 0; v < SCORE_THREADS / 64; v++) {
            S += L.red_s[v];
            E += (double)L.red_e[v];
            O += (double)L.red_o[v];
            Y = __builtin_fmax(Y, (double)L.red_y[v]);
            B |= L.red_bad[v];
        }
        const double u = 0x1p-24;
        /* the bound of the header; sum(ob) and sum(ex) were accumulated in float by n / 256 additions per lane + the
         * reduction (relative error below (n / 256 + 16) 2u), everything left out is covered by 1 % */
        const double fa = 1.0 + 2.0 * u * ((double)n / SCORE_THREADS + 16.0) * 2.0;
        const double bound = 1.01 * (u * fa * (O * (4.4 * Y + 2.72 * (double)cy) + E * (6.3 * Y + 6.3 * (double)cy + 11.2)) +
                                     (double)n * (4.2 * u * (double)pzc_max + 0x1p-32));
        /* The contract clamps a term to |t| < 2^20 before it is quantised (ig_quantize); the screening term does not, and it
         * leaves log10(ob!) out.  Counts below 2^14 and |yy| <= 18 (P below 2^18) keep every exact term inside the clamp:
         * |t| <= 2^14 18 log10(2) + 2^18 + log10(2^14 !) + pzc < 2^20.  Beyond: the column's bound is void. */
        const bool in_clamp = ((B & 0x7fffffffu) >> 8) < (1u << 14) && Y <= 18.0 && (double)pzc_max < 1e5;
        const bool ok = !(B & 0x80000000u) && in_clamp && (__builtin_fabs(S) < 1e15) && (bound < 1e12); /* false for NaN / inf as well */
        if (!ok) {
            atomicOr(&scr_void[cw], 1u << k);
        } else {
            atomic_add_ll(&scr[cw * NSLOT + k].s_fix, (long long)__builtin_rint(S * SCR_FIX));
            atomic_add_ll(&scr[cw * NSLOT + k].b_fix, (long long)__builtin_ceil(bound * SCR_FIX) + 2); /* + the rounding of s_fix */
        }
    }
}

/* k_screen: one workgroup = (segment of the slice list, TWO columns 2g and 2g + 1, candidate cw).  k_screen's traffic is the
 * lists, re-read from L2 once per column (1.7 GB per launch at the headline shape: the L2 delivers little more): two plain
 * columns -- staged, no ring, P_z table inside its LDS copy -- share one pass over the entries (screen_pair2: one unpack,
 * two columns' terms); anything else takes the one-column routine, column by column. */
struct alignas(16) ScreenLds2 {
    ScreenLds one;            /* the one-column routine's block; its pzc table and col[] serve the pair loop as column A */
    uint2 colB[LDS_COL_SMALL]; /* column B of the pair loop (directly behind one.col) */
    double red_s2[SCORE_THREADS / 64];
    float red_e2[SCORE_THREADS / 64], red_y2[SCORE_THREADS / 64];
    unsigned red_void[SCORE_THREADS / 64]; /* bit 0 / 1: column A / B holds a ring pair whose count is beyond the linear bound */
};
/* the arithmetic of screen_pair on precomputed LDS byte offsets (o = 8 x local index) for one column */
/* ABL != 0: probe instances (IG_SCREEN_PROBE, wrong sums into scratch words): what the loop costs without its parts, timed on the
 * real trajectory next to the real launch -- bit 0: no P_z gather, bit 1: the partner's record made up from the row's (one LDS gather
 * per entry and column instead of two), bit 2: no transcendental functions, bit 3: no entries (set-up, staging, reduction and
 * publication only), bit 4: leave behind the first round of loads, bit 5: leave behind the staging's barrier */
/* CIRC (round 4): a column with a ring on its window (a candidate that closes a contig) through the same pass -- a pair on the ring
 * contributes the UPPER bound of its term instead of the term (screen_term has the derivation): with ob <= d_max ln 10 the bound is
 * linear in the count, ub = ob log10(d_max) - d_max + zc_ub (+ margins), clamped below like the contract's term at -2^20 (+ log10(ob!)
 * <= ob log10(2^14), which the screened sums leave out); a larger count voids the column (bit 31 of `bad`).  The error sums take the
 * ring pairs' linear-contig values along: too large, which a bound may be.  Until round 4 such columns went through the one-column
 * routine, staged again behind a barrier: 23 us of the launch's 172 at the headline shape (tools/screen_probe.py). */
struct RingUb {
    unsigned mask; /* contig codes that are rings in this column */
    float k1, k0;  /* ub = ob k1 + k0 */
    float ob_max;  /* counts above it: the linear form does not hold */
};
template <int ABL = 0, bool CIRC = false>
__device__ __forceinline__ void screen_pair_col(unsigned oi0, unsigned oj0, unsigned oi1, unsigned oj1, f32x2 obf, const char* colb, const char* pzb,
                                                float slope, float la, float lv, float d_max, float c10, double& acc, f32x2& exs2, float& ymax,
                                                const RingUb ru = RingUb{0u, 0.0f, 0.0f, 0.0f}, unsigned* bad = nullptr)
{
    const uint2 a0 = *(const uint2*)(colb + oi0), a1 = *(const uint2*)(colb + oi1);
    uint2 b0, b1;
    if (ABL & 2) {
        b0 = make_uint2(a0.x ^ oj0, a0.y + (oj0 & 0xff8u));
        b1 = make_uint2(a1.x ^ oj1, a1.y + (oj1 & 0xff8u));
    } else {
        b0 = *(const uint2*)(colb + oj0);
        b1 = *(const uint2*)(colb + oj1);
    }
    const unsigned d0 = abs_diff_u32(a0.y, b0.y), d1 = abs_diff_u32(a1.y, b1.y);
    f32x2 sv; /* scalar subtractions, see screen_pair */
    __asm__("v_sub_f32 %0, %1, %2" : "=v"(sv.x) : "v"(__uint_as_float(a0.x)), "v"(__uint_as_float(b0.x)));
    __asm__("v_sub_f32 %0, %1, %2" : "=v"(sv.y) : "v"(__uint_as_float(a1.x)), "v"(__uint_as_float(b1.x)));
    const bool in0 = (d0 < (1u << 27)) && (sv.x != 0.0f) && (fabsf(sv.x) < d_max);
    const bool in1 = (d1 < (1u << 27)) && (sv.y != 0.0f) && (fabsf(sv.y) < d_max);
    f32x2 pzc, lg2;
    if (ABL & 1) pzc = f32x2{__uint_as_float(d0 | 0x3f800000u), __uint_as_float(d1 | 0x3f800000u)};
    else pzc = f32x2{*(const float*)(pzb + min(d0, 4u * LDS_PZ)), *(const float*)(pzb + min(d1, 4u * LDS_PZ))};
    if (ABL & 4) lg2 = f32x2{fabsf(sv.x) * 0.001f, fabsf(sv.y) * 0.001f};
    else lg2 = f32x2{__builtin_amdgcn_logf(fabsf(sv.x)), __builtin_amdgcn_logf(fabsf(sv.y))};
    const f32x2 y = __builtin_elementwise_fma(f32x2{slope, slope}, lg2, f32x2{la, la});
    float m0, m1;
    __asm__("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(y.x), "v"(lv));
    __asm__("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(y.y), "v"(lv));
    const f32x2 yy = {in0 ? m0 : lv, in1 ? m1 : lv};
    f32x2 ex;
    if (ABL & 4) ex = f32x2{yy.x * 1.5f, yy.y * 1.5f};
    else ex = f32x2{__builtin_amdgcn_exp2f(yy.x), __builtin_amdgcn_exp2f(yy.y)};
    const f32x2 m = obf * yy;
    f32x2 t = __builtin_elementwise_fma(m, f32x2{c10, c10}, -ex) + pzc;
    if (CIRC) {
        const bool r0 = (d0 < (1u << 27)) && ((ru.mask >> (a0.y >> 28)) & 1u), r1 = (d1 < (1u << 27)) && ((ru.mask >> (a1.y >> 28)) & 1u);
        const float u0 = fmaxf(__builtin_fmaf(obf.x, ru.k1, ru.k0), __builtin_fmaf(obf.x, 4.2145f, -1048575.9f));
        const float u1 = fmaxf(__builtin_fmaf(obf.y, ru.k1, ru.k0), __builtin_fmaf(obf.y, 4.2145f, -1048575.9f));
        t.x = r0 ? u0 : t.x;
        t.y = r1 ? u1 : t.y;
        *bad |= ((r0 && obf.x > ru.ob_max) || (r1 && obf.y > ru.ob_max)) ? 0x80000000u : 0u;
    }
    float tsum; /* (kept scalar: the vectoriser pairs it with the other column's sum through three v_mov) */
    __asm__("v_add_f32 %0, %1, %2" : "=v"(tsum) : "v"(t.x), "v"(t.y));
    acc += (double)tsum;
    exs2 += ex;
    __asm__("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ymax) : "v"(yy.x), "v"(yy.y));
}
/* publish one column's sum and bound (thread 0 of the workgroup, after the reduction) */
__device__ __forceinline__ void screen_publish(const ScreenConst* __restrict__ sc, ScreenSum* __restrict__ scr, unsigned* __restrict__ scr_void,
                                               int cw, int k, long long n, double S, double E, double O, double Y, unsigned B)
{
    const double u = 0x1p-24, cy = sc->cy, pzc_max = sc->pzc_max;
    const double fa = 1.0 + 2.0 * u * ((double)n / SCORE_THREADS + 16.0) * 2.0;
    const double bound = 1.01 * (u * fa * (O * (4.4 * Y + 2.72 * cy) + E * (6.3 * Y + 6.3 * cy + 11.2)) + (double)n * (4.2 * u * pzc_max + 0x1p-32));
    const bool in_clamp = ((B & 0x7fffffffu) >> 8) < (1u << 14) && Y <= 18.0 && pzc_max < 1e5;
    const bool ok = !(B & 0x80000000u) && in_clamp && (__builtin_fabs(S) < 1e15) && (bound < 1e12);
    if (!ok) {
        atomicOr(&scr_void[cw], 1u << k);
    } else {
        atomic_add_ll(&scr[cw * NSLOT + k].s_fix, (long long)__builtin_rint(S * SCR_FIX));
        atomic_add_ll(&scr[cw * NSLOT + k].b_fix, (long long)__builtin_ceil(bound * SCR_FIX) + 2);
    }
}
/* one workgroup of the screening pass: segment `seg` of the slice list of candidate zc (= slot * max_c + c), columns 2 ypair, 2 ypair + 1 */
template <int ABL = 0>
__device__ __forceinline__ void screen_block(const ScreenConst* __restrict__ sc, const MoveBuf& mb, ScreenSum* __restrict__ scr,
                                             unsigned* __restrict__ scr_void, unsigned* __restrict__ scr_ub, int max_c, int w_begin, int seg, int ypair,
                                             int zc, ScreenLds2& L2, int use_order, int q = 0, int Q = 1)
{
    ScreenLds& L = L2.one;
    /* (use_order: the launch covers the slots k_offsets ordered -- the long lists first) */
    const int oc = use_order ? mb.order[zc] : (((w_begin + zc / max_c) << 8) | (zc % max_c));
    const int w = oc >> 8, c = oc & 255;
    if (KEPT(w)) return; /* (a slot of the window scored by an earlier launch and still valid) */
    const int cw = CW(w, c);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned step = 64 * SCREEN_BATCH, stride = step * (SCORE_THREADS / 64);
    /* What a workgroup needs is requested in rounds: everything whose address follows from the block index first (descriptors,
     * the uniq list, the constants, the P_z table), unconditionally; then the changed flags behind the uniq list; then the two
     * columns, their ring flags and the first entries.
     * The pairs are pairs of LIVE columns: a column whose genome is the current genome on the window (k_mutate's changed flag:
     * a third of them at the headline shape) has column 0's sums and is not screened; paired by index (2 y, 2 y + 1) two pairs in
     * five had one such column and took the one-column routine -- 80 of the launch's 209 us (tools/screen_probe.py).  Pair y is the
     * (2 y)-th and (2 y + 1)-th live column instead: only a candidate's last, odd column is left alone. */
    const CandMeta* mp = &mb.meta[cw];
    const int C = mb.ctl[PS(w)].C;
    const int n_uniq = mp->n_uniq, m_loc = mp->m_loc;
    const unsigned livecol = mb.livecol[cw]; /* bit k: column k's genome differs from the current one (k_mutate) */
    long long n = mb.part[(size_t)cw * P_STRIDE + P_CNT + seg];
    long long off = mb.sloff[(size_t)cw * SLICE_SEG + seg];
    if (off >= 0) screen_chunk(n, off, q, Q);
    const float slope = sc->slope, la_s = sc->la, lv_s = sc->lv, d_max = sc->d_max;
    const int fast = sc->fast;
    float4 pz_v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float pz_t = 0.0f;
    if (!(ABL & 64)) { /* (probe bit 6: no P_z table) */
        pz_v = ((const float4*)sc->pzc)[threadIdx.x]; /* (LDS_PZ + 2) / 4 = 256 vectors and two more words */
        pz_t = sc->pzc[(LDS_PZ + 2) / 4 * 4 + (threadIdx.x & 1)];
    }
    if (c >= C || 2 * ypair > n_uniq || n == 0 || off < 0) return;
    if (ABL & 16) return; /* (probe: dispatch and the first round of loads only) */
    /* which columns are live: column 0 and the changed ones (one word from k_mutate; until round 4 the changed flags of the
     * mutation slots behind the uniq list: a dependent round of loads in front of the columns) */
    const unsigned live = 1u | (livecol & (n_uniq >= 31 ? 0xfffffffeu : ((2u << n_uniq) - 2u)));
    const int n_live = __popc(live);
    if (2 * ypair >= n_live) return;
    unsigned rest = live;
    for (int q = 0; q < 2 * ypair; q++) rest &= rest - 1; /* drop the 2 y lowest set bits */
    const int kA = __ffs(rest) - 1;
    rest &= rest - 1;
    const int kB_real = rest ? __ffs(rest) - 1 : -1;
    /* a candidate's last, odd column goes through the pair loop next to itself (its second copy is not published): the one-column
     * routine -- staging again behind a barrier, half the work per fixed cost -- took twice a pair's time for it */
    const bool doA = true, doB = kB_real >= 0;
    const int kB = doB ? kB_real : kA;
    /* third round */
    const float ring_a = lane < NCODE ? mb.cmeta[(size_t)(cw * NSLOT + kA) * NCODE + lane].stot : 0.0f;
    const float ring_b = lane < NCODE ? mb.cmeta[(size_t)(cw * NSLOT + max(kB, 0)) * NCODE + lane].stot : 0.0f;
    const unsigned long long* slp = mb.sl_pk + off;
    const unsigned long long* ptr = slp + wave * step + lane;
    unsigned long long nx[SCREEN_BATCH];
#pragma unroll
    for (int u = 0; u < SCREEN_BATCH; u++) nx[u] = (ABL & 256) ? 0ull : ptr[u * 64]; /* (the pool has slack behind its last entry; probe bit 8: no entries requested) */
    const bool small = m_loc <= LDS_COL_SMALL;
    if (small && !(ABL & 128)) { /* (probe bit 7: no columns) */
        const uint2* gA = mb.coords + (size_t)(cw * NSLOT + kA) * mb.sM;
        const uint2* gB = mb.coords + (size_t)(cw * NSLOT + max(kB, 0)) * mb.sM;
        for (int i = threadIdx.x; i < m_loc; i += SCORE_THREADS) { /* ranks x 4, see screen_pair */
            uint2 va = gA[i], vb = gB[i];
            va.y = (va.y & 0xf0000000u) | ((va.y & 0x0fffffffu) << 2);
            vb.y = (vb.y & 0xf0000000u) | ((vb.y & 0x0fffffffu) << 2);
            L.col[i] = va;
            L2.colB[i] = vb;
        }
    }
    if (!(ABL & 64)) {
        ((float4*)L.pzc)[threadIdx.x] = pz_v;
        if (threadIdx.x < (LDS_PZ + 2) % 4) L.pzc[(LDS_PZ + 2) / 4 * 4 + threadIdx.x] = pz_t;
    }
    const unsigned circA = (unsigned)__ballot(lane < NCODE && ring_a != 0.0f), circB = doB ? (unsigned)__ballot(lane < NCODE && ring_b != 0.0f) : 0u;
    const bool ring = (circA | circB) != 0u; /* (every wave asks the same eight codes) */
    /* a ring goes through the pair loop's second instance where it has an upper bound (zc_ub) and is not on the current genome's own
     * column (k = 0 needs both bounds: void, the one-column routine says so); badA / badB: the columns are voided apart */
    const float zc_ub_s = sc->zc_ub;
    const bool ring_ok = zc_ub_s >= 0.0f && !(kA == 0 && circA) && !(ABL & (1024 | 2048 | 4096));
    const bool plain = fast && small && (!ring || ring_ok); /* staged: the pair loop */
    if (ABL & 512) return; /* (probe: everything requested and staged, no barrier, no pass of either kind) */
    if (!plain) {
        if (ABL & 1024) return; /* (probe: the pairs that take the one-column routine leave) */
        if ((ABL & 2048) && ring) return; /* (probe: ... those with a ring) */
        if ((ABL & 4096) && !ring) return; /* (probe: ... those without) */
        __syncthreads(); /* (everybody is through with the staging above: the one-column routine stages again) */
        screen_column(L, sc, mb, scr, scr_void, scr_ub, w, c, kA, seg, q, Q);
        if (doB) {
            __syncthreads(); /* the second column restages the LDS */
            screen_column(L, sc, mb, scr, scr_void, scr_ub, w, c, kB, seg, q, Q);
        }
        return;
    }
    __syncthreads();
    if (ABL & 32) return; /* (probe: ... and the second round, the staging, the barrier) */
    const float c10 = (float)IG_LOG2_10_INV;
    float la, lv;
    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(la) : "s"(la_s));
    __asm__ volatile("v_mov_b32 %0, %1" : "=v"(lv) : "s"(lv_s));
    const char* colA = (const char*)L.col;
    const char* colBb = (const char*)L2.colB;
    const char* pzb = (const char*)L.pzc;
    double accA = 0.0, accB = 0.0;
    float ymaxA = 0.0f, ymaxB = 0.0f, exA = 0.0f, exB = 0.0f, obs = 0.0f;
    unsigned bad = 0, badA = 0, badB = 0;
    f32x2 exsA = {0.0f, 0.0f}, exsB = {0.0f, 0.0f}, obs2 = {0.0f, 0.0f};
    unsigned s0 = wave * step;
    const unsigned nn = (ABL & 8) ? 0u : (unsigned)n; /* (probe bit 3: no entries at all -- what a workgroup costs before and after its loop) */
    /* the ring pairs' bound, linear in the count while 0.434 ob <= d_max (screen_term: em = max(d_max, 0.434 ob)) */
    const float lem_s = __builtin_amdgcn_logf(fmaxf(d_max, 1e-30f)) * c10;
    const RingUb ruA{circA, lem_s + 1e-5f * fabsf(lem_s), zc_ub_s - d_max + 1e-5f * d_max, d_max * 2.3f},
                 ruB{circB, lem_s + 1e-5f * fabsf(lem_s), zc_ub_s - d_max + 1e-5f * d_max, d_max * 2.3f};
    if (ring && threadIdx.x == 0) { /* (every segment's workgroup: an empty one left before) */
        if (circA) atomicOr(&scr_ub[cw], 1u << kA);
        if (circB) atomicOr(&scr_ub[cw], 1u << kB);
    }
    if (ring) { /* the same pass, the ring pairs' upper bounds selected in (a second instance of the loop: the common one stays as it is) */
        for (; s0 + step <= nn; s0 += stride) {
            ptr += stride;
#pragma unroll
            for (int u = 0; u < SCREEN_BATCH; u += 2) {
                const unsigned lo0 = (unsigned)nx[u], hi0 = (unsigned)(nx[u] >> 32), lo1 = (unsigned)nx[u + 1], hi1 = (unsigned)(nx[u + 1] >> 32);
                bad |= hi0 | hi1;
                const unsigned oi0 = (lo0 << 3) & 0x7ffff8u, oj0 = __builtin_amdgcn_alignbit(hi0, lo0, 17) & 0x7ffff8u;
                const unsigned oi1 = (lo1 << 3) & 0x7ffff8u, oj1 = __builtin_amdgcn_alignbit(hi1, lo1, 17) & 0x7ffff8u;
                const f32x2 obf = {(float)(hi0 >> 8), (float)(hi1 >> 8)};
                nx[u] = ptr[u * 64];
                nx[u + 1] = ptr[(u + 1) * 64];
                obs2 += obf;
                screen_pair_col<0, true>(oi0, oj0, oi1, oj1, obf, colA, pzb, slope, la, lv, d_max, c10, accA, exsA, ymaxA, ruA, &badA);
                screen_pair_col<0, true>(oi0, oj0, oi1, oj1, obf, colBb, pzb, slope, la, lv, d_max, c10, accB, exsB, ymaxB, ruB, &badB);
            }
        }
    }
    for (; s0 + step <= nn; s0 += stride) { /* full steps: both columns from one pass over the entries */
        ptr += stride;
#pragma unroll
        for (int u = 0; u < SCREEN_BATCH; u += 2) {
            const unsigned lo0 = (unsigned)nx[u], hi0 = (unsigned)(nx[u] >> 32), lo1 = (unsigned)nx[u + 1], hi1 = (unsigned)(nx[u + 1] >> 32);
            bad |= hi0 | hi1;
            const unsigned oi0 = (lo0 << 3) & 0x7ffff8u, oj0 = __builtin_amdgcn_alignbit(hi0, lo0, 17) & 0x7ffff8u;
            const unsigned oi1 = (lo1 << 3) & 0x7ffff8u, oj1 = __builtin_amdgcn_alignbit(hi1, lo1, 17) & 0x7ffff8u;
            const f32x2 obf = {(float)(hi0 >> 8), (float)(hi1 >> 8)};
            /* unpacked: the next step's entries straight into the same registers (no copy of the look-ahead) */
            if (!(ABL & 8192)) { /* (probe bit 13: the first step's entries again and again -- no loads inside the loop) */
                nx[u] = ptr[u * 64];
                nx[u + 1] = ptr[(u + 1) * 64];
            }
            obs2 += obf;
            screen_pair_col<ABL>(oi0, oj0, oi1, oj1, obf, colA, pzb, slope, la, lv, d_max, c10, accA, exsA, ymaxA);
            screen_pair_col<ABL>(oi0, oj0, oi1, oj1, obf, colBb, pzb, slope, la, lv, d_max, c10, accB, exsB, ymaxB);
        }
    }
    exA += exsA.x + exsA.y;
    exB += exsB.x + exsB.y;
    obs += obs2.x + obs2.y;
    if (s0 < nn) { /* the partly filled step: one term at a time, masked */
        const unsigned long long safe = slp[0];
        float obs_dummy = 0.0f;
        unsigned bad_dummy = 0;
#pragma unroll
        for (int u = 0; u < SCREEN_BATCH; u++) {
            const bool live = s0 + u * 64 + lane < nn;
            const unsigned long long e = live ? nx[u] : safe;
            if (ring) screen_term<true, false, true, true, true>(e, live, nullptr, L, slope, la, lv, d_max, c10, 0xffffffffu, accA, exA, obs, ymaxA, bad, circA, zc_ub_s);
            else screen_term<true, false, true, false, true>(e, live, nullptr, L, slope, la, lv, d_max, c10, 0xffffffffu, accA, exA, obs, ymaxA, bad);
            /* column B: the same routine on a view of the block whose col[] is column B */
            {
                const unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
                const unsigned oi = (lo << 3) & 0x7ffff8u, oj = __builtin_amdgcn_alignbit(hi, lo, 17) & 0x7ffff8u;
                const uint2 ai = *(const uint2*)(colBb + oi), bj = *(const uint2*)(colBb + oj);
                const unsigned dq = abs_diff_u32(ai.y, bj.y);
                const bool cis = dq < (1u << 27);
                const float sv = fabsf(__uint_as_float(ai.x) - __uint_as_float(bj.x));
                const bool in = cis && (sv > 0.0f) && (sv < d_max);
                const float pzc = *(const float*)(pzb + min(dq, 4u * LDS_PZ));
                const float y = __builtin_fmaf(slope, __builtin_amdgcn_logf(sv), la);
                float ymx;
                __asm__("v_max_f32 %0, %1, %2" : "=v"(ymx) : "v"(y), "v"(lv));
                const float yy = in ? ymx : lv;
                const float ex = __builtin_amdgcn_exp2f(yy);
                float t = __builtin_fmaf((float)(hi >> 8) * yy, c10, -ex) + pzc;
                if (ring && cis && ((circB >> (ai.y >> 28)) & 1u)) { /* (screen_term's bound, as column A's masked step takes it) */
                    const float obf1 = (float)(hi >> 8);
                    const float em = fmaxf(d_max, 0.43429448f * obf1);
                    const float lem = __builtin_amdgcn_logf(em) * c10;
                    const float ub0 = __builtin_fmaf(obf1, lem, -em) + zc_ub_s + 1e-5f * (obf1 * fabsf(lem) + em);
                    const float lgf_ub = obf1 * __builtin_amdgcn_logf(fmaxf(obf1, 1.0f)) * c10 * 1.00001f + 1e-3f;
                    t = fmaxf(ub0, -1048576.0f + lgf_ub);
                }
                accB += (double)(live ? t : 0.0f);
                exB += live ? ex : 0.0f;
                const float yya = live ? yy : 0.0f;
                __asm__("v_max_f32 %0, %0, |%1|" : "+v"(ymaxB) : "v"(yya));
            }
            (void)obs_dummy;
            (void)bad_dummy;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        accA += __shfl_down(accA, o, 64);
        accB += __shfl_down(accB, o, 64);
        exA += __shfl_down(exA, o, 64);
        exB += __shfl_down(exB, o, 64);
        obs += __shfl_down(obs, o, 64);
        ymaxA = fmaxf(ymaxA, __shfl_down(ymaxA, o, 64));
        ymaxB = fmaxf(ymaxB, __shfl_down(ymaxB, o, 64));
        bad |= __shfl_down(bad, o, 64);
        badA |= __shfl_down(badA, o, 64);
        badB |= __shfl_down(badB, o, 64);
    }
    if (lane == 0) {
        L.red_s[wave] = accA;
        L.red_e[wave] = exA;
        L.red_o[wave] = obs;
        L.red_y[wave] = ymaxA;
        L.red_bad[wave] = bad;
        L2.red_void[wave] = (badA >> 31) | ((badB >> 31) << 1);
        L2.red_s2[wave] = accB;
        L2.red_e2[wave] = exB;
        L2.red_y2[wave] = ymaxB;
    }
    __syncthreads();
    if (threadIdx.x < (doB ? 2 : 1)) {
        const bool isB = threadIdx.x == 1;
        double S = 0.0, E = 0.0, O = 0.0, Y = 0.0;
        unsigned B = 0;
        for (int v = 0; v < SCORE_THREADS / 64; v++) {
            S += isB ? L2.red_s2[v] : L.red_s[v];
            E += (double)(isB ? L2.red_e2[v] : L.red_e[v]);
            O += (double)L.red_o[v];
            Y = __builtin_fmax(Y, (double)(isB ? L2.red_y2[v] : L.red_y[v]));
            B |= L.red_bad[v];
        }
        for (int v = 0; v < SCORE_THREADS / 64; v++) B |= ((L2.red_void[v] >> (isB ? 1 : 0)) & 1u) ? 0x80000000u : 0u;
        screen_publish(sc, scr, scr_void, cw, isB ? kB : kA, n, S, E, O, Y, B);
    }
}
/* (round 5's MULTI-COLUMN pass -- a workgroup = (segment, a group of up to 8 live columns staged side by side, candidate) -- was built, measured
 * at 250 - 310 us against this pair pass's 170 and dropped; the numbers are in DESIGN 4.3 and profiles/r05e_screen_multi_probe.txt, the code
 * in the history before round 6.) */

#ifndef SCREEN_MIN_WAVES
#define SCREEN_MIN_WAVES 7 /* seven workgroups per CU (what the LDS admits) need <= 96 SGPRs (106 admit six) */
#endif
template <int ABL = 0>
__global__ void __launch_bounds__(SCORE_THREADS, SCREEN_MIN_WAVES)
    k_screen(const ScreenConst* __restrict__ sc, MoveBuf mb, ScreenSum* __restrict__ scr, unsigned* __restrict__ scr_void, unsigned* __restrict__ scr_ub,
             int max_c, int w_begin, int use_order, int Q)
{
    __shared__ ScreenLds2 L2;
    screen_block<ABL>(sc, mb, scr, scr_void, scr_ub, max_c, w_begin, blockIdx.x % mb.nseg, blockIdx.y, blockIdx.z, L2, use_order, blockIdx.x / mb.nseg, Q);
}

/* k_screen and k_tail in ONE launch.  The Q5 tail walk (prefinal_tail: one workgroup per candidate, a chain of dependent
 * loads of ~80 us) needs the slice lists only and used to run on a second stream next to k_screen: the event record / wait
 * pairs that ordered the two streams cost ~10 us of idle queue each, three times per batch.  Here the first n_tail
 * workgroups of a 1-D grid are the tail walks (dispatched first, done long before the screening workgroups are), the rest
 * the (segment, column pair, candidate) workgroups of k_screen in the same order as its 3-D grid (block -> XCD affinity of a
 * segment is kept: n_tail shifts every segment by the same amount). */
__global__ void __launch_bounds__(SCORE_THREADS, SCREEN_MIN_WAVES)
    k_screen_tail(const ScreenConst* __restrict__ sc, MoveBuf mb, ScreenSum* __restrict__ scr, unsigned* __restrict__ scr_void,
                  unsigned* __restrict__ scr_ub, int max_c, int w_begin, int n_tail, const long long* __restrict__ rowptr,
                  const int2* __restrict__ cc, Tables tab, Glob* g, const double* __restrict__ lgf_tab, int tail_quirk, PzTab pz, int use_order, int Q,
                  int tsplit)
{
    /* one block of LDS for either kind of workgroup (the tail walk's 2.3 KB on top of the screening block's 20.2 KB cost the eighth
     * workgroup per CU) */
    __shared__ __align__(16) unsigned char lds_raw[sizeof(ScreenLds2) > sizeof(TailLds) ? sizeof(ScreenLds2) : sizeof(TailLds)];
    const int b = (int)blockIdx.x;
    if (b < n_tail) {
        const int bb = b / tsplit; /* tsplit workgroups per (candidate, slot): the columns dealt out among them (prefinal_tail) */
        prefinal_tail(rowptr, cc, tab, g, mb, lgf_tab, tail_quirk, pz, w_begin, bb % max_c, bb / max_c, *(TailLds*)lds_raw, b % tsplit, tsplit);
        return;
    }
    const int s = b - n_tail, ny = (NSLOT + 1) / 2, nseg = mb.nseg, nxq = nseg * Q; /* x: segment fastest (block -> XCD), then the part of it */
    screen_block(sc, mb, scr, scr_void, scr_ub, max_c, w_begin, s % nseg, (s / nxq) % ny, s / (nxq * ny), *(ScreenLds2*)lds_raw, use_order,
                 (s % nxq) / nseg, Q);
}

/* k_contend: one workgroup per move slot.  From the screened sums, the exact zero-pixel sums and the exact tail sums: an
 * interval [lo, hi] for every scored (candidate, slot) that contains the score the decide step will compute; the
 * contenders are the columns with hi >= the best lo among the columns that are scored whatever the stale insert flags turn
 * out to be (candidate 0 of a slot w > 0 is screened with ALL block-insert slots, quirk Q4).  cont[cw]: bit k = column k
 * goes through the exact kernel (bit 0, the current genome's column, with any other bit). */
__global__ void __launch_bounds__(256) k_contend(Glob* g, MoveBuf mb, const ScreenSum* __restrict__ scr, const unsigned* __restrict__ scr_void,
                                                 const unsigned* __restrict__ scr_ub, unsigned* __restrict__ cont, int w_begin, int force_all, int grid_cap, int chunk0,
                                                 int inline_worklist)
{
    __shared__ double s_lo[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT], s_hi[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT];
    __shared__ int s_kind[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT]; /* 0 not scored, 1 always scored, 2 depends on the flags, 3 void bound */
    __shared__ double s_best[4];
    __shared__ unsigned s_mask[IG_MAX_CANDIDATES], s_ident[IG_MAX_CANDIDATES];
    const int w = w_begin + blockIdx.x, tid = threadIdx.x;
    if (KEPT(w)) { /* (nothing of this slot on the exact kernel's work list: k_worklist adds up the slots in front of a slot) */
        if (tid < 8) mb.slot_items[w * 8 + tid] = 0;
        return;
    }
    const MoveCtl& mc = mb.ctl[PS(w)];
    const int C = mc.C, n = C * IG_N_TMP_STRUCT;
    const ig_params p = g->par[0];
    const double log_e = IG_LOG_E_F, n_tot_pxl = g->n_tot_pxl;
    const long long z_hi = g->z_hi, z_lo = g->z_lo, n_intra = g->n_intra;
    const double cur_nz = ig_acc_to_double(g->nz_hi, g->nz_lo);
    if (tid < IG_MAX_CANDIDATES) s_mask[tid] = s_ident[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
        const int cw = CW(w, c);
        const CandMeta& m = mb.meta[cw];
        const int k = m.kidx[slot];
        int kind = 0;
        double lo = 0.0, hi = 0.0;
        if (k > 0) {
            const bool sup = (c == 0) && mc.superset0 && (slot >= 12);
            kind = sup ? 2 : 1;
            const unsigned vd = scr_void[cw];
            const bool same = force_all != 1 && mb.sinfo[cw * NSLOT + slot].x == 0; /* the current genome again: nz - ext = 0 exactly */
            if (same) atomicOr(&s_ident[c], 1u << k);
            if (!same && (force_all == 1 || ((vd >> k) & 1u) || (vd & 1u))) {
                kind = 3;
            } else {
                const long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
                const ScreenSum a = scr[cw * NSLOT + k], b = scr[cw * NSLOT];
                const double D = same ? 0.0 : (double)(a.s_fix - b.s_fix) * (1.0 / SCR_FIX);
                const double Bd = same ? 0.0 : (double)(a.b_fix + b.b_fix) * (1.0 / SCR_FIX);
                const long long dz_hi = qp[Q_Z + 2 * k] - qp[Q_Z], dz_lo = qp[Q_Z + 2 * k + 1] - qp[Q_Z + 1];
                const long long dni = qp[Q_NI + k] - qp[Q_NI];
                const double val_inter = -1.0 * log_e * (n_tot_pxl - (double)(n_intra + dni)) * p.v_inter;
                const double val_intra = ig_acc_to_double(z_hi + dz_hi, z_lo + dz_lo) * log_e;
                const double full = D + (val_intra + val_inter) + cur_nz;
                const double tail = ig_acc_to_double(qp[Q_TAIL + 2 * k], qp[Q_TAIL + 2 * k + 1]);
                const int r = (int)(slice_total(mb.part + (size_t)cw * P_STRIDE) % 64);
                double vmin, vmax;
                if (r == 0) {
                    vmin = vmax = full;
                } else if (sup) { /* its position in the list, hence whether the tail counts, depends on the flags (quirk Q5) */
                    vmin = __builtin_fmin(full, full - tail);
                    vmax = __builtin_fmax(full, full - tail);
                } else {
                    vmin = vmax = (k - 1 >= r) ? full - tail : full;
                }
                /* slack: the decide step assembles the score in double from the LIVE scalars (a shift common to all
                 * columns of the move, up to roundings of numbers of this size) */
                const double slack = 1e-3 + 1e-12 * (__builtin_fabs(cur_nz) + __builtin_fabs(val_intra) + __builtin_fabs(val_inter));
                lo = vmin - Bd - slack;
                hi = vmax + Bd + slack;
                if ((scr_ub[cw] >> k) & 1u) { /* a ring on the window: the screened sum is an upper bound only */
                    lo = -1e299;
                    if (kind == 1) kind = 2; /* bounds nothing */
                }
                if (!(lo == lo) || !(hi == hi) || !(__builtin_fabs(lo) < 1e300) || !(__builtin_fabs(hi) < 1e300)) kind = 3;
                /* an exact score of 0.0 counts as "not scored" in the argmax (CL:1435-1440): such a column bounds nothing */
                else if (lo <= 0.0 && hi >= 0.0 && kind == 1) kind = 2;
            }
        }
        if (kind == 3 && force_all == 2) kind = 0; /* timing experiment only (IG_CONTEND_ALL=2): void columns dropped -- WRONG results */
        if (kind == 3) atomic_add_ll(&g->scr_void_cols, 1);
        s_kind[i] = kind;
        s_lo[i] = lo;
        s_hi[i] = hi;
    }
    __syncthreads();
    double best = -IG_INF;
    for (int i = tid; i < n; i += blockDim.x)
        if (s_kind[i] == 1) best = __builtin_fmax(best, s_lo[i]);
    for (int o = 32; o > 0; o >>= 1) best = __builtin_fmax(best, __shfl_down(best, o, 64));
    if ((tid & 63) == 0) s_best[tid >> 6] = best;
    __syncthreads();
    best = __builtin_fmax(__builtin_fmax(s_best[0], s_best[1]), __builtin_fmax(s_best[2], s_best[3]));
    for (int i = tid; i < n; i += blockDim.x) {
        const int kind = s_kind[i];
        if (kind == 0) continue;
        if (kind == 3 || s_hi[i] >= best) {
            const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
            atomicOr(&s_mask[c], (1u << mb.meta[CW(w, c)].kidx[slot]) | 1u);
        }
    }
    __syncthreads();
    if (tid < C) {
        cont[CW(w, tid)] = s_mask[tid];
        mb.ident[CW(w, tid)] = s_ident[tid];
    }
    /* what this slot will put on the exact kernel's work list (k_worklist): items of at most ch entries; ch is chosen so
     * that a slot alone never needs more than half of the grid */
    if (tid == 0) {
        long long tot = 0;
        for (int c = 0; c < C; c++) tot += slice_total(mb.part + (size_t)CW(w, c) * P_STRIDE) * __popc(s_mask[c] & ~s_ident[c]);
        /* (the constant part -- at most one partly filled item per (candidate, column, segment) -- does not shrink with ch: the
         * host keeps grid_cap above twice that for the widest move (exact_grid_floor); should it ever not be, the loop ends at
         * whole segments, k_worklist flags the slot (overflow 2) and the host enlarges the grid) */
        long long ch = chunk0 > 0 ? chunk0 : EXACT_CHUNK;
        while (ch < (1LL << 40) && tot / ch + (long long)C * NSLOT * mb.nseg > grid_cap / 2) ch *= 2;
        mb.ctl[PS(w)].exact_chunk = (int)ch;
        s_kind[0] = (int)ch; /* (the intervals are done with) */
    }
    /* ... and how many items that is per sub-list (segment s -> sub-list s % 8): k_worklist places the slots in order from these */
    __shared__ int s_items[8];
    if (tid < 8) s_items[tid] = 0;
    __syncthreads();
    {
        const long long ch = s_kind[0];
        for (int u = tid; u < C * SLICE_SEG; u += blockDim.x) {
            const int c = u / SLICE_SEG, seg = u % SLICE_SEG;
            const long long n_seg = mb.part[(size_t)CW(w, c) * P_STRIDE + P_CNT + seg];
            const int it = (int)((n_seg + ch - 1) / ch) * __popc(s_mask[c] & ~s_ident[c]);
            if (it) atomicAdd(&s_items[seg & 7], it);
        }
    }
    __syncthreads();
    if (tid < 8) mb.slot_items[w * 8 + tid] = s_items[tid];
    if (tid == 0) { /* diagnostics: columns screened / columns sent to the exact kernel (column 0 of a candidate included) */
        long long cols = 0, cnt = 0;
        long long tcols = 0, tcnt = 0;
        for (int c = 0; c < C; c++) {
            const long long Sc = slice_total(mb.part + (size_t)CW(w, c) * P_STRIDE);
            cols += mb.meta[CW(w, c)].n_uniq + 1;
            cnt += __popc(s_mask[c] & ~s_ident[c]);
            tcols += Sc * (mb.meta[CW(w, c)].n_uniq + 1);
            tcnt += Sc * __popc(s_mask[c] & ~s_ident[c]);
        }
        atomic_add_ll(&g->scr_cols, cols);
        atomic_add_ll(&g->scr_cont, cnt);
        atomic_add_ll(&g->scr_terms, tcols);
        atomic_add_ll(&g->scr_terms_exact, tcnt);
    }
    /* a launch of ONE slot (ig_step_draw: one move per call): the slot's part of the work list right here, from the masks and counts
     * in LDS -- k_worklist's arithmetic with nothing in front of the slot; its launch (5 us of a 143 us call) is left out (round 6) */
    if (inline_worklist) {
        __shared__ int wl_cnt[8], wl_fit;
        if (tid == 0) {
            bool fit = true;
            for (int x = 0; x < 8; x++) fit &= 8LL * s_items[x] <= (long long)grid_cap;
            wl_fit = fit;
            if (!fit && mb.ctl[PS(w)].overflow == 0) mb.ctl[PS(w)].overflow = 2; /* (never over a 1: k_worklist) */
            for (int x = 0; x < 8; x++) {
                if (fit) atomicMax(&mb.work[x], (unsigned long long)s_items[x]);
                atomicMax(&mb.work[8 + x], (unsigned long long)s_items[x]);
            }
        }
        if (tid < 8) wl_cnt[tid] = 0;
        __syncthreads();
        if (!wl_fit) return;
        const long long ch = s_kind[0];
        for (int u = tid; u < C * SLICE_SEG; u += blockDim.x) {
            const int c = u / SLICE_SEG, seg = u % SLICE_SEG;
            const int cw = CW(w, c);
            const long long n_seg = mb.part[(size_t)cw * P_STRIDE + P_CNT + seg];
            const int nch = (int)((n_seg + ch - 1) / ch);
            const unsigned mask = s_mask[c] & ~s_ident[c];
            if (!nch || !mask) continue;
            const int x = seg & 7;
            int j = atomicAdd(&wl_cnt[x], nch * __popc(mask));
            for (int q = 0; q < nch; q++) {
                unsigned m = mask;
                while (m) {
                    const int k = __ffs(m) - 1;
                    m &= m - 1;
                    mb.work[16 + 8 * (size_t)(j++) + x] =
                        ((unsigned long long)q << 32) | ((unsigned long long)cw << 12) | ((unsigned long long)k << 4) | (unsigned)seg;
                }
            }
        }
    }
}

/* k_worklist: the exact kernel's work list of a batch, one workgroup per slot (after k_contend).  Items of at most ch
 * entries (MoveCtl.exact_chunk) of one segment under one contender column, in eight interleaved sub-lists: the items of
 * segment s go to sub-list s % 8, item j of sub-list x sits at index 8 j + x -- the index is the exact kernel's block index,
 * block b runs on XCD b % 8 (observed; a placement for speed, nothing depends on it), the XCD whose L2 k_screen just pulled
 * that segment's lists through; the contender columns of a chunk are consecutive in their sub-list.  The slots take their
 * places in slot order (the counts per slot and sub-list come from k_contend), so a slot
 * fits or not regardless of the ones behind it: the slots that fit are a prefix of the batch, the first one always fits
 * (k_contend's choice of ch), a slot that does not fit is flagged like a slot whose slice overflowed the pool -- the decide
 * step stops before it and it is re-run.  grid_cap = the blocks the host launches the exact kernel with (sized from what
 * the previous batches needed: launching the list's full capacity costs more in dispatch than the contenders cost to score). */
__global__ void __launch_bounds__(256) k_worklist(MoveBuf mb, const unsigned* __restrict__ cont, int w_begin, int grid_cap)
{
    __shared__ int s_base[8], s_cnt[8], s_fit;
    const int w = w_begin + blockIdx.x, tid = threadIdx.x;
    if (KEPT(w)) return;
    /* the items of the slots before this one, per sub-list (k_contend counted them) */
    if (tid < 8) {
        int base = 0;
        for (int ws = w_begin; ws < w; ws++) base += mb.slot_items[ws * 8 + tid];
        s_base[tid] = base;
        s_cnt[tid] = mb.slot_items[w * 8 + tid];
    }
    __syncthreads();
    if (tid == 0) {
        bool fit = true;
        for (int x = 0; x < 8; x++) fit &= 8LL * (s_base[x] + s_cnt[x]) <= (long long)grid_cap;
        s_fit = fit;
        /* 2: the exact kernel's grid (1: the slice pool, k_offsets).  NEVER over a 1: the grid is dealt out again whenever the
         * parameter half of a slot is re-scored (k_rescore_prepare takes a 2 back), the pool is not -- a slot without lists whose 1
         * had become a 2 here (every slot behind the first one that does not fit the grid is flagged) came back from the next
         * re-scoring as "fits", with empty lists and sums of zero, and was decided from them (found by tools/fuzz_chains.py in
         * round 5: a nuisance run on a small pool AND a small grid) */
        if (!fit && mb.ctl[PS(w)].overflow == 0) mb.ctl[PS(w)].overflow = 2;
        for (int x = 0; x < 8; x++) {
            if (fit) atomicMax(&mb.work[x], (unsigned long long)(s_base[x] + s_cnt[x])); /* the sub-lists end behind the last slot that fits */
            atomicMax(&mb.work[8 + x], (unsigned long long)(s_base[x] + s_cnt[x])); /* what the grid would have to cover (the host sizes the next one) */
        }
    }
    if (tid < 8) s_cnt[tid] = 0; /* now: the slot's append cursors */
    __syncthreads();
    if (!s_fit) return;
    const int C = mb.ctl[PS(w)].C;
    const long long ch = mb.ctl[PS(w)].exact_chunk;
    for (int u = tid; u < C * SLICE_SEG; u += blockDim.x) {
        const int c = u / SLICE_SEG, seg = u % SLICE_SEG;
        const int cw = CW(w, c);
        const long long n_seg = mb.part[(size_t)cw * P_STRIDE + P_CNT + seg];
        const int nch = (int)((n_seg + ch - 1) / ch);
        const unsigned mask = cont[cw] & ~mb.ident[cw];
        if (!nch || !mask) continue;
        const int x = seg & 7;
        int j = s_base[x] + atomicAdd(&s_cnt[x], nch * __popc(mask));
        for (int q = 0; q < nch; q++) {
            unsigned m = mask;
            while (m) {
                const int k = __ffs(m) - 1;
                m &= m - 1;
                mb.work[16 + 8 * (size_t)(j++) + x] =
                    ((unsigned long long)q << 32) | ((unsigned long long)cw << 12) | ((unsigned long long)k << 4) | (unsigned)seg;
            }
        }
    }
}

/* IG_SCREEN_VERIFY=1: every column was scored exactly as well; check the bound column by column */
__global__ void k_screen_verify(Glob* g, MoveBuf mb, const ScreenSum* __restrict__ scr, const unsigned* __restrict__ scr_void,
                                const unsigned* __restrict__ scr_ub, int w_begin, double* worst)
{
    const int w = w_begin + blockIdx.x;
    if (KEPT(w)) return;
    const MoveCtl& mc = mb.ctl[PS(w)];
    for (int i = threadIdx.x; i < mc.C * NSLOT; i += blockDim.x) {
        const int c = i / NSLOT, k = i % NSLOT;
        const int cw = CW(w, c);
        if (k == 0 || k > mb.meta[cw].n_uniq) continue;
        const unsigned vd = scr_void[cw];
        if (((vd >> k) & 1u) || (vd & 1u) || ((mb.ident[cw] >> k) & 1u)) continue;
        const long long* part = mb.part + (size_t)cw * P_STRIDE;
        const double exact = ig_acc_to_double(part[P_NZ + 2 * k] - part[P_NZ], part[P_NZ + 2 * k + 1] - part[P_NZ + 1]);
        const ScreenSum a = scr[cw * NSLOT + k], b = scr[cw * NSLOT];
        const double D = (double)(a.s_fix - b.s_fix) * (1.0 / SCR_FIX);
        const double Bd = (double)(a.b_fix + b.b_fix) * (1.0 / SCR_FIX);
        if ((scr_ub[cw] >> k) & 1u) { /* upper bound only */
            if (!(exact <= D + Bd)) g->error = 7;
            continue;
        }
        const double err = __builtin_fabs(D - exact);
        if (!(err <= Bd)) g->error = 7;
        if (worst && Bd > 0) { /* largest used fraction of a bound, largest bound (diagnostics; races are harmless) */
            if (err / Bd > worst[0]) worst[0] = err / Bd;
            if (Bd > worst[1]) worst[1] = Bd;
        }
    }
}

/* The two hardware functions the bound leans on, over their whole domain: v_log_f32 on every positive normal float,
 * v_exp_f32 on every float in [-150, 128), against the contract's double functions.
 * out[0] = max |L - log2 s| / (2u (|L| + 1)), out[1] = max |E - 2^y| / (2u 2^y) over normal results (2u = 2^-23). */
__global__ void k_transcendental_error(double* out)
{
    __shared__ double red[2][4];
    const double* T = ig_tab();
    double e_log = 0.0, e_exp = 0.0;
    const unsigned long long tid = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x, nth = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = 0x00800000ull + tid; b < 0x7f800000ull; b += nth) {
        const float s = __uint_as_float((unsigned)b);
        const double L = (double)__builtin_amdgcn_logf(s);
        const double ref = ig_log2_pos((double)s, T);
        e_log = __builtin_fmax(e_log, __builtin_fabs(L - ref) / (0x1p-23 * (__builtin_fabs(L) + 1.0)));
    }
    /* y in [-126, 128): normal results; both signs of the float encoding */
    for (unsigned long long b = tid; b < 0x43000000ull; b += nth) { /* +0 .. 128 */
        const float y = __uint_as_float((unsigned)b);
        const double E = (double)__builtin_amdgcn_exp2f(y), ref = ig_exp2((double)y, T);
        e_exp = __builtin_fmax(e_exp, __builtin_fabs(E - ref) / (0x1p-23 * ref));
    }
    for (unsigned long long b = 0x80000000ull + tid; b <= 0xc2fc0000ull; b += nth) { /* -0 .. -126 */
        const float y = __uint_as_float((unsigned)b);
        const double E = (double)__builtin_amdgcn_exp2f(y), ref = ig_exp2((double)y, T);
        e_exp = __builtin_fmax(e_exp, __builtin_fabs(E - ref) / (0x1p-23 * ref));
    }
    for (int o = 32; o > 0; o >>= 1) {
        e_log = __builtin_fmax(e_log, __shfl_down(e_log, o, 64));
        e_exp = __builtin_fmax(e_exp, __shfl_down(e_exp, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = e_log;
        red[1][threadIdx.x >> 6] = e_exp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int v = 1; v < (int)(blockDim.x >> 6); v++) {
            red[0][0] = __builtin_fmax(red[0][0], red[0][v]);
            red[1][0] = __builtin_fmax(red[1][0], red[1][v]);
        }
        /* doubles >= 0: the integer order of their bit patterns is their order */
        atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(red[0][0]));
        atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(red[1][0]));
    }
}
