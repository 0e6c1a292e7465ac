/* ig_model.cuh -- device helpers: wave reductions, the per-contact term (include/ig_detmath.h) with its lookup
 * tables, get_bounds and the slice predicate. */
#pragma once

/* ------------------------------------------------------------------ device helpers */

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ void atomic_add_ll(long long* p, long long v)
{
    atomicAdd((unsigned long long*)p, (unsigned long long)v);
}

__device__ __noinline__ double lgfact_big(int ob);
__device__ __forceinline__ double lgfact_dev(int ob, const double* __restrict__ lgf_tab)
{
    if (ob <= 0) return 0.0;
    if (ob < LGF_TAB) return lgf_tab[ob];
    return lgfact_big(ob);
}

/* quantised likelihood term of one contact under one coordinate column */
/* P(s) of the zero-pixel companion term only ever sees s_z = d * mean_kb for an INTEGER rank distance d
 * (KA:4324-4334): pz[d] holds exactly the value the direct evaluation would produce (built by k_build_pz with
 * the same functions), for d < PZ_MAX; beyond the table s_z >= d_max by construction, i.e. v_inter. */
#define PZ_MAX 4096
struct PzTab {
    const float* v;
    int n;
};

/* rare paths, kept out of line so that the hot loop stays small (I-cache) */
__device__ __noinline__ float pz_direct(const ig_params p, float mean_kb, int d)
{
    const float s_z = (float)d * mean_kb;
    return (s_z < p.d_max) ? ig_rippe(s_z, p, ig_tab()) : p.v_inter;
}
__device__ __noinline__ void expected_circ(const ig_params p, float mean_kb, float s, float s_tot, int d, int len_j, float* ex,
                                           float* ex_z)
{
    const float s_z = (float)d * mean_kb;
    *ex = ig_rippe_circ(s, s_tot, p, ig_tab());
    *ex_z = (s_z < p.d_max) ? ig_rippe_circ(s_z, (float)len_j * mean_kb, p, ig_tab()) : p.v_inter;
}
__device__ __noinline__ double lgfact_big(int ob)
{
    const double* T = ig_tab();
    double o = (double)ob;
    return (o * ig_log10(o, T) - o) + 0.5 * ig_log10(o * 2.0 * 3.14159265358979323846, T);
}

__device__ __forceinline__ float pz_lookup(const PzTab& t, const ig_params& p, float mean_kb, int d)
{
    if (d < t.n) return t.v[d];
    return pz_direct(p, mean_kb, d);
}

/* T: the log2/exp2 table of ig_detmath.h (a kernel passes its LDS copy, everything else ig_tab()) */
__device__ __forceinline__ long long eval_q(const ig_params& p, const ig_hot& h, float mean_kb, uint2 a, uint2 b,
                                            const ColMeta* __restrict__ cm, int ob, double lgf, const PzTab& pz, const double* T)
{
    const float di = __uint_as_float(a.x), dj = __uint_as_float(b.x);
    const int pi = (int)(a.y & 0x0fffffffu), pj = (int)(b.y & 0x0fffffffu);
    const int ci = (int)(a.y >> 28), cj = (int)(b.y >> 28);
    float ex, ex_z;
    if (ci == cj) {
        const float s = fabsf(di - dj);
        const float s_tot = cm[ci].stot;
        const int d = pi > pj ? pi - pj : pj - pi;
        if (s_tot == 0) {
            ex_z = pz_lookup(pz, p, mean_kb, d);
            if (h.fast && ob > 0) return ig_quantize(ig_term_hot(s, 0, ob, lgf, ex_z, &h, T));
            ex = ig_rippe(s, p, T);
        } else {
            expected_circ(p, mean_kb, s, s_tot, d, cm[cj].len, &ex, &ex_z);
        }
    } else {
        ex = p.v_inter;
        ex_z = p.v_inter;
        if (h.fast && ob > 0) return ig_quantize(ig_term_hot(0.0f, 1, ob, lgf, ex_z, &h, T));
    }
    return ig_quantize(ig_pixel_term(ex, ex_z, ob, lgf, T));
}

__global__ void k_build_pz(const Glob* g, float* pz, int n, int which)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    const ig_params p = g->par[which];
    const float s_z = (float)d * g->mean_kb;
    pz[d] = (s_z < p.d_max) ? ig_rippe(s_z, p, ig_tab()) : p.v_inter;
}

/* one sub-fragment's zero-pixel term: KA:3882-3899 */
__device__ __forceinline__ long long zero_q(const ig_params& p, int pos, int len_cont, float s_tot, float mean_kb,
                                            const float* __restrict__ pz, int pz_n)
{
    const float s = (float)pos * mean_kb;
    const float s_tot_z = (float)len_cont * mean_kb;
    double ve;
    if (s < p.d_max) {
        if (s_tot == 0) ve = (double)((pz && pos < pz_n) ? pz[pos] : ig_rippe(s, p, ig_tab()));
        else ve = (double)ig_rippe_circ(s, s_tot_z, p, ig_tab());
    } else {
        ve = (double)p.v_inter;
    }
    return ig_quantize(0.0 - (ve * (double)(len_cont - pos)));
}

/* cut positions + validity flags of get_bounds (KA:2124-2252), scalar part */
__device__ inline void bounds_scalar(const State& st, const Glob* g, int P, int I, int* pos_up, int* pos_down, int* valid)
{
    const int cP = st.cid[P], cI = st.cid[I];
    const int same = (cP == cI);
    const int pP = st.pos[P], pI = st.pos[I];
    const int LP = st.L[P], LI = st.L[I];
    const int ins_is_ext = (pI == 0) || (pI == (LI - 1));
    const int nb = IG_N_INSERT_BLOCKS;
    for (int i = 0; i < nb; i++) {
        int up, down;
        if (i == 0) {
            if (same && (pI < pP - 1)) {
                up = pI + 1;
                down = pP;
            } else if (same && (pI > pP + 1)) {
                down = pI - 1;
                up = pP;
            } else {
                up = pP;
                down = pP;
            }
        } else if (i < nb - 1) {
            up = max(0, pP - g->list_bounds[i - 1]);
            down = min(LP - 1, pP + g->list_bounds[i - 1]);
        } else {
            up = 0;
            down = LP - 1;
        }
        if (same && (pI <= pP) && (pI >= up)) {
            pos_up[i] = -1;
            valid[i * 2] = -1;
        } else {
            pos_up[i] = up;
            valid[i * 2] = 1;
            if (up == 0 && (((pP - up) == 1) || ins_is_ext)) {
                valid[i * 2] = -1;
                pos_up[i] = -1;
            }
        }
        if (same && (((pI >= pP) && (pI <= down)) || (pI == (pP - 1)))) {
            pos_down[i] = -1;
            valid[i * 2 + 1] = -1;
        } else {
            pos_down[i] = down;
            valid[i * 2 + 1] = 1;
            if (down == LP - 1 && (((down - pP) == 1) || ins_is_ext)) {
                valid[i * 2 + 1] = -1;
                pos_down[i] = -1;
            }
        }
    }
}

/* slice predicate of slice_sp_mat (KA:557-593) for a contact whose row lies in a touched contig */
__device__ __forceinline__ bool slice_keep(const CandMeta& m, int c1, int c2, int p1, int p2, int ob, bool unwindowed)
{
    if (ob <= 0) return false;
    if ((c2 == c1) && m.same && m.windowed && !unwindowed) {
        const int px = min(p1, p2), py = max(p1, p2);
        const bool ca = (px <= m.down_fa) && (py >= m.up_fa);
        const bool cb = (py >= m.up_fb) && (px <= m.down_fb);
        return ca || cb;
    }
    if (m.same) return c2 == m.ctgB; /* ctgA == ctgB */
    return (c2 == m.ctgA) || (c2 == m.ctgB);
}

/* S_c: kept contacts of a candidate's slice = sum of its segment cursors */
__host__ __device__ inline long long slice_total(const long long* part)
{
    long long n = 0;
    for (int s = 0; s < SLICE_SEG; s++) n += part[P_CNT + s];
    return n;
}
