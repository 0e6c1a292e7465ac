/*
 * ig_oracle_lik.c -- coordinates, slice and likelihood kernels of the CPU oracle
 * (TEST INFRASTRUCTURE ONLY; see ig_oracle.h).
 *
 * Follows /root/reference/src/instagraal/kernels/kernel_sparse_adapt.cu ("KA"):
 *   fill_vect_dist / uni_fill_vect_dist        KA:3699-3822
 *   slice_sp_mat                                KA:485-607
 *   prepare_sparse_call                         KA:4048-4097
 *   eval_likelihood_on_zero (+ _1st, _2nd)      KA:3850-4027
 *   eval_all_scores                             KA:4029-4046
 *   extract_sub_likelihood / eval_sub_likelihood KA:4099-4370
 *   evaluate_likelihood_sparse                  KA:4374-4488
 * Launch geometry is the one issued by cuda_lib_gl_single.py ("CL") wherever it
 * changes a result: block 64 for the slice kernels (CL:200, quirk Q5), 1024 for
 * the M- and Z-length reductions (CL:726, 856, 1204).
 *
 * Repeats are dead in the reference (simu_single.py:513 forces candidates_dup=[]),
 * so the dispatcher/collector tables are the identity (x=i, y=i+1) and are not
 * passed here.
 */
#include "ig_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_mode = IG_MODE_LIBM;
static int g_threads = 1;
static int64_t g_limb_hi[IGO_N_TMP_STRUCT + 1], g_limb_lo[IGO_N_TMP_STRUCT + 1];
static double g_lgf_det[15];
static int g_lgf_ready = 0;

void igo_set_mode(int mode) { g_mode = mode; }
int igo_get_mode(void) { return g_mode; }
void igo_set_threads(int n) { g_threads = n > 0 ? n : 1; }
void igo_last_limbs(int64_t* hi, int64_t* lo, int n)
{
    for (int i = 0; i < n && i <= IGO_N_TMP_STRUCT; i++) {
        hi[i] = g_limb_hi[i];
        lo[i] = g_limb_lo[i];
    }
}

/* ------------------------------------------------------------------ model */

/* KA:111-124 with libm */
static float factorial_libm(float n)
{
    float result = 1;
    n = floorf(n);
    if (n < 10) {
        for (int c = 1; c <= n; c++) result = result * c;
    } else {
        result = powf(n, n) * expf(-n) * sqrtf(2 * M_PI * n);
    }
    return result;
}
/* KA:111-124 with the deterministic functions (host-side table only) */
static float factorial_det(float n)
{
    float result = 1;
    if (n < 10) {
        for (int c = 1; c <= n; c++) result = result * c;
    } else {
        result = ig_powf(n, n, ig_tab()) * ig_expf(-n, ig_tab()) * sqrtf((float)(2 * M_PI * n));
    }
    return result;
}
void igo_lgf_table(double* out15)
{
    for (int k = 0; k < 15; k++) out15[k] = ig_log10((double)factorial_det((float)k), ig_tab());
}
static void lgf_init(void)
{
    if (!g_lgf_ready) {
        igo_lgf_table(g_lgf_det);
        g_lgf_ready = 1;
    }
}

/* KA:153-163 with libm */
static float rippe_libm(float s, const ig_params p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max)) {
        result = (p.c1 * powf(s, p.slope) * expf((p.d - 2) / (powf(s * p.lm / p.kuhn, 2.0f) + p.d))) * p.fact;
    }
    return fmaxf(result, p.v_inter);
}
/* KA:200-225 with libm */
static float rippe_circ_libm(float s, float s_tot, const ig_params p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max)) {
        float K = p.lm / p.kuhn;
        float n = K * s * (s_tot - s) / s_tot;
        result = (powf(p.kuhn, -3.0f) * powf(n, p.slope) * expf((p.d - 2.0f) / (powf(n, 2.0f) + p.d))) * p.fact;
    }
    return fmaxf(result, p.d_max);
}
/* KA:251-270 with libm */
static double pxl_libm(double ex, double ob)
{
    double res = 0;
    double lim = 15;
    if (ex != 0) {
        if (ob >= lim) {
            res = ob * log10(ex) - ex - (ob * log10(ob) - ob + log10(sqrt(ob * 2.0 * M_PI)));
        } else if ((ob > 0) && (ob < lim)) {
            res = ob * log10(ex) - ex - log10((double)factorial_libm((float)ob));
        } else if (ob == 0) {
            res = -ex;
        }
    }
    return res;
}

static inline float m_rippe(float s, const ig_params p) { return g_mode ? ig_rippe(s, p, ig_tab()) : rippe_libm(s, p); }
static inline float m_rippe_circ(float s, float st, const ig_params p)
{
    return g_mode ? ig_rippe_circ(s, st, p, ig_tab()) : rippe_circ_libm(s, st, p);
}

/* expected contacts of one (i,j) pair: KA:4430-4459 (same text at 4182-4206, 4327-4351) */
static inline void expected_pair(const ig_params p, int cis, float s, float s_z, float s_tot, float s_tot_z, float* ex,
                                 float* ex_z)
{
    if (cis) {
        if (s_tot == 0) {
            *ex = m_rippe(s, p);
            *ex_z = (s_z < p.d_max) ? m_rippe(s_z, p) : p.v_inter;
        } else {
            *ex = m_rippe_circ(s, s_tot, p);
            *ex_z = (s_z < p.d_max) ? m_rippe_circ(s_z, s_tot_z, p) : p.v_inter;
        }
    } else {
        *ex = p.v_inter;
        *ex_z = p.v_inter;
    }
}

/* KA:4462: pxl(ex, ob) + ex_z * 0.43429448190325182f */
static inline double pixel_term(float ex, float ex_z, int ob)
{
    if (g_mode) return ig_pixel_term(ex, ex_z, ob, ob > 0 ? ig_lgfact(ob, g_lgf_det, ig_tab()) : 0.0, ig_tab());
    return pxl_libm((double)ex, (double)ob) + (double)ex_z * 0.43429448190325182f;
}

/* the term of one contact, KA:4430-4462: LIBM mode composes the reference's functions literally; DET mode is the
 * arithmetic contract's definition (include/ig_detmath.h ig_pair_term: one log2 + one exp2 for the linear-contig case) */
static inline double pair_term(const ig_params p, const ig_hot* h, int cis, float s, float s_z, float s_tot, float s_tot_z, int ob)
{
    if (g_mode) return ig_pair_term(p, h, cis, s, s_z, s_tot, s_tot_z, ob, ob > 0 ? ig_lgfact(ob, g_lgf_det, ig_tab()) : 0.0, ig_tab());
    float ex, ex_z;
    expected_pair(p, cis, s, s_z, s_tot, s_tot_z, &ex, &ex_z);
    return pixel_term(ex, ex_z, ob);
}

/* one sub-fragment's zero-pixel contribution: KA:3882-3899 / 3955-3972 */
static inline double zero_term(const ig_params p, int pos, int len_cont, float s_tot, float mean_size_frag)
{
    float s = (float)pos * mean_size_frag;
    float s_tot_z = (float)len_cont * mean_size_frag;
    double val_expected;
    if (s < p.d_max) {
        if (s_tot == 0) val_expected = (double)m_rippe(s, p);
        else val_expected = (double)m_rippe_circ(s, s_tot_z, p);
    } else {
        val_expected = (double)p.v_inter;
    }
    double n_tmp_vals = (double)(len_cont - pos);
    return 0.0 - (val_expected * n_tmp_vals);
}

/* len*(len-1)/2 in the reference's int32 arithmetic (KA:3879-3880, 3950-3953) */
static inline int32_t half_pairs_i32(int32_t len)
{
    int32_t t = (int32_t)((uint32_t)len * (uint32_t)(len - 1));
    return t / 2;
}

void igo_eval_terms(const float* s, const float* s_tot, const int32_t* ob, int64_t n, const ig_params* P, float* ex,
                    float* ex_circ, double* term, int64_t* q)
{
    lgf_init();
    const ig_hot hot = ig_hot_make(*P, ig_tab());
    for (int64_t i = 0; i < n; i++) {
        ex[i] = m_rippe(s[i], *P);
        ex_circ[i] = m_rippe_circ(s[i], s_tot[i], *P);
        /* DET mode: the contract's term with P_z := ex_circ (any float will do for a bit-parity probe) */
        if (g_mode && hot.fast && ob[i] > 0) term[i] = ig_term_hot(s[i], 0, ob[i], ig_lgfact(ob[i], g_lgf_det, ig_tab()), ex_circ[i], &hot, ig_tab());
        else term[i] = pixel_term(ex[i], ex_circ[i], ob[i]);
        q[i] = ig_quantize(term[i]);
    }
}

/* ------------------------------------------------- reductions (two shapes) */

/* the in-block tree of KA:3903-3913, 4218-4228, 4473-4483 */
static double tree_reduce(double* sdata, int block)
{
    for (int offset = block / 2; offset > 0; offset >>= 1)
        for (int t = 0; t < offset; t++) sdata[t] += sdata[t + offset];
    return sdata[0];
}

static void publish(int k, ig_acc a)
{
    ig_acc_normalize(&a.hi, &a.lo);
    g_limb_hi[k] = a.hi;
    g_limb_lo[k] = a.lo;
}

/* ------------------------------------------------------------ coordinates */

/* KA:3699-3760.  s_tot goes through an int local (quirk Q7). */
void igo_fill_vect_dist(const ig_float4* sub2frag, const frag* f, float* dist, int32_t* id_c, float* s_tot, int32_t* pos,
                        int32_t* len, int n_sub_frags, int id_mut)
{
    for (int k = 0; k < n_sub_frags; k++) {
        ig_float4 info = sub2frag[k];
        int fi = (int)info.x;
        int sub_pos_i = (int)info.w;
        int or_fi = f->ori[fi];
        int pos_i = f->sub_pos[fi];
        int sub_len = f->sub_len[fi];
        int s_tot_i = (int)((float)f->circ[fi] * (float)f->l_cont_bp[fi] / 1000.0f);
        float fi_start_bp = (float)f->start_bp[fi];
        float dfi;
        int frag_sub_pos;
        if (or_fi == 1) {
            dfi = info.y;
            frag_sub_pos = pos_i + sub_pos_i;
        } else {
            dfi = info.z;
            frag_sub_pos = pos_i + sub_len - (sub_pos_i + 1);
        }
        int64_t o = (int64_t)k * IGO_N_TMP_STRUCT + id_mut;
        dist[o] = fi_start_bp / 1000.0f + dfi;
        id_c[o] = f->id_c[fi];
        s_tot[o] = (float)s_tot_i;
        pos[o] = frag_sub_pos;
        len[o] = f->sub_l_cont[fi];
    }
}

/* KA:3763-3822 */
void igo_uni_fill_vect_dist(const ig_float4* sub2frag, const frag* f, float* dist, int32_t* id_c, float* s_tot, int32_t* pos,
                            int32_t* len, int n_sub_frags)
{
    for (int k = 0; k < n_sub_frags; k++) {
        ig_float4 info = sub2frag[k];
        int fi = (int)info.x;
        int sub_pos_i = (int)info.w;
        int or_fi = f->ori[fi];
        int pos_i = f->sub_pos[fi];
        int sub_len = f->sub_len[fi] - 1;
        int is_circle = f->circ[fi] == 1;
        int s_tot_i = (int)((float)is_circle * (float)f->l_cont_bp[fi] / 1000.0f);
        float fi_start_bp = (float)f->start_bp[fi];
        float dfi;
        int frag_sub_pos;
        if (or_fi == 1) {
            dfi = info.y;
            frag_sub_pos = pos_i + sub_pos_i;
        } else {
            dfi = info.z;
            frag_sub_pos = pos_i + sub_len - sub_pos_i;
        }
        dist[k] = fi_start_bp / 1000.0f + dfi;
        id_c[k] = f->id_c[fi];
        s_tot[k] = (float)s_tot_i;
        pos[k] = frag_sub_pos;
        len[k] = f->sub_l_cont[fi];
    }
}

/* ------------------------------------------------------------------ slice */

/* KA:485-607.  Canonical compaction order = input (COO) order; entries with
 * dat <= 0 occupy a slot that is never written (KA:602) -- reproduced. */
void igo_slice_sp_mat(const int32_t* dat, const int32_t* row, const int32_t* col, const frag* f, const int32_t* vect_id_c,
                      const int32_t* vect_pos, int32_t* sub_row, int32_t* sub_col, int32_t* sub_dat, int id_ctg1, int id_ctg2,
                      int id_frag_a, int id_frag_b, int n_bounds, int32_t* counter, int64_t size_arr)
{
    const int same_contigs = id_ctg1 == id_ctg2;
    const int tmp_pos_fa = f->sub_pos[id_frag_a], tmp_pos_fb = f->sub_pos[id_frag_b];
    const int ori_fa = f->ori[id_frag_a], ori_fb = f->ori[id_frag_b];
    const int sub_len_fa = f->sub_len[id_frag_a], sub_len_fb = f->sub_len[id_frag_b];
    int pos_fa = tmp_pos_fa * (ori_fa == 1) + (tmp_pos_fa - sub_len_fa) * (ori_fa == -1);
    int pos_fb = tmp_pos_fb * (ori_fb == 1) + (tmp_pos_fb - sub_len_fb) * (ori_fb == -1);
    if (pos_fa < 0) pos_fa = 0;
    if (pos_fb < 0) pos_fb = 0;
    const int is_circ = f->circ[id_frag_a];
    const int l_ctg_fa = f->sub_l_cont[id_frag_a], l_ctg_fb = f->sub_l_cont[id_frag_b];
    int up_bound_fa = pos_fa - n_bounds - sub_len_fa;
    if (up_bound_fa < 0) up_bound_fa = 0;
    int down_bound_fa = pos_fa + n_bounds + sub_len_fa;
    if (down_bound_fa > l_ctg_fa - 1) down_bound_fa = l_ctg_fa - 1;
    int up_bound_fb = pos_fb - sub_len_fb;
    if (up_bound_fb < 0) up_bound_fb = 0;
    int down_bound_fb = pos_fb + sub_len_fb;
    if (down_bound_fb > l_ctg_fb - 1) down_bound_fb = l_ctg_fb - 1;

#define IGO_SLICE_KEEP(k, keep)                                                                                  \
    do {                                                                                                          \
        int fi_ = row[k], fj_ = col[k];                                                                           \
        int c1_ = vect_id_c[fi_];                                                                                 \
        (keep) = 0;                                                                                               \
        if ((c1_ == id_ctg1) || (c1_ == id_ctg2)) {                                                               \
            int c2_ = vect_id_c[fj_];                                                                             \
            if ((c2_ == c1_) && (same_contigs == 1) && (is_circ == 0)) { /* KA:565-586 */                         \
                int pos_fi = vect_pos[fi_], pos_fj = vect_pos[fj_];                                               \
                int pos_x = pos_fi < pos_fj ? pos_fi : pos_fj;                                                    \
                int pos_y = pos_fi < pos_fj ? pos_fj : pos_fi;                                                    \
                int c_a = (pos_x <= down_bound_fa) && (pos_y >= up_bound_fa);                                     \
                int c_b = (pos_y >= up_bound_fb) && (pos_x <= down_bound_fb);                                     \
                (keep) = c_a || c_b;                                                                              \
            } else if (((same_contigs == 0) && (c2_ == id_ctg1)) || (c2_ == id_ctg2)) { /* KA:587, precedence Q10 */ \
                (keep) = 1;                                                                                       \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

    int64_t n = counter[0];
#ifdef _OPENMP
    if (g_threads > 1 && size_arr > (1 << 16)) {
        /* the same compaction in the same (COO) order on g_threads cores: count per chunk, prefix, fill */
        const int T = g_threads;
        int64_t* cnt = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));
        const int64_t per = (size_arr + T - 1) / T;
#pragma omp parallel num_threads(T)
        {
            const int t = omp_get_thread_num();
            const int64_t b = (int64_t)t * per, e = (b + per < size_arr) ? b + per : size_arr;
            int64_t c = 0;
            for (int64_t k = b; k < e; k++) {
                int keep;
                IGO_SLICE_KEEP(k, keep);
                c += keep;
            }
            cnt[t + 1] = c;
#pragma omp barrier
#pragma omp single
            {
                cnt[0] = n;
                for (int q = 0; q < T; q++) cnt[q + 1] += cnt[q];
            }
            int64_t at = cnt[t];
            for (int64_t k = b; k < e; k++) {
                int keep;
                IGO_SLICE_KEEP(k, keep);
                if (keep) {
                    if (dat[k] > 0) {
                        sub_dat[at] = dat[k];
                        sub_row[at] = row[k];
                        sub_col[at] = col[k];
                    }
                    at++;
                }
            }
        }
        counter[0] = (int32_t)cnt[T];
        free(cnt);
        return;
    }
#endif
    for (int64_t k = 0; k < size_arr; k++) {
        int keep;
        IGO_SLICE_KEEP(k, keep);
        if (keep) {
            if (dat[k] > 0) {
                sub_dat[n] = dat[k];
                sub_row[n] = row[k];
                sub_col[n] = col[k];
            }
            n++;
        }
    }
    counter[0] = (int32_t)n;
}

/* KA:4048-4097, block = 64 (CL:1011) */
void igo_prepare_sparse_call(const int32_t* row, ig_int3* info_block, int32_t* block_csr, int32_t* counter, int size_arr)
{
    const int B = IGO_SIZE_BLOCK_4_SUB;
    int n_blocks = size_arr / B + 1; /* CL:1052 */
    int glob = counter[0];
    for (int b = 0; b < n_blocks; b++) {
        int local = 0;
        for (int t = 0; t < B; t++) {
            int idx = b * B + t;
            if (idx >= size_arr) continue;
            int id_next = idx + 1 < size_arr - 1 ? idx + 1 : size_arr - 1;
            int curr = row[idx], next = row[id_next];
            if ((curr != next) || (t == 0) || (t == B - 1) || (idx == size_arr - 1)) {
                block_csr[glob + local] = curr;
                local++;
            }
        }
        info_block[b].x = local;
        info_block[b].y = glob;
        info_block[b].z = b * B;
        glob += local;
    }
    counter[0] = glob;
}

/* ------------------------------------------------------------ zero pixels */

/* KA:3850-3917, block 1024 */
void igo_eval_likelihood_on_zero(const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                 const ig_params* P, float mean_size_frag, double* vect_likelihood, int32_t* n_vals_intra,
                                 int n_frags)
{
    (void)id_c;
    const ig_params p = *P;
    const int B = 1024;
    ig_acc acc = {0, 0};
    double total = vect_likelihood[0];
    double* sdata = (double*)malloc(sizeof(double) * B);
    for (int b0 = 0; b0 < n_frags; b0 += B) {
        for (int t = 0; t < B; t++) {
            int k = b0 + t;
            double v = 0.0;
            if (k < n_frags) {
                if (pos[k] == 0) n_vals_intra[0] += half_pairs_i32(len[k]);
                if (pos[k] > 0) {
                    v = zero_term(p, pos[k], len[k], s_tot[k], mean_size_frag);
                    if (g_mode) ig_acc_add(&acc, ig_quantize(v));
                }
            }
            sdata[t] = v;
        }
        total += tree_reduce(sdata, B);
    }
    free(sdata);
    if (g_mode) {
        publish(0, acc);
        vect_likelihood[0] += ig_acc_to_double(acc.hi, acc.lo);
    } else {
        vect_likelihood[0] = total;
    }
}

/* KA:3919-4002, block 1024, one sub-frag per thread (CL:856-880) */
void igo_eval_all_likelihood_on_zero_1st(const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                         const ig_params* P, float mean_size_frag, const int32_t* list_uniq,
                                         const int32_t* n_uniq, double* vect_likelihood, int32_t* n_vals_intra, int n_frags)
{
    (void)id_c;
    const ig_params p = *P;
    const int B = 1024, T = IGO_N_TMP_STRUCT;
    const int nu = n_uniq[0];
    double* sdata = (double*)malloc(sizeof(double) * B);
    for (int k = 0; k < nu; k++) {
        const int m = list_uniq[k];
        ig_acc acc = {0, 0};
        double total = vect_likelihood[m];
        for (int b0 = 0; b0 < n_frags; b0 += B) {
            for (int t = 0; t < B; t++) {
                int s = b0 + t;
                double v = 0.0;
                if (s < n_frags) {
                    int64_t o = (int64_t)s * T + m;
                    if (pos[o] == 0) n_vals_intra[m] += half_pairs_i32(len[o]);
                    if (pos[o] > 0) {
                        v = zero_term(p, pos[o], len[o], s_tot[o], mean_size_frag);
                        if (g_mode) ig_acc_add(&acc, ig_quantize(v));
                    }
                }
                sdata[t] = v;
            }
            total += tree_reduce(sdata, B);
        }
        if (g_mode) {
            publish(m, acc);
            vect_likelihood[m] += ig_acc_to_double(acc.hi, acc.lo);
        } else {
            vect_likelihood[m] = total;
        }
    }
    free(sdata);
}

/* KA:4005-4027 */
void igo_eval_all_likelihood_on_zero_2nd(const int32_t* list_uniq, const int32_t* n_uniq, const ig_params* P,
                                         double* vect_likelihood, const int32_t* n_vals_intra, const double* n_tot_pxl)
{
    const ig_params p = *P;
    const double log_e = 0.43429448190325182f;
    for (int t = 0; t < n_uniq[0]; t++) {
        int m = list_uniq[t];
        double intra_vals = (double)n_vals_intra[m];
        double val_inter = -1.0 * log_e * (n_tot_pxl[0] - intra_vals) * p.v_inter;
        double val_intra = vect_likelihood[m] * log_e;
        vect_likelihood[m] = val_intra + val_inter;
    }
}

/* KA:4029-4046 */
void igo_eval_all_scores(const int32_t* list_uniq, const int32_t* n_uniq, const double* z, const double* nz,
                         const double* extract, const double* curr_nz, double* all_score)
{
    for (int t = 0; t < n_uniq[0]; t++) {
        int m = list_uniq[t];
        all_score[m] = nz[m] + z[m] + curr_nz[0] - extract[0];
    }
}

/* ------------------------------------------------------------ slice sums */

/* the row-side cache of KA:4144-4162 / 4281-4302: the LAST matching entry of the
 * block's row list wins; entry 0 if none matches */
static inline int find_row_slot(const int32_t* block_csr, ig_int3 pb, int curr_fi)
{
    int local = 0;
    for (int i = 0; i < pb.x; i++)
        if (curr_fi == block_csr[pb.y + i]) local = i;
    return local;
}

/* KA:4099-4233, block 64.  The blocks are independent (a CUDA block each): they are evaluated on g_threads cores
 * and their results taken in block order, i.e. summed exactly as the serial loop sums them. */
void igo_extract_sub_likelihood(const int32_t* dat, const ig_int3* info_block, const int32_t* block_csr, const int32_t* row,
                                const int32_t* col, const ig_params* P, float mean_size_frag, const float* pos_bp,
                                const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                double* vect_likelihood, int n_data, int n_sub_frags)
{
    (void)n_sub_frags;
    lgf_init();
    const ig_params p = *P;
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const int B = IGO_SIZE_BLOCK_4_SUB;
    const int n_blocks = n_data / B + 1;
    const int mode = g_mode;
    double* per_block = (double*)malloc(sizeof(double) * (size_t)n_blocks);
    int64_t hi = 0, lo = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : hi, lo) num_threads(g_threads) schedule(static)
#endif
    for (int b = 0; b < n_blocks; b++) {
        double sdata[IGO_SIZE_BLOCK_4_SUB];
        ig_int3 pb = info_block[b];
        for (int t = 0; t < B; t++) {
            int g = b * B + t;
            double v = 0.0;
            if (g < n_data) {
                int curr_fi = row[g];
                int slot = find_row_slot(block_csr, pb, curr_fi);
                int fi = block_csr[pb.y + slot]; /* row whose cached data are used */
                int fj = col[g];
                int contig_i = (int)floorf((float)id_c[fi]);
                int contig_j = id_c[fj];
                float st = s_tot[fi];
                float si = pos_bp[fi];
                float pos_i = (float)pos[curr_fi];
                float sj = pos_bp[fj];
                float pos_j = (float)pos[fj];
                float s = fabsf(si - sj);
                float s_z = fabsf(pos_i - pos_j) * mean_size_frag;
                float s_tot_z = (float)len[fj] * mean_size_frag;
                v = pair_term(p, &hot, contig_i == contig_j, s, s_z, st, s_tot_z, dat[g]);
                if (mode) {
                    int64_t q = ig_quantize(v);
                    hi += q >> 32;
                    lo += (int64_t)(uint32_t)q;
                }
            }
            sdata[t] = v;
        }
        per_block[b] = tree_reduce(sdata, B);
    }
    if (mode) {
        ig_acc acc = {hi, lo};
        publish(IGO_N_TMP_STRUCT, acc);
        vect_likelihood[0] += ig_acc_to_double(acc.hi, acc.lo);
    } else {
        double total = vect_likelihood[0];
        for (int b = 0; b < n_blocks; b++)
            if (b * B < n_data) total += per_block[b]; /* tid 0 && condition, KA:4230 */
        vect_likelihood[0] = total;
    }
    free(per_block);
}

/* KA:4236-4370, block 64.  Column k of the uniq list is summed by thread k of
 * each block and only if THAT thread holds a valid entry (KA:4362): in the last,
 * partially filled block columns with k >= n_data % 64 drop the block (quirk Q5).
 * Blocks on g_threads cores, their column sums added in block order (LIBM: the serial loop's double sums;
 * DET: integer limbs, any order). */
void igo_eval_sub_likelihood(const int32_t* dat, const ig_int3* info_block, const int32_t* block_csr, const int32_t* row,
                             const int32_t* col, const ig_params* P, float mean_size_frag, const float* pos_bp,
                             const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                             const int32_t* list_uniq, const int32_t* n_uniq, double* vect_likelihood, int n_data,
                             int n_sub_frags)
{
    (void)n_sub_frags;
    lgf_init();
    const ig_params p = *P;
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const int B = IGO_SIZE_BLOCK_4_SUB, T = IGO_N_TMP_STRUCT;
    const int n_blocks = n_data / B + 1;
    const int nu = n_uniq[0];
    const int mode = g_mode;
    ig_acc acc[IGO_N_TMP_STRUCT];
    memset(acc, 0, sizeof acc);
    double* per_block = mode ? NULL : (double*)malloc(sizeof(double) * (size_t)n_blocks * T);
#ifdef _OPENMP
#pragma omp parallel num_threads(g_threads)
#endif
    {
        double loc[IGO_SIZE_BLOCK_4_SUB * IGO_N_TMP_STRUCT];
        int64_t locq[IGO_SIZE_BLOCK_4_SUB * IGO_N_TMP_STRUCT];
        ig_acc mine[IGO_N_TMP_STRUCT];
        memset(mine, 0, sizeof mine);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int b = 0; b < n_blocks; b++) {
            ig_int3 pb = info_block[b];
            for (int t = 0; t < B; t++) {
                int g = b * B + t;
                if (g < n_data) {
                    int curr_fi = row[g];
                    int slot = find_row_slot(block_csr, pb, curr_fi);
                    int64_t fiT = (int64_t)block_csr[pb.y + slot] * T;
                    int curr_fj = col[g];
                    for (int k = 0; k < nu; k++) {
                        int m = list_uniq[k];
                        int64_t fj = (int64_t)curr_fj * T + m;
                        int contig_i = (int)floorf((float)id_c[fiT + m]);
                        int contig_j = id_c[fj];
                        float st = s_tot[fiT + m];
                        float si = pos_bp[fiT + m];
                        float pos_i = (float)pos[fiT + m];
                        float sj = pos_bp[fj];
                        float pos_j = (float)pos[fj];
                        float s = fabsf(si - sj);
                        float s_z = fabsf(pos_i - pos_j) * mean_size_frag;
                        float s_tot_z = (float)len[fj] * mean_size_frag;
                        double v = pair_term(p, &hot, contig_i == contig_j, s, s_z, st, s_tot_z, dat[g]);
                        loc[t * T + m] = v;
                        locq[t * T + m] = mode ? ig_quantize(v) : 0;
                    }
                } else {
                    for (int k = 0; k < nu; k++) {
                        int m = list_uniq[k];
                        loc[t * T + m] = 0.0;
                        locq[t * T + m] = 0;
                    }
                }
            }
            for (int t = 0; t < nu && t < B; t++) {
                int m = list_uniq[t];
                if (b * B + t >= n_data) { /* (tid < n_uniq) && condition */
                    if (!mode) per_block[(size_t)b * T + m] = 0.0;
                    continue;
                }
                double tmp = 0.0;
                for (int i = 0; i < B; i++) {
                    tmp += loc[i * T + m];
                    if (mode) ig_acc_add(&mine[m], locq[i * T + m]);
                }
                if (!mode) per_block[(size_t)b * T + m] = tmp;
            }
        }
        if (mode) {
#ifdef _OPENMP
#pragma omp critical
#endif
            for (int m = 0; m < T; m++) {
                acc[m].hi += mine[m].hi;
                acc[m].lo += mine[m].lo;
            }
        }
    }
    if (mode) {
        for (int k = 0; k < nu; k++) {
            int m = list_uniq[k];
            publish(m, acc[m]);
            vect_likelihood[m] += ig_acc_to_double(acc[m].hi, acc[m].lo);
        }
    } else {
        for (int b = 0; b < n_blocks; b++)
            for (int t = 0; t < nu && t < B; t++) {
                if (b * B + t >= n_data) continue;
                int m = list_uniq[t];
                vect_likelihood[m] += per_block[(size_t)b * T + m];
            }
        free(per_block);
    }
}

/* KA:4374-4488, block 1024, grid Z/1024+1 (CL:1204-1211): one pixel per thread */
void igo_evaluate_likelihood_sparse(const int32_t* dat, const int32_t* row, const int32_t* col, const ig_params* P,
                                    float mean_size_frag, const float* pos_bp, const int32_t* id_c, const float* s_tot,
                                    const int32_t* pos, const int32_t* len, double* vect_likelihood, int64_t n_data_pxl)
{
    lgf_init();
    const ig_params p = *P;
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const int B = 1024;
    if (g_mode) {
        int64_t hi = 0, lo = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : hi, lo) num_threads(g_threads) schedule(static)
#endif
        for (int64_t k = 0; k < n_data_pxl; k++) {
            int fi = row[k], fj = col[k];
            float s = fabsf(pos_bp[fi] - pos_bp[fj]);
            int dp = pos[fi] - pos[fj];
            float s_z = (float)(dp < 0 ? -dp : dp) * mean_size_frag;
            float s_tot_z = (float)len[fi] * mean_size_frag;
            int64_t q = ig_quantize(pair_term(p, &hot, id_c[fi] == id_c[fj], s, s_z, s_tot[fi], s_tot_z, dat[k]));
            hi += q >> 32;
            lo += (int64_t)(uint32_t)q;
        }
        ig_acc a = {hi, lo};
        publish(0, a);
        vect_likelihood[0] += ig_acc_to_double(a.hi, a.lo);
        return;
    }
    double total = vect_likelihood[0];
    double* sdata = (double*)malloc(sizeof(double) * B);
    for (int64_t b0 = 0; b0 < n_data_pxl + 1; b0 += B) { /* Z/1024+1 blocks */
        for (int t = 0; t < B; t++) {
            int64_t k = b0 + t;
            double v = 0.0;
            if (k < n_data_pxl) {
                int fi = row[k], fj = col[k];
                float s = fabsf(pos_bp[fi] - pos_bp[fj]);
                int dp = pos[fi] - pos[fj];
                float s_z = (float)(dp < 0 ? -dp : dp) * mean_size_frag;
                float s_tot_z = (float)len[fi] * mean_size_frag;
                v = pair_term(p, &hot, id_c[fi] == id_c[fj], s, s_z, s_tot[fi], s_tot_z, dat[k]);
            }
            sdata[t] = v;
        }
        total += tree_reduce(sdata, B);
    }
    free(sdata);
    vect_likelihood[0] = total;
}
