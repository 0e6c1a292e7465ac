/*
 * ig_oracle.h -- CPU restatement of instaGRAAL's per-move scoring kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (instagraal_amd/).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / baseline.
 *
 * One C function per live reference kernel of
 *   /root/reference/src/instagraal/kernels/kernel_sparse_adapt.cu  ("KA")
 * with the reference's argument order, operating on whole-genome arrays exactly
 * as the CUDA grid does (one loop iteration == one CUDA thread).  Where the
 * CUDA code is order-dependent (shared/global atomics) the canonical order is
 * "threads in index order, blocks in index order" (SURVEY.md Appendix D, Q1-Q3).
 *
 * PARITY: the reference's own tests hold no golden vector for this path and the
 * CUDA source cannot be compiled or run here (no nvcc / no CUDA device), so the
 * kernel arithmetic is "parity unpinned" against a CUDA run.  What IS pinned:
 * the host orchestration (candidate draw, stale flags, argmax, apply, RNG use),
 * by driving the reference's own `sampler` Python over these functions through
 * a functional fake pycuda (tools/gen_golden.py -> tests/golden/).
 *
 * Two arithmetic modes:
 *   IG_MODE_LIBM : glibc powf/expf/log10 and double sums in the reference's
 *                  block/tree/atomic shape -- the closest CPU stand-in for CUDA.
 *   IG_MODE_DET  : include/ig_detmath.h functions and exact fixed-point sums --
 *                  what the HIP kernels must reproduce bit for bit.
 */
#ifndef IG_ORACLE_H
#define IG_ORACLE_H

#include <stdint.h>
#include "../include/ig_detmath.h"

#ifdef __cplusplus
extern "C" {
#endif

/* KA:40-58 -- struct of 17 int* in this order (also the packed pointer block
 * that gpustruct.py:142-160 builds, so the fake pycuda can hand it over as is). */
typedef struct frag {
    int32_t* pos;
    int32_t* sub_pos;
    int32_t* id_c;
    int32_t* start_bp;
    int32_t* len_bp;
    int32_t* sub_len;
    int32_t* circ;
    int32_t* id;
    int32_t* prev;
    int32_t* next;
    int32_t* l_cont;
    int32_t* sub_l_cont;
    int32_t* l_cont_bp;
    int32_t* ori;
    int32_t* rep;
    int32_t* activ;
    int32_t* id_d;
} frag;

typedef struct { float x, y, z, w; } ig_float4;
typedef struct { int32_t x, y, z; } ig_int3;

enum { IG_MODE_LIBM = 0, IG_MODE_DET = 1 };
void igo_set_mode(int mode);
int igo_get_mode(void);
void igo_set_threads(int n); /* OpenMP threads for the Z-length loops (DET mode only) */

/* constants the reference injects by text substitution (CL:1528-1537) */
#define IGO_N_TMP_STRUCT 24
#define IGO_SIZE_BLOCK_4_SUB 64
#define IGO_N_TO_CUT 6

/* ---- genome-state kernels (KA:357-482, 612-3693, 4566-4626, 4657-4692) ---- */
void igo_select_uniq_id_c(const frag* f, int32_t* list_uniq_id_c, int32_t* list_uniq_len, int32_t* counter, int n_frags);
void igo_make_old_2_new_id_c(const int32_t* list_uniq_id_c, int32_t* old_2_new, int n_contigs);
void igo_count_num(const int32_t* vals, int value, int32_t* counter, int n_values);
void igo_renumber_id_c(frag* f, const int32_t* old_2_new, int32_t* id_contigs, float max_id, int n_frags); /* id part of gl_update_pos */
void igo_explode_genome(frag* f, const int32_t* shuffle_order, int n_frags);
void igo_flip_frag(frag* out, const frag* in, int id_f_flip, int n_frags);
void igo_pop_out_frag(frag* out, const frag* in, int32_t* pop_id_contigs, int id_f_pop, int max_id_contig, int n_frags);
void igo_pop_in_frag_1(frag* out, const frag* in, int id_f_pop, int id_f_ins, int max_id_contig, int ori_f_pop, int n_frags);
void igo_pop_in_frag_2(frag* out, const frag* in, int id_f_pop, int id_f_ins, int max_id_contig, int ori_f_pop, int n_frags);
void igo_pop_in_frag_3(frag* out, const frag* in, int id_f_pop, int id_f_ins, int max_id_contig, int ori_f_pop, int n_frags);
void igo_split_contig(frag* out, const frag* in, int32_t* split_id_contigs, int id_f_cut, int upstream, int max_id_contig, int n_frags);
void igo_paste_contigs(frag* out, const frag* in, int id_fA, int id_fB, int max_id_contig, int n_frags);
void igo_get_bounds(const frag* f, int id_f_pop, int id_f_ins, int32_t* list_valid_insert, const int32_t* list_bounds,
                    int32_t* id_f_cut_upstream, int32_t* id_f_cut_downstream, int n_bounds, int n_frags);
void igo_extract_block(frag* out, const frag* in, int32_t* split_id_contigs, int id_f_cut_a, const int32_t* list_id_f_cut_b,
                       int id_fb, int upstream, int max_id_contig, int n_frags);
void igo_insert_block(frag* out, const frag* o, const frag* init, int id_f_pop, int id_f_ins, const int32_t* list_id_bounds,
                      const int32_t* list_valid_insert, int id_mutation, int id_bound, int upstream, int n_frags);
void igo_simple_copy(frag* out, const frag* in, int n_frags);
void igo_copy_struct(frag* out, const frag* in, int32_t* id_contigs, int n_frags);
void igo_extract_uniq_mutations(const frag* f, int frag_a, int frag_b, int32_t* list_uniq_mutations,
                                const int32_t* list_valid_insert, int32_t* n_uniq, int flip_eject);

/* ---- coordinates (KA:3699-3822) ---- */
void igo_fill_vect_dist(const ig_float4* sub2frag, const frag* f, float* dist, int32_t* id_c, float* s_tot, int32_t* pos,
                        int32_t* len, int n_sub_frags, int id_mut);
void igo_uni_fill_vect_dist(const ig_float4* sub2frag, const frag* f, float* dist, int32_t* id_c, float* s_tot, int32_t* pos,
                            int32_t* len, int n_sub_frags);

/* ---- likelihood (KA:485-607, 3850-4488) ---- */
void igo_slice_sp_mat(const int32_t* dat, const int32_t* row, const int32_t* col, const frag* f, const int32_t* vect_id_c,
                      const int32_t* vect_pos, int32_t* sub_row, int32_t* sub_col, int32_t* sub_dat, int id_ctg1, int id_ctg2,
                      int id_frag_a, int id_frag_b, int n_bounds, int32_t* counter, int64_t size_arr);
void igo_prepare_sparse_call(const int32_t* row, ig_int3* info_block, int32_t* block_csr, int32_t* counter, int size_arr);
void igo_eval_likelihood_on_zero(const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                 const ig_params* P, float mean_size_frag, double* vect_likelihood, int32_t* n_vals_intra,
                                 int n_frags);
void igo_eval_all_likelihood_on_zero_1st(const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                         const ig_params* P, float mean_size_frag, const int32_t* list_uniq,
                                         const int32_t* n_uniq, double* vect_likelihood, int32_t* n_vals_intra, int n_frags);
void igo_eval_all_likelihood_on_zero_2nd(const int32_t* list_uniq, const int32_t* n_uniq, const ig_params* P,
                                         double* vect_likelihood, const int32_t* n_vals_intra, const double* n_tot_pxl);
void igo_eval_all_scores(const int32_t* list_uniq, const int32_t* n_uniq, const double* z, const double* nz,
                         const double* extract, const double* curr_nz, double* all_score);
void igo_extract_sub_likelihood(const int32_t* dat, const ig_int3* info_block, const int32_t* block_csr, const int32_t* row,
                                const int32_t* col, const ig_params* P, float mean_size_frag, const float* pos_bp,
                                const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                                double* vect_likelihood, int n_data, int n_sub_frags);
void igo_eval_sub_likelihood(const int32_t* dat, const ig_int3* info_block, const int32_t* block_csr, const int32_t* row,
                             const int32_t* col, const ig_params* P, float mean_size_frag, const float* pos_bp,
                             const int32_t* id_c, const float* s_tot, const int32_t* pos, const int32_t* len,
                             const int32_t* list_uniq, const int32_t* n_uniq, double* vect_likelihood, int n_data,
                             int n_sub_frags);
void igo_evaluate_likelihood_sparse(const int32_t* dat, const int32_t* row, const int32_t* col, const ig_params* P,
                                    float mean_size_frag, const float* pos_bp, const int32_t* id_c, const float* s_tot,
                                    const int32_t* pos, const int32_t* len, double* vect_likelihood, int64_t n_data_pxl);

/* In DET mode every sum above is also left behind as normalised limbs
 * (value = hi*2^32 + lo, units of 2^-32) so tests can compare integers. */
void igo_last_limbs(int64_t* hi, int64_t* lo, int n); /* n <= 24: limbs of the last vector-valued sum */

/* scalar entry points used by the det-math parity test */
void igo_eval_terms(const float* s, const float* s_tot, const int32_t* ob, int64_t n, const ig_params* P, float* ex,
                    float* ex_circ, double* term, int64_t* q);
void igo_lgf_table(double* out15);

#ifdef __cplusplus
}
#endif
#endif
