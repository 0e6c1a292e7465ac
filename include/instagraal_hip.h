/*
 * instagraal_hip.h -- C ABI of the MI355X-native per-move scoring path of instaGRAAL.
 *
 * The reference has no FFI of its own for this path: its Python reaches the GPU only
 * through pycuda (cuda_lib_gl_single.py, "CL").  This library replaces that layer --
 *   pycuda.compiler.SourceModule(...).get_function(name)   CL:1521-1600
 *   gpuarray / mem_alloc / memcpy_htod / memcpy_dtoh         CL:321-424, 551-646
 *   GPUStruct (struct of 17 int*)                            gpustruct.py:8-186
 * -- with a flat C ABI: one opaque handle per sampler, library-owned device memory,
 * caller-owned host memory, int return codes (0 = ok, <0 = error; text via
 * ig_last_error()).  No exceptions, callbacks or torch types cross the boundary.
 * A handle is not thread-safe; distinct handles are independent.
 * Unless stated otherwise a call returns after its work is complete (synchronous).
 *
 * Fragment state is exchanged as int32 soa[17][N] in the member order of
 * kernel_sparse_adapt.cu:40-58 ("KA"):
 *   pos sub_pos id_c start_bp len_bp sub_len circ id prev next l_cont sub_l_cont
 *   l_cont_bp ori rep activ id_d
 */
#ifndef INSTAGRAAL_HIP_H
#define INSTAGRAAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IG_N_TMP_STRUCT 24 /* candidate mutation slots per (frag, candidate) pair, CL:192-194 */
#define IG_MAX_CANDIDATES 16
#define IG_N_INSERT_BLOCKS 6 /* CL:192 */

typedef struct ig_ctx ig_ctx;

/* outcome of one move: the 6-tuple of sampler.step_sampler (CL:1458-1465) + diagnostics */
typedef struct ig_move_result {
    double o;            /* score of the applied candidate = new likelihood_t (CL:1454-1457) */
    double dist;         /* dist_inter_genome (CL:665-716) */
    double mean_len;     /* mean_length_contigs (CL:2742), float32 value widened */
    int32_t op_sampled;  /* 0..23 */
    int32_t id_f_sampled;/* partner fragment of the applied candidate */
    int32_t n_contigs;
    int32_t n_candidates;
    int64_t n_slice;     /* sum over candidates of sliced contacts S_c (CL:1043) */
    int64_t n_evals;     /* sum over candidates of S_c * (n_uniq + 1) term evaluations */
    int64_t bytes_min;   /* compulsory-traffic model B_min of this move */
    int32_t error;       /* 0, or a device-side consistency failure code */
    int32_t pad;         /* 0 in every record a caller sees (inside the library: 1 = the one-move launches found the move's lists too long
                          * for the slice pool and applied nothing; the entry point grows the pool and repeats the move) */
} ig_move_result;

/* ---- lifetime --------------------------------------------------------- */
int ig_create(int device_id, ig_ctx** out);
void ig_destroy(ig_ctx* ctx);
const char* ig_last_error(void);
int ig_sync(ig_ctx* ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = library-owned stream */
int ig_set_stream(ig_ctx* ctx, void* hip_stream);

/* ---- problem upload (replaces sparse_data_2_gpu CL:551-646, create_gpu_struct CL:429-549) */
/* level-(L-1) contacts: strict upper triangle COO, row-major sorted, as CL:592-615 uploads them.
 * rank/world: contact shard of this handle (rows r with r % world == rank); (0,1) = all. */
int ig_upload_contacts(ig_ctx* ctx, const int32_t* row, const int32_t* col, const int32_t* cnt, int64_t Z, int32_t M,
                       int32_t rank, int32_t world);
/* np_sub_frags_2_frags: M x float4 (parent bin, watson kb, crick kb, index in bin), simu_single.py:701-717 */
int ig_upload_subfrag_table(ig_ctx* ctx, const float* xyzw, int32_t M);
int ig_upload_state(ig_ctx* ctx, const int32_t* soa17, int32_t N);
/* contig ids are returned renumbered exactly as modify_gl_cuda_buffer leaves them (CL:2715-2881) */
int ig_download_state(ig_ctx* ctx, int32_t* soa17);
/* which: 0 = param_simu, 1 = param_simu_test (CL:2343-2349, 3019-3023); p in KA:91-100 order */
int ig_set_params(ig_ctx* ctx, const float p[8], float mean_subfrag_kb, int which);
/* list_bounds (CL:417-422) and max_bounds_insert (CL:418-420) */
int ig_set_insert_config(ig_ctx* ctx, const int32_t list_bounds[IG_N_INSERT_BLOCKS], int32_t max_bounds_insert);
/* initial prev/next/orientable for the genome distance (CL:269-276); blacklist may be NULL */
int ig_set_initial_genome(ig_ctx* ctx, const int32_t* init_prev, const int32_t* init_next, const int32_t* orientable,
                          const int32_t* blacklisted, int32_t n_blacklisted);

/* ---- likelihood (evaluate_likelihood_sparse KA:4374-4488, eval_likelihood_on_zero KA:3850-3917) */
/* nz, z: the two scalars eval_likelihood()/approx_single_likelihood_on_zeros() leave behind
 * (CL:1245-1292, 718-760).  which_params as ig_set_params.  use_prev_tables != 0 evaluates on the
 * coordinates of the state BEFORE the last applied move (what eval_likelihood_4_nuisance sees,
 * CL:1296-1344, quirk Q12).  limbs (may be NULL): exact sums {nz_hi, nz_lo, z_hi, z_lo, n_intra}. */
int ig_full_likelihood(ig_ctx* ctx, int which_params, int use_prev_tables, double* nz, double* z, int64_t* limbs5);

/* ---- the move (step_sampler CL:1401-1465) ------------------------------ */
/* scores: C x 24 doubles laid out as all_scores (CL:1414, 1431); slots that are not scored are 0. */
int ig_score_move(ig_ctx* ctx, int32_t frag_a, const int32_t* cands, int32_t C, double* scores);
/* re-materialise (frag_a, frag_b, op) into the live state (test_copy_struct CL:2094-2151) */
int ig_apply(ig_ctx* ctx, int32_t frag_a, int32_t frag_b, int32_t op);
/* score + device argmax (CL:1435-1446) + apply + bookkeeping, one small D2H */
int ig_step(ig_ctx* ctx, int32_t frag_a, const int32_t* cands, int32_t C, ig_move_result* out, double* scores_or_null);
/* n_moves consecutive moves (the reference's inner loop over step_sampler calls, main.py full_em / CL:1401-1465).
 * cands: n_moves x max_c, -1 padded.  results: n_moves entries.
 * Moves are scored W at a time against the same state ("speculative batches": candidate draws do not depend on the
 * genome) and committed in order on the device; a move whose contigs an earlier move of its batch modified is
 * re-scored in the next batch, so the results are identical to n_moves calls of ig_step for every W. */
int ig_step_batch(ig_ctx* ctx, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c,
                  ig_move_result* results);
/* ---- the candidate draw (step 1 of step_sampler, CL:1403-1408: return_neighbours CL:3103-3141, candidates.sort()) ----
 * Host code.  The reference draws with numpy's global legacy generator -- np.random.choice(xk, k, p=pk, replace=False), or
 * np.random.choice(n_frags, k, replace=False) for a bin without hetero contacts -- and everything stochastic that follows
 * continues that stream, so the draw is restated on a COPY of the MT19937 state: the caller passes key[624] and pos from
 * np.random.get_state() and puts them back with set_state() (same lists, same generator state as numpy: tests/).
 * Distributions (setup_distri_frags, CL:3053-3101): CSR over the level-L bins, bin i -> partners xk[indptr[i]..indptr[i+1])
 * with float32 probabilities pk; an empty row = no hetero contact (the uniform draw).  Lists come out sorted, without the
 * focal bin (quirk Q13) and without blacklisted bins, -1 padded to n_neighbours. */
typedef struct ig_neighbours ig_neighbours;
int ig_neighbours_create(const int64_t* indptr, const int32_t* xk, const float* pk, int32_t n_frags, const int32_t* blacklisted,
                         int32_t n_blacklisted, ig_neighbours** out);
void ig_neighbours_destroy(ig_neighbours* nb);
int ig_neighbours_draw(ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, const int32_t* frags, int32_t n_moves,
                       int32_t n_neighbours, int32_t* cands_out);
/* the same for a run of moves with nuisance sampling (instagraal.py:217-262, cycles > 4): per move the neighbour draw, then the
 * three draws of step_nuisance_parameters (CL:2976-3028): choice(4), the STANDARD normal behind normal(0, sigma) -- numpy's
 * legacy polar method with its one-value cache (has_gauss, gauss of get_state()) --, rand().  skip_normal_3: no normal is
 * drawn for modifier 3 (the reference's branch for a non-positive sigma of the trans level). */
int ig_neighbours_draw_nuisance(ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, int32_t* has_gauss, double* gauss,
                                const int32_t* frags, int32_t n_moves, int32_t n_neighbours, int32_t skip_normal_3, int32_t* cands_out,
                                int32_t* id_modif_out, double* normal_out, double* uniform_out);
/* ONE complete step_sampler call (CL:1401-1465) as the reference's loop makes it (instagraal.py:221-228), candidate draw of
 * return_neighbours (CL:3103-3141) included: nb != NULL draws cands[0 .. n_neighbours) (-1 padded, *n_cands = the list's length) on
 * numpy's MT19937 state as ig_neighbours_draw does; nb == NULL scores the caller's cands[0 .. *n_cands).  Same results as ig_step,
 * less around them: the lists and the results travel through mapped host memory instead of five copies, and the move is decided
 * and applied by the batch path's fused commit kernel (a batch of one) instead of the five kernels of the one-move tail.
 * scores_or_null == NULL (the reference's loop reads all_scores nowhere outside step_sampler, CL:1414-1454): the move may be scored
 * in two tiers like a wider batch -- every column through the float screen, the exact kernel for the columns that can still win.
 * IG_STEP_DRAW_FAST=0: ig_step's way. */
int ig_step_draw(ig_ctx* ctx, ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, int32_t frag_a, int32_t n_neighbours,
                 int32_t* cands, int32_t* n_cands, ig_move_result* out, double* scores_or_null);
/* n_moves complete step_sampler calls: draw (on a host thread, ahead of the launches) + ig_step_batch.  cands_out
 * [n_moves x n_neighbours] receives the drawn lists; the generator state is advanced past all n_moves draws. */
int ig_step_batch_draw(ig_ctx* ctx, ig_neighbours* nb, uint32_t* mt_key624, int32_t* mt_pos, int32_t n_moves, const int32_t* frags,
                       int32_t n_neighbours, int32_t* cands_out, ig_move_result* results);
/* The same in steps, for a caller that splits the slots of a batch over several GPUs (one process per GPU, every rank
 * holds the full problem): upload the lists once; per batch every rank builds all W candidate-genome sets but slices and
 * scores only slots [slot_begin, slot_end); the slot-major score records (ig_batch_records: one device buffer, a fixed
 * number of bytes per slot) are all-gathered by the caller; then every rank commits the batch -- identical integer
 * inputs, identical decisions, no further communication.  W <= max_w <= 64. */
int ig_batch_max_width(ig_ctx* ctx, int32_t max_c); /* largest W whose work buffers fit (<= 64; the per-slot window arrays are strided by 3 x the longest contig: ~12 MB per slot at 50 k bins in 1 000 contigs, gigabytes late in an assembly) */
int ig_batch_upload(ig_ctx* ctx, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c, int32_t max_w);
int ig_batch_score(ig_ctx* ctx, int32_t move0, int32_t W, int32_t slot_begin, int32_t slot_end); /* asynchronous */
int ig_batch_records(ig_ctx* ctx, void** records, int64_t* bytes_per_slot); /* slot w: records + w * bytes_per_slot */
int ig_batch_commit(ig_ctx* ctx, int32_t move0, int32_t W, int32_t* n_committed); /* moves move0 .. move0+n_committed-1 are done */
int ig_batch_results(ig_ctx* ctx, int32_t n_moves, ig_move_result* results);
int ig_set_batch_width(int w);                        /* W in 1..64 (default 24, env IG_BATCH_W); 1 = no speculation */
/* The WINDOW rule of ig_step_batch / ig_step_batch_draw (round 5): the scored slots of the moves ahead stay scored from launch to launch
 * while no contig they read is written; a launch re-scores the stale ones and fills the window up; decisions strictly in order, results
 * those of one move at a time (CL:1401-1465).  w = 0: off (every launch scores a fresh batch of ig_set_batch_width slots and drops what
 * lies behind its first conflict: rounds 1 - 4); w in 2..64: slots of the window (default 48, env IG_WINDOW). */
int ig_set_window(int w);
int ig_batch_stats(ig_ctx* ctx, int64_t out4[4]);     /* {batches, moves committed in-batch, one-move tails, predicted deltas used} */
int ig_scratch_bytes(ig_ctx* ctx, int64_t out3[3]);   /* move buffers: {per-window arrays, slice pool, per-slot records and lists} */

/* ---- a move and the nuisance step behind it, in flight together (instagraal.py:217-262 for cycles > 4: step_sampler, then
 * step_nuisance_parameters CL:2961-3051) ----------------------------------
 * The nuisance step evaluates the full likelihood under its test parameters on the coordinates of the state BEFORE the move
 * just applied (eval_likelihood_4_nuisance CL:1296-1344, quirk Q12) and needs only that move's score besides: its pass over
 * all contacts runs next to the move.  ig_nuis_begin: asynchronous -- the move (score + apply) and, on a second stream, the
 * pass under p_test (KA:91-100 order).  ig_nuis_end: waits for both; nz_test / z_test as ig_full_likelihood.  ig_nuis_accept:
 * the test parameters of the last step become param_simu (maintained sums recomputed under them, CL:3032-3036). */
int ig_nuis_begin(ig_ctx* ctx, int32_t frag_a, const int32_t* cands, int32_t C, const float p_test[8], float mean_subfrag_kb);
int ig_nuis_end(ig_ctx* ctx, ig_move_result* out, double* nz_test, double* z_test, int64_t* limbs5);
int ig_nuis_accept(ig_ctx* ctx);
/* The same for a RUN of (move, nuisance step) pairs -- the loop of instagraal.py:217-262 itself.  A rejected step changes
 * nothing a move reads, so the moves behind it are scored ahead, in batches (width: env IG_NUIS_W, default: follows the run
 * lengths, at most IG_NUIS_WMAX = 24): ig_nuis_run_begin uploads the lists of the run (as ig_batch_upload);
 * ig_nuis_step_begin(move), move = 0, 1, ... in order: asynchronous -- the step's pass under p_test and the decision + apply
 * of that ONE move from the batch it was scored in (a batch starting at `move` is scored first when there is none: first
 * step, after an accepted step, after a conflict with an earlier move of the batch, batch used up); ig_nuis_end /
 * ig_nuis_accept as above.  Results as one move and one step at a time.  Any other call that runs moves or changes state or
 * param_simu ends the run. */
int ig_links_inverse(ig_ctx* ctx); /* 1: speculative batches (ig_step_batch with W > 1, ig_batch_*, ig_nuis_run_begin) are available for this initial genome */
int ig_set_nuis_width(int w); /* moves scored ahead per launch: 0 = follow the run lengths (default, env IG_NUIS_W); results do not depend on it */
int ig_nuis_run_begin(ig_ctx* ctx, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c);
int ig_nuis_step_begin(ig_ctx* ctx, int32_t move, const float p_test[8], float mean_subfrag_kb);
/* ig_nuis_end + the acceptance test (CL:3026-3036: exp((L_test - L_move) / temperature) >= u) + ig_nuis_accept + the next
 * move's ig_nuis_step_begin (test parameters for both outcomes supplied) in one call; *accepted = 0 / 1, or 2: a close call
 * (within 1e-9 relative), nothing decided or enqueued, the caller does it with its own exp(). */
int ig_nuis_step_next(ig_ctx* ctx, double temperature, double u, const float p_next_rejected[8], const float p_next_accepted[8],
                      float mean_subfrag_kb, int32_t has_next, ig_move_result* out, double* nz_test, double* z_test, int32_t* accepted);
/* The steps of a run decide their Metropolis test (CL:3026-3036) from a SCREENED pass where they can: the change of the
 * likelihood between the model's and the test parameters, term by term in float with a rigorous bound
 * (csrc/ig_kernels_nuis.cuh); a step whose whole interval lies below T ln u is rejected without the exact pass (its *nz_test is
 * then the interval's midpoint: eval_likelihood_4_nuisance's value is only read inside the method, CL:3023-3036), every other
 * step runs the exact pass as well.  0 = the exact pass on every step (env IG_NUIS_SCREEN); env IG_NUIS_SCREEN_VERIFY=1: both
 * on every step, the bound checked. */
int ig_set_nuis_screen(int on);
/* The screened pass has two tiers.  The first reads no contact at all: the same change of the likelihood from a histogram of the
 * cis contacts over log2 of their distance (in which the term's exponent is piecewise linear), kept up to date by the moves that
 * change the genome, with its own rigorous bound; the pass over the contacts runs only where that interval does not decide
 * (csrc/ig_kernels_nuis.cuh, "tier 0").  1 (default): where it pays -- a cost model on the host (contacts, contigs, share of the
 * moves that change the genome) switches it off where few long contigs make the walks dearer than the pass; 2: always; 0 = start
 * with the pass over the contacts (env IG_NUIS_HIST); IG_NUIS_SCREEN_VERIFY=1 checks both tiers against the exact pass on every
 * step. */
int ig_set_nuis_hist(int on);
/* Chains: the pairs move .. move + n_sets - 1 of a run, as far as the device can take them WITHOUT the host (csrc/ig_common.cuh,
 * ChainIn): per segment the Metropolis intervals of the next 8 steps' test sets from the histogram tier in one launch, then one decide
 * wave that takes the moves from the batch's score records in order and tests each step (CL:3026-3036) against its interval with
 * the live likelihood; it stops in front of the first pair that needs the host (a step that is not a certain rejection, a conflict,
 * a pending windowed winner, an overflow) -- that pair is untouched and goes through ig_nuis_step_begin / ig_nuis_step_next -- and
 * goes on behind a move that changed the genome once the histogram has followed it.  p_tests [n_sets][8] in KA:91-100 order, u /
 * temperature [n_sets]: the acceptance uniforms and temperatures of those steps (n_sets <= 64).  Asynchronous: a helper thread
 * drives the segments; ig_nuis_chain_done polls (1: ended), ig_nuis_chain_end waits: *n_done pairs completed (each a move decided
 * exactly as ig_step_batch decides it and a step rejected with the margins of ig_nuis_step_next; their records: ig_batch_results),
 * *reason: 0 sets used up, 1 test, 2 conflict, 3 pending, 4 overflow, 5 no slot scored under the model's parameters, 6 the
 * histogram tier is not in use.  ig_set_nuis_chain(0) / env IG_NUIS_CHAIN=0: the runs keep to one pair per call (and score their
 * batches without winner prediction).  Results do not depend on any of it. */
int ig_set_nuis_chain(int on);
int ig_nuis_chain_begin(ig_ctx* ctx, int32_t move, int32_t n_sets, const float* p_tests, const double* u, const double* temperature,
                        float mean_subfrag_kb);
int ig_nuis_chain_done(ig_ctx* ctx);
int ig_nuis_chain_end(ig_ctx* ctx, int32_t* n_done, int32_t* reason);
/* *accepted = 3 from ig_nuis_step_next: the step was ACCEPTED from the screened interval alone (every L_test in it gives a ratio
 * above u); its exact pass -- the promotion of the maintained sum needs it, the decision does not -- runs behind the decision, next
 * to the re-scoring of the moves ahead; *nz_test was the interval's midpoint.  The exact value (what eval_likelihood_4_nuisance
 * returns, CL:1296-1344, and likelihood_t becomes, CL:3036): here, any time before the next step is accepted that way. */
int ig_nuis_exact_result(ig_ctx* ctx, double* nz_test);

/* ---- bookkeeping -------------------------------------------------------- */
int ig_renumber_contigs(ig_ctx* ctx, int32_t* n_contigs, float* mean_len, int32_t* max_id); /* CL:2715-2881 */
int ig_bomb(ig_ctx* ctx, const int32_t* shuffle);                                            /* CL:1925-1948 */
int ig_genome_distance(ig_ctx* ctx, double* d);                                              /* CL:665-716 */
int ig_get_valid_insert(ig_ctx* ctx, int32_t out12[12]); /* gpu_list_valid_insert, stale-flag state (Q4) */

/* ---- multi-GPU (contact shards) -------------------------- */
/* Two-phase move: partial sums over this handle's contact shard are left in a device buffer of
 * ig_partials_count() int64 values; the caller all-reduces (SUM) it across ranks, then finishes. */
int ig_set_shard(ig_ctx* ctx, int32_t rank, int32_t world); /* score rows r with r % world == rank */
int64_t ig_partials_count(ig_ctx* ctx);
void* ig_partials_device_ptr(ig_ctx* ctx);
int ig_step_begin(ig_ctx* ctx, int32_t frag_a, const int32_t* cands, int32_t C);
int ig_step_finish(ig_ctx* ctx, ig_move_result* out, double* scores_or_null);

/* ---- timing / roofline -------------------------------------------------- */
/* average duration (ms) of the named kernel over the launches since the last reset, measured
 * with hipEvents on the library's stream; name in {"score","mutate","gather","finalize","apply","post"} */
int ig_kernel_time_ms(ig_ctx* ctx, const char* name, double* avg_ms, int64_t* n_launches);
int ig_reset_timers(ig_ctx* ctx, int enable);
int ig_set_timer_sampling(ig_ctx* ctx, int every); /* hipEvent pairs around every n-th launch only (an event record costs ~6 us of idle queue) */

/* ---- debug ABI (kernel-granularity parity tests) ------------------------ */
/* evaluate the model on arrays: ex = rippe(s), exc = rippe_circ(s, s_tot), term, quantised term */
int ig_debug_eval_terms(ig_ctx* ctx, const float* s, const float* s_tot, const int32_t* ob, int64_t n, float* ex, float* exc,
                        double* term, int64_t* q);
/* candidate genome (cand index c, slot) of the last ig_score_move, as soa17 (ids internal) */
int ig_debug_candidate_state(ig_ctx* ctx, int32_t c, int32_t slot, int32_t* soa17);
/* per-(candidate, slot) exact sums of the last scored move: nz limbs, z limbs, n_intra, extract limbs, S_c */
int ig_debug_last_sums(ig_ctx* ctx, int64_t* nz_hi, int64_t* nz_lo, int64_t* z_hi, int64_t* z_lo, int64_t* n_intra,
                       int64_t* ext_hi, int64_t* ext_lo, int64_t* n_slice, int32_t* n_uniq, int32_t* uniq);
int ig_debug_tables(ig_ctx* ctx, float* dist, int32_t* id_c, float* s_tot, int32_t* pos, int32_t* len);
/* maintained exact sums {nz_hi, nz_lo, z_hi, z_lo, n_intra}; {n_contigs, next_cid, chosen c, k, slot, windowed} */
int ig_debug_globals(ig_ctx* ctx, int64_t* sums5, int32_t* ints6);
int ig_debug_dbg(ig_ctx* ctx, int32_t* out8, int32_t clear); /* Glob.dbg: the eight words a device-side consistency failure leaves (tuning builds: tick counters) */
/* two-tier scoring of the batches (csrc/ig_kernels_screen.cuh): the hardware log2 / exp2 the screening bound leans on, measured
 * over their whole domain {max |v_log_f32(s) - log2 s| / (2^-23 (|result| + 1)), max |v_exp_f32(y) - 2^y| / (2^-23 2^y)};
 * and {largest |screened - exact| / bound, largest bound} of the runs under IG_SCREEN_VERIFY=1, {columns screened, columns
 * scored exactly} */
int ig_debug_transcendental_error(ig_ctx* ctx, double out2[2]);
int ig_debug_screen_stats(ig_ctx* ctx, double out6[6]); /* ..., terms screened, terms scored exactly */
/* 0 disables the reference's dropped-tail behaviour of eval_sub_likelihood (quirk Q5); default 1 */
int ig_debug_set_tail_quirk(int on);
int ig_debug_tile_trace(ig_ctx* ctx, int64_t* out4n, int64_t cap, int64_t* n_items); /* per-workgroup clocks of one from-scratch pass */
int ig_debug_nuis_wait(ig_ctx* ctx, double* seconds); /* time ig_nuis_end has waited for the device */
/* per-workgroup clocks of one screened nuisance pass under p_test (8 words per workgroup; see ig_hip.hip) and its output words */
int ig_debug_diff_trace(ig_ctx* ctx, const float p_test[8], float mean_subfrag_kb, int64_t* out8n, int64_t cap, int64_t* n, int64_t* sums8);
/* the screened nuisance pass: {steps screened, rejected from the interval alone, exact passes behind a screened one, void,
 * largest bound, largest |screened - exact| / bound seen, sum of the bounds, steps whose interval did not decide, void because of
 * {the parameter pair, a contact, a workgroup's sums, a move record that did not come from the batch commit}} */
int ig_debug_nuis_screen_stats(ig_ctx* ctx, double out12[12]);
/* its histogram tier: {evaluations, steps rejected there, accepted there, void, sum of its bounds, largest |screened - exact| / bound
 * seen, moves walked into the histogram, builds from scratch, void because of {as above}} */
int ig_debug_nuis_hist_stats(ig_ctx* ctx, double out12[12]);
/* the maintained histogram against one built from scratch (the last move of the run is walked in first): words that differ, -1: none kept */
int ig_debug_nuis_hist_check(ig_ctx* ctx, int64_t* mismatches);
/* two-tier scoring, the decide step's zero-score rule (a score of exactly 0.0 counts as "not scored", CL:1435-1440: a move whose
 * contenders hold one under the live scalars is scored again with every column exact): fault injection for the tests -- every n-th
 * move of a two-tier batch takes that path (0 = off) -- and how often a handle has taken it */
int ig_debug_set_zero_inject(int every);
int ig_debug_zero_fallbacks(ig_ctx* ctx, int64_t* fallbacks);
/* tests: moves of the one-move path (ig_step, ig_score_move, ig_apply, ig_step_begin, ig_step_batch at width 1, ig_nuis_begin) whose
 * lists did not fit the slice pool and were repeated with a larger one -- the counterpart of the batch path's re-run slots
 * (replaces nothing in the reference: its sort buffers are sized for the whole matrix, CL:1009-1069) */
int ig_debug_pool_retries(ig_ctx* ctx, int64_t* n);
int ig_debug_step_stats(ig_ctx* ctx, int64_t out2[2]); /* ig_step_draw: {calls that went through mapped memory, of them finished by the one-move tail} */
int ig_debug_nuis_chain_stats(ig_ctx* ctx, int64_t out10[10]); /* chains: {calls, segments, pairs completed, ends by reason [7]} */
int ig_debug_set_full_hist(int on); /* from-scratch pass: all-trans tiles from their count histograms (1, default) or contact by contact (0) */

#ifdef __cplusplus
}
#endif
#endif
