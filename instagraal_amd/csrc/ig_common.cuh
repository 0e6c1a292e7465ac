/* ig_common.cuh -- includes, error plumbing and the device-side data structures shared by every kernel of
 * libinstagraal_hip.so (one translation unit: ig_hip.hip includes the parts in order). */
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/ig_detmath.h"
#include "../../include/instagraal_hip.h"
#include "ig_ops.cuh"

#define NSLOT 25          /* 24 mutation slots + the current genome */
#define NCODE 8           /* contig codes inside a candidate: A, B, fresh0..fresh3 (+spare) */
#define NFRESH 4
#define LGF_TAB 1024
#define SCORE_THREADS 256

static thread_local std::string g_err;
static int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
int ig_fail_msg(const char* msg) { return fail("%s", msg); } /* for ig_draw.cpp */
#define HIPCK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

/* ------------------------------------------------------------------ device data */

struct SubTab {
    int parent;
    float wat, cri;
    int w;
};

struct State { /* N-length arrays */
    int *pos, *spos, *cid, *sbp, *circ, *prev, *next, *L, *SL, *LB, *ori; /* dynamic, order = Loc */
    int *lb, *sl, *sub_first, *rep, *activ, *id_d;                        /* constant */
};
#define NDYN 11

struct Tables { /* M-length current coordinates (uni_fill_vect_dist, KA:3763-3822) */
    float* dist;
    float* stot;
    int* len;
    int2* cp; /* (contig id, rank in contig) packed: one 8-byte gather per contact endpoint */
};

struct CandMeta {
    int B, ctgA, ctgB, same, windowed;
    int LA, LB, SLA, SLB, n_loc, m_loc;
    int lA, lB; /* local indices of A and B */
    int n_uniq, uniq[24], kidx[NSLOT];
    int flags[12], pos_up[6], pos_down[6];
    /* slice windows (KA:530-548) */
    int pos_fa, pos_fb, up_fa, down_fa, up_fb, down_fb;
};

struct ColMeta {
    float stot;
    int len;
};

struct Glob {
    ig_params par[2];
    float mean_kb;
    int slice_nb;
    int list_bounds[6];
    long long nz_hi, nz_lo, z_hi, z_lo, n_intra;
    long long credit2, credit2_acc;
    double n_tot_pxl;
    double lgf[15];
    int n_contigs, next_cid, n_black, N, M;
    int max_L, max_SL; /* upper bounds of the longest contig in fragments / sub-fragments (exact after a recount, raised by every
                        * committed move that changed the genome: its window's total) */
    int n_prev_touched;
    int valid_insert[12];
    int error;
    int stamp_ctr;
    int retry_pool; /* one-move path: the lists of a move did not fit the slice pool (k_offsets: MoveCtl.overflow) -- set by the chooser,
                     * nothing is applied while it stands (every later one-move launch skips as well); the host grows the pool, clears it
                     * and repeats the move (ig_move_result.pad = 1 tells it) */
    int retry_pad;
    long long scr_cols, scr_cont; /* two-tier scoring: columns screened, columns scored exactly */
    long long scr_void_cols; /* columns whose screening bound was void */
    long long scr_terms, scr_terms_exact; /* ... and the (contact, column) terms in them */
    int dbg[8]; /* what raised Glob.error (diagnostics: printed with a consistency failure) */
};

/* one move slot of a batch (W = 1: the move in flight) */
#define IG_MAX_BATCH 64
struct MoveCtl {
    int A, C, force_slot, fresh; /* fresh: first of the NFRESH contig ids this move may create */
    int ch_c, ch_k, ch_slot, ch_windowed;
    int superset0; /* candidate 0 was scored with every insert slot (its stale flags were not known yet) */
    int overflow;  /* the slice pool could not hold this slot: it is re-run at the head of the next batch */
    int n_dirty, pad;
    int exact_chunk, pad2; /* two-tier scoring: entries per work item of the exact kernel (k_contend) */
    double ch_score;
    long long n_slice_tot, n_eval_tot, bytes_min;
    long long d_hi, d_lo; /* k_delta accumulator */
    long long nzb_hi, nzb_lo; /* the maintained exact sum when this move was decided, i.e. of the state before it (k_decide_batch) */
    /* the predicted winner (k_predict: scores under the batch-start scalars), when it is a windowed candidate that changes
     * the genome, and its exact full-contig delta, computed before the decisions so that the batch need not pause for it */
    int pred, pred_c, pred_k, pred_pad; /* pred = c * 24 + slot, or -1 */
    long long pd_hi, pd_lo;
};

/* what the decide step needs about one (candidate, mutation slot), written slot-major by k_records: 64 bytes (the decide
 * wave is bound by the loads and instructions per move, so everything that does not depend on the live scalars is
 * prepared here and nothing else is stored) */
struct SlotPre {
    long long nz_hi, nz_lo;      /* slice sum under this slot's genome (all sliced contacts) */
    double nz_d, nz_cut_d;       /* ig_acc_to_double of it, whole and without the tail quirk Q5 drops (list position >= S_c mod 64) */
    long long dz_hi, dz_lo, dni; /* zero-pixel sum and intra pair count: this genome minus the current one, on the window */
    int k;                       /* coordinate column (0 = not scored) */
    unsigned info;               /* bit 0: the mutated window differs from the current genome; bits 1..: contigs on the mutated window */
};
/* ... and about one candidate: its slice under the current genome and the few fields of its window the decisions read */
struct CandPre {
    long long ext_hi, ext_lo; /* slice sum under the current genome */
    double ext_d;             /* ig_acc_to_double(ext_hi, ext_lo) */
    long long n_slice;
    int r;                    /* S_c mod 64 */
    int base_cnt;             /* list entries before the block-insert slots */
    int ctgA, ctgB, m_loc, n_loc, n_uniq, B;
    int same_windowed;        /* bit 0: same contig, bit 1: windowed slice */
    unsigned flag_mask;       /* get_bounds validity of the 12 block-insert slots (bit i: flags[i] != -1) */
    int overflow;             /* the slot's slice did not fit the pool: re-run */
    int pred;                 /* candidate 0 of a slot carries the slot's prediction (MoveCtl.pred, pd_hi, pd_lo) */
    long long pd_hi, pd_lo;
};

struct MoveBuf {
    int* Lloc;      /* [capW*capC][N] global ids of local fragments */
    int* lbloc;     /* [..][N] */
    int* slloc;     /* [..][N] */
    int* subs;      /* [..][M] global sub-frag id of local sub index */
    int* rowcnt;    /* [..][M] sliced contacts per local row */
    int4* rowbe;    /* [..][M] the local row's CSR range {begin lo, begin hi, length, global sub-frag id}: k_slice starts from it (one round trip instead of subs -> rowptr) */
    int* sl_li;     /* slice pool: candidate cw's list starts at slice_offset(w, c): local row index, */
    int* sl_lj;     /*           local column index, */
    int* sl_ob;     /*           observed count (order = arrival, sums are order-free) */
    unsigned long long* sl_pk; /* packed form used instead when M < 2^20 and every count < 2^24 (8 instead of 12 bytes per
                                * entry: k_slice is bound by writing the lists): row | column << 20 | count << 40 */
    int packed;
    /* a candidate's slice list is kept in SLICE_SEG segments (local row r -> segment r % SLICE_SEG): one append cursor
     * per segment instead of one per candidate (same-address atomics were k_slice's bottleneck), and workgroup x of
     * k_score_list streams exactly segment x */
    unsigned* touched;  /* ig_ctx.touched_bits */
    long long* slbound; /* [..][SLICE_SEG] upper bound of a segment = contacts in its rows */
    long long* sloff;   /* [..][SLICE_SEG] start of the segment in the pool, -1 = does not fit (k_offsets) */
    long long pool_cap;
    uint2* coords;  /* [..][NSLOT][M] column k: {dist bits, pos | code<<28} per local sub index */
    int* loc;       /* [..][NSLOT][NDYN][N] candidate genomes on the local window */
    CandMeta* meta; /* [..] */
    ColMeta* cmeta; /* [..][NSLOT][NCODE] */
    long long* part;/* [..][P_STRIDE] partial sums (all-reduced across ranks when sharded) */
    long long* qpart;/* [..][Q_STRIDE] sums every rank computes redundantly */
    double* scores; /* [..][24] */
    MoveCtl* ctl;   /* [capW] */
    int2* sinfo;    /* [..][NSLOT] (changed, contig heads) of each candidate genome (k_mutate) */
    /* the score records of one slot are contiguous -- capC x 24 SlotPre, then capC CandPre -- so that the records of a range
     * of slots are ONE block of memory (one all-gather per batch when the slots are split over GPUs): pre_at / cpre_at */
    char* rec;
    size_t rec_stride; /* bytes per slot */
    /* two-tier scoring of a batch (ig_kernels_screen.cuh): screened sums + bounds per (candidate, column), the columns whose
     * bound is void, the contender masks */
    struct ScreenSum* scr; /* [..][NSLOT] */
    unsigned* scr_void;    /* [..] bit k */
    unsigned* scr_ub;      /* [..] bit k: the screened sum is an upper bound only (a ring on the window) */
    unsigned* cont;        /* [..] bit k: column k can still win (k_contend) */
    unsigned* livecol;     /* [..] bit k: column k's genome differs from the current genome on the window (k_mutate: the screening kernel pairs
                            * the live columns; from the mutation slots' changed flags it took a dependent round of loads per workgroup) */
    unsigned* ident;       /* [..] bit k: column k's genome IS the current genome on the window (its sums are column 0's: neither
                            * screened nor scored) */
    /* two-tier scoring: the exact kernel's work list of a batch: eight interleaved sub-lists (k_contend), work[0..8) = their
     * lengths, work[8..16) = the lengths they would have had if every slot had fitted, item j of sub-list x at work[16 + 8 j + x]:
     * (chunk << 32 | cw << 12 | k << 4 | segment): entries [chunk * ch, (chunk + 1) * ch) of that segment of candidate cw's
     * slice list under column k, ch = MoveCtl.exact_chunk of the slot (whole segments would leave the launch waiting for
     * the few longest ones) */
    unsigned long long* work;
    int work_cap;
    int* slot_items; /* [capW][8] work items a slot puts on each of the eight sub-lists (k_contend -> k_worklist) */
    int* pred_list;  /* [1 + capW] k_predict -> k_delta: how many positions have a predicted windowed winner, and which (round 6: k_delta's grid
                      * covers those -- one or two of a chain's 36 slots -- instead of every slot: 9 216 workgroups that left on their first load) */
    int* order;      /* [capW * capC] the (slot, candidate) pairs of the batch by falling size of their slice lists (w << 8 | c; k_offsets):
                      * the screening launch hands out the long ones first */
    /* set while the parameter-dependent half of slots scored EARLIER is redone (enqueue_score, par_only): {n, contig ids modified by
     * the moves of this batch committed meanwhile}.  The Q5 tail walk is the one scoring step that reads the live tables: it
     * skips a slot whose contigs are on the list (the decide step stops in front of such a slot anyway) */
    const int* stale;
    /* the Q5 tail of a candidate's slice list -- its last S_c mod 64 contacts as (local row, local column, count) -- found once per
     * structural scoring (a radix descent over the window's rows and a walk of their ends: the longest chain of the scoring
     * launches) and kept for the re-evaluations under other parameters: tail_n[cw] = -1 not walked yet (k_gather), else the number
     * of entries in tail_ent[cw][3][64] */
    int* tail_n;
    int* tail_ent;
    int N, M, capC, capW;
    int nseg; /* list segments of the batch in the buffers: 8, or 16 = SLICE_SEG where a batch is a few slots with long lists (few long
               * contigs, late in an assembly: the screening launch needs its workgroups -- bigctg 5.9 -> 6.3 k moves/s); set by the
               * host with the structural half of a scoring (enqueue_score), a power of two */
    /* strides of the per-window arrays above (Lloc .. loc: sN fragments, subs / rowcnt / coords: sM sub-fragments): the
     * largest window the genome can produce right now -- two contigs of the current maximum length, with headroom -- not
     * the whole genome; the host grows them when the maximum grows (ensure_window_buffers) */
    int sN, sM;
    /* the WINDOW of scored slots (round 5; ig_host_batch.inc: run_moves_window).  A kernel's slot index w is a POSITION in the window
     * (position p <-> move `done + p`); the buffers of position p live in physical slot PS(p) = (p + rot) & wmask, so that a slot keeps
     * its buffers while the window moves on (rot = done & wmask).  keep: bit p = position p holds a slot scored by an EARLIER launch
     * that is still valid (no contig it reads was written since): every scoring kernel leaves it alone, the decide step uses it.
     * Everywhere else (one move per call, the nuisance runs, slots split over GPUs): rot = 0, wmask = ~0, keep = 0 -- PS is the identity. */
    int rot, wmask;
    unsigned long long keep;
    int ring; /* 1: the window rule (every slot's candidate 0 is scored with all block-insert slots: its stale flags are not known yet) */
};
#define PS(w) (((w) + mb.rot) & mb.wmask)
#define KEPT(w) ((int)((mb.keep >> (w)) & 1ull))
#define LW(ps) (((ps) - mb.rot) & mb.wmask) /* the position of physical slot ps */
/* layout of MoveBuf.part per candidate (int64 units) */
#define P_NZ 0                 /* [NSLOT][2] slice sums per column k (k=0: current = "extract") */
#ifndef SLICE_SEG
#define SLICE_SEG 16 /* the MOST segments (array strides); a batch uses MoveBuf.nseg of them: 8 (16 until round 3: the screening kernel's workgroups (one per segment, pair of columns and candidate) live as long
                     * around their pass as in it (tools/screen_probe.py); 32 / 16 / 8 / 4 / 2 segments: k_screen 264 / 197 / 168 / 164 / 242 us,
                     * 36.9 / 41.3 / 43.3 / 42.6 / 36.0 k moves/s (fewer append cursors cost k_slice)) */
#endif
#define P_CNT (NSLOT * 2)      /* [SLICE_SEG] kept entries per segment; S_c = their sum (slice_total) */
#define P_STRIDE (NSLOT * 2 + SLICE_SEG)
/* not all-reduced (computed redundantly on every rank) */
#define Q_Z 0                  /* [NSLOT][2] zero-pixel sums on the local window, per column k */
#define Q_NI (NSLOT * 2)       /* [NSLOT] intra pair counts */
#define Q_NZFULL (NSLOT * 3)   /* [NSLOT][2] slice sums before the tail correction */
#define Q_TAIL (NSLOT * 5)     /* [NSLOT][2] sum of the last S_c mod 64 sliced contacts' terms (quirk Q5) */
#define Q_STRIDE (NSLOT * 7)

/* a nuisance step's results on the host (pinned, mapped where possible) */
struct NuisHost {
    ig_move_result res;
    long long sums[8];
    int frag, cands[IG_MAX_CANDIDATES]; /* the move's lists: the asynchronous upload reads them after ig_nuis_begin returned */
    int max_L, max_SL;                  /* Glob.max_L / max_SL as of the move */
    volatile int res_seq, sums_seq;     /* written last, by the kernel that wrote the record (k_commit_batch) / the sums (k_full_nz_tiled) */
    /* the screened pass (ig_kernels_nuis.cuh): its sums {exact limbs of the all-trans tiles' change, screened sum and bound in
     * 2^-20 units, void flags, contacts read}; the maintained exact sum as of BEFORE the move of the record (k_commit_batch) */
    long long diff[8];
    long long nzb[2];
    volatile int diff_seq;
    /* an accepted step whose exact pass ran BEHIND the decision (the screened interval was decisive): the pass's exact limbs, left by
     * the promotion of the sums (k_nuis_promote, mode 2) */
    long long exact[2];
    volatile int exact_seq;
    int changed; /* the record's move changed the genome (k_commit_batch, with the record) */
};

/* ig_step_draw (one reference-shaped step_sampler call, CL:1401-1465): mapped, coherent host memory the call's kernels read the
 * move's lists from and write its scores to -- no copy in either direction, no stream synchronisation.  The record comes back the
 * same way: through NuisHost.res from the batch commit kernels, through `fin` from k_commit when the move was finished by the one-move
 * tail (a windowed winner that changes the genome). */
struct StepHost {
    int in[1 + IG_MAX_CANDIDATES]; /* focal bin, candidates */
    ig_move_result fin;
    int max_L, max_SL;             /* with `fin`: Glob.max_L / max_SL behind the move */
    volatile int fin_seq;          /* written last, behind a system-scope fence */
    double scores[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT];
};

/* The cis contacts of the state BEFORE the last move as a histogram over log2 of their distance (ig_kernels_nuis.cuh, tier 0 of
 * the screened nuisance pass): bin = floor(x 2^NH_OCT_BITS), x = log2 s in units of 2^-NH_FRAC_BITS; per bin
 * {contacts, sum of the offsets inside the bin, sum of the counts, sum of count x offset} -- integers, maintained with atomics by
 * the moves that change the genome, so independent of any order. */
#define NH_OCT_BITS 10
#define NH_FRAC_BITS 20
#define NH_LMAX 24
#define NH_NB ((2 * NH_LMAX) << NH_OCT_BITS)
#define NH_SUB (1 << (NH_FRAC_BITS - NH_OCT_BITS)) /* offsets inside a bin: 0 .. NH_SUB - 1 */
#define NH_MISC 16
struct NuisHist {
    long long* bins; /* [NH_NB][4] */
    long long* dh;   /* [dh_n + 1]: cis contacts by rank distance (the last entry: that far or further -- with dh_n = the number of
                      * sub-fragments no contact gets there).  Up to LDS_PZ the evaluation reads P_z from the staged tables, beyond
                      * (contigs of more than LDS_PZ sub-fragments under a table longer than that) from the tables / the formula */
    long long* misc; /* {cis at distance 0: contacts, counts; cis on a ring; cis outside the binned range; all contacts, their counts;
                      * contacts with a count beyond the screening term's domain; the largest rank distance ever entered (the
                      * evaluation's loop bound: it never comes down)} */
    int dh_n;
};
#define NH_DH_BLOCKS 8 /* workgroups of k_hist_eval that share the rank distances */

/* ---- chains of (move, nuisance step) pairs decided on the device (ig_nuis_chain_begin) ------------------------
 * Once a chain has settled, 97 % of its nuisance steps are rejected and most moves leave the genome alone: nothing a pair does
 * depends on the host.  A SEGMENT evaluates the Metropolis intervals of the next CHAIN_SEG steps' test parameter sets from the
 * histogram of the cis contacts' distances in one launch (k_chain_hist_eval: the histogram is that of the current state, valid
 * for every step up to and including the first move that changes the genome), then ONE decide wave takes the moves from the
 * batch's score records in order, tests each step against its interval with the live likelihood in registers, and stops in
 * front of the first pair that needs the host -- a test that is not a certain rejection, a conflict, a pending windowed winner,
 * an overflow -- or behind the first move that changes the genome (the histogram has to follow it before the next segment). */
#ifndef CHAIN_SEG
#define CHAIN_SEG 8    /* test parameter sets evaluated per segment */
#endif
#define CHAIN_MAX 64   /* ... uploaded per call */
#define DIFF_FIX 1048576.0 /* 2^20: the screened nuisance passes publish sums and bounds as integers (deterministic totals) */
struct ChainIn {       /* host -> device, per step: the test parameters (KA:91-100 order) and ln of the acceptance uniform */
    float p[8];
    double ln_u;
};
struct ChainTest {     /* k_chain_hist_eval -> the decide wave, per set: the histogram tier's interval of D (2^-20 units), its void flags,
                        * the zero-pixel likelihood of the test set */
    long long s_fix, b_fix, flags;
    double z;
};

struct ig_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    hipStream_t stream2;           /* k_tail next to k_score_list */
    hipEvent_t ev_slice, ev_tail;
    hipStream_t stream3;           /* the nuisance step's full pass, next to the move it follows (ig_nuis_begin) */
    hipEvent_t ev_main;            /* everything queued on the library stream before the step in flight */
    hipEvent_t ev_gathered;        /* k_gather of the move in flight is done: tab_prev holds the state before that move */
    long long* scratch_accept;     /* ig_nuis_accept's accumulators (zero between calls) */
    long long* scratch_nuis;       /* 8 x int64 reduction scratch of that pass */
    struct NuisHost* host_nuis;    /* pinned: its results and the move's */
    struct NuisHost *host_nuis_dev, *pub_sums; /* its device address when mapped; set while a pass that publishes its sums is enqueued */
    int res_seq, sums_seq;         /* launch numbers the flags in host_nuis are compared with */
    bool nuis_pub_res, nuis_pub_sums; /* the step in flight publishes its record / its sums itself */
    /* the Metropolis test from a screened pass over the parameter DIFFERENCE (ig_kernels_nuis.cuh) */
    struct DiffConst* diff_const;
    long long* scratch_diff;      /* its 8 output words */
    long long* tile_partial0;     /* k_tile_trans: the histogram sums under the model's current set */
    int diff_seq;
    bool nuis_diff;               /* the step in flight ran the screened pass (the exact one only if it does not decide) */
    bool nuis_exact_queued;       /* ... and the exact pass behind it already (verify mode) */
    bool nuis_screen_rejected;    /* the last ig_nuis_end: rejected from the screened interval, no exact pass */
    /* a decisively accepted step: the exact pass (needed for the promotion of the maintained sum and for the returned likelihood,
     * not for the decision) runs on the side stream while the library stream promotes the parameters and re-scores the moves
     * ahead; the sums are promoted in front of the next kernel that reads them (flush_pending_sums) */
    long long* scratch_exact;
    hipEvent_t ev_exact;
    bool nuis_sums_pending, nuis_accept_certain;
    int exact_seq;
    bool nuis_nzb_copied;         /* the step's move was finished outside the batch commit: its NuisHost.nzb was copied from the control block */
    /* tier 0 of the screened pass: the histogram of the cis contacts' distances (NuisHist), valid for the state before the last
     * move of a run once nh_pending_slot's move has been walked (nh_flush_pending) */
    struct ScreenSum* probe_scr; /* IG_SCREEN_PROBE: scratch outputs of the probe launches of k_screen */
    unsigned* probe_void;
    struct NuisWorker* worker; /* the helper thread that enqueues a run's next step (ig_hip.hip) */
    bool last_moved;   /* the move of the step just ended changed the genome (or nobody said it did not) */
    long long n_accepts;
    NuisHist nh;
    long long* scratch_hist; /* k_hist_eval's 8 output words (zero between two launches) */
    bool nh_valid;
    int nh_pending_slot;     /* the slot of the last move of the run, not yet in the histogram (-1: none) */
    int nuis_tier;           /* the step in flight: 0 the histogram decided / is deciding, 1 the pass over the contacts */
    bool nuis_tiles_listed;  /* ... k_tile_trans has run (the passes over the tiles need its list) */
    hipEvent_t ev_walk;
    double nhs[12];          /* statistics: evaluations, rejected / accepted there, void, sum of bounds, largest used fraction, walks, builds, void because of {parameters, a contact, sums, no record} */
    bool nh_tracking;        /* a step of a run is being enqueued / ended: the moves applied now are followed by the histogram */
    bool nh_policy_on;       /* the histogram tier is worth its walks at the moment (nuis_hist_usable's cost model) */
    double nh_p_changed;     /* moving average: share of a run's moves that change the genome */
    double nscr[12];               /* statistics: steps screened, rejected from the interval, exact passes, void, largest bound, largest used fraction, sum of bounds */
    bool nuis_in_flight;
    bool side_busy;      /* launch_full_nz on a side stream: the library stream is busy with a batch (one workgroup per CU for the pass) */
    bool tail_fused;     /* the batch in flight: the Q5 tail walk ran inside the screening kernel's launch (no second stream, no events) */
    bool no_predict;     /* enqueue_score: no k_predict / predicted k_delta for this batch */
    bool main_drained;   /* nothing is queued on the library stream (end of a run's step, until something is enqueued there) */
    bool nuis_caught_up; /* tab_prev is the state before the next move already and ev_gathered recorded (ig_nuis_step_next) */
    double nuis_wait_s; /* time ig_nuis_end spent waiting for the device (ig_debug_nuis_wait) */
    /* moves of a run of (move, nuisance step) pairs scored ahead in batches (ig_nuis_run_begin / ig_nuis_step_begin): the batch
     * in the buffers starts at move spec_base, has spec_W slots of which [0, spec_next) are decided; spec_valid: its
     * undecided slots were scored under the model's current parameters */
    bool nuis_spec, spec_valid, spec_prev_pending;
    int spec_base, spec_W, spec_next, spec_move, spec_slot;
    /* the parameter-dependent half (screening, exact kernel, records) of the batch's slots [spec_par_begin, spec_par_end) is
     * valid; the structural half (windows, candidate genomes, slice lists) of all spec_W slots: an accepted step only voids
     * the former (ig_kernels: k_rescore_prepare) */
    int spec_par_begin, spec_par_end;
    double spec_ema, spec_struct_ema; /* moves decided between two accepted steps / per structural batch: set the widths */
    int since_accept;                 /* moves decided since the last accepted step */
    ig_params nuis_test;           /* the test parameters of the step in flight */
    ig_params par_model;           /* host copy of the model's parameters (set 0): ig_set_params, ig_nuis_accept */
    float nuis_mean_kb;
    int N, M;
    long long Z;
    int max_count; /* largest contact count (packed slice entries need it below 2^24) */
    int rank, world;
    State st;
    int* st_block; /* one allocation for all state arrays */
    Tables tab, tab_prev;
    SubTab* sub_tab;
    long long* rowptr;
    int2* cc; /* (col, count) */
    int* crow;    /* row of every contact (COO companion of cc: k_full_nz is contact-parallel) */
    int4* tabrec; /* k_pack_tab: (dist, s_tot, contig, rank) per sub-fragment */
    /* the tiled copy of the contacts k_full_nz_tiled streams (ig_upload_contacts), nullptr: not built */
    uint2* tiled_cc;
    struct TileWork* tile_work;
    int n_tile_work;
    long long* tile_trace; /* ig_debug_tile_trace */
    long long* diff_trace; /* ig_debug_diff_trace */
    int n_tile_static, n_tile_info; /* work items k_full_nz_tiled is launched over; off-diagonal tiles with a histogram */
    struct TileInfo* tile_info;
    long long* tile_partial;        /* k_tile_trans: one (hi, lo) pair per workgroup, summed by k_full_nz_tiled */
    int *tile_dyn, *tile_dyn_list;  /* {count, cursor} and the work items of the tiles whose contacts have to be read this pass */
    unsigned *tile_hist, *tile_sig; /* count histograms of the off-diagonal tiles (static); contig signatures of the blocks (per pass) */
    int* init_prev;
    int* init_next;
    int* orientable;
    unsigned char* black;
    double* lgf_tab;
    Glob* glob;
    long long* scratch8; /* 8 x int64 reduction scratch of the from-scratch passes */
    MoveBuf mb;
    int* batch_out; /* committed moves, pending slot, (unused), candidates, predicted deltas used, contigs */
    int *host_bo, *host_bo_dev; /* the same in mapped host memory (+ [7] = sequence number of the decide launch), and its device address */
    int bo_seq;
    int last_stop; /* what the last decided batch stopped at: 0 a conflict / its end, 1 the slice pool, 2 the exact kernel's grid, 3 a score of exactly 0.0 */
    bool exact_next; /* ... 3: the next scoring leaves the screening tier out (enqueue_score) */
    int nuis_one_C; /* candidates of the move ig_nuis_begin enqueued (repeated by ig_nuis_end if its lists did not fit the pool) */
    long long n_pool_retries; /* one-move path: moves repeated with a larger slice pool (retry_with_larger_pool; ig_debug_pool_retries) */
    /* chains (see ChainIn): per-set constants of a segment, the uploaded sets, the intervals, scratch words; statistics */
    struct ChainSet* chain_sets; /* [CHAIN_SEG] */
    ChainIn* chain_in;           /* [CHAIN_MAX] device */
    ChainIn* chain_in_host;      /* pinned staging */
    ChainTest* chain_tests;      /* [CHAIN_SEG] */
    long long *chain_out16, *chain_zs; /* [CHAIN_SEG][16], [CHAIN_SEG][8] */
    long long n_chain_calls, n_chain_segments, n_chain_pairs, n_chain_stops[8];
    int chain_done, chain_reason; /* of the last chain: pairs completed, why it ended */
    bool chain_busy;
    long long n_zero_fallbacks;
    int n_contigs_seen; /* contigs after the last batch (0: none yet): picks k_mutate's launch shape */
    double w_ema; /* moving average of the moves a batch gets through: sets the width of the next one */
    int* dirty_buf; /* [1 + 2 * IG_MAX_BATCH + 2] contigs modified by the committed moves of the batch in flight */
    int *own_tag, *own_idx; /* [N] which committed move of the current batch owns a fragment, and where in its window */
    ig_move_result* d_results;
    int results_cap;
    int* d_frags;
    int* d_cands;
    int cands_cap;
    int* prev_touched;
    unsigned* touched_bits; /* 2 x one bit per sub-fragment: in a window of the batch being sliced (set by k_gather, read by k_slice; the
                             * other half is cleared by the same k_gather for the next batch) */
    int touched_flip;
    unsigned timing_mask;
    int timing_every; /* ig_set_timer_sampling */
    struct ScoreConst* score_const; /* tables and constants k_score_list stages (parameter set 0) */
    struct ScoreConst* full_const;  /* the same for the parameter set a k_full_nz launch evaluates */
    struct ScreenConst* screen_const; /* what k_screen stages (parameter set 0) */
    double* screen_worst;             /* IG_SCREEN_VERIFY: {largest used fraction of a bound, largest bound} */
    long long n_screen_cols, n_screen_cont; /* diagnostics of the verify mode */
    float* pz_tab;  /* P_z table of parameter set 0 (the model in use) */
    int pz_n;
    float* pz_tab1; /* and of set 1 (the nuisance step's test parameters) */
    int pz_n1;
    /* timers */
    bool timing;
    struct Timer {
        const char* name;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        double total_ms;
        long long n;
        long long seen; /* launches since the timers were reset (timing_every) */
    } timers[13];
    std::vector<hipEvent_t> ev_pool; /* recycled timer events */
    long long n_batches, n_batch_committed, n_batch_pending, n_batch_predicted;
    int up_moves, up_max_c; /* the uploaded move lists */
    void* h_stage;          /* pinned staging of lists and result records (ensure_io) */
    size_t h_stage_bytes;
    int own_begin, own_end; /* slots whose candidate genomes this handle built for the batch in flight */
    int own_screened;       /* the batch in flight was scored in two tiers (1), or verified (2) */
    unsigned long long last_stale; /* the window rule: positions the last decide launch found stale (wait_commit) */
    long long n_window_slots;      /* ... slots scored by its launches */
    int max_L, max_SL;      /* host copies of Glob.max_L / max_SL as of the last synchronisation */
    int win_slots = 0;      /* physical slots the per-window arrays hold (ensure_window_buffers; <= mb.capW) */
    bool full_windows;      /* window strides = the whole genome (runs of moves enqueued one at a time without a host round trip) */
    int* host_max;          /* pinned: {max_L, max_SL} copied back with every one-move call's result */
    struct StepHost *host_step, *host_step_dev; /* ig_step_draw: mapped host memory and its device address (null: not available) */
    int step_seq;
    struct StepHost* pub_step; /* set while ig_step_draw enqueues: k_commit publishes there */
    double* pub_scores;        /* ... and the decide step of its one-move batch writes the move's scores there */
    bool screen_w1;            /* ... whose caller asked for no scores: two-tier scoring at width one (enqueue_score) */
    long long n_step_tail;     /* ig_step_draw calls finished by the one-move tail */
    int exact_grid;         /* two-tier scoring: blocks of the exact kernel's launch (follows what the last batches needed) */
    bool have_contacts, have_sub, have_state, have_init, have_params;
    bool init_links_inverse; /* initial prev / next are mutually inverse (k_commit_batch's de-duplication relies on it; else W = 1) */
};

__host__ __device__ inline size_t rec_bytes_per_slot(int capC) { return (size_t)capC * (IG_N_TMP_STRUCT * sizeof(SlotPre) + sizeof(CandPre)); }
__device__ __forceinline__ SlotPre& pre_at(const MoveBuf& mb, int cw, int slot)
{
    const int w = cw / mb.capC, c = cw % mb.capC;
    return ((SlotPre*)(mb.rec + (size_t)w * mb.rec_stride))[c * IG_N_TMP_STRUCT + slot];
}
/* the same by (slot w, entry i = c * IG_N_TMP_STRUCT + column) / (slot w, candidate c): no division by the runtime capC */
/* (w: a position of the window, PS(w) its physical slot; pre_at / cpre_at take the physical cw = CW(w, c)) */
__device__ __forceinline__ SlotPre& pre_w(const MoveBuf& mb, int w, int i) { return ((SlotPre*)(mb.rec + (size_t)PS(w) * mb.rec_stride))[i]; }
__device__ __forceinline__ CandPre& cpre_w(const MoveBuf& mb, int w, int c)
{
    return ((CandPre*)(mb.rec + (size_t)PS(w) * mb.rec_stride + (size_t)mb.capC * IG_N_TMP_STRUCT * sizeof(SlotPre)))[c];
}
__device__ __forceinline__ CandPre& cpre_at(const MoveBuf& mb, int cw)
{
    const int w = cw / mb.capC, c = cw % mb.capC;
    return ((CandPre*)(mb.rec + (size_t)w * mb.rec_stride + (size_t)mb.capC * IG_N_TMP_STRUCT * sizeof(SlotPre)))[c];
}
