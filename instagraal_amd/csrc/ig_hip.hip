/*
 * ig_hip.hip -- MI355X (gfx950) implementation of instaGRAAL's per-move scoring path behind the
 * C ABI of include/instagraal_hip.h.  Design notes: DESIGN.md.  Reference being replaced:
 *   /root/reference/src/instagraal/kernels/kernel_sparse_adapt.cu  ("KA", 38 CUDA kernels)
 *   /root/reference/src/instagraal/cuda_lib_gl_single.py           ("CL", pycuda host code)
 *
 * One move (CL:1401-1465) is eight launches on one stream, no host round trip in between:
 *   k_gather   O(N)        local fragment lists of the touched contigs, uniq-mutation lists, flags
 *   k_mutate   C x 25 WGs  one candidate genome per workgroup, operators applied in place on the
 *                          local window, coordinate columns + zero-pixel sums emitted
 *   k_score    C x G WGs   CSR rows of the touched contigs streamed once per column, Rippe P(s),
 *                          Poisson term, exact fixed-point sums (wave shuffles + 64-bit atomics)
 *   k_finalize 1 WG        tail quirk (Q5), scores, argmax
 *   k_delta    G WGs       exact update of the full likelihood when the slice was windowed
 *   k_apply    WGs         winner scattered into the live state + coordinate tables
 *   k_post     O(N)        genome distance credits
 *   k_commit   1 lane      result record
 */
#include <hip/hip_runtime.h>

#include <algorithm>
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
#define SCORE_BLOCKS 64   /* workgroups per candidate in k_score */
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
    int n_prev_touched;
    int valid_insert[12];
    int error;
    int stamp_ctr;
};

/* one move slot of a batch (W = 1: the move in flight) */
#define IG_MAX_BATCH 64
struct MoveCtl {
    int A, C, force_slot, fresh; /* fresh: first of the NFRESH contig ids this move may create */
    int ch_c, ch_k, ch_slot, ch_windowed;
    int superset0; /* candidate 0 was scored with every insert slot (its stale flags were not known yet) */
    int overflow;  /* the slice pool could not hold this slot: it is re-run at the head of the next batch */
    int n_dirty, pad;
    double ch_score;
    long long n_slice_tot, n_eval_tot, bytes_min;
    long long d_hi, d_lo; /* k_delta accumulator */
};

/* what the commit step needs about one (candidate, mutation slot), written slot-major by k_prefinal */
struct SlotPre {
    long long nz_hi, nz_lo;     /* slice sum under this slot's genome (all sliced contacts) */
    long long tail_hi, tail_lo; /* the part quirk Q5 drops when the slot's list position is >= S_c mod 64 */
    long long dz_hi, dz_lo, dni; /* zero-pixel sum and intra pair count: this genome minus the current one, on the window */
    int k;                      /* coordinate column (0 = not scored) */
    int changed;                /* the mutated window differs from the current genome */
    int heads;                  /* contigs on the mutated window */
    int pad;
};
struct CandPre {
    long long ext_hi, ext_lo; /* slice sum under the current genome */
    long long n_slice;
    int r;                    /* S_c mod 64 */
    int base_cnt;             /* list entries before the block-insert slots */
    int n_uniq_basic;         /* == base_cnt (kept for the statistics) */
    int pad;
};

struct MoveBuf {
    int* Lloc;      /* [capW*capC][N] global ids of local fragments */
    int* lbloc;     /* [..][N] */
    int* slloc;     /* [..][N] */
    int* subs;      /* [..][M] global sub-frag id of local sub index */
    int* rowcnt;    /* [..][M] sliced contacts per local row */
    int* sl_li;     /* slice pool: candidate cw's list starts at slice_offset(w, c): local row index, */
    int* sl_lj;     /*           local column index, */
    int* sl_ob;     /*           observed count (order = arrival, sums are order-free) */
    long long* slbound; /* [..] upper bound of the list length = contacts in the rows of the touched contigs */
    long long* sloff;   /* [..] start of the list in the pool, -1 = does not fit (k_offsets) */
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
    SlotPre* pre;   /* [..][24] */
    CandPre* cpre;  /* [..] */
    int N, M, capC, capW;
};
/* layout of MoveBuf.part per candidate (int64 units) */
#define P_NZ 0                 /* [NSLOT][2] slice sums per column k (k=0: current = "extract") */
#define P_CNT (NSLOT * 2)      /* [1] kept entries S_c */
#define P_STRIDE (NSLOT * 2 + 2)
/* not all-reduced (computed redundantly on every rank) */
#define Q_Z 0                  /* [NSLOT][2] zero-pixel sums on the local window, per column k */
#define Q_NI (NSLOT * 2)       /* [NSLOT] intra pair counts */
#define Q_NZFULL (NSLOT * 3)   /* [NSLOT][2] slice sums before the tail correction */
#define Q_TAIL (NSLOT * 5)     /* [NSLOT][2] sum of the last S_c mod 64 sliced contacts' terms (quirk Q5) */
#define Q_STRIDE (NSLOT * 7)

struct ig_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    hipStream_t stream2;           /* k_tail next to k_score_list */
    hipEvent_t ev_slice, ev_tail;
    int N, M;
    long long Z;
    int rank, world;
    State st;
    int* st_block; /* one allocation for all state arrays */
    Tables tab, tab_prev;
    SubTab* sub_tab;
    long long* rowptr;
    int2* cc; /* (col, count) */
    int* init_prev;
    int* init_next;
    int* orientable;
    unsigned char* black;
    double* lgf_tab;
    Glob* glob;
    MoveBuf mb;
    int* stamp;     /* [N] claim stamps of the incremental genome distance */
    int* batch_out; /* [4] committed moves, pending slot, windows above LDS_COL_SMALL, candidates */
    int* dirty_buf; /* [1 + 2 * IG_MAX_BATCH + 2] contigs modified by the committed moves of the batch in flight */
    int *own_tag, *own_idx; /* [N] which committed move of the current batch owns a fragment, and where in its window */
    ig_move_result* d_results;
    int results_cap;
    int* d_frags;
    int* d_cands;
    int cands_cap;
    int* prev_touched;
    unsigned timing_mask;
    float* pz_tab;
    int pz_n;
    /* timers */
    bool timing;
    struct Timer {
        const char* name;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        double total_ms;
        long long n;
    } timers[10];
    long long n_batches, n_batch_committed, n_batch_pending;
    int large_seen;
    int up_moves, up_max_c; /* the uploaded move lists */
    bool have_contacts, have_sub, have_state, have_init, have_params;
};

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

/* expected contacts of one pair: KA:4430-4459 (== 4182-4206, 4327-4351) */
__device__ __forceinline__ void expected_pair(const ig_params& p, bool cis, float s, float s_z, float s_tot, float s_tot_z,
                                              float& ex, float& ex_z)
{
    if (cis) {
        if (s_tot == 0) {
            ex = ig_rippe(s, p, ig_tab());
            ex_z = (s_z < p.d_max) ? ig_rippe(s_z, p, ig_tab()) : p.v_inter;
        } else {
            ex = ig_rippe_circ(s, s_tot, p, ig_tab());
            ex_z = (s_z < p.d_max) ? ig_rippe_circ(s_z, s_tot_z, p, ig_tab()) : p.v_inter;
        }
    } else {
        ex = p.v_inter;
        ex_z = p.v_inter;
    }
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

__global__ void k_build_pz(const Glob* g, float* pz, int n)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    const ig_params p = g->par[0];
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

/* ------------------------------------------------------------------ set-up kernels */

__global__ void k_lgf_table(double* tab, const double* small15)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < LGF_TAB) tab[i] = ig_lgfact(i < 1 ? 1 : i, small15, ig_tab());
    if (i == 0) tab[0] = 0.0;
}

/* KA:3763-3822 for every sub-fragment */
__global__ void k_fill_tables(State st, const SubTab* __restrict__ sub, Tables t, int M)
{
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M) return;
    const SubTab b = sub[s];
    const int f = b.parent;
    const int ori = st.ori[f];
    const int sp = st.spos[f], sl = st.sl[f];
    const int stot_i = (int)((float)(st.circ[f] == 1) * (float)st.LB[f] / 1000.0f);
    const float dfi = (ori == 1) ? b.wat : b.cri;
    t.dist[s] = (float)st.sbp[f] / 1000.0f + dfi;
    t.stot[s] = (float)stot_i;
    t.cp[s] = make_int2(st.cid[f], (ori == 1) ? sp + b.w : sp + (sl - 1) - b.w);
    t.len[s] = st.SL[f];
}

/* evaluate_likelihood_sparse (KA:4374-4488) over the whole CSR, exact sums -> out[0..1] */
__global__ void k_full_nz(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables t, const Glob* g, int which,
                          const double* __restrict__ lgf_tab, int M, int rank, int world, long long* out)
{
    const ig_params p = g->par[which];
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const float mean = g->mean_kb;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    long long hi = 0, lo = 0;
    for (int i = wave; i < M; i += nwaves) {
        if (world > 1 && (i % world) != rank) continue;
        const long long b = rowptr[i], e = rowptr[i + 1];
        if (b == e) continue;
        const float di = t.dist[i], sti = t.stot[i];
        const int2 cpi = t.cp[i];
        const int ci = cpi.x, pi = cpi.y, li = t.len[i];
        for (long long k = b + lane; k < e; k += 64) {
            const int2 v = cc[k];
            const int j = v.x;
            const int2 cpj = t.cp[j];
            const float s = fabsf(di - t.dist[j]);
            const int dp = pi - cpj.y;
            const float s_z = (float)(dp < 0 ? -dp : dp) * mean;
            const long long q = ig_quantize(ig_pair_term(p, &hot, ci == cpj.x, s, s_z, sti, (float)li * mean, v.y, lgfact_dev(v.y, lgf_tab),
                                                         ig_tab()));
            hi += q >> 32;
            lo += (long long)(unsigned int)q;
        }
    }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    if (lane == 0) {
        atomic_add_ll(&out[0], hi);
        atomic_add_ll(&out[1], lo);
    }
}

/* eval_likelihood_on_zero (KA:3850-3917) over all sub-fragments -> out[0..2] = hi, lo, n_intra */
__global__ void k_full_zero(Tables t, const Glob* g, int which, int M, long long* out)
{
    const ig_params p = g->par[which];
    const float mean = g->mean_kb;
    long long hi = 0, lo = 0, ni = 0;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < M; s += gridDim.x * blockDim.x) {
        const int pos = t.cp[s].y, len = t.len[s];
        if (pos == 0) ni += ((long long)len * (long long)(len - 1)) / 2;
        if (pos > 0) {
            const long long q = zero_q(p, pos, len, t.stot[s], mean, nullptr, 0);
            hi += q >> 32;
            lo += (long long)(unsigned int)q;
        }
    }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    ni = wave_sum_ll(ni);
    if ((threadIdx.x & 63) == 0) {
        atomic_add_ll(&out[0], hi);
        atomic_add_ll(&out[1], lo);
        atomic_add_ll(&out[2], ni);
    }
}

__global__ void k_count_heads(State st, int N, int* out)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    int h = (f < N && st.pos[f] == 0) ? 1 : 0;
    h = wave_sum_i(h);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}

/* explode_genome (KA:409-426); internal contig id = fragment index (ori is NOT reset) */
__global__ void k_explode(State st, int N)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= N) return;
    st.pos[f] = 0;
    st.sbp[f] = 0;
    st.spos[f] = 0;
    st.cid[f] = f;
    st.prev[f] = -1;
    st.next[f] = -1;
    st.L[f] = 1;
    st.LB[f] = st.lb[f];
    st.SL[f] = st.sl[f];
}

/* dist_inter_genome (CL:665-716): credits in half units, summed exactly */
__device__ __forceinline__ int credit2_of(const State& st, const int* ip, const int* in, const int* orientable, int f)
{
    const int p0 = ip[f], n0 = in[f];
    int p1 = st.prev[f], n1 = st.next[f];
    const int o1 = st.ori[f];
    int c2 = 0;
    if (((p1 == p0) && (n1 == n0)) || ((p1 == n0) && (n1 == p0))) c2 += 2;
    if (orientable[f]) {
        int swap = 1;
        if (1 != o1) { /* init ori is +1 (CL:276) */
            int t = p1;
            p1 = n1;
            n1 = t;
            swap = -1;
        }
        if (p0 == p1) {
            if (p0 == -1) c2 += 2;
            else if (!orientable[p1]) c2 += 2;
            else c2 += 1 + ((1 == swap * st.ori[p1]) ? 1 : 0);
        }
        if (n0 == n1) {
            if (n0 == -1) c2 += 2;
            else if (!orientable[n1]) c2 += 2;
            else c2 += 1 + ((1 == swap * st.ori[n1]) ? 1 : 0);
        }
    } else {
        if ((p1 == p0) || (p1 == n0)) c2 += 2;
        if ((n1 == n0) || (n1 == p0)) c2 += 2;
    }
    return c2;
}

__global__ void k_post(State st, const int* __restrict__ ip, const int* __restrict__ in, const int* __restrict__ orientable,
                       const unsigned char* __restrict__ black, Glob* g, int N)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    int c2 = 0;
    if (f < N && !black[f]) c2 = credit2_of(st, ip, in, orientable, f);
    c2 = wave_sum_i(c2);
    if ((threadIdx.x & 63) == 0 && c2) atomic_add_ll(&g->credit2_acc, (long long)c2);
}

/* ------------------------------------------------------------------ the move(s)
 *
 * Candidate draws do not depend on the genome (CL:3103-3141 reads fixed distributions), so W consecutive
 * moves can be SCORED against the same base state in single launches (slot dimension w below) and then
 * COMMITTED in order by one workgroup (k_commit_batch) that stops at the first move whose contigs were
 * modified by an earlier move of the batch.  W = 1 is the plain one-move-at-a-time path.
 * Buffers of candidate c of slot w live at index cw = w * capC + c. */

#define CW(w, c) ((w) * mb.capC + (c))

/* uniq-mutation list of extract_uniq_mutations (KA:4492-4553); vf == nullptr -> every insert slot (superset) */
__device__ inline int build_uniq(int* u, bool first, int LA, int LB, const int* vf)
{
    int n = 0;
    if (first) {
        u[n++] = 0;
        u[n++] = 1;
    }
    u[n++] = 2;
    u[n++] = 3;
    if (LB != 1)
        for (int k = 4; k < 8; k++) u[n++] = k;
    if (LA != 1)
        for (int k = 8; k < 12; k++) u[n++] = k;
    for (int k = 12; k < IG_N_TMP_STRUCT; k++)
        if (!vf || vf[k - 12] != -1) u[n++] = k;
    return n;
}

/* k_gather: every fragment of a touched contig drops itself at its rank (no compaction needed);
 * block w also derives the metadata of move slot w: get_bounds flags (KA:2124-2252), slice windows
 * (KA:530-548) and the uniq-mutation lists with the STALE flags of quirk Q4.  For slots w > 0 the flags
 * the first candidate will see depend on the outcome of move w-1, so that candidate is scored with the
 * superset list and the commit step selects the actual one. */
__global__ void __launch_bounds__(256)
    k_gather(State st, Glob* g, MoveBuf mb, const int* __restrict__ cands_all, const int* __restrict__ frags_all, int move0, int W,
             int max_c, Tables tab, Tables tab_prev, const int* __restrict__ prev_touched, int force_slot)
{
    /* tab_prev := coordinates before the LAST applied move (eval_likelihood_4_nuisance reads tables that
     * were filled before the move was applied, CL:1296-1344 / quirk Q12): catch up the entries that move touched */
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g->n_prev_touched; i += gridDim.x * blockDim.x) {
        const int s = prev_touched[i];
        tab_prev.dist[s] = tab.dist[s];
        tab_prev.stot[s] = tab.stot[s];
        tab_prev.cp[s] = tab.cp[s];
        tab_prev.len[s] = tab.len[s];
    }
    __shared__ int sh_cA[IG_MAX_BATCH], sh_LA[IG_MAX_BATCH], sh_C[IG_MAX_BATCH];
    __shared__ int sh_cB[IG_MAX_BATCH * IG_MAX_CANDIDATES];
    __shared__ int sh_flags[IG_MAX_CANDIDATES][12];
    const int N = mb.N;
    for (int i = threadIdx.x; i < W; i += blockDim.x) {
        const int A = frags_all[move0 + i];
        sh_cA[i] = st.cid[A];
        sh_LA[i] = st.L[A];
        int C = 0;
        for (int q = 0; q < max_c; q++) C += (cands_all[(size_t)(move0 + i) * max_c + q] >= 0);
        sh_C[i] = C;
    }
    for (int i = threadIdx.x; i < W * max_c; i += blockDim.x) {
        const int b = cands_all[(size_t)move0 * max_c + i];
        sh_cB[(i / max_c) * IG_MAX_CANDIDATES + (i % max_c)] = b >= 0 ? st.cid[b] : -1;
    }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < N) {
        const int cf = st.cid[f], pf = st.pos[f];
        const int lb = st.lb[f], sl = st.sl[f];
        for (int w = 0; w < W; w++) {
            const int cA = sh_cA[w], LA = sh_LA[w], C = sh_C[w];
            for (int c = 0; c < C; c++) {
                int slot = -1;
                if (cf == cA) slot = pf;
                else if (cf == sh_cB[w * IG_MAX_CANDIDATES + c]) slot = LA + pf;
                if (slot >= 0) {
                    const size_t o = (size_t)CW(w, c) * N + slot;
                    mb.Lloc[o] = f;
                    mb.lbloc[o] = lb;
                    mb.slloc[o] = sl;
                }
            }
        }
    }
    const int w = blockIdx.x;
    if (w >= W) return;
    const int t = threadIdx.x;
    const int A = frags_all[move0 + w];
    const int* cands = cands_all + (size_t)(move0 + w) * max_c;
    const int C = sh_C[w];
    const int cA = sh_cA[w], LA = sh_LA[w];
    if (t == 0) {
        MoveCtl mc;
        mc.A = A;
        mc.C = C;
        mc.force_slot = force_slot;
        mc.fresh = g->next_cid + NFRESH * w;
        mc.ch_c = mc.ch_k = mc.ch_slot = mc.ch_windowed = 0;
        mc.ch_score = 0.0;
        mc.n_slice_tot = mc.n_eval_tot = mc.bytes_min = 0;
        mc.d_hi = mc.d_lo = 0;
        mc.superset0 = (w > 0 && force_slot < 0) ? 1 : 0;
        mc.n_dirty = 0;
        mc.overflow = 0;
        mc.pad = 0;
        mb.ctl[w] = mc;
    }
    for (int i = t; i < C * P_STRIDE; i += blockDim.x) mb.part[(size_t)CW(w, 0) * P_STRIDE + i] = 0;
    for (int i = t; i < C * Q_STRIDE; i += blockDim.x) mb.qpart[(size_t)CW(w, 0) * Q_STRIDE + i] = 0;
    for (int i = t; i < C * IG_N_TMP_STRUCT; i += blockDim.x) mb.scores[(size_t)CW(w, 0) * IG_N_TMP_STRUCT + i] = 0.0;
    if (t < C) {
        CandMeta m;
        const int B = cands[t];
        m.B = B;
        m.ctgA = cA;
        m.ctgB = st.cid[B];
        m.same = (m.ctgA == m.ctgB);
        m.LA = LA;
        m.LB = st.L[B];
        m.SLA = st.SL[A];
        m.SLB = st.SL[B];
        m.n_loc = m.same ? m.LA : m.LA + m.LB;
        m.m_loc = m.same ? m.SLA : m.SLA + m.SLB;
        m.lA = st.pos[A];
        m.lB = (m.same ? 0 : m.LA) + st.pos[B];
        /* slice windows, KA:530-548 */
        const int sa = st.spos[A], sb = st.spos[B], oa = st.ori[A], ob = st.ori[B];
        const int sla = st.sl[A], slb = st.sl[B];
        m.pos_fa = max(0, sa * (oa == 1) + (sa - sla) * (oa == -1));
        m.pos_fb = max(0, sb * (ob == 1) + (sb - slb) * (ob == -1));
        m.up_fa = max(0, m.pos_fa - g->slice_nb - sla);
        m.down_fa = min(m.SLA - 1, m.pos_fa + g->slice_nb + sla);
        m.up_fb = max(0, m.pos_fb - slb);
        m.down_fb = min(m.SLB - 1, m.pos_fb + slb);
        m.windowed = m.same && (st.circ[A] == 0);
        /* a window that spans the whole contig keeps every pair: the slice is then the full contig */
        if (m.windowed && ((m.up_fa == 0 && m.down_fa == m.SLA - 1) || (m.up_fb == 0 && m.down_fb == m.SLA - 1))) m.windowed = 0;
        bounds_scalar(st, g, A, B, m.pos_up, m.pos_down, m.flags);
        for (int i = 0; i < 12; i++) sh_flags[t][i] = m.flags[i];
        mb.meta[CW(w, t)] = m;
    }
    __syncthreads();
    if (t < C) {
        CandMeta* m = &mb.meta[CW(w, t)];
        int n = 0;
        int* u = m->uniq;
        for (int k = 0; k < NSLOT; k++) m->kidx[k] = -1;
        if (force_slot >= 0) {
            u[n++] = force_slot;
        } else if (t == 0) {
            n = build_uniq(u, true, m->LA, m->LB, (w == 0) ? g->valid_insert : nullptr);
        } else {
            n = build_uniq(u, false, m->LA, m->LB, sh_flags[t - 1]);
        }
        m->n_uniq = n;
        m->kidx[IG_N_TMP_STRUCT] = 0; /* current genome = column 0 */
        for (int k = 0; k < n; k++) m->kidx[u[k]] = k + 1;
    }
}

/* k_mutate: one workgroup = one candidate genome on the local window. */
__global__ void __launch_bounds__(256) k_mutate(State st, Tables tab, const SubTab* __restrict__ sub, const long long* __restrict__ rowptr,
                                                Glob* g, MoveBuf mb, PzTab pz)
{
    const int slot = blockIdx.x, c = blockIdx.y, w = blockIdx.z;
    const MoveCtl& mc = mb.ctl[w];
    if (c >= mc.C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int k = m.kidx[slot];
    if (k < 0) return;
    const int N = mb.N, M = mb.M, n = m.n_loc;
    int* base = mb.loc + ((size_t)(cw * NSLOT + slot) * NDYN) * N;
    igd::Loc S;
    S.pos = base;
    S.spos = base + (size_t)N;
    S.cid = base + (size_t)2 * N;
    S.sbp = base + (size_t)3 * N;
    S.circ = base + (size_t)4 * N;
    S.prev = base + (size_t)5 * N;
    S.next = base + (size_t)6 * N;
    S.L = base + (size_t)7 * N;
    S.SL = base + (size_t)8 * N;
    S.LB = base + (size_t)9 * N;
    S.ori = base + (size_t)10 * N;
    S.gid = mb.Lloc + (size_t)cw * N;
    S.lb = mb.lbloc + (size_t)cw * N;
    S.sl = mb.slloc + (size_t)cw * N;
    S.n = n;
    for (int x = threadIdx.x; x < n; x += blockDim.x) {
        const int f = S.gid[x];
        S.pos[x] = st.pos[f];
        S.spos[x] = st.spos[f];
        S.cid[x] = st.cid[f];
        S.sbp[x] = st.sbp[f];
        S.circ[x] = st.circ[f];
        S.prev[x] = st.prev[f];
        S.next[x] = st.next[f];
        S.L[x] = st.L[f];
        S.SL[x] = st.SL[f];
        S.LB[x] = st.LB[f];
        S.ori[x] = st.ori[f];
    }
    __syncthreads();
    const int A = m.lA, B = m.lB;
    const int fresh = mc.fresh;
    if (slot == 0) { /* CL:1672 */
        igd::op_pop_out(S, A, fresh);
    } else if (slot == 1) { /* CL:1680 */
        igd::op_flip(S, A);
    } else if (slot < 8) { /* CL:1689-1760 */
        igd::op_pop_out(S, A, fresh);
        const int ori = (slot & 1) ? -1 : 1;
        if (slot < 4) igd::op_pop_in_1(S, A, B, fresh + 1, ori);
        else if (slot < 6) igd::op_pop_in_2(S, A, B, fresh + 1, ori);
        else igd::op_pop_in_3(S, A, B, ori);
    } else if (slot < 12) { /* CL:1780-1841: (upA, upB) = (0,0),(0,1),(1,0),(1,1) */
        igd::op_split(S, A, (slot - 8) >> 1, fresh);
        igd::op_split(S, B, (slot - 8) & 1, fresh + 1);
        igd::op_paste(S, A, B);
    } else if (slot < IG_N_TMP_STRUCT) { /* CL:1843-1916: slot = 12 + 2 i + (j == 1 ? 0 : 1) */
        const int i = (slot - 12) >> 1;
        const int up = ((slot - 12) & 1) ? 0 : 1;
        const int cutpos = up ? m.pos_up[i] : m.pos_down[i];
        const int g_ext = cutpos >= 0 ? S.gid[cutpos] : -1;
        igd::op_extract_block(S, A, cutpos, up, fresh);
        igd::op_insert_block(S, A, B, g_ext, m.flags[slot - 12], up);
    }
    /* ---- does this slot change the genome at all, and how many contigs does the window hold afterwards */
    __syncthreads();
    {
        int ch = 0, hd = 0;
        for (int x = threadIdx.x; x < n; x += blockDim.x) {
            const int f = S.gid[x];
            ch |= (S.pos[x] != st.pos[f]) | (S.spos[x] != st.spos[f]) | (S.cid[x] != st.cid[f]) | (S.sbp[x] != st.sbp[f]) |
                  (S.circ[x] != st.circ[f]) | (S.prev[x] != st.prev[f]) | (S.next[x] != st.next[f]) | (S.L[x] != st.L[f]) |
                  (S.SL[x] != st.SL[f]) | (S.LB[x] != st.LB[f]) | (S.ori[x] != st.ori[f]);
            hd += (S.pos[x] == 0);
        }
        __shared__ int sh_ch, sh_hd;
        if (threadIdx.x == 0) {
            sh_ch = 0;
            sh_hd = 0;
        }
        __syncthreads();
        hd = wave_sum_i(hd);
        if ((threadIdx.x & 63) == 0 && hd) atomicAdd(&sh_hd, hd);
        if (ch) atomicOr(&sh_ch, 1);
        __syncthreads();
        if (threadIdx.x == 0) mb.sinfo[cw * NSLOT + slot] = make_int2(sh_ch, sh_hd);
    }
    /* ---- coordinate column k (fill_vect_dist, KA:3699-3760) + zero-pixel sums on the window */
    const ig_params p = g->par[0];
    const float mean = g->mean_kb;
    uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
    ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    int* subs = mb.subs + (size_t)cw * M;
    long long hi = 0, lo = 0, ni = 0, bound = 0;
    for (int x = threadIdx.x; x < n; x += blockDim.x) {
        const int f = S.gid[x];
        const int cid = S.cid[x];
        const int code = (cid == m.ctgA) ? 0 : ((cid == m.ctgB) ? 1 : 2 + (cid - fresh));
        const int ori = S.ori[x], sp = S.spos[x], sl = S.sl[x], SLc = S.SL[x];
        const float stot = (float)(int)((float)S.circ[x] * (float)S.LB[x] / 1000.0f);
        if (S.pos[x] == 0) {
            cm[code].stot = stot;
            cm[code].len = SLc;
        }
        const float sbp_kb = (float)S.sbp[x] / 1000.0f;
        const int sf = st.sub_first[f];
        const int lbase = (x < m.LA) ? 0 : m.SLA;
        for (int q = 0; q < sl; q++) {
            const int s = sf + q;
            const SubTab b = sub[s];
            const float dist = sbp_kb + ((ori == 1) ? b.wat : b.cri);
            const int npos = (ori == 1) ? sp + q : sp + sl - (q + 1);
            const int ls = lbase + tab.cp[s].y;
            uint2 v;
            v.x = __float_as_uint(dist);
            v.y = (unsigned)npos | ((unsigned)code << 28);
            col[ls] = v;
            if (k == 0) {
                subs[ls] = s;
                bound += rowptr[s + 1] - rowptr[s]; /* upper bound of this candidate's slice */
            }
            if (npos == 0) ni += ((long long)SLc * (long long)(SLc - 1)) / 2;
            if (npos > 0) {
                const long long q2 = zero_q(p, npos, SLc, stot, mean, pz.v, pz.n);
                hi += q2 >> 32;
                lo += (long long)(unsigned int)q2;
            }
        }
    }
    __shared__ long long red[4][4];
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    ni = wave_sum_ll(ni);
    bound = wave_sum_ll(bound);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
        red[2][wv] = ni;
        red[3][wv] = bound;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long* q = mb.qpart + (size_t)cw * Q_STRIDE;
        q[Q_Z + 2 * k] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        q[Q_Z + 2 * k + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        q[Q_NI + k] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        if (k == 0) mb.slbound[cw] = red[3][0] + red[3][1] + red[3][2] + red[3][3];
    }
}

/* k_offsets: where each candidate's slice list starts in the pool = exclusive prefix sum of the upper bounds
 * (one small workgroup; a slot whose lists do not fit is flagged and re-run at the head of the next batch) */
__global__ void __launch_bounds__(64) k_offsets(MoveBuf mb, int W, int w_begin, int w_end)
{
    /* one wave: lane l owns the `per` consecutive (slot, candidate) entries l*per .. ; exclusive scan across lanes */
    const int lane = threadIdx.x;
    const int n = W * mb.capC;
    const int per = (n + 63) / 64;
    long long b[(IG_MAX_BATCH * IG_MAX_CANDIDATES + 63) / 64];
    long long sum = 0;
#pragma unroll
    for (int q = 0; q < (IG_MAX_BATCH * IG_MAX_CANDIDATES + 63) / 64; q++) {
        const int i = lane * per + q;
        long long v = 0;
        if (q < per && i < n) {
            const int w = i / mb.capC, c = i % mb.capC;
            if (w >= w_begin && w < w_end && c < mb.ctl[w].C) v = mb.slbound[i];
        }
        b[q] = v;
        sum += v;
    }
    long long incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    long long run = incl - sum;
#pragma unroll
    for (int q = 0; q < (IG_MAX_BATCH * IG_MAX_CANDIDATES + 63) / 64; q++) {
        const int i = lane * per + q;
        if (q < per && i < n) {
            const int w = i / mb.capC, c = i % mb.capC;
            if (w >= w_begin && w < w_end && c < mb.ctl[w].C) {
                if (run + b[q] > mb.pool_cap) {
                    mb.sloff[i] = -1;
                    mb.ctl[w].overflow = 1;
                } else {
                    mb.sloff[i] = run;
                }
            }
            run += b[q];
        }
    }
}

#define LDS_COL_CAP 4096  /* local sub-fragments whose column fits the 32 KB LDS stage */
#define DELTA_RB 128

/* k_slice: slice_sp_mat (KA:485-607) restricted to the CSR rows of the touched contigs (instead of a scan of
 * all Z contacts).  One wave per row: up to SLICE_UNROLL x 64 contacts are loaded back to back (coalesced
 * 8-byte loads, then one packed (contig, rank) gather each), the predicate is evaluated, and the kept ones
 * are appended to the candidate's list with ONE wave-aggregated atomic per batch (ballot + popcount ranks).
 * No sort afterwards: the reference sorted by row only to feed its shared-memory row cache (CL:1045-1050). */
#define SLICE_RB 128
#define SLICE_UNROLL 4
__global__ void __launch_bounds__(256) k_slice(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab,
                                               Glob* g, MoveBuf mb, int rank, int world, int w_begin)
{
    const int c = blockIdx.y, w = w_begin + blockIdx.z;
    if (c >= mb.ctl[w].C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int M = mb.M, m_loc = m.m_loc;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const long long off = mb.sloff[cw];
    if (off < 0) return; /* slice pool exhausted */
    const int* subs = mb.subs + (size_t)cw * M;
    int* rowcnt = mb.rowcnt + (size_t)cw * M;
    int* sli = mb.sl_li + off;
    int* slj = mb.sl_lj + off;
    int* slo = mb.sl_ob + off;
    unsigned long long* cursor = (unsigned long long*)(mb.part + (size_t)cw * P_STRIDE + P_CNT);
    const int nrw = gridDim.x * 4;
    for (int r = blockIdx.x * 4 + wv; r < m_loc; r += nrw) {
        const int i = subs[r];
        const long long b = rowptr[i], e = rowptr[i + 1];
        const bool mine = (world <= 1) || ((r % world) == rank);
        int rc = 0;
        if (b != e) {
            const int2 cp1 = tab.cp[i];
            for (long long q0 = b; q0 < e; q0 += 64 * SLICE_UNROLL) {
                int2 v[SLICE_UNROLL], cp2[SLICE_UNROLL];
                bool keep[SLICE_UNROLL];
                unsigned long long mask[SLICE_UNROLL];
#pragma unroll
                for (int u = 0; u < SLICE_UNROLL; u++) {
                    const long long qi = q0 + u * 64 + lane;
                    v[u] = (qi < e) ? cc[qi] : make_int2(-1, 0);
                }
#pragma unroll
                for (int u = 0; u < SLICE_UNROLL; u++) cp2[u] = (v[u].x >= 0) ? tab.cp[v[u].x] : make_int2(-1, -1);
                int add = 0;
#pragma unroll
                for (int u = 0; u < SLICE_UNROLL; u++) {
                    keep[u] = (v[u].x >= 0) && slice_keep(m, cp1.x, cp2[u].x, cp1.y, cp2[u].y, v[u].y, false);
                    mask[u] = __ballot(keep[u]);
                    add += __popcll(mask[u]);
                }
                if (add) {
                    rc += add;
                    if (mine) {
                        unsigned long long base = 0;
                        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)add);
                        base = __shfl(base, 0, 64);
                        int o2 = 0;
#pragma unroll
                        for (int u = 0; u < SLICE_UNROLL; u++) {
                            if (keep[u]) {
                                const long long at = (long long)base + o2 + __popcll(mask[u] & lt_mask);
                                sli[at] = r;
                                slj[at] = ((m.same || cp2[u].x == m.ctgA) ? 0 : m.SLA) + cp2[u].y;
                                slo[at] = v[u].y;
                            }
                            o2 += __popcll(mask[u]);
                        }
                    }
                }
            }
        }
        if (lane == 0) rowcnt[r] = rc; /* every rank knows every row's count: the tail walk needs them */
    }
}

#define SCORE_EB 16
#define LDS_PZ 1024
#define LDS_LGF 256

/* general (checked) evaluation of a linear-cis / trans pair, out of line: counts >= LDS_LGF, rank distances beyond the
 * LDS P_z table, parameters outside the one-log domain */
__device__ __noinline__ double term_general(const ig_params p, float mean_kb, float s, int dkey, int ob, double lgf, PzTab pz)
{
    const ig_hot h = ig_hot_make(p, ig_tab()); /* rare path: recomputed rather than passed */
    const int inter = dkey < 0;
    const float ex_z = inter ? p.v_inter : pz_lookup(pz, p, mean_kb, dkey);
    if (h.fast && ob > 0) return ig_term_hot(s, inter, ob, lgf, ex_z, &h, ig_tab());
    const float ex = inter ? p.v_inter : ig_rippe(s, p, ig_tab());
    return ig_pixel_term(ex, ex_z, ob, lgf, ig_tab());
}

/* q = ig_quantize(t) as (q >> 32, (uint32) q): the same integer, split without 64-bit conversions */
__device__ __forceinline__ void quantize_split(double t, int& qh, unsigned& ql)
{
    t = (t != t) ? 0.0 : t;
    t = __builtin_fmin(__builtin_fmax(t, -IG_QCLAMP), IG_QCLAMP); /* t is a number here: same as the two compares */
    const double Q = __builtin_rint(t * IG_QSCALE);
    const double H = __builtin_floor(Q * (1.0 / IG_QSCALE));
    qh = (int)H;
    ql = (unsigned)ig_fma(H, -IG_QSCALE, Q);
}

/* circular contigs (rare): the general functions, out of line */
__device__ __noinline__ double term_circ(const ig_params p, float mean_kb, float s, float s_tot, int d, int len_j, int ob, double lgf)
{
    float ex, ex_z;
    expected_circ(p, mean_kb, s, s_tot, d, len_j, &ex, &ex_z);
    return ig_pixel_term(ex, ex_z, ob, lgf, ig_tab());
}

/* dkey: rank distance d of a linear cis pair; -1 for a trans pair; d | code << 27 | 1 << 30 for a pair on a circular contig.
 * The hot case is the contract's ig_term_hot (one log2, one exp2) with the count's log-factorial and P_z from the LDS
 * tables; everything else (circular contig, count >= 256, rank distance beyond the LDS table, parameters outside the
 * one-log domain) is fixed up afterwards behind a wave-uniform branch that is almost never taken. */
#define DKEY_CIRC 0x40000000
__device__ __forceinline__ void term_hot(const ig_hot& h, const ig_params& p, float mean_kb, float s, int dkey, int ob,
                                         const float* pz_s, int pzn_s, const PzTab& pz, const double* lgf_s,
                                         const double* __restrict__ lgf_tab, const ColMeta* cm_s, const double* T, int& qh,
                                         unsigned& ql)
{
    const bool inter = dkey < 0;
    double lgf = lgf_s[min(ob, LDS_LGF - 1)];
    float ex_z = pz_s[min(max(dkey, 0), pzn_s - 1)];
    /* a P_z table shorter than PZ_MAX ends where s_z reaches d_max (ig_set_params): beyond it P_z is the trans level */
    const bool past_table = (pz.n < PZ_MAX) && (dkey >= pz.n) && !(dkey & DKEY_CIRC);
    ex_z = (inter || past_table) ? h.v_inter : ex_z;
    double t = ig_term_hot(s, inter, ob, lgf, ex_z, &h, T);
    const bool rare = (ob >= LDS_LGF) || (dkey >= pzn_s && !past_table) || !h.fast || (ob <= 0);
    if (__any(rare)) {
        if (rare) {
            if (ob >= LDS_LGF) lgf = lgfact_dev(ob, lgf_tab);
            if (!inter && (dkey & DKEY_CIRC)) {
                const int code = (dkey >> 27) & 7;
                t = term_circ(p, mean_kb, s, cm_s[code].stot, dkey & 0x07ffffff, cm_s[code].len, ob, lgf);
            } else {
                t = term_general(p, mean_kb, s, dkey, ob, lgf, pz);
            }
        }
    }
    quantize_split(t, qh, ql);
}

/* classification of one slice entry under one coordinate column: what its term is computed from */
__device__ __forceinline__ void classify_pair(uint2 ai, uint2 bj, unsigned circ_mask, float& sv, int& dkey)
{
    const unsigned ci = ai.y >> 28, cj = bj.y >> 28;
    const int pi = (int)(ai.y & 0x0fffffffu), pj = (int)(bj.y & 0x0fffffffu);
    const bool cis = ci == cj;
    sv = cis ? fabsf(__uint_as_float(ai.x) - __uint_as_float(bj.x)) : 0.0f;
    dkey = cis ? (pi > pj ? pi - pj : pj - pi) : -1;
    if (cis && ((circ_mask >> ci) & 1u)) dkey = (dkey & 0x07ffffff) | ((int)ci << 27) | DKEY_CIRC;
}

struct ScoreArgs {
    const int *sli, *slj, *slo; /* the candidate's slice list (one candidate: < 2^31 entries, 32-bit offsets from a uniform base) */
    unsigned n;
    const uint2* gcol; /* column k in global memory */
    const uint2* lcol; /* and its LDS copy */
    const float* pz_s;
    const double *lgf_s, *mt_s;
    const ColMeta* cm_s;
    const double* lgf_tab;
    PzTab pz;
    int pzn;
    unsigned circ_mask;
    float mean;
    int ablate;
};

/* the streaming loop of k_score_list; STAGED: the column fits the LDS stage (ds_read), else 8-byte gathers from L2 */
#define SCORE_BATCH 4
template <bool STAGED>
__device__ __forceinline__ void score_loop(const ScoreArgs& a, const ig_hot& hp, const ig_params& p, long long& hi, long long& lo)
{
    const unsigned stride = gridDim.x * SCORE_THREADS;
    for (unsigned e0 = blockIdx.x * SCORE_THREADS + threadIdx.x; e0 < a.n; e0 += stride * SCORE_BATCH) {
        int li[SCORE_BATCH], lj[SCORE_BATCH], ob[SCORE_BATCH];
#pragma unroll
        for (int u = 0; u < SCORE_BATCH; u++) {
            const unsigned e = e0 + u * stride;
            const bool ok = e < a.n;
            li[u] = ok ? a.sli[e] : -1;
            lj[u] = ok ? a.slj[e] : 0;
            ob[u] = ok ? a.slo[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < SCORE_BATCH; u++) { /* unrolled: the batch stays in registers */
            const bool valid = li[u] >= 0; /* lanes past the end evaluate entry 0 and drop the result */
            if (!__any(valid)) break;      /* wave-uniform */
            const int l_i = valid ? li[u] : 0, l_j = lj[u], o_b = valid ? ob[u] : 1;
            const uint2 ai = STAGED ? a.lcol[l_i] : a.gcol[l_i];
            const uint2 bj = STAGED ? a.lcol[l_j] : a.gcol[l_j];
            float sv;
            int dkey;
            classify_pair(ai, bj, a.circ_mask, sv, dkey);
            int qh;
            unsigned ql;
            if (a.ablate & 1) {
                qh = (int)__float_as_uint(sv) >> 12;
                ql = (unsigned)(dkey + o_b);
            } else {
                term_hot(hp, p, a.mean, sv, dkey, o_b, a.pz_s, a.pzn, a.pz, a.lgf_s, a.lgf_tab, a.cm_s, a.mt_s, qh, ql);
            }
            hi += valid ? qh : 0;
            lo += (long long)(valid ? ql : 0u);
        }
    }
}

/* k_score_list: the hot kernel.  One workgroup = (entry block, coordinate column k, candidate cw).
 * Staged in LDS: the column (8 B per local sub-fragment), the P_z table, the log10(ob!) table, the log2/exp2
 * tables of the arithmetic contract and the per-contig constants.  Lanes stream the slice list (coalesced
 * 4-byte loads, SCORE_BATCH contacts in flight), read both endpoints' coordinates from LDS, evaluate the
 * Rippe / Poisson term (term_hot: the arithmetic contract of ig_detmath.h with the argument checks hoisted) and add
 * it as an exact integer.  Wave shuffles, one LDS step, two atomics per workgroup. */
#define LDS_COL_SMALL 1024
/* two instantiations per launch site: windows of <= LDS_COL_SMALL sub-fragments (8 KB column: more workgroups per CU)
 * and the rest (<= LDS_COL_CAP staged, larger ones gathered from L2); each workgroup serves its own class only */
template <int CAP>
__global__ void __launch_bounds__(SCORE_THREADS)
    k_score_list(const Glob* g, MoveBuf mb, const double* __restrict__ lgf_tab, PzTab pz, int ablate, int max_c, int large_on,
                 int w_begin)
{
    __shared__ uint2 lcol[CAP];
    __shared__ float pz_s[LDS_PZ];
    __shared__ double lgf_s[LDS_LGF];
    __shared__ double mt_s[IG_TAB_SIZE];
    __shared__ ColMeta cm_s[NCODE];
    __shared__ long long red[2][SCORE_THREADS / 64];
    const int w = w_begin + blockIdx.z / max_c, c = blockIdx.z % max_c;
    if (c >= mb.ctl[w].C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int k = blockIdx.y;
    if (k > m.n_uniq) return;
    const long long n = mb.part[(size_t)cw * P_STRIDE + P_CNT];
    if ((long long)blockIdx.x * SCORE_THREADS >= n) return;
    const long long off = mb.sloff[cw];
    if (off < 0) return;
    const int M = mb.M, m_loc = m.m_loc;
    if (large_on && ((CAP == LDS_COL_SMALL) != (m_loc <= LDS_COL_SMALL))) return;
    const ig_params p = g->par[0];
    const ig_hot hp = ig_hot_make(p, ig_tab());
    const float mean = g->mean_kb;
    const uint2* gcol = mb.coords + (size_t)(cw * NSLOT + k) * M;
    const bool staged = m_loc <= CAP;
    if (staged)
        for (int i = threadIdx.x; i < m_loc; i += SCORE_THREADS) lcol[i] = gcol[i];
    const int pzn = min(pz.n, LDS_PZ);
    for (int i = threadIdx.x; i < pzn; i += SCORE_THREADS) pz_s[i] = pz.v[i];
    for (int i = threadIdx.x; i < LDS_LGF; i += SCORE_THREADS) lgf_s[i] = lgf_tab[i];
    {
        const double* T0 = ig_tab();
        for (int i = threadIdx.x; i < IG_TAB_SIZE; i += SCORE_THREADS) mt_s[i] = T0[i];
    }
    if (threadIdx.x < NCODE) cm_s[threadIdx.x] = mb.cmeta[(size_t)(cw * NSLOT + k) * NCODE + threadIdx.x];
    __syncthreads();
    unsigned circ_mask = 0;
#pragma unroll
    for (int q = 0; q < NCODE; q++) circ_mask |= (cm_s[q].stot != 0) ? (1u << q) : 0u;
    long long hi = 0, lo = 0;
    const ScoreArgs sa{mb.sl_li + off, mb.sl_lj + off, mb.sl_ob + off, (unsigned)n, gcol, lcol, pz_s, lgf_s, mt_s, cm_s, lgf_tab, pz, pzn, circ_mask, mean, ablate};
    if (staged) score_loop<true>(sa, hp, p, hi, lo);
    else score_loop<false>(sa, hp, p, hi, lo);
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hi = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        lo = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (hi | lo) {
            long long* part = mb.part + (size_t)cw * P_STRIDE;
            atomic_add_ll(&part[P_NZ + 2 * k], hi);
            atomic_add_ll(&part[P_NZ + 2 * k + 1], lo);
        }
    }
}

/* k_delta: exact update of the full likelihood when the winner's slice was windowed (KA:565-586 keeps only
 * pairs near A and B): sum over ALL pairs of the contig of (term under the winner - term under the current
 * genome).  Row-parallel with a per-wave compaction queue; two columns (current, winner). */
__global__ void __launch_bounds__(SCORE_THREADS)
    k_delta(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, Tables tab_prev,
            const int* __restrict__ prev_touched, Glob* g, MoveBuf mb, const double* __restrict__ lgf_tab, PzTab pz, int w)
{
    /* tab_prev catches up with the last applied move before k_apply replaces the touched list (quirk Q12) */
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < g->n_prev_touched;
         i += gridDim.x * gridDim.y * blockDim.x) {
        const int s = prev_touched[i];
        tab_prev.dist[s] = tab.dist[s];
        tab_prev.stot[s] = tab.stot[s];
        tab_prev.cp[s] = tab.cp[s];
        tab_prev.len[s] = tab.len[s];
    }
    __shared__ uint2 lcol[LDS_COL_CAP];
    __shared__ long long red[2][SCORE_THREADS / 64];
    __shared__ int q_li[SCORE_THREADS / 64][128], q_lj[SCORE_THREADS / 64][128], q_ob[SCORE_THREADS / 64][128];
    MoveCtl& mc = mb.ctl[w];
    if (!mc.ch_windowed || g->error) return;
    const int c = mc.ch_c;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int kk = blockIdx.y;
    const int k = (kk == 0) ? 0 : mc.ch_k;
    const int M = mb.M, m_loc = m.m_loc;
    const ig_params p = g->par[0];
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const float mean = g->mean_kb;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint2* gcol = mb.coords + (size_t)(cw * NSLOT + k) * M;
    const bool staged = m_loc <= LDS_COL_CAP;
    if (staged) {
        for (int i = threadIdx.x; i < m_loc; i += SCORE_THREADS) lcol[i] = gcol[i];
        __syncthreads();
    }
    const int* subs = mb.subs + (size_t)cw * M;
    const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    int* qli = q_li[wv];
    int* qlj = q_lj[wv];
    int* qob = q_ob[wv];
    long long hi = 0, lo = 0;
    int qn = 0;
    auto drain = [&](int n_take) {
        if (lane < n_take) {
            const int li = qli[lane], lj = qlj[lane], ob = qob[lane];
            const uint2 ai = staged ? lcol[li] : gcol[li];
            const uint2 bj = staged ? lcol[lj] : gcol[lj];
            const long long q = eval_q(p, hot, mean, ai, bj, cm, ob, lgfact_dev(ob, lgf_tab), pz, ig_tab());
            hi += q >> 32;
            lo += (long long)(unsigned int)q;
        }
    };
    const int nrw = gridDim.x * (SCORE_THREADS / 64);
    for (int r = blockIdx.x * (SCORE_THREADS / 64) + wv; r < m_loc; r += nrw) {
        const int i = subs[r];
        const long long b = rowptr[i], e = rowptr[i + 1];
        if (b == e) continue;
        const int2 cp1 = tab.cp[i];
        for (long long q0 = b; q0 < e; q0 += 64) {
            const long long qi = q0 + lane;
            bool keep = false;
            int lj = 0, ob = 0;
            if (qi < e) {
                const int2 v = cc[qi];
                const int2 cp2 = tab.cp[v.x];
                keep = slice_keep(m, cp1.x, cp2.x, cp1.y, cp2.y, v.y, true);
                lj = ((m.same || cp2.x == m.ctgA) ? 0 : m.SLA) + cp2.y;
                ob = v.y;
            }
            const unsigned long long mask = __ballot(keep);
            if (mask) {
                if (keep) {
                    const int at = qn + __popcll(mask & lt_mask);
                    qli[at] = r;
                    qlj[at] = lj;
                    qob[at] = ob;
                }
                qn += __popcll(mask);
                __builtin_amdgcn_wave_barrier();
                if (qn >= 64) {
                    drain(64);
                    __builtin_amdgcn_wave_barrier();
                    const int rem = qn - 64;
                    int t0 = 0, t1 = 0, t2 = 0;
                    if (lane < rem) {
                        t0 = qli[64 + lane];
                        t1 = qlj[64 + lane];
                        t2 = qob[64 + lane];
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rem) {
                        qli[lane] = t0;
                        qlj[lane] = t1;
                        qob[lane] = t2;
                    }
                    __builtin_amdgcn_wave_barrier();
                    qn = rem;
                }
            }
        }
    }
    drain(qn);
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    if (lane == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hi = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        lo = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (hi | lo) {
            atomic_add_ll(&mc.d_hi, kk == 0 ? -hi : hi);
            atomic_add_ll(&mc.d_lo, kk == 0 ? -lo : lo);
        }
    }
}

__device__ __forceinline__ int wave_max_i(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

/* k_prefinal: one workgroup per (candidate, slot).  Keeps the uncorrected slice sums and computes, for
 * every column, T[k] = sum of the terms of the LAST r = S_c mod 64 sliced contacts (canonical order = COO
 * order, so "last" = highest rows, found by bisection on the row id).  Quirk Q5 (KA:4362, block 64 CL:200):
 * a column at list position >= r never receives those contacts; which columns that applies to is decided
 * when the uniq list is known (k_scores / k_commit_batch). */
__device__ void prefinal_tail(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, Glob* g, MoveBuf mb,
                              const double* __restrict__ lgf_tab, int tail_quirk, PzTab pz, int w_begin)
{
    __shared__ int t_li[64], t_lj[64], t_ob[64], t_rows[64];
    __shared__ int sh_n_rows, sh_n_tail, sh_cnt;
    __shared__ long long sh_red[4];
    const int c = blockIdx.x, w = w_begin + blockIdx.y;
    if (c >= mb.ctl[w].C) return;
    const int cw = CW(w, c);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = mb.M;
    const ig_params p = g->par[0];
    const ig_hot hot = ig_hot_make(p, ig_tab());
    const float mean = g->mean_kb;
    const CandMeta& m = mb.meta[cw];
    const long long* part = mb.part + (size_t)cw * P_STRIDE;
    long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
    const int* subs = mb.subs + (size_t)cw * M;
    const int* rowcnt = mb.rowcnt + (size_t)cw * M;
    const int ncol = m.n_uniq + 1;
    for (int k = tid; k < ncol; k += blockDim.x) {
        qp[Q_TAIL + 2 * k] = 0;
        qp[Q_TAIL + 2 * k + 1] = 0;
    }
    __syncthreads();
    const long long Sc = part[P_CNT];
    const int r = (int)(Sc % 64);
    if (!(tail_quirk && r > 0)) return;
    int lo_t = 0, hi_t = M; /* count(lo_t) >= r, count(hi_t) < r */
    while (hi_t - lo_t > 1) {
        const int mid = lo_t + (hi_t - lo_t) / 2;
        long long s = 0;
        for (int ls = tid; ls < m.m_loc; ls += blockDim.x)
            if (subs[ls] >= mid) s += rowcnt[ls];
        s = wave_sum_ll(s);
        if (lane == 0) sh_red[wv] = s;
        __syncthreads();
        const long long tot = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
        __syncthreads();
        if (tot >= r) lo_t = mid;
        else hi_t = mid;
    }
    const int T = lo_t;
    if (tid == 0) {
        sh_n_rows = 0;
        sh_n_tail = 0;
        sh_cnt = 0;
    }
    __syncthreads();
    int above = 0; /* kept contacts in rows > T */
    for (int ls = tid; ls < m.m_loc; ls += blockDim.x) {
        const int s = subs[ls];
        if (s >= T && rowcnt[ls] > 0) {
            const int slot = atomicAdd(&sh_n_rows, 1);
            if (slot < 64) t_rows[slot] = ls;
            if (s > T) above += rowcnt[ls];
        }
    }
    above = wave_sum_i(above);
    if (lane == 0 && above) atomicAdd(&sh_cnt, above);
    __syncthreads();
    const int n_rows = min(sh_n_rows, 64);
    const int need_T = r - sh_cnt; /* contacts to take from the END of row T */
    for (int ri = wv; ri < n_rows; ri += 4) {
        const int ls = t_rows[ri];
        const int i = subs[ls];
        const int2 cp1 = tab.cp[i];
        const long long b = rowptr[i], e = rowptr[i + 1];
        int remaining = (i == T) ? need_T : 0x7fffffff;
        for (long long end = e; end > b && remaining > 0; end -= 64) {
            const long long q0 = end - 1 - lane;
            bool keep = false;
            int2 v = make_int2(0, 0);
            int lj = 0;
            if (q0 >= b) {
                v = cc[q0];
                const int2 cp2 = tab.cp[v.x];
                keep = slice_keep(m, cp1.x, cp2.x, cp1.y, cp2.y, v.y, false);
                lj = ((m.same || cp2.x == m.ctgA) ? 0 : m.SLA) + cp2.y;
            }
            const unsigned long long mask = __ballot(keep);
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            const int took = min((int)__popcll(mask), remaining);
            int basei = 0;
            if (lane == 0 && took) basei = atomicAdd(&sh_n_tail, took);
            basei = __shfl(basei, 0, 64);
            if (keep && rank < remaining && basei + rank < 64) {
                t_li[basei + rank] = ls;
                t_lj[basei + rank] = lj;
                t_ob[basei + rank] = v.y;
            }
            remaining -= took;
        }
    }
    __syncthreads();
    const int n_tail = min(sh_n_tail, 64);
    if (tid == 0 && n_tail != r) g->error = 5; /* the walk must find exactly r contacts */
    for (int k = 1 + wv; k < ncol; k += 4) {
        const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
        const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
        long long hi = 0, lo = 0;
        if (lane < n_tail) {
            const long long q = eval_q(p, hot, mean, col[t_li[lane]], col[t_lj[lane]], cm, t_ob[lane], lgfact_dev(t_ob[lane], lgf_tab), pz,
                                       ig_tab());
            hi = q >> 32;
            lo = (long long)(unsigned int)q;
        }
        hi = wave_sum_ll(hi);
        lo = wave_sum_ll(lo);
        if (lane == 0) {
            qp[Q_TAIL + 2 * k] = hi;
            qp[Q_TAIL + 2 * k + 1] = lo;
        }
    }
}

/* k_tail: needs the slice only (list length, per-row counts), not the column sums: it runs on a second stream next to
 * k_score_list */
__global__ void __launch_bounds__(256) k_tail(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, Glob* g,
                                              MoveBuf mb, const double* __restrict__ lgf_tab, int tail_quirk, PzTab pz, int w_begin)
{
    prefinal_tail(rowptr, cc, tab, g, mb, lgf_tab, tail_quirk, pz, w_begin);
}

/* k_records: after k_score_list and k_tail: the slot-major records of the commit step */
__global__ void __launch_bounds__(64) k_records(MoveBuf mb, int w_begin)
{
    const int c = blockIdx.x, w = w_begin + blockIdx.y;
    if (c >= mb.ctl[w].C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
    const int t = threadIdx.x;
    {
        const long long* part = mb.part + (size_t)cw * P_STRIDE;
        if (t <= m.n_uniq) {
            qp[Q_NZFULL + 2 * t] = part[P_NZ + 2 * t];
            qp[Q_NZFULL + 2 * t + 1] = part[P_NZ + 2 * t + 1];
        }
    }
    __syncthreads();
    if (t < IG_N_TMP_STRUCT) {
        SlotPre r;
        const int k = m.kidx[t];
        r.k = k > 0 ? k : 0;
        r.nz_hi = r.nz_lo = r.tail_hi = r.tail_lo = r.dz_hi = r.dz_lo = r.dni = 0;
        r.changed = r.heads = r.pad = 0;
        if (k > 0) {
            r.nz_hi = qp[Q_NZFULL + 2 * k];
            r.nz_lo = qp[Q_NZFULL + 2 * k + 1];
            r.tail_hi = qp[Q_TAIL + 2 * k];
            r.tail_lo = qp[Q_TAIL + 2 * k + 1];
            r.dz_hi = qp[Q_Z + 2 * k] - qp[Q_Z];
            r.dz_lo = qp[Q_Z + 2 * k + 1] - qp[Q_Z + 1];
            r.dni = qp[Q_NI + k] - qp[Q_NI];
            const int2 si = mb.sinfo[cw * NSLOT + t];
            r.changed = si.x;
            r.heads = si.y;
        }
        mb.pre[(size_t)cw * IG_N_TMP_STRUCT + t] = r;
    }
    if (t == 0) {
        CandPre cp;
        cp.ext_hi = qp[Q_NZFULL];
        cp.ext_lo = qp[Q_NZFULL + 1];
        cp.n_slice = mb.part[(size_t)cw * P_STRIDE + P_CNT];
        cp.r = (int)(cp.n_slice % 64);
        int nb = 0;
        for (int q = 0; q < m.n_uniq; q++) nb += (m.uniq[q] < 12);
        cp.base_cnt = nb;
        cp.n_uniq_basic = nb;
        cp.pad = mb.ctl[w].overflow; /* travels with the records: the slot must be re-run */
        mb.cpre[cw] = cp;
    }
}

/* scores of one move slot (eval_all_likelihood_on_zero_2nd KA:4005-4027, eval_all_scores KA:4029-4046) and the
 * host argmax of CL:1435-1446 (zeros -> -inf, shifted/clipped scores, FIRST index of the maximum).  Executed by
 * one workgroup; `vf0` = the stale insert flags the first candidate sees (quirk Q4). */
__device__ void score_and_choose(Glob* g, const MoveBuf& mb, int w, const int* vf0, double* sc_lds /* [C*24] */)
{
    const int tid = threadIdx.x;
    MoveCtl& mc = mb.ctl[w];
    const int C = mc.C;
    const ig_params p = g->par[0];
    const double log_e = IG_LOG_E_F;
    const double cur_nz = ig_acc_to_double(g->nz_hi, g->nz_lo);
    const int n = C * IG_N_TMP_STRUCT;
    for (int i = tid; i < n; i += blockDim.x) sc_lds[i] = 0.0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
        const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
        const int cw = CW(w, c);
        const CandMeta& m = mb.meta[cw];
        const int k = m.kidx[slot];
        if (k <= 0) continue;
        /* position of this slot in the ACTUAL uniq list (the scored list may be a superset for c == 0) */
        int pos;
        if (c == 0 && mc.superset0) {
            if (slot >= 12 && vf0[slot - 12] == -1) continue; /* not scored by the reference */
            pos = 0;
            for (int q = 0; q < m.n_uniq; q++) {
                const int s2 = m.uniq[q];
                if (s2 >= slot) break;
                if (s2 < 12 || vf0[s2 - 12] != -1) pos++;
            }
        } else {
            pos = k - 1;
        }
        const long long* part = mb.part + (size_t)cw * P_STRIDE;
        const long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
        const int r = (int)(part[P_CNT] % 64);
        long long nh = qp[Q_NZFULL + 2 * k], nl = qp[Q_NZFULL + 2 * k + 1];
        if (r > 0 && pos >= r) { /* quirk Q5 */
            nh -= qp[Q_TAIL + 2 * k];
            nl -= qp[Q_TAIL + 2 * k + 1];
        }
        const double ext = ig_acc_to_double(qp[Q_NZFULL], qp[Q_NZFULL + 1]);
        const long long zhi = g->z_hi + qp[Q_Z + 2 * k] - qp[Q_Z];
        const long long zlo = g->z_lo + qp[Q_Z + 2 * k + 1] - qp[Q_Z + 1];
        const long long ni = g->n_intra + qp[Q_NI + k] - qp[Q_NI];
        const double val_inter = -1.0 * log_e * (g->n_tot_pxl - (double)ni) * p.v_inter;
        const double val_intra = ig_acc_to_double(zhi, zlo) * log_e;
        const double z = val_intra + val_inter;
        const double nz = ig_acc_to_double(nh, nl);
        sc_lds[i] = nz + z + cur_nz - ext;
    }
    __syncthreads();
    if (tid < 64) {
        const int lane = tid;
        double mx = -IG_INF;
        for (int i = lane; i < n; i += 64) {
            const double s = sc_lds[i];
            const double ok = (s == 0.0) ? -IG_INF : s;
            mx = ok > mx ? ok : mx;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(mx, off, 64);
            mx = o > mx ? o : mx;
        }
        double bestv = -IG_INF;
        int best = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {
            const double s = sc_lds[i];
            const double ok = (s == 0.0) ? -IG_INF : s;
            double fs = ok - (mx - 30.0);
            if (fs < 0) fs = 0;
            if (fs > bestv) { /* strictly greater: the first index wins inside a lane */
                bestv = fs;
                best = i;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bestv, off, 64);
            const int oi = __shfl_xor(best, off, 64);
            if (ov > bestv || (ov == bestv && oi < best)) {
                bestv = ov;
                best = oi;
            }
        }
        if (lane == 0) {
            if (best >= n) best = 0;
            const int cc_ = best / IG_N_TMP_STRUCT, slot = best % IG_N_TMP_STRUCT;
            long long tot_slice = 0, tot_eval = 0, bytes = 0;
            for (int c = 0; c < C; c++) {
                const CandMeta& m = mb.meta[CW(w, c)];
                const long long Sc = mb.part[(size_t)CW(w, c) * P_STRIDE + P_CNT];
                tot_slice += Sc;
                int nu = m.n_uniq;
                if (c == 0 && mc.superset0) /* the list the reference would have scored */
                    for (int q = 0; q < m.n_uniq; q++) nu -= (m.uniq[q] >= 12 && vf0[m.uniq[q] - 12] == -1);
                tot_eval += Sc * (nu + 1);
                bytes += 12 * Sc + 20LL * m.m_loc * nu + 8LL * nu;
            }
            const CandMeta& mch = mb.meta[CW(w, cc_)];
            mc.ch_c = cc_;
            mc.ch_slot = slot;
            mc.ch_k = mch.kidx[slot] < 0 ? 0 : mch.kidx[slot];
            mc.ch_windowed = mch.windowed;
            mc.ch_score = sc_lds[best];
            mc.n_slice_tot = tot_slice;
            mc.n_eval_tot = tot_eval;
            mc.bytes_min = bytes;
            if (mch.kidx[slot] < 0) g->error = 3; /* an unscored slot won: cannot happen */
        }
    }
    __syncthreads();
}

/* one-move path: scores + argmax of slot w (then k_delta / k_apply / k_post / k_commit) */
__global__ void __launch_bounds__(256) k_scores(Glob* g, MoveBuf mb, int w)
{
    __shared__ double sc[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT];
    __shared__ int vf[12];
    if (threadIdx.x < 12) vf[threadIdx.x] = g->valid_insert[threadIdx.x];
    __syncthreads();
    score_and_choose(g, mb, w, vf, sc);
    const int n = mb.ctl[w].C * IG_N_TMP_STRUCT;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mb.scores[(size_t)CW(w, 0) * IG_N_TMP_STRUCT + i] = sc[i];
}

/* forced choice for ig_apply (test_copy_struct / apply_replay_simu, CL:2094-2151, 2546-2553) */
__global__ void k_force_choice(Glob* g, MoveBuf mb, int slot)
{
    MoveCtl& mc = mb.ctl[0];
    mc.ch_c = 0;
    mc.ch_slot = slot;
    mc.ch_k = mb.meta[0].kidx[slot];
    mc.ch_windowed = 1; /* always take the exact-delta path */
    mc.ch_score = 0.0;
    mc.n_slice_tot = 0;
    mc.n_eval_tot = 0;
    mc.bytes_min = 0;
    if (mc.ch_k < 0) g->error = 4;
}

/* the winner becomes the live genome (copy_struct KA:4566-4591) and the coordinate tables of the touched
 * sub-fragments are refreshed from its column; executed cooperatively by the calling threads (tid/nth). */
__device__ void apply_winner(State st, Tables tab, Glob* g, const MoveBuf& mb, int w, int forced, int* prev_touched, int tid, int nth,
                             bool single_block)
{
    MoveCtl& mc = mb.ctl[w];
    const int c = mc.ch_c, slot = mc.ch_slot, k = mc.ch_k;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int N = mb.N, M = mb.M;
    const int* base = mb.loc + ((size_t)(cw * NSLOT + slot) * NDYN) * N;
    const int* gid = mb.Lloc + (size_t)cw * N;
    int heads = 0;
    for (int x = tid; x < m.n_loc; x += nth) {
        const int f = gid[x];
        const int np_ = base[x];
        st.pos[f] = np_;
        st.spos[f] = base[(size_t)N + x];
        st.cid[f] = base[(size_t)2 * N + x];
        st.sbp[f] = base[(size_t)3 * N + x];
        st.circ[f] = base[(size_t)4 * N + x];
        st.prev[f] = base[(size_t)5 * N + x];
        st.next[f] = base[(size_t)6 * N + x];
        st.L[f] = base[(size_t)7 * N + x];
        st.SL[f] = base[(size_t)8 * N + x];
        st.LB[f] = base[(size_t)9 * N + x];
        st.ori[f] = base[(size_t)10 * N + x];
        heads += (np_ == 0);
    }
    heads = wave_sum_i(heads);
    if ((threadIdx.x & 63) == 0 && heads) atomicAdd(&g->n_contigs, heads);
    const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
    const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    const int* subs = mb.subs + (size_t)cw * M;
    const int fresh = mc.fresh;
    for (int ls = tid; ls < m.m_loc; ls += nth) {
        const int s = subs[ls];
        const uint2 v = col[ls];
        const int code = (int)(v.y >> 28);
        tab.dist[s] = __uint_as_float(v.x);
        tab.cp[s] = make_int2(code == 0 ? m.ctgA : (code == 1 ? m.ctgB : fresh + (code - 2)), (int)(v.y & 0x0fffffffu));
        tab.stot[s] = cm[code].stot;
        tab.len[s] = cm[code].len;
        prev_touched[ls] = s;
    }
    if (tid == 0) {
        const long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
        g->n_prev_touched = m.m_loc;
        atomicAdd(&g->n_contigs, m.same ? -1 : -2);
        long long dh, dl;
        if (mc.ch_windowed) {
            dh = mc.d_hi;
            dl = mc.d_lo;
        } else {
            dh = qp[Q_NZFULL + 2 * k] - qp[Q_NZFULL];
            dl = qp[Q_NZFULL + 2 * k + 1] - qp[Q_NZFULL + 1];
        }
        long long h = g->nz_hi + dh, l = g->nz_lo + dl;
        ig_acc_normalize((int64_t*)&h, (int64_t*)&l);
        g->nz_hi = h;
        g->nz_lo = l;
        h = g->z_hi + qp[Q_Z + 2 * k] - qp[Q_Z];
        l = g->z_lo + qp[Q_Z + 2 * k + 1] - qp[Q_Z + 1];
        ig_acc_normalize((int64_t*)&h, (int64_t*)&l);
        g->z_hi = h;
        g->z_lo = l;
        g->n_intra += qp[Q_NI + k] - qp[Q_NI];
        /* stale-flag state (quirk Q4): flags of the last candidate, or of the winner when its
         * family re-ran get_bounds in test_copy_struct (op >= 12, CL:2125-2126) */
        if (!forced || slot >= 12) {
            const int* fl = (slot >= 12) ? m.flags : mb.meta[CW(w, mc.C - 1)].flags;
            for (int i = 0; i < 12; i++) g->valid_insert[i] = fl[i];
        }
    }
    (void)single_block;
}

__global__ void k_apply(State st, Tables tab, Glob* g, MoveBuf mb, int w, int forced, int* prev_touched)
{
    if (g->error) return;
    apply_winner(st, tab, g, mb, w, forced, prev_touched, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, false);
}

__device__ __forceinline__ void write_result(Glob* g, const MoveBuf& mb, int w, ig_move_result* out)
{
    const MoveCtl& mc = mb.ctl[w];
    ig_move_result r;
    const double norm = 3.0 * (double)(g->N - g->n_black);
    r.o = mc.ch_score;
    r.dist = (norm - 0.5 * (double)g->credit2) / norm;
    r.mean_len = (double)((float)g->N / (float)g->n_contigs);
    r.op_sampled = mc.ch_slot;
    r.id_f_sampled = mb.meta[CW(w, mc.ch_c)].B;
    r.n_contigs = g->n_contigs;
    r.n_candidates = mc.C;
    r.n_slice = mc.n_slice_tot;
    r.n_evals = mc.n_eval_tot;
    r.bytes_min = mc.bytes_min + 68LL * mb.meta[CW(w, mc.ch_c)].n_loc;
    r.error = g->error;
    r.pad = 0;
    *out = r;
}

__global__ void k_commit(Glob* g, MoveBuf mb, ig_move_result* res, int move, int w)
{
    g->next_cid += NFRESH;
    g->credit2 = g->credit2_acc;
    g->credit2_acc = 0;
    write_result(g, mb, w, res + move);
}

/* credit of fragment f (dist_inter_genome, CL:665-716) when the genome is read through an accessor: V(x) returns
 * (prev, next, ori) of x as of the moment being evaluated */
template <class V>
__device__ __forceinline__ int credit2_view(V view, const int* ip, const int* in, const int* orientable, int f)
{
    const int p0 = ip[f], n0 = in[f];
    const int3 sf = view(f);
    int p1 = sf.x, n1 = sf.y;
    const int o1 = sf.z;
    int c2 = 0;
    if (((p1 == p0) && (n1 == n0)) || ((p1 == n0) && (n1 == p0))) c2 += 2;
    if (orientable[f]) {
        int swap = 1;
        if (1 != o1) {
            int t = p1;
            p1 = n1;
            n1 = t;
            swap = -1;
        }
        if (p0 == p1) {
            if (p0 == -1) c2 += 2;
            else if (!orientable[p1]) c2 += 2;
            else c2 += 1 + ((1 == swap * view(p1).z) ? 1 : 0);
        }
        if (n0 == n1) {
            if (n0 == -1) c2 += 2;
            else if (!orientable[n1]) c2 += 2;
            else c2 += 1 + ((1 == swap * view(n1).z) ? 1 : 0);
        }
    } else {
        if ((p1 == p0) || (p1 == n0)) c2 += 2;
        if ((n1 == n0) || (n1 == p0)) c2 += 2;
    }
    return c2;
}

/* k_commit_batch: the sequential half of a batch, one workgroup.
 *
 * 1. DECIDE (wave 0, no barriers): for w = 0, 1, ...: stop if a contig of move w was modified by an earlier move of
 *    this batch (its scores were computed against a stale state) or if its slice did not fit the pool; otherwise score
 *    and argmax with the LIVE scalars (kept in registers) from the slot-major records of k_prefinal, update the scalars,
 *    write the result record.  A winner whose slice was windowed and that changes the genome needs the exact k_delta
 *    pass: the batch stops BEFORE it (pending) and the host finishes that move with the one-move kernels.
 * 2. APPLY (whole workgroup): the committed moves touch pairwise disjoint contigs, so their winners are applied
 *    together: ownership marks, exact genome-distance deltas (each move's credits evaluated on the genome as of just
 *    before / just after that move, read through the marks), state + coordinate tables, the distance column of the
 *    results.  tab_prev receives every committed move but the last one (quirk Q12: tables before the last move). */
#define COMMIT_THREADS 1024
/* step 1 of the batch commit: ONE wave (it may use the whole register file: the data of the next move is held in
 * registers while the current one is decided) */
__global__ void __launch_bounds__(64)
    k_decide_batch(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out)
{
    /* w_start > 0: slot w_start - 1 was the pending move, meanwhile applied by the one-move kernels; the rest of the batch
     * is still valid wherever it does not touch a contig modified so far (dirty_buf carries the list across the calls) */
    __shared__ int dirty[IG_MAX_BATCH * 2 + 2];
    const int tid = threadIdx.x, lane = tid & 63;
    {
        /* ------------------------------------------------------------ 1. decide */
        long long nz_hi = g->nz_hi, nz_lo = g->nz_lo, z_hi = g->z_hi, z_lo = g->z_lo, n_intra = g->n_intra;
        int n_contigs = g->n_contigs, next_cid = g->next_cid;
        const int err0 = g->error;
        unsigned vmask = 0;
        {
            const int v = (lane < 12) ? g->valid_insert[lane] : -1;
            vmask = (unsigned)__ballot(lane < 12 && v != -1);
        }
        const ig_params p = g->par[0];
        const double log_e = IG_LOG_E_F;
        const double n_tot_pxl = g->n_tot_pxl;
        int n_dirty = 0, committed = w_start, pending = -1, n_large = 0, n_cand = 0;
        if (w_start > 0) {
            n_dirty = dirty_buf[0];
            for (int q = lane; q < n_dirty; q += 64) dirty[q] = dirty_buf[1 + q];
            const MoveCtl pm = mb.ctl[w_start - 1];
            const CandMeta& m = mb.meta[CW(w_start - 1, pm.ch_c)];
            if (lane == 0) {
                dirty[n_dirty] = m.ctgA;
                dirty[n_dirty + 1] = m.ctgB;
            }
            n_dirty += 2;
        }
        /* Everything a decision reads is loaded ONE MOVE AHEAD (none of it depends on earlier decisions, only its
         * interpretation does): while move w is decided from registers with wave shuffles only, the loads of move w + 1
         * are in flight.  Moves with more than 5 candidates (> 2 score records per lane) take the unpipelined path. */
        struct MoveData {
            int C, superset0;                                       /* uniform */
            int cA, cB, mloc, same, windowed, B, n_loc, n_uniq;     /* lane c < C: candidate c */
            long long c_ext_hi, c_ext_lo, c_n_slice;
            int c_base_cnt, c_overflow;
            int flag;                                               /* lane < 12 * min(C, 5): flags[lane % 12] of candidate lane / 12 */
            SlotPre rec[2];                                         /* score records lane, lane + 64 */
            long long e_ext_hi[2], e_ext_lo[2];                     /* their candidates' slice sum under the current genome, */
            int e_r[2], e_base[2];                                  /* S_c mod 64, list entries before the block inserts */
        };
        auto load_move = [&](int w) {
            MoveData d;
            const MoveCtl& mc = mb.ctl[w];
            d.C = mc.C;
            d.superset0 = mc.superset0;
            d.cA = d.cB = -1;
            d.mloc = d.same = d.windowed = d.B = d.n_loc = d.n_uniq = 0;
            d.c_ext_hi = d.c_ext_lo = d.c_n_slice = 0;
            d.c_base_cnt = d.c_overflow = 0;
            if (lane < d.C) {
                const CandMeta& m = mb.meta[CW(w, lane)];
                d.cA = m.ctgA;
                d.cB = m.ctgB;
                d.mloc = m.m_loc;
                d.same = m.same;
                d.windowed = m.windowed;
                d.B = m.B;
                d.n_loc = m.n_loc;
                d.n_uniq = m.n_uniq;
                const CandPre& cp = mb.cpre[CW(w, lane)];
                d.c_ext_hi = cp.ext_hi;
                d.c_ext_lo = cp.ext_lo;
                d.c_n_slice = cp.n_slice;
                d.c_base_cnt = cp.base_cnt;
                d.c_overflow = cp.pad;
            }
            d.flag = -1;
            if (lane < 12 * min(d.C, 5)) d.flag = mb.meta[CW(w, lane / 12)].flags[lane % 12];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = lane + 64 * j;
                d.rec[j].k = 0;
                d.e_ext_hi[j] = d.e_ext_lo[j] = 0;
                d.e_r[j] = d.e_base[j] = 0;
                if (i < d.C * IG_N_TMP_STRUCT) {
                    const int cw = CW(w, i / IG_N_TMP_STRUCT);
                    d.rec[j] = mb.pre[(size_t)cw * IG_N_TMP_STRUCT + i % IG_N_TMP_STRUCT];
                    const CandPre& cp = mb.cpre[cw];
                    d.e_ext_hi[j] = cp.ext_hi;
                    d.e_ext_lo[j] = cp.ext_lo;
                    d.e_r[j] = cp.r;
                    d.e_base[j] = cp.base_cnt;
                }
            }
            return d;
        };
        auto rl = [](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
        auto rl64 = [](long long v, int src) {
            const int lo = __builtin_amdgcn_readlane((int)(unsigned)v, src), hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
            return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        };
        auto rld = [&](double v, int src) { return __longlong_as_double(rl64(__double_as_longlong(v), src)); };
        MoveData cur = load_move(w_start < W ? w_start : W - 1);
        for (int w = w_start; w < W; w++) {
            const MoveData d = cur;
            if (w + 1 < W) cur = load_move(w + 1);
            const int C = d.C;
            /* conflict with an earlier move of this batch?  slice pool overflow? */
            bool hitd = false;
            for (int q = 0; q < n_dirty; q++) hitd |= (dirty[q] == d.cA) | (dirty[q] == d.cB);
            if (err0 || rl(d.c_overflow, 0) || __any(hitd && lane < C)) break;
            n_large += __popcll(__ballot(lane < C && d.mloc > LDS_COL_SMALL));
            n_cand += C;
            /* scores (eval_all_likelihood_on_zero_2nd KA:4005-4027, eval_all_scores KA:4029-4046) with the live scalars */
            const double cur_nz = ig_acc_to_double(nz_hi, nz_lo);
            const int n = C * IG_N_TMP_STRUCT;
            constexpr int NJ = (IG_MAX_CANDIDATES * IG_N_TMP_STRUCT + 63) / 64;
            double sc[NJ];
            /* host argmax of CL:1435-1446: zeros -> -inf, scores shifted by (max - 30) and clipped at 0, FIRST index of the
             * maximum.  The clipped maximum is 30 > 0 and is reached exactly where the score is maximal, so this is the first
             * index of the maximal score (all scores zero: index 0) -- one reduction of (score, index). */
            double bestv = -IG_INF;
            int best = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int i = lane + 64 * j;
                double v = 0.0;
                if (j < 2 || i < n) { /* j >= 2: only moves with more than 5 candidates get here with i < n */
                    SlotPre r;
                    long long ext_hi, ext_lo;
                    int cr, cbase;
                    if (j < 2) {
                        r = d.rec[j];
                        ext_hi = d.e_ext_hi[j];
                        ext_lo = d.e_ext_lo[j];
                        cr = d.e_r[j];
                        cbase = d.e_base[j];
                    } else {
                        const int cw = CW(w, i / IG_N_TMP_STRUCT);
                        r = mb.pre[(size_t)cw * IG_N_TMP_STRUCT + i % IG_N_TMP_STRUCT];
                        const CandPre cp = mb.cpre[cw];
                        ext_hi = cp.ext_hi;
                        ext_lo = cp.ext_lo;
                        cr = cp.r;
                        cbase = cp.base_cnt;
                    }
                    const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
                    const bool sup = (c == 0) && d.superset0 && (slot >= 12);
                    const bool scored = (i < n) && (r.k > 0) && !(sup && !((vmask >> (slot - 12)) & 1u));
                    if (scored) {
                        const int pos = sup ? cbase + __popc(vmask & ((1u << (slot - 12)) - 1u)) : r.k - 1;
                        long long nh = r.nz_hi, nl = r.nz_lo;
                        if (cr > 0 && pos >= cr) { /* quirk Q5 */
                            nh -= r.tail_hi;
                            nl -= r.tail_lo;
                        }
                        const double ext = ig_acc_to_double(ext_hi, ext_lo);
                        const double val_inter = -1.0 * log_e * (n_tot_pxl - (double)(n_intra + r.dni)) * p.v_inter;
                        const double val_intra = ig_acc_to_double(z_hi + r.dz_hi, z_lo + r.dz_lo) * log_e;
                        const double z = val_intra + val_inter;
                        v = ig_acc_to_double(nh, nl) + z + cur_nz - ext;
                    }
                }
                sc[j] = v;
                const double ok = (v == 0.0) ? -IG_INF : v;
                if (i < n && ok > bestv) { /* strictly greater: the lower index wins inside a lane */
                    bestv = ok;
                    best = i;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bestv, off, 64);
                const int oi = __shfl_xor(best, off, 64);
                if (ov > bestv || (ov == bestv && oi < best)) {
                    bestv = ov;
                    best = oi;
                }
            }
            best = rl(best, 0);
            if (best >= n) best = 0;
            const int bc = best / IG_N_TMP_STRUCT, bslot = best % IG_N_TMP_STRUCT;
            const int owner = best & 63, bj = best >> 6; /* the lane and register that hold the winner's record */
            SlotPre br;
            double bests;
            if (bj < 2) {
                const SlotPre mine = (bj == 0) ? d.rec[0] : d.rec[1];
                br.nz_hi = rl64(mine.nz_hi, owner);
                br.nz_lo = rl64(mine.nz_lo, owner);
                br.dz_hi = rl64(mine.dz_hi, owner);
                br.dz_lo = rl64(mine.dz_lo, owner);
                br.dni = rl64(mine.dni, owner);
                br.k = rl(mine.k, owner);
                br.changed = rl(mine.changed, owner);
                br.heads = rl(mine.heads, owner);
                bests = rld((bj == 0) ? sc[0] : sc[1], owner);
            } else {
                br = mb.pre[(size_t)CW(w, bc) * IG_N_TMP_STRUCT + bslot];
                double sv = 0.0;
#pragma unroll
                for (int j = 2; j < NJ; j++) sv = (bj == j) ? sc[j] : sv;
                bests = rld(sv, owner);
            }
            const int windowed = rl(d.windowed, bc), b_same = rl(d.same, bc), b_B = rl(d.B, bc), b_nloc = rl(d.n_loc, bc);
            const int b_cA = rl(d.cA, bc), b_cB = rl(d.cB, bc);
            const long long b_ext_hi = rl64(d.c_ext_hi, bc), b_ext_lo = rl64(d.c_ext_lo, bc);
            /* statistics of the move: off the critical path, k_commit_batch fills them in from the flag mask kept here
             * (a pending move needs them now: its record is written by the one-move kernels) */
            long long Sc = 0, ev = 0, by = 0;
            const bool is_pending = windowed && br.changed;
            if (is_pending) {
                if (lane < C) {
                    int nu = d.n_uniq;
                    if (lane == 0 && d.superset0) nu = d.c_base_cnt + __popc(vmask); /* the list the reference would have scored */
                    Sc = d.c_n_slice;
                    ev = Sc * (nu + 1);
                    by = 12 * Sc + 20LL * d.mloc * nu + 8LL * nu;
                }
                Sc = rl64(wave_sum_ll(Sc), 0);
                ev = rl64(wave_sum_ll(ev), 0);
                by = rl64(wave_sum_ll(by), 0);
            }
            if (lane == 0) {
                MoveCtl& o = mb.ctl[w];
                o.ch_c = bc;
                o.ch_slot = bslot;
                o.ch_k = br.k;
                o.ch_windowed = windowed;
                o.ch_score = bests;
                o.n_slice_tot = Sc;
                o.n_eval_tot = ev;
                o.bytes_min = by;
                o.d_hi = 0;
                o.d_lo = 0;
                o.n_dirty = br.changed;
                o.pad = (int)vmask; /* the stale flags this move was scored under */
                if (br.k <= 0) g->error = 3; /* an unscored slot won: cannot happen */
            }
            if (is_pending) { /* needs k_delta: hand this move to the one-move tail */
                pending = w;
                break;
            }
            /* commit: scalars (exact), stale-flag state (quirk Q4), fresh ids */
            nz_hi += br.nz_hi - b_ext_hi;
            nz_lo += br.nz_lo - b_ext_lo;
            ig_acc_normalize((int64_t*)&nz_hi, (int64_t*)&nz_lo);
            z_hi += br.dz_hi;
            z_lo += br.dz_lo;
            ig_acc_normalize((int64_t*)&z_hi, (int64_t*)&z_lo);
            n_intra += br.dni;
            n_contigs += br.heads - (b_same ? 1 : 2);
            next_cid += NFRESH;
            {
                const int sel = (bslot >= 12) ? bc : C - 1; /* the family of the winner re-ran get_bounds (CL:2125-2126) */
                if (sel < 5) {
                    vmask = (unsigned)((__ballot(d.flag != -1) >> (12 * sel)) & 0xfffull);
                } else {
                    const int v = (lane < 12) ? mb.meta[CW(w, sel)].flags[lane] : -1;
                    vmask = (unsigned)__ballot(lane < 12 && v != -1);
                }
            }
            if (lane == 0) {
                ig_move_result r;
                r.o = bests;
                r.dist = 0.0; /* step 2 */
                r.mean_len = (double)((float)g->N / (float)n_contigs);
                r.op_sampled = bslot;
                r.id_f_sampled = b_B;
                r.n_contigs = n_contigs;
                r.n_candidates = C;
                r.n_slice = 0; /* step 2 */
                r.n_evals = 0;
                r.bytes_min = 68LL * b_nloc;
                r.error = err0;
                r.pad = 0;
                res[move0 + w] = r;
                if (br.changed) {
                    dirty[n_dirty] = b_cA;
                    dirty[n_dirty + 1] = b_cB;
                }
            }
            if (br.changed) n_dirty += 2;
            committed = w + 1;
        }
        if (lane == 0) {
            g->nz_hi = nz_hi;
            g->nz_lo = nz_lo;
            g->z_hi = z_hi;
            g->z_lo = z_lo;
            g->n_intra = n_intra;
            g->n_contigs = n_contigs;
            g->next_cid = next_cid;
            dirty_buf[0] = n_dirty;
            for (int q = 0; q < n_dirty; q++) dirty_buf[1 + q] = dirty[q];
            batch_out[0] = committed;
            batch_out[1] = pending;
            batch_out[2] = n_large;
            batch_out[3] = n_cand;
        }
        if (lane < 12) g->valid_insert[lane] = ((vmask >> lane) & 1u) ? 1 : -1;
    }
}

/* step 2 of the batch commit: one workgroup applies the moves [w_start, batch_out[0]) k_decide_batch committed */
__global__ void __launch_bounds__(COMMIT_THREADS)
    k_commit_batch(State st, Tables tab, Tables tab_prev, Glob* g, MoveBuf mb, const int* __restrict__ ip, const int* __restrict__ in,
                   const int* __restrict__ orientable, const unsigned char* __restrict__ black, int* stamp, int* own_tag, int* own_idx,
                   int* prev_touched, ig_move_result* res, int move0, int W, int w_start, const int* batch_out)
{
    __shared__ long long sh_delta[IG_MAX_BATCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tag_base = g->stamp_ctr; /* tags/stamps of this batch: tag_base + w */
    if (tid < IG_MAX_BATCH) sh_delta[tid] = 0;
    const int committed = batch_out[0];
    __syncthreads();
    if (committed == w_start) return;
    /* ---------------------------------------------------------------- 2. apply */
    const int N = mb.N, M = mb.M;
    auto winner_loc = [&](int w) -> const int* {
        const MoveCtl& mc = mb.ctl[w];
        return mb.loc + ((size_t)(CW(w, mc.ch_c) * NSLOT + mc.ch_slot) * NDYN) * N;
    };
    /* 2a. ownership marks of the fragments whose state changes */
    for (int w = w_start; w < committed; w++) {
        const MoveCtl& mc = mb.ctl[w];
        if (!mc.n_dirty) continue;
        const int cw = CW(w, mc.ch_c);
        const int n_loc = mb.meta[cw].n_loc;
        const int* gid = mb.Lloc + (size_t)cw * N;
        for (int x = tid; x < n_loc; x += blockDim.x) {
            const int f = gid[x];
            own_tag[f] = tag_base + w;
            own_idx[f] = x;
        }
    }
    __syncthreads();
    /* 2b. genome distance: credit(f) depends on prev/next/ori of f and on the orientation of its INITIAL neighbours
     * (CL:665-716), so move w can change the credits of its window and of the window's initial neighbours only; each is
     * evaluated on the genome as of move w-1 and as of move w (moves < t applied, read through the marks) */
    for (int w = w_start; w < committed; w++) {
        const MoveCtl& mc = mb.ctl[w];
        if (!mc.n_dirty) continue;
        const int cw = CW(w, mc.ch_c);
        const int n_loc = mb.meta[cw].n_loc;
        const int* gid = mb.Lloc + (size_t)cw * N;
        const int stampv = tag_base + w + 1; /* != 0 */
        long long d = 0;
        for (int item = tid; item < 3 * n_loc; item += blockDim.x) {
            const int f0 = gid[item / 3];
            const int q = item % 3;
            const int f = (q == 0) ? f0 : ((q == 1) ? ip[f0] : in[f0]);
            if (f < 0 || black[f]) continue;
            if (atomicExch(&stamp[f], stampv) == stampv) continue; /* claimed by another item of this move */
            auto view_at = [&](int t) {
                return [=](int x) -> int3 {
                    const int tg = own_tag[x] - tag_base;
                    if (tg >= 0 && tg <= t) {
                        const int* b = winner_loc(tg);
                        const int xi = own_idx[x];
                        return make_int3(b[(size_t)5 * N + xi], b[(size_t)6 * N + xi], b[(size_t)10 * N + xi]);
                    }
                    return make_int3(st.prev[x], st.next[x], st.ori[x]);
                };
            };
            d += credit2_view(view_at(w), ip, in, orientable, f) - credit2_view(view_at(w - 1), ip, in, orientable, f);
        }
        d = wave_sum_ll(d);
        if (lane == 0 && d) atomic_add_ll(&sh_delta[w], d);
    }
    __syncthreads();
    /* 2c. the winners become the live genome (copy_struct KA:4566-4591); coordinate tables of the touched sub-fragments.
     * First tab_prev catches up with the move applied last before this call. */
    for (int i = tid; i < g->n_prev_touched; i += blockDim.x) {
        const int s2 = prev_touched[i];
        tab_prev.dist[s2] = tab.dist[s2];
        tab_prev.stot[s2] = tab.stot[s2];
        tab_prev.cp[s2] = tab.cp[s2];
        tab_prev.len[s2] = tab.len[s2];
    }
    __syncthreads();
    for (int w = w_start; w < committed; w++) {
        const MoveCtl& mc = mb.ctl[w];
        const int cw = CW(w, mc.ch_c);
        const CandMeta& m = mb.meta[cw];
        const bool last = (w == committed - 1);
        if (last && tid == 0) g->n_prev_touched = mc.n_dirty ? m.m_loc : 0;
        if (!mc.n_dirty) continue;
        const int* base = winner_loc(w);
        const int* gid = mb.Lloc + (size_t)cw * N;
        for (int x = tid; x < m.n_loc; x += blockDim.x) {
            const int f = gid[x];
            st.pos[f] = base[x];
            st.spos[f] = base[(size_t)N + x];
            st.cid[f] = base[(size_t)2 * N + x];
            st.sbp[f] = base[(size_t)3 * N + x];
            st.circ[f] = base[(size_t)4 * N + x];
            st.prev[f] = base[(size_t)5 * N + x];
            st.next[f] = base[(size_t)6 * N + x];
            st.L[f] = base[(size_t)7 * N + x];
            st.SL[f] = base[(size_t)8 * N + x];
            st.LB[f] = base[(size_t)9 * N + x];
            st.ori[f] = base[(size_t)10 * N + x];
        }
        const int k = mc.ch_k;
        const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
        const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
        const int* subs = mb.subs + (size_t)cw * M;
        const int fresh = mc.fresh;
        for (int ls = tid; ls < m.m_loc; ls += blockDim.x) {
            const int s = subs[ls];
            const uint2 v = col[ls];
            const int code = (int)(v.y >> 28);
            const float dist = __uint_as_float(v.x);
            const int2 cp = make_int2(code == 0 ? m.ctgA : (code == 1 ? m.ctgB : fresh + (code - 2)), (int)(v.y & 0x0fffffffu));
            const float stot = cm[code].stot;
            const int len = cm[code].len;
            tab.dist[s] = dist;
            tab.cp[s] = cp;
            tab.stot[s] = stot;
            tab.len[s] = len;
            if (last) {
                prev_touched[ls] = s;
            } else {
                tab_prev.dist[s] = dist;
                tab_prev.cp[s] = cp;
                tab_prev.stot[s] = stot;
                tab_prev.len[s] = len;
            }
        }
    }
    __syncthreads();
    /* 2d. the statistics columns (one thread per move), the distance column */
    if (tid >= w_start && tid < committed) {
        const int w = tid;
        const MoveCtl& mc = mb.ctl[w];
        const unsigned vmask = (unsigned)mc.pad;
        long long Sc = 0, ev = 0, by = 0;
        for (int c = 0; c < mc.C; c++) {
            const CandMeta& m = mb.meta[CW(w, c)];
            const CandPre& cp = mb.cpre[CW(w, c)];
            int nu = m.n_uniq;
            if (c == 0 && mc.superset0) nu = cp.base_cnt + __popc(vmask); /* the list the reference would have scored */
            Sc += cp.n_slice;
            ev += cp.n_slice * (nu + 1);
            by += 12 * cp.n_slice + 20LL * m.m_loc * nu + 8LL * nu;
        }
        res[move0 + w].n_slice = Sc;
        res[move0 + w].n_evals = ev;
        res[move0 + w].bytes_min += by;
    }
    if (tid == 0) {
        long long c2 = g->credit2;
        const double norm = 3.0 * (double)(g->N - g->n_black);
        for (int w = w_start; w < committed; w++) {
            c2 += sh_delta[w];
            res[move0 + w].dist = (norm - 0.5 * (double)c2) / norm;
        }
        g->credit2 = c2;
        g->stamp_ctr = tag_base + W + 2;
    }
}

__global__ void k_debug_terms(const float* s, const float* stot, const int* ob, long long n, const Glob* g,
                              const double* __restrict__ lgf_tab, float* ex, float* exc, double* term, long long* q)
{
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ig_params p = g->par[0];
    const ig_hot hot = ig_hot_make(p, ig_tab());
    ex[i] = ig_rippe(s[i], p, ig_tab());
    exc[i] = ig_rippe_circ(s[i], stot[i], p, ig_tab());
    /* the contract's term with P_z := exc (same probe as the oracle's igo_eval_terms) */
    if (hot.fast && ob[i] > 0) term[i] = ig_term_hot(s[i], 0, ob[i], lgfact_dev(ob[i], lgf_tab), exc[i], &hot, ig_tab());
    else term[i] = ig_pixel_term(ex[i], exc[i], ob[i], lgfact_dev(ob[i], lgf_tab), ig_tab());
    q[i] = ig_quantize(term[i]);
}

/* ================================================================== host side */

template <class T>
static int dalloc(T** p, size_t n)
{
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    if (e != hipSuccess) return fail("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}
#define DALLOC(p, n)                     \
    do {                                 \
        if (dalloc(&(p), (n))) return -1; \
    } while (0)

enum { T_GATHER = 0, T_MUTATE, T_SCORE, T_FINALIZE, T_DELTA, T_APPLY, T_POST, T_COMMIT, T_SLICE, T_ARGMAX, T_COUNT };
static const char* kTimerNames[T_COUNT] = {"gather", "mutate", "score", "finalize", "delta", "apply", "post", "commit", "slice", "argmax"};

struct TimedLaunch {
    ig_ctx* c;
    int id;
    hipEvent_t a, b;
    TimedLaunch(ig_ctx* ctx, int which) : c(ctx), id(which), a(nullptr), b(nullptr)
    {
        if (c->timing && !((c->timing_mask >> id) & 1u)) return;
        if (c->timing) {
            hipEventCreate(&a);
            hipEventCreate(&b);
            hipEventRecord(a, c->stream);
        }
    }
    ~TimedLaunch()
    {
        if (c->timing && a) {
            hipEventRecord(b, c->stream);
            c->timers[id].ev.emplace_back(a, b);
        }
    }
};

static void drain_timers(ig_ctx* c)
{
    for (int i = 0; i < T_COUNT; i++) {
        for (auto& pr : c->timers[i].ev) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->timers[i].total_ms += ms;
                c->timers[i].n++;
            }
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        c->timers[i].ev.clear();
    }
}

extern "C" const char* ig_last_error(void) { return g_err.c_str(); }

extern "C" int ig_create(int device_id, ig_ctx** out)
{
    if (!out) return fail("ig_create: out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail("ig_create: no HIP device (the MI355X path has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail("ig_create: device %d out of range (%d devices)", device_id, n);
    HIPCK(hipSetDevice(device_id));
    ig_ctx* c = new ig_ctx();
    memset((void*)&c->st, 0, sizeof c->st);
    memset((void*)&c->tab, 0, sizeof c->tab);
    memset((void*)&c->tab_prev, 0, sizeof c->tab_prev);
    memset((void*)&c->mb, 0, sizeof c->mb);
    c->device = device_id;
    c->own_stream = true;
    c->rank = 0;
    c->world = 1;
    c->N = c->M = 0;
    c->Z = 0;
    c->st_block = nullptr;
    c->sub_tab = nullptr;
    c->rowptr = nullptr;
    c->cc = nullptr;
    c->init_prev = c->init_next = c->orientable = nullptr;
    c->black = nullptr;
    c->stamp = nullptr;
    c->batch_out = nullptr;
    c->own_tag = c->own_idx = nullptr;
    c->dirty_buf = nullptr;
    c->d_results = nullptr;
    c->results_cap = 0;
    c->d_frags = c->d_cands = nullptr;
    c->cands_cap = 0;
    c->prev_touched = nullptr;
    c->pz_tab = nullptr;
    c->pz_n = 0;
    c->timing_mask = 0xffff;
    c->timing = false;
    c->n_batches = c->n_batch_committed = c->n_batch_pending = 0;
    c->large_seen = 1;
    c->up_moves = c->up_max_c = 0;
    for (int i = 0; i < T_COUNT; i++) {
        c->timers[i].name = kTimerNames[i];
        c->timers[i].total_ms = 0;
        c->timers[i].n = 0;
    }
    c->have_contacts = c->have_sub = c->have_state = c->have_init = c->have_params = false;
    HIPCK(hipStreamCreate(&c->stream));
    HIPCK(hipStreamCreate(&c->stream2));
    HIPCK(hipEventCreateWithFlags(&c->ev_slice, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    DALLOC(c->glob, 1);
    HIPCK(hipMemset(c->glob, 0, sizeof(Glob)));
    DALLOC(c->lgf_tab, LGF_TAB);
    /* log10(ob!) table (KA:111-124, 251-270): the 15 float-factorial constants on the host, the rest on the device */
    double small[15];
    for (int k = 0; k < 15; k++) {
        float r = 1;
        if (k < 10) {
            for (int q = 1; q <= k; q++) r = r * q;
        } else {
            r = ig_powf((float)k, (float)k, ig_tab()) * ig_expf(-(float)k, ig_tab()) * __builtin_sqrtf((float)(2 * 3.14159265358979323846 * (float)k));
        }
        small[k] = ig_log10((double)r, ig_tab());
    }
    Glob hg;
    memset(&hg, 0, sizeof hg);
    for (int k = 0; k < 15; k++) hg.lgf[k] = small[k];
    const int lb[6] = {1, 3, 5, 10, 20, 50}; /* CL:417 */
    for (int k = 0; k < 6; k++) hg.list_bounds[k] = lb[k];
    hg.slice_nb = 50 * 4;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_lgf_table, dim3((LGF_TAB + 255) / 256), dim3(256), 0, c->stream, c->lgf_tab, c->glob->lgf);
    HIPCK(hipStreamSynchronize(c->stream));
    *out = c;
    return 0;
}

static void free_move_buffers(ig_ctx* c)
{
    MoveBuf& m = c->mb;
    hipFree(m.Lloc);
    hipFree(m.lbloc);
    hipFree(m.slloc);
    hipFree(m.subs);
    hipFree(m.rowcnt);
    hipFree(m.sl_li);
    hipFree(m.sl_lj);
    hipFree(m.sl_ob);
    hipFree(m.coords);
    hipFree(m.loc);
    hipFree(m.meta);
    hipFree(m.cmeta);
    hipFree(m.part);
    hipFree(m.scores);
    hipFree(m.slbound);
    hipFree(m.sloff);
    hipFree(m.qpart);
    hipFree(m.ctl);
    hipFree(m.sinfo);
    hipFree(m.pre);
    hipFree(m.cpre);
    hipFree(c->own_tag);
    hipFree(c->own_idx);
    c->own_tag = c->own_idx = nullptr;
    hipFree(c->stamp);
    hipFree(c->batch_out);
    hipFree(c->dirty_buf);
    c->dirty_buf = nullptr;
    memset((void*)&m, 0, sizeof m);
    c->stamp = nullptr;
    c->batch_out = nullptr;
}

extern "C" void ig_destroy(ig_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->stream2);
    hipStreamDestroy(c->stream2);
    hipEventDestroy(c->ev_slice);
    hipEventDestroy(c->ev_tail);
    drain_timers(c);
    free_move_buffers(c);
    hipFree(c->st_block);
    hipFree(c->tab.dist);
    hipFree(c->tab_prev.dist);
    hipFree(c->sub_tab);
    hipFree(c->rowptr);
    hipFree(c->cc);
    hipFree(c->init_prev);
    hipFree(c->init_next);
    hipFree(c->orientable);
    hipFree(c->black);
    hipFree(c->lgf_tab);
    hipFree(c->glob);
    hipFree(c->d_results);
    hipFree(c->d_frags);
    hipFree(c->d_cands);
    hipFree(c->prev_touched);
    hipFree(c->pz_tab);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int ig_sync(ig_ctx* c)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    drain_timers(c);
    return 0;
}

extern "C" int ig_set_stream(ig_ctx* c, void* s)
{
    HIPCK(hipStreamSynchronize(c->stream));
    if (c->own_stream) hipStreamDestroy(c->stream);
    if (s) {
        c->stream = (hipStream_t)s;
        c->own_stream = false;
    } else {
        HIPCK(hipStreamCreate(&c->stream));
        c->own_stream = true;
    }
    return 0;
}

static int ensure_move_buffers(ig_ctx* c, int capC, int capW = 1)
{
    if (c->mb.capC >= capC && c->mb.capW >= capW && c->mb.N == c->N && c->mb.M == c->M) return 0;
    if (c->N == 0 || c->M == 0) return 0;
    capC = std::max(capC, c->mb.capC);
    capW = std::max(capW, c->mb.capW);
    free_move_buffers(c);
    MoveBuf& m = c->mb;
    const size_t N = c->N, M = c->M, C = (size_t)capC * capW;
    DALLOC(m.Lloc, C * N);
    DALLOC(m.lbloc, C * N);
    DALLOC(m.slloc, C * N);
    DALLOC(m.subs, C * M);
    DALLOC(m.rowcnt, C * M);
    {
        /* one slot never needs more than capC x Z entries; a batch shares the pool and the slots that do not fit
         * are re-run (MoveCtl.overflow) */
        size_t Zc = (size_t)std::max<long long>(c->Z, 1) * (size_t)std::max(capC, capW > 1 ? 16 : 1);
        if (const char* e = getenv("IG_POOL_ENTRIES")) /* tests: a small pool forces the overflow / re-run path */
            Zc = std::max<size_t>((size_t)atoll(e), (size_t)std::max<long long>(c->Z, 1) * (size_t)capC);
        DALLOC(m.sl_li, Zc);
        DALLOC(m.sl_lj, Zc);
        DALLOC(m.sl_ob, Zc);
        m.pool_cap = (long long)Zc;
    }
    DALLOC(m.slbound, C);
    DALLOC(m.sloff, C);
    DALLOC(m.coords, C * M * NSLOT);
    DALLOC(m.loc, C * NSLOT * NDYN * N);
    DALLOC(m.meta, C);
    DALLOC(m.cmeta, C * NSLOT * NCODE);
    DALLOC(m.part, C * P_STRIDE);
    DALLOC(m.qpart, C * Q_STRIDE);
    DALLOC(m.scores, C * IG_N_TMP_STRUCT);
    DALLOC(m.ctl, (size_t)capW);
    DALLOC(m.sinfo, C * NSLOT);
    DALLOC(m.pre, C * IG_N_TMP_STRUCT);
    DALLOC(m.cpre, C);
    DALLOC(c->own_tag, N);
    DALLOC(c->own_idx, N);
    HIPCK(hipMemset(c->own_tag, 0xff, N * sizeof(int)));
    DALLOC(c->stamp, N);
    DALLOC(c->batch_out, 4);
    DALLOC(c->dirty_buf, 2 * IG_MAX_BATCH + 4);
    HIPCK(hipMemset(m.cmeta, 0, C * NSLOT * NCODE * sizeof(ColMeta)));
    HIPCK(hipMemset(m.slbound, 0, C * sizeof(long long)));
    HIPCK(hipMemset(m.ctl, 0, (size_t)capW * sizeof(MoveCtl)));
    HIPCK(hipMemset(c->stamp, 0, N * sizeof(int)));
    m.N = c->N;
    m.M = c->M;
    m.capC = capC;
    m.capW = capW;
    return 0;
}

extern "C" int ig_upload_contacts(ig_ctx* c, const int32_t* row, const int32_t* col, const int32_t* cnt, int64_t Z, int32_t M,
                                  int32_t rank, int32_t world)
{
    HIPCK(hipSetDevice(c->device));
    if (Z < 0 || M <= 0) return fail("ig_upload_contacts: bad sizes");
    if (world < 1 || rank < 0 || rank >= world) return fail("ig_upload_contacts: bad shard %d/%d", rank, world);
    if (c->M && c->M != M) return fail("ig_upload_contacts: M=%d does not match the sub-fragment table (%d)", M, c->M);
    std::vector<long long> rp((size_t)M + 1, 0);
    std::vector<int2> cc((size_t)Z);
    for (int64_t k = 0; k < Z; k++) {
        const int r = row[k], q = col[k];
        if (r < 0 || r >= M || q <= r || q >= M) return fail("ig_upload_contacts: entry %lld (%d,%d) is not strict upper triangle", (long long)k, r, q);
        if (k > 0 && (row[k - 1] > r || (row[k - 1] == r && col[k - 1] >= q)))
            return fail("ig_upload_contacts: entries must be row-major sorted and distinct (at %lld)", (long long)k);
        rp[(size_t)r + 1]++;
        cc[(size_t)k] = make_int2(q, cnt[k]);
    }
    for (int i = 0; i < M; i++) rp[(size_t)i + 1] += rp[(size_t)i];
    hipFree(c->rowptr);
    hipFree(c->cc);
    DALLOC(c->rowptr, (size_t)M + 1);
    DALLOC(c->cc, (size_t)Z);
    HIPCK(hipMemcpy(c->rowptr, rp.data(), ((size_t)M + 1) * sizeof(long long), hipMemcpyHostToDevice));
    if (Z) HIPCK(hipMemcpy(c->cc, cc.data(), (size_t)Z * sizeof(int2), hipMemcpyHostToDevice));
    c->Z = Z;
    c->M = M;
    c->rank = rank;
    c->world = world;
    c->have_contacts = true;
    return 0;
}

extern "C" int ig_upload_subfrag_table(ig_ctx* c, const float* xyzw, int32_t M)
{
    HIPCK(hipSetDevice(c->device));
    if (M <= 0) return fail("ig_upload_subfrag_table: M <= 0");
    if (c->M && c->M != M) return fail("ig_upload_subfrag_table: M=%d does not match the contacts (%d)", M, c->M);
    std::vector<SubTab> t((size_t)M);
    for (int s = 0; s < M; s++) {
        t[s].parent = (int)xyzw[4 * (size_t)s];
        t[s].wat = xyzw[4 * (size_t)s + 1];
        t[s].cri = xyzw[4 * (size_t)s + 2];
        t[s].w = (int)xyzw[4 * (size_t)s + 3];
        if (s > 0 && (t[s].parent < t[s - 1].parent || (t[s].parent == t[s - 1].parent && t[s].w != t[s - 1].w + 1)))
            return fail("ig_upload_subfrag_table: sub-fragments of a bin must be contiguous and ordered (at %d)", s);
    }
    hipFree(c->sub_tab);
    DALLOC(c->sub_tab, (size_t)M);
    HIPCK(hipMemcpy(c->sub_tab, t.data(), (size_t)M * sizeof(SubTab), hipMemcpyHostToDevice));
    hipFree(c->tab.dist);
    hipFree(c->tab_prev.dist);
    for (Tables* tb : {&c->tab, &c->tab_prev}) {
        int* blk;
        DALLOC(blk, 6 * (size_t)M + 2);
        tb->dist = (float*)blk;
        tb->stot = (float*)(blk + (size_t)M);
        tb->len = blk + 2 * (size_t)M;
        tb->cp = (int2*)(blk + 4 * (size_t)M + ((4 * (size_t)M) & 1)); /* keep the int2 array 8-byte aligned */
    }
    hipFree(c->prev_touched);
    DALLOC(c->prev_touched, (size_t)M);
    c->M = M;
    c->have_sub = true;
    return 0;
}

static int launch_recompute(ig_ctx* c);

extern "C" int ig_upload_state(ig_ctx* c, const int32_t* soa, int32_t N)
{
    HIPCK(hipSetDevice(c->device));
    if (N <= 0) return fail("ig_upload_state: N <= 0");
    if (!c->have_sub) return fail("ig_upload_state: upload the sub-fragment table first");
    const size_t n = N;
    std::vector<int> host(17 * n);
    /* soa member order (KA:40-58): 0 pos 1 sub_pos 2 id_c 3 start_bp 4 len_bp 5 sub_len 6 circ 7 id 8 prev 9 next
     * 10 l_cont 11 sub_l_cont 12 l_cont_bp 13 ori 14 rep 15 activ 16 id_d */
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(&host[k * n], soa + dyn_src[k] * n, n * sizeof(int));
    memcpy(&host[11 * n], soa + 4 * n, n * sizeof(int));  /* len_bp */
    memcpy(&host[12 * n], soa + 5 * n, n * sizeof(int));  /* sub_len */
    memcpy(&host[14 * n], soa + 14 * n, n * sizeof(int)); /* rep */
    memcpy(&host[15 * n], soa + 15 * n, n * sizeof(int)); /* activ */
    memcpy(&host[16 * n], soa + 16 * n, n * sizeof(int)); /* id_d */
    /* first sub-fragment of each bin + consistency with the table */
    std::vector<SubTab> t((size_t)c->M);
    HIPCK(hipMemcpy(t.data(), c->sub_tab, (size_t)c->M * sizeof(SubTab), hipMemcpyDeviceToHost));
    long long acc = 0;
    for (size_t f = 0; f < n; f++) {
        host[13 * n + f] = (int)acc;
        const int sl = soa[5 * n + f];
        if (sl < 1 || acc + sl > c->M) return fail("ig_upload_state: sub_len of bin %zu inconsistent with the sub-fragment table", f);
        for (int w = 0; w < sl; w++)
            if (t[(size_t)acc + w].parent != (int)f || t[(size_t)acc + w].w != w)
                return fail("ig_upload_state: sub-fragment %lld is not (bin %zu, index %d)", acc + w, f, w);
        acc += sl;
        if (soa[15 * n + f] != 1) return fail("ig_upload_state: inactive fragments are not supported (dead in the reference)");
    }
    if (acc != c->M) return fail("ig_upload_state: bins cover %lld sub-fragments, table has %d", acc, c->M);
    /* internal contig ids: any injective relabelling works; keep the caller's, they are >= 0 */
    int max_c = 0;
    for (size_t f = 0; f < n; f++) {
        if (soa[2 * n + f] < 0) return fail("ig_upload_state: negative contig id");
        max_c = std::max(max_c, soa[2 * n + f]);
    }
    hipFree(c->st_block);
    DALLOC(c->st_block, 17 * n);
    HIPCK(hipMemcpy(c->st_block, host.data(), 17 * n * sizeof(int), hipMemcpyHostToDevice));
    int** sp = (int**)&c->st;
    for (int k = 0; k < 17; k++) sp[k] = c->st_block + k * n;
    const bool size_changed = (c->N != N);
    c->N = N;
    if (size_changed || !c->have_init) { /* default initial genome for the distance = this state (CL:269-276) */
        hipFree(c->init_prev);
        hipFree(c->init_next);
        hipFree(c->orientable);
        hipFree(c->black);
        DALLOC(c->init_prev, n);
        DALLOC(c->init_next, n);
        DALLOC(c->orientable, n);
        DALLOC(c->black, n);
        std::vector<int> orient(n);
        for (size_t f = 0; f < n; f++) orient[f] = soa[5 * n + f] > 1;
        HIPCK(hipMemcpy(c->init_prev, soa + 8 * n, n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(c->init_next, soa + 9 * n, n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(c->orientable, orient.data(), n * sizeof(int), hipMemcpyHostToDevice));
        HIPCK(hipMemset(c->black, 0, n));
        c->have_init = true;
    }
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    hg.N = N;
    hg.M = c->M;
    hg.next_cid = max_c + 1;
    hg.n_tot_pxl = (double)c->M * ((double)c->M - 1.0) / 2.0; /* CL:366 */
    for (int i = 0; i < 12; i++) hg.valid_insert[i] = 0; /* CL:421 */
    hg.error = 0;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    c->have_state = true;
    if (ensure_move_buffers(c, 8)) return -1;
    return launch_recompute(c);
}

/* tables, head count, genome-distance credits and (when parameters are known) the exact
 * likelihood sums of the current state */
static int launch_recompute(ig_ctx* c)
{
    if (!c->have_state || !c->have_sub) return 0;
    const int N = c->N, M = c->M;
    hipLaunchKernelGGL(k_fill_tables, dim3((M + 255) / 256), dim3(256), 0, c->stream, c->st, c->sub_tab, c->tab, M);
    HIPCK(hipMemcpyAsync(c->tab_prev.dist, c->tab.dist, (6 * (size_t)M + 2) * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    long long* scratch;
    DALLOC(scratch, 8);
    HIPCK(hipMemsetAsync(scratch, 0, 8 * sizeof(long long), c->stream));
    int* heads = (int*)(scratch + 6);
    hipLaunchKernelGGL(k_count_heads, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, N, heads);
    HIPCK(hipMemsetAsync(&c->glob->credit2_acc, 0, sizeof(long long), c->stream));
    hipLaunchKernelGGL(k_post, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, c->init_prev, c->init_next, c->orientable,
                       c->black, c->glob, N);
    if (c->have_params && c->have_contacts) {
        hipLaunchKernelGGL(k_full_nz, dim3(1024), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, 0, c->lgf_tab, M, 0, 1,
                           scratch);
        hipLaunchKernelGGL(k_full_zero, dim3(256), dim3(256), 0, c->stream, c->tab, c->glob, 0, M, scratch + 2);
    }
    long long h[8];
    HIPCK(hipMemcpyAsync(h, scratch, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    hipFree(scratch);
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    ig_acc_normalize((int64_t*)&h[0], (int64_t*)&h[1]);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    hg.nz_hi = h[0];
    hg.nz_lo = h[1];
    hg.z_hi = h[2];
    hg.z_lo = h[3];
    hg.n_intra = h[4];
    hg.n_contigs = ((int*)&h[6])[0];
    hg.credit2 = hg.credit2_acc;
    hg.credit2_acc = 0;
    hg.n_prev_touched = 0;
    HIPCK(hipMemcpy(c->glob, &hg, sizeof hg, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ig_set_params(ig_ctx* c, const float p[8], float mean_subfrag_kb, int which)
{
    HIPCK(hipSetDevice(c->device));
    if (which != 0 && which != 1) return fail("ig_set_params: which must be 0 or 1");
    HIPCK(hipStreamSynchronize(c->stream));
    ig_params hp = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
    HIPCK(hipMemcpy(&c->glob->par[which], &hp, sizeof hp, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->mean_kb, &mean_subfrag_kb, sizeof(float), hipMemcpyHostToDevice));
    if (which == 0) {
        c->have_params = true;
        if (!c->pz_tab) DALLOC(c->pz_tab, PZ_MAX);
        /* table length: first rank distance whose s_z reaches d_max (+1), capped */
        double need = (mean_subfrag_kb > 0) ? (double)p[5] / (double)mean_subfrag_kb + 2.0 : 0.0;
        c->pz_n = (need > 0 && need < (double)PZ_MAX) ? (int)need : ((need >= (double)PZ_MAX) ? PZ_MAX : 0);
        if (c->pz_n > 0)
            hipLaunchKernelGGL(k_build_pz, dim3((c->pz_n + 255) / 256), dim3(256), 0, c->stream, c->glob, c->pz_tab, c->pz_n);
        return launch_recompute(c); /* the maintained exact sums depend on param_simu */
    }
    return 0;
}

extern "C" int ig_set_insert_config(ig_ctx* c, const int32_t list_bounds[IG_N_INSERT_BLOCKS], int32_t max_bounds_insert)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(c->glob->list_bounds, list_bounds, 6 * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->slice_nb, &max_bounds_insert, sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int ig_set_initial_genome(ig_ctx* c, const int32_t* ip, const int32_t* in, const int32_t* orientable,
                                     const int32_t* blacklisted, int32_t nb)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_set_initial_genome: upload the state first");
    const size_t n = c->N;
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(c->init_prev, ip, n * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->init_next, in, n * sizeof(int), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(c->orientable, orientable, n * sizeof(int), hipMemcpyHostToDevice));
    std::vector<unsigned char> b(n, 0);
    for (int i = 0; i < nb; i++) {
        if (blacklisted[i] < 0 || (size_t)blacklisted[i] >= n) return fail("ig_set_initial_genome: blacklisted id out of range");
        b[blacklisted[i]] = 1;
    }
    int cnt = 0;
    for (size_t i = 0; i < n; i++) cnt += b[i];
    HIPCK(hipMemcpy(c->black, b.data(), n, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(&c->glob->n_black, &cnt, sizeof(int), hipMemcpyHostToDevice));
    return launch_recompute(c);
}

/* canonical contig numbering of modify_gl_cuda_buffer (CL:2715-2881): contigs enumerated by
 * ascending index of their pos==0 fragment (the order an in-order select_uniq_id_c produces,
 * KA:357-406), stable sort by length descending (CL:69-77), id = (n-1) - rank (KA:4689-4692). */
static void canonical_ids(const int* pos, const int* cid, const int* L, size_t n, std::vector<int>& out, int* n_contigs)
{
    std::vector<int> heads;
    for (size_t f = 0; f < n; f++)
        if (pos[f] == 0) heads.push_back((int)f);
    std::vector<int> order(heads.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return L[heads[a]] > L[heads[b]]; });
    int max_c = 0;
    for (size_t f = 0; f < n; f++) max_c = std::max(max_c, cid[f]);
    std::vector<int> map((size_t)max_c + 1, -1);
    const int nc = (int)heads.size();
    for (int r = 0; r < nc; r++) map[cid[heads[order[r]]]] = (nc - 1) - r;
    out.resize(n);
    for (size_t f = 0; f < n; f++) out[f] = map[cid[f]];
    *n_contigs = nc;
}

extern "C" int ig_download_state(ig_ctx* c, int32_t* soa)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_download_state: no state");
    const size_t n = c->N;
    HIPCK(hipStreamSynchronize(c->stream));
    std::vector<int> host(17 * n);
    HIPCK(hipMemcpy(host.data(), c->st_block, 17 * n * sizeof(int), hipMemcpyDeviceToHost));
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(soa + dyn_src[k] * n, &host[k * n], n * sizeof(int));
    memcpy(soa + 4 * n, &host[11 * n], n * sizeof(int));
    memcpy(soa + 5 * n, &host[12 * n], n * sizeof(int));
    memcpy(soa + 14 * n, &host[14 * n], n * sizeof(int));
    memcpy(soa + 15 * n, &host[15 * n], n * sizeof(int));
    memcpy(soa + 16 * n, &host[16 * n], n * sizeof(int));
    for (size_t f = 0; f < n; f++) soa[7 * n + f] = (int)f;
    std::vector<int> ids;
    int nc;
    canonical_ids(&host[0], &host[2 * n], &host[7 * n], n, ids, &nc);
    memcpy(soa + 2 * n, ids.data(), n * sizeof(int));
    return 0;
}

extern "C" int ig_renumber_contigs(ig_ctx* c, int32_t* n_contigs, float* mean_len, int32_t* max_id)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    int nc;
    HIPCK(hipMemcpy(&nc, &c->glob->n_contigs, sizeof(int), hipMemcpyDeviceToHost));
    if (n_contigs) *n_contigs = nc;
    if (mean_len) *mean_len = (float)c->N / (float)nc;
    if (max_id) *max_id = nc - 1;
    return 0;
}

extern "C" int ig_bomb(ig_ctx* c, const int32_t* shuffle)
{
    (void)shuffle; /* explode_genome writes id_c = shuffle[i] (KA:419); the renumbering that follows (CL:1948) erases it */
    HIPCK(hipSetDevice(c->device));
    if (!c->have_state) return fail("ig_bomb: no state");
    hipLaunchKernelGGL(k_explode, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->st, c->N);
    int next = c->N;
    HIPCK(hipMemcpyAsync(&c->glob->next_cid, &next, sizeof(int), hipMemcpyHostToDevice, c->stream));
    return launch_recompute(c);
}

extern "C" int ig_genome_distance(ig_ctx* c, double* d)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    const double norm = 3.0 * (double)(hg.N - hg.n_black);
    *d = (norm - 0.5 * (double)hg.credit2) / norm;
    return 0;
}

extern "C" int ig_get_valid_insert(ig_ctx* c, int32_t out12[12])
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(out12, c->glob->valid_insert, 12 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ig_full_likelihood(ig_ctx* c, int which, int use_prev, double* nz, double* z, int64_t* limbs5)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->have_contacts || !c->have_state || !c->have_params) return fail("ig_full_likelihood: contacts, state and parameters are required");
    if (which != 0 && which != 1) return fail("ig_full_likelihood: which must be 0 or 1");
    long long* scratch;
    DALLOC(scratch, 8);
    HIPCK(hipMemsetAsync(scratch, 0, 8 * sizeof(long long), c->stream));
    Tables& t = use_prev ? c->tab_prev : c->tab;
    hipLaunchKernelGGL(k_full_nz, dim3(1024), dim3(256), 0, c->stream, c->rowptr, c->cc, t, c->glob, which, c->lgf_tab, c->M, 0, 1,
                       scratch);
    hipLaunchKernelGGL(k_full_zero, dim3(256), dim3(256), 0, c->stream, t, c->glob, which, c->M, scratch + 2);
    long long h[8];
    HIPCK(hipMemcpyAsync(h, scratch, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    hipFree(scratch);
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    ig_acc_normalize((int64_t*)&h[0], (int64_t*)&h[1]);
    ig_acc_normalize((int64_t*)&h[2], (int64_t*)&h[3]);
    if (nz) *nz = ig_acc_to_double(h[0], h[1]);
    if (z) { /* CL:755-759 with the float log_e of the kernels replaced by the host's double constant */
        const double log_e = 0.43429448190325182;
        const double val_intra = ig_acc_to_double(h[2], h[3]) * log_e;
        const double val_inter = log_e * (hg.n_tot_pxl - (double)h[4]) * -1.0 * (double)hg.par[which].v_inter;
        *z = val_intra + val_inter;
    }
    if (limbs5)
        for (int i = 0; i < 5; i++) limbs5[i] = h[i];
    return 0;
}

/* ------------------------------------------------------------------ move driver */

static int ensure_io(ig_ctx* c, int n_moves, int max_c)
{
    if (c->results_cap < n_moves) {
        hipFree(c->d_results);
        hipFree(c->d_frags);
        DALLOC(c->d_results, (size_t)n_moves);
        DALLOC(c->d_frags, (size_t)n_moves);
        c->results_cap = n_moves;
    }
    if (c->cands_cap < n_moves * max_c) {
        hipFree(c->d_cands);
        DALLOC(c->d_cands, (size_t)n_moves * max_c);
        c->cands_cap = n_moves * max_c;
    }
    return 0;
}

static int check_ready(ig_ctx* c)
{
    if (!c->have_contacts || !c->have_sub || !c->have_state || !c->have_params)
        return fail("contacts, sub-fragment table, state and parameters must be uploaded before a move");
    return 0;
}

static int g_tail_quirk = 1;

/* enqueue the scoring launches of W move slots (moves move0 .. move0+W-1 of the uploaded lists);
 * phase 0 = up to k_score_list (the sums that are all-reduced when sharded), 1 = the rest, 2 = both */
static void enqueue_score(ig_ctx* c, int move0, int W, int max_c, int force_slot, int phase, int w_begin = 0, int w_end = -1)
{
    /* slots [w_begin, w_end) are sliced and scored here (multi-GPU: the other ranks score the rest and the slot-major
     * records are all-gathered); the candidate genomes of EVERY slot are built on every rank, the commit step needs them */
    if (w_end < 0) w_end = W;
    const int nW = w_end - w_begin;
    const int N = c->N;
    const int gN = std::max((N + 255) / 256, W);
    const PzTab pz{c->pz_tab, c->pz_n};
    if (phase == 0 || phase == 2) {
        {
            TimedLaunch t(c, T_GATHER);
            hipLaunchKernelGGL(k_gather, dim3(gN), dim3(256), 0, c->stream, c->st, c->glob, c->mb, c->d_cands, c->d_frags, move0, W, max_c,
                               c->tab, c->tab_prev, c->prev_touched, force_slot);
        }
        {
            TimedLaunch t(c, T_MUTATE);
            hipLaunchKernelGGL(k_mutate, dim3(NSLOT, max_c, W), dim3(256), 0, c->stream, c->st, c->tab, c->sub_tab, c->rowptr, c->glob,
                               c->mb, pz);
        }
        if (force_slot < 0 && nW > 0) {
            {
                TimedLaunch t(c, T_SLICE);
                hipLaunchKernelGGL(k_offsets, dim3(1), dim3(64), 0, c->stream, c->mb, W, w_begin, w_end);
                hipLaunchKernelGGL(k_slice, dim3(SLICE_RB, max_c, nW), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, c->mb,
                                   c->rank, c->world, w_begin);
            }
            if (phase == 2) { /* the Q5 tail walk only needs the slice: second stream, next to k_score_list */
                hipEventRecord(c->ev_slice, c->stream);
                hipStreamWaitEvent(c->stream2, c->ev_slice, 0);
                hipLaunchKernelGGL(k_tail, dim3(max_c, nW), dim3(256), 0, c->stream2, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->lgf_tab,
                                   g_tail_quirk, pz, w_begin);
                hipEventRecord(c->ev_tail, c->stream2);
            }
            TimedLaunch t(c, T_SCORE);
            static int s_eb = getenv("IG_SCORE_EB") ? atoi(getenv("IG_SCORE_EB")) : SCORE_EB;
            static int s_abl = getenv("IG_ABLATE") ? atoi(getenv("IG_ABLATE")) : 0;
            /* the large-window variant is launched only when the previous batch saw windows above LDS_COL_SMALL
             * sub-fragments; without it the small variant serves every window (unstaged above its cap) */
            static int s_large = getenv("IG_LARGE") ? atoi(getenv("IG_LARGE")) : -1;
            const int large_on = s_large >= 0 ? s_large : c->large_seen;
            hipLaunchKernelGGL(k_score_list<LDS_COL_SMALL>, dim3(s_eb, NSLOT, max_c * nW), dim3(SCORE_THREADS), 0, c->stream, c->glob,
                               c->mb, c->lgf_tab, pz, s_abl, max_c, large_on, w_begin);
            if (large_on)
                hipLaunchKernelGGL(k_score_list<LDS_COL_CAP>, dim3(s_eb, NSLOT, max_c * nW), dim3(SCORE_THREADS), 0, c->stream, c->glob,
                                   c->mb, c->lgf_tab, pz, s_abl, max_c, large_on, w_begin);
        }
    }
    if (phase == 1 || phase == 2) {
        if (force_slot < 0 && nW > 0) {
            if (phase == 1) /* after the all-reduce of the list lengths (contact shards): no overlap */
                hipLaunchKernelGGL(k_tail, dim3(max_c, nW), dim3(256), 0, c->stream, c->rowptr, c->cc, c->tab, c->glob, c->mb, c->lgf_tab,
                                   g_tail_quirk, pz, w_begin);
            else
                hipStreamWaitEvent(c->stream, c->ev_tail, 0);
            TimedLaunch t(c, T_FINALIZE);
            hipLaunchKernelGGL(k_records, dim3(max_c, nW), dim3(64), 0, c->stream, c->mb, w_begin);
        }
    }
}

/* scores + argmax of slot w (or the forced choice of ig_apply) */
static void enqueue_choose(ig_ctx* c, int w, int force_slot)
{
    TimedLaunch t(c, T_ARGMAX);
    if (force_slot < 0) hipLaunchKernelGGL(k_scores, dim3(1), dim3(256), 0, c->stream, c->glob, c->mb, w);
    else hipLaunchKernelGGL(k_force_choice, dim3(1), dim3(1), 0, c->stream, c->glob, c->mb, force_slot);
}

/* one-move tail: exact delta, apply, genome distance, result record */
static void enqueue_apply(ig_ctx* c, int move, int w, int forced)
{
    const int N = c->N;
    const PzTab pz{c->pz_tab, c->pz_n};
    {
        TimedLaunch t(c, T_DELTA);
        hipLaunchKernelGGL(k_delta, dim3(DELTA_RB, 2, 1), dim3(SCORE_THREADS), 0, c->stream, c->rowptr, c->cc, c->tab, c->tab_prev,
                           c->prev_touched, c->glob, c->mb, c->lgf_tab, pz, w);
    }
    {
        TimedLaunch t(c, T_APPLY);
        hipLaunchKernelGGL(k_apply, dim3(64), dim3(256), 0, c->stream, c->st, c->tab, c->glob, c->mb, w, forced, c->prev_touched);
    }
    {
        TimedLaunch t(c, T_POST);
        hipLaunchKernelGGL(k_post, dim3((N + 255) / 256), dim3(256), 0, c->stream, c->st, c->init_prev, c->init_next, c->orientable,
                           c->black, c->glob, N);
    }
    {
        TimedLaunch t(c, T_COMMIT);
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, c->stream, c->glob, c->mb, c->d_results, move, w);
    }
}

static void enqueue_move(ig_ctx* c, int move, int max_c, int force_slot, int phase)
{
    enqueue_score(c, move, 1, max_c, force_slot, phase);
    if (phase == 1 || phase == 2) enqueue_choose(c, 0, force_slot);
}

static int validate_move(ig_ctx* c, int frag_a, const int32_t* cands, int C)
{
    if (C < 1 || C > IG_MAX_CANDIDATES) return fail("a move needs 1..%d candidates (got %d)", IG_MAX_CANDIDATES, C);
    if (frag_a < 0 || frag_a >= c->N) return fail("fragment %d out of range", frag_a);
    for (int i = 0; i < C; i++) {
        if (cands[i] < 0 || cands[i] >= c->N) return fail("candidate %d out of range", cands[i]);
        if (cands[i] == frag_a)
            return fail("candidate == focal fragment (%d): the reference reads stale buffers here (quirk Q13), not supported", frag_a);
    }
    return 0;
}

/* commit the scored batch [move0, move0 + w_now): k_commit_batch, the one-move tail for a windowed winner, resume */
static int commit_loop(ig_ctx* c, int done, int w_now, int* next_out)
{
    int next = 0; /* slots [0, next) of this batch are committed */
    for (;;) {
        {
            TimedLaunch t(c, T_COMMIT);
            hipLaunchKernelGGL(k_decide_batch, dim3(1), dim3(64), 0, c->stream, c->glob, c->mb, c->d_results, done, w_now, next,
                               c->dirty_buf, c->batch_out);
            hipLaunchKernelGGL(k_commit_batch, dim3(1), dim3(COMMIT_THREADS), 0, c->stream, c->st, c->tab, c->tab_prev, c->glob,
                               c->mb, c->init_prev, c->init_next, c->orientable, c->black, c->stamp, c->own_tag, c->own_idx,
                               c->prev_touched, c->d_results, done, w_now, next, c->batch_out);
        }
        int bo[4];
        HIPCK(hipMemcpyAsync(bo, c->batch_out, sizeof bo, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        if (next == 0) {
            c->n_batches++;
            c->large_seen = (bo[2] * 4 > bo[3]); /* a quarter of the windows above LDS_COL_SMALL: launch the large variant too */
        }
        c->n_batch_committed += bo[0] - next;
        next = bo[0];
        if (bo[1] >= 0) { /* slot bo[1] chose a windowed winner: delta + apply with the one-move kernels, then go on */
            enqueue_apply(c, done + bo[1], bo[1], 0);
            c->n_batch_pending++;
            next = bo[1] + 1;
            if (next < w_now) continue;
        }
        break;
    }
    if (next == 0) {
        Glob hg;
        HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
        return fail("device-side consistency failure %d in a batch at move %d", hg.error, done);
    }
    *next_out = next;
    return 0;
}

static int g_batch_w = -1; /* moves scored per launch in ig_step_batch: env IG_BATCH_W, default 24; 1 = one move at a time */

/* the per-slot work buffers are sized for the worst case (a window = the whole genome): keep them under ~64 GB */
static int max_batch_width(ig_ctx* c, int max_c)
{
    const double per_slot = (double)std::max(8, max_c) *
                            ((double)NSLOT * NDYN * c->N * 4.0 + (double)c->M * NSLOT * 8.0 + 3.0 * c->N * 4.0 + 2.0 * c->M * 4.0);
    const int fit = (int)std::max(1.0, 64e9 / std::max(per_slot, 1.0));
    return std::min(fit, IG_MAX_BATCH);
}

static int batch_width(ig_ctx* c, int max_c)
{
    if (g_batch_w < 0) {
        const char* e = getenv("IG_BATCH_W");
        g_batch_w = e ? atoi(e) : 24;
        g_batch_w = std::min(std::max(g_batch_w, 1), IG_MAX_BATCH);
    }
    return std::min(g_batch_w, max_batch_width(c, max_c));
}

extern "C" int ig_batch_max_width(ig_ctx* c, int32_t max_c) { return max_batch_width(c, max_c); }

/* validate and upload the pre-drawn (fragment, candidates) lists of a run of moves */
static int upload_moves(ig_ctx* c, int n_moves, const int32_t* frags, const int32_t* cands, int max_c)
{
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("max_c out of range");
    for (int i = 0; i < n_moves; i++) {
        int C = 0;
        while (C < max_c && cands[(size_t)i * max_c + C] >= 0) C++;
        for (int k = C; k < max_c; k++)
            if (cands[(size_t)i * max_c + k] >= 0) return fail("candidates must be packed before the -1 padding");
        if (validate_move(c, frags[i], cands + (size_t)i * max_c, C)) return -1;
    }
    if (ensure_io(c, n_moves, max_c)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, frags, (size_t)n_moves * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)n_moves * max_c * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->up_moves = n_moves;
    c->up_max_c = max_c;
    return 0;
}

static int download_results(ig_ctx* c, int n_moves, ig_move_result* results)
{
    HIPCK(hipMemcpyAsync(results, c->d_results, (size_t)n_moves * sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    for (int i = 0; i < n_moves; i++)
        if (results[i].error) return fail("device-side consistency failure %d at move %d", results[i].error, i);
    return 0;
}

extern "C" int ig_step_batch(ig_ctx* c, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c,
                             ig_move_result* results)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (n_moves <= 0) return 0;
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_step_batch: max_c out of range");
    const int Wmax = (c->world > 1) ? 1 : batch_width(c, max_c);
    if (ensure_move_buffers(c, std::max(8, (int)max_c), Wmax)) return -1;
    if (upload_moves(c, n_moves, frags, cands, max_c)) return -1;
    if (Wmax == 1) {
        for (int i = 0; i < n_moves; i++) {
            enqueue_move(c, i, max_c, -1, 2);
            enqueue_apply(c, i, 0, 0);
        }
    } else {
        /* speculative batches: score W moves against the same state, commit the conflict-free prefix on the device,
         * finish a winner that needs the exact delta pass with the one-move tail, continue after it */
        int done = 0;
        while (done < n_moves) {
            const int w_now = std::min(Wmax, n_moves - done);
            enqueue_score(c, done, w_now, max_c, -1, 2);
            int next = 0;
            if (commit_loop(c, done, w_now, &next)) return -1;
            done += next;
        }
    }
    return download_results(c, n_moves, results);
}

/* ---- the same, one step at a time, for callers that split the slots of a batch over several GPUs ---------- */

extern "C" int ig_batch_upload(ig_ctx* c, int32_t n_moves, const int32_t* frags, const int32_t* cands, int32_t max_c, int32_t max_w)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (n_moves <= 0) return fail("ig_batch_upload: no moves");
    if (max_w < 1 || max_w > max_batch_width(c, max_c))
        return fail("ig_batch_upload: batch width %d out of 1..%d (ig_batch_max_width)", max_w, max_batch_width(c, max_c));
    if (c->world > 1) return fail("ig_batch_upload: contact shards (ig_set_shard) and slot splitting are exclusive");
    if (max_c < 1 || max_c > IG_MAX_CANDIDATES) return fail("ig_batch_upload: max_c out of range");
    if (ensure_move_buffers(c, std::max(8, (int)max_c), max_w)) return -1;
    return upload_moves(c, n_moves, frags, cands, max_c);
}

extern "C" int ig_batch_score(ig_ctx* c, int32_t move0, int32_t W, int32_t slot_begin, int32_t slot_end)
{
    HIPCK(hipSetDevice(c->device));
    if (W < 1 || W > c->mb.capW || move0 < 0 || move0 + W > c->up_moves) return fail("ig_batch_score: batch out of range");
    if (slot_begin < 0 || slot_end > W || slot_begin > slot_end) return fail("ig_batch_score: slot range out of range");
    enqueue_score(c, move0, W, c->up_max_c, -1, 2, slot_begin, slot_end);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int ig_batch_records(ig_ctx* c, void** pre, int64_t* pre_bytes_per_slot, void** cpre, int64_t* cpre_bytes_per_slot)
{
    if (!c->mb.pre) return fail("ig_batch_records: no batch buffers yet (ig_batch_upload first)");
    *pre = c->mb.pre;
    *pre_bytes_per_slot = (int64_t)c->mb.capC * IG_N_TMP_STRUCT * (int64_t)sizeof(SlotPre);
    *cpre = c->mb.cpre;
    *cpre_bytes_per_slot = (int64_t)c->mb.capC * (int64_t)sizeof(CandPre);
    return 0;
}

extern "C" int ig_batch_commit(ig_ctx* c, int32_t move0, int32_t W, int32_t* n_committed)
{
    HIPCK(hipSetDevice(c->device));
    if (W < 1 || W > c->mb.capW || move0 < 0 || move0 + W > c->up_moves) return fail("ig_batch_commit: batch out of range");
    int next = 0;
    if (commit_loop(c, move0, W, &next)) return -1;
    *n_committed = next;
    return 0;
}

extern "C" int ig_batch_results(ig_ctx* c, int32_t n_moves, ig_move_result* results)
{
    HIPCK(hipSetDevice(c->device));
    if (n_moves < 0 || n_moves > c->up_moves) return fail("ig_batch_results: out of range");
    return download_results(c, n_moves, results);
}

extern "C" int ig_set_batch_width(int w)
{
    g_batch_w = std::min(std::max(w, 1), IG_MAX_BATCH);
    return 0;
}

extern "C" int ig_batch_stats(ig_ctx* c, int64_t out3[3])
{
    out3[0] = c->n_batches;
    out3[1] = c->n_batch_committed;
    out3[2] = c->n_batch_pending;
    return 0;
}

extern "C" int ig_step(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C, ig_move_result* out, double* scores)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 2);
    if (scores) HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    enqueue_apply(c, 0, 0, 0);
    HIPCK(hipMemcpyAsync(out, c->d_results, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    if (out->error) return fail("device-side consistency failure %d", out->error);
    return 0;
}

extern "C" int ig_score_move(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C, double* scores)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 2);
    HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    return 0;
}

extern "C" int ig_apply(ig_ctx* c, int32_t frag_a, int32_t frag_b, int32_t op)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (op < 0 || op >= IG_N_TMP_STRUCT) return fail("ig_apply: op out of range");
    if (validate_move(c, frag_a, &frag_b, 1)) return -1;
    if (ensure_move_buffers(c, 8)) return -1;
    if (ensure_io(c, 1, 1)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, &frag_b, sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, 1, op, 2);
    enqueue_apply(c, 0, 0, 1);
    ig_move_result r;
    HIPCK(hipMemcpyAsync(&r, c->d_results, sizeof r, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    if (r.error) return fail("device-side consistency failure %d", r.error);
    return 0;
}

extern "C" int ig_set_shard(ig_ctx* c, int32_t rank, int32_t world)
{
    if (world < 1 || rank < 0 || rank >= world) return fail("ig_set_shard: bad shard %d/%d", rank, world);
    HIPCK(hipStreamSynchronize(c->stream));
    c->rank = rank;
    c->world = world;
    return 0;
}

extern "C" int64_t ig_partials_count(ig_ctx* c) { return (int64_t)c->mb.capC * P_STRIDE; }
extern "C" void* ig_partials_device_ptr(ig_ctx* c) { return c->mb.part; }

extern "C" int ig_step_begin(ig_ctx* c, int32_t frag_a, const int32_t* cands, int32_t C)
{
    HIPCK(hipSetDevice(c->device));
    if (check_ready(c)) return -1;
    if (validate_move(c, frag_a, cands, C)) return -1;
    if (ensure_move_buffers(c, std::max(8, (int)C))) return -1;
    if (ensure_io(c, 1, C)) return -1;
    HIPCK(hipMemcpyAsync(c->d_frags, &frag_a, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCK(hipMemcpyAsync(c->d_cands, cands, (size_t)C * sizeof(int), hipMemcpyHostToDevice, c->stream));
    enqueue_move(c, 0, C, -1, 0);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int ig_step_finish(ig_ctx* c, ig_move_result* out, double* scores)
{
    HIPCK(hipSetDevice(c->device));
    int C = 0;
    enqueue_move(c, 0, IG_MAX_CANDIDATES, -1, 1);
    (void)C;
    if (scores) {
        MoveCtl mc;
        HIPCK(hipMemcpyAsync(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        HIPCK(hipMemcpyAsync(scores, c->mb.scores, (size_t)mc.C * IG_N_TMP_STRUCT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    enqueue_apply(c, 0, 0, 0);
    HIPCK(hipMemcpyAsync(out, c->d_results, sizeof(ig_move_result), hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipGetLastError());
    drain_timers(c);
    if (out->error) return fail("device-side consistency failure %d", out->error);
    return 0;
}

extern "C" int ig_kernel_time_ms(ig_ctx* c, const char* name, double* avg_ms, int64_t* n)
{
    drain_timers(c);
    for (int i = 0; i < T_COUNT; i++)
        if (!strcmp(name, c->timers[i].name)) {
            if (avg_ms) *avg_ms = c->timers[i].n ? c->timers[i].total_ms / (double)c->timers[i].n : 0.0;
            if (n) *n = c->timers[i].n;
            return 0;
        }
    return fail("ig_kernel_time_ms: unknown kernel '%s'", name);
}

extern "C" int ig_reset_timers(ig_ctx* c, int enable)
{
    HIPCK(hipStreamSynchronize(c->stream));
    drain_timers(c);
    for (int i = 0; i < T_COUNT; i++) {
        c->timers[i].total_ms = 0;
        c->timers[i].n = 0;
    }
    c->timing = enable != 0;
    c->timing_mask = enable > 1 ? (unsigned)(enable >> 1) : 0xffffu; /* enable = 1 | (mask << 1) */
    return 0;
}

/* ------------------------------------------------------------------ debug ABI */

extern "C" int ig_debug_eval_terms(ig_ctx* c, const float* s, const float* s_tot, const int32_t* ob, int64_t n, float* ex, float* exc,
                                   double* term, int64_t* q)
{
    HIPCK(hipSetDevice(c->device));
    if (!c->have_params) return fail("ig_debug_eval_terms: set parameters first");
    float *ds, *dst, *dex, *dexc;
    int* dob;
    double* dterm;
    long long* dq;
    DALLOC(ds, (size_t)n);
    DALLOC(dst, (size_t)n);
    DALLOC(dob, (size_t)n);
    DALLOC(dex, (size_t)n);
    DALLOC(dexc, (size_t)n);
    DALLOC(dterm, (size_t)n);
    DALLOC(dq, (size_t)n);
    HIPCK(hipMemcpy(ds, s, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dst, s_tot, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dob, ob, n * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_terms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ds, dst, dob, (long long)n, c->glob,
                       c->lgf_tab, dex, dexc, dterm, dq);
    HIPCK(hipStreamSynchronize(c->stream));
    HIPCK(hipMemcpy(ex, dex, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(exc, dexc, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(term, dterm, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(q, dq, n * sizeof(long long), hipMemcpyDeviceToHost));
    hipFree(ds);
    hipFree(dst);
    hipFree(dob);
    hipFree(dex);
    hipFree(dexc);
    hipFree(dterm);
    hipFree(dq);
    return 0;
}

extern "C" int ig_debug_candidate_state(ig_ctx* c, int32_t cand, int32_t slot, int32_t* soa)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    if (cand < 0 || cand >= c->mb.capC || slot < 0 || slot > IG_N_TMP_STRUCT) return fail("ig_debug_candidate_state: bad index");
    const size_t n = c->N;
    CandMeta m;
    HIPCK(hipMemcpy(&m, c->mb.meta + cand, sizeof m, hipMemcpyDeviceToHost));
    if (m.kidx[slot] < 0) return fail("ig_debug_candidate_state: slot %d was not materialised for candidate %d", slot, cand);
    /* start from the live genome (internal ids), overlay the local window */
    std::vector<int> host(17 * n);
    HIPCK(hipMemcpy(host.data(), c->st_block, 17 * n * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> loc((size_t)NDYN * n), gid(n);
    HIPCK(hipMemcpy(loc.data(), c->mb.loc + ((size_t)(cand * NSLOT + slot) * NDYN) * n, (size_t)NDYN * n * sizeof(int), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(gid.data(), c->mb.Lloc + (size_t)cand * n, n * sizeof(int), hipMemcpyDeviceToHost));
    for (int x = 0; x < m.n_loc; x++)
        for (int k = 0; k < NDYN; k++) host[k * n + gid[x]] = loc[k * n + x];
    static const int dyn_src[NDYN] = {0, 1, 2, 3, 6, 8, 9, 10, 11, 12, 13};
    for (int k = 0; k < NDYN; k++) memcpy(soa + dyn_src[k] * n, &host[k * n], n * sizeof(int));
    memcpy(soa + 4 * n, &host[11 * n], n * sizeof(int));
    memcpy(soa + 5 * n, &host[12 * n], n * sizeof(int));
    memcpy(soa + 14 * n, &host[14 * n], n * sizeof(int));
    memcpy(soa + 15 * n, &host[15 * n], n * sizeof(int));
    memcpy(soa + 16 * n, &host[16 * n], n * sizeof(int));
    for (size_t f = 0; f < n; f++) soa[7 * n + f] = (int)f;
    return 0;
}

extern "C" int ig_debug_last_sums(ig_ctx* c, int64_t* nz_hi, int64_t* nz_lo, int64_t* z_hi, int64_t* z_lo, int64_t* n_intra,
                                  int64_t* ext_hi, int64_t* ext_lo, int64_t* n_slice, int32_t* n_uniq, int32_t* uniq)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    MoveCtl mc;
    HIPCK(hipMemcpy(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost));
    const int C = mc.C;
    std::vector<long long> part((size_t)C * P_STRIDE), qp((size_t)C * Q_STRIDE);
    std::vector<CandMeta> meta(C);
    HIPCK(hipMemcpy(part.data(), c->mb.part, part.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(qp.data(), c->mb.qpart, qp.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(meta.data(), c->mb.meta, C * sizeof(CandMeta), hipMemcpyDeviceToHost));
    for (int cc_ = 0; cc_ < C; cc_++) {
        const long long* p = &part[(size_t)cc_ * P_STRIDE];
        const long long* q = &qp[(size_t)cc_ * Q_STRIDE];
        n_uniq[cc_] = meta[cc_].n_uniq;
        n_slice[cc_] = p[P_CNT];
        int64_t eh = p[P_NZ], el = p[P_NZ + 1];
        ig_acc_normalize(&eh, &el);
        ext_hi[cc_] = eh;
        ext_lo[cc_] = el;
        for (int s = 0; s < IG_N_TMP_STRUCT; s++) {
            const int o = cc_ * IG_N_TMP_STRUCT + s;
            uniq[o] = -1;
            nz_hi[o] = nz_lo[o] = z_hi[o] = z_lo[o] = n_intra[o] = 0;
        }
        for (int k = 1; k <= meta[cc_].n_uniq; k++) {
            const int s = meta[cc_].uniq[k - 1];
            const int o = cc_ * IG_N_TMP_STRUCT + s;
            uniq[cc_ * IG_N_TMP_STRUCT + (k - 1)] = s;
            int64_t h = q[Q_NZFULL + 2 * k], l = q[Q_NZFULL + 2 * k + 1];
            const int r = (int)(p[P_CNT] % 64);
            if (r > 0 && (k - 1) >= r) { /* quirk Q5 */
                h -= q[Q_TAIL + 2 * k];
                l -= q[Q_TAIL + 2 * k + 1];
            }
            ig_acc_normalize(&h, &l);
            nz_hi[o] = h;
            nz_lo[o] = l;
            h = hg.z_hi + q[Q_Z + 2 * k] - q[Q_Z];
            l = hg.z_lo + q[Q_Z + 2 * k + 1] - q[Q_Z + 1];
            ig_acc_normalize(&h, &l);
            z_hi[o] = h;
            z_lo[o] = l;
            n_intra[o] = hg.n_intra + q[Q_NI + k] - q[Q_NI];
        }
    }
    return 0;
}

extern "C" int ig_debug_tables(ig_ctx* c, float* dist, int32_t* id_c, float* s_tot, int32_t* pos, int32_t* len)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    const size_t m = c->M;
    HIPCK(hipMemcpy(dist, c->tab.dist, m * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(s_tot, c->tab.stot, m * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(len, c->tab.len, m * 4, hipMemcpyDeviceToHost));
    std::vector<int2> cp(m);
    HIPCK(hipMemcpy(cp.data(), c->tab.cp, m * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; i++) {
        id_c[i] = cp[i].x;
        pos[i] = cp[i].y;
    }
    return 0;
}

extern "C" int ig_debug_set_tail_quirk(int on)
{
    g_tail_quirk = on;
    return 0;
}

/* maintained exact sums {nz_hi, nz_lo, z_hi, z_lo, n_intra} and {n_contigs, next_cid, ch_c, ch_k, ch_slot, ch_windowed} */
extern "C" int ig_debug_globals(ig_ctx* c, int64_t* sums5, int32_t* ints6)
{
    HIPCK(hipSetDevice(c->device));
    HIPCK(hipStreamSynchronize(c->stream));
    Glob hg;
    HIPCK(hipMemcpy(&hg, c->glob, sizeof hg, hipMemcpyDeviceToHost));
    sums5[0] = hg.nz_hi;
    sums5[1] = hg.nz_lo;
    sums5[2] = hg.z_hi;
    sums5[3] = hg.z_lo;
    sums5[4] = hg.n_intra;
    ints6[0] = hg.n_contigs;
    ints6[1] = hg.next_cid;
    MoveCtl mc;
    memset(&mc, 0, sizeof mc);
    if (c->mb.ctl) HIPCK(hipMemcpy(&mc, c->mb.ctl, sizeof mc, hipMemcpyDeviceToHost));
    ints6[2] = mc.ch_c;
    ints6[3] = mc.ch_k;
    ints6[4] = mc.ch_slot;
    ints6[5] = mc.ch_windowed;

    return 0;
}
