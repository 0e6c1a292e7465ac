/* ig_kernels_commit.cuh -- choosing and applying: the one-move kernels (k_scores, k_apply, k_commit) and the batch
 * commit (k_decide_batch, k_commit_batch). */
#pragma once

/* scores of one move slot (eval_all_likelihood_on_zero_2nd KA:4005-4027, eval_all_scores KA:4029-4046) and the
 * host argmax of CL:1435-1446 (zeros -> -inf, shifted/clipped scores, FIRST index of the maximum).  Executed by
 * one workgroup; `vf0` = the stale insert flags the first candidate sees (quirk Q4). */
__device__ void score_and_choose(Glob* g, const MoveBuf& mb, int w, const int* vf0, double* sc_lds /* [C*24] */)
{
    const int tid = threadIdx.x;
    MoveCtl& mc = mb.ctl[PS(w)];
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
        const int r = (int)(slice_total(part) % 64);
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
                const long long Sc = slice_total(mb.part + (size_t)CW(w, c) * P_STRIDE);
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
    if (threadIdx.x == 0 && mb.ctl[PS(w)].overflow) g->retry_pool = 1; /* (the scores below miss the lists that did not fit: nothing is applied) */
    __syncthreads();
    score_and_choose(g, mb, w, vf, sc);
    const int n = mb.ctl[PS(w)].C * IG_N_TMP_STRUCT;
    for (int i = threadIdx.x; i < n; i += blockDim.x) mb.scores[(size_t)CW(w, 0) * IG_N_TMP_STRUCT + i] = sc[i];
}

/* forced choice for ig_apply (test_copy_struct / apply_replay_simu, CL:2094-2151, 2546-2553) */
__global__ void k_force_choice(Glob* g, MoveBuf mb, int slot)
{
    MoveCtl& mc = mb.ctl[0];
    if (mc.overflow) g->retry_pool = 1;
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
    MoveCtl& mc = mb.ctl[PS(w)];
    const int c = mc.ch_c, slot = mc.ch_slot, k = mc.ch_k;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int N = mb.sN, M = mb.sM; /* strides of the window arrays */
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
        atomicMax(&g->max_L, m.n_loc); /* no contig made by this move is longer than its window */
        atomicMax(&g->max_SL, m.m_loc);
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
    if (g->error || g->retry_pool) return;
    apply_winner(st, tab, g, mb, w, forced, prev_touched, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, false);
}

__device__ __forceinline__ void write_result(Glob* g, const MoveBuf& mb, int w, ig_move_result* out)
{
    const MoveCtl& mc = mb.ctl[PS(w)];
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

__device__ __forceinline__ void publish_final(const Glob* g, StepHost* hs, int seq, const ig_move_result* r)
{
    if (!hs) return;
    hs->fin = *r;
    hs->max_L = g->max_L;
    hs->max_SL = g->max_SL;
    __threadfence_system();
    hs->fin_seq = seq;
}

__global__ void k_commit(Glob* g, MoveBuf mb, ig_move_result* res, int move, int w, int* dirty, StepHost* hs, int seq)
{
    if (g->retry_pool) { /* the move's lists did not fit the slice pool: nothing was applied, the host repeats it with a larger one */
        g->credit2_acc = 0;
        write_result(g, mb, w, res + move);
        res[move].pad = 1;
        publish_final(g, hs, seq, res + move);
        return;
    }
    /* dirty (a move of a batch finished with the one-move kernels, decided one move per call: ig_nuis_step_begin): its contigs
     * onto the batch's list of modified contigs, as k_decide_batch does for the moves it commits itself */
    if (dirty) {
        const MoveCtl& mc = mb.ctl[PS(w)];
        const CandMeta& m = mb.meta[CW(w, mc.ch_c)];
        const int n = dirty[0];
        if (n + 2 <= 2 * IG_MAX_BATCH + 2) {
            dirty[1 + n] = m.ctgA;
            dirty[2 + n] = m.ctgB;
            dirty[0] = n + 2;
        }
    }
    g->next_cid += NFRESH;
    g->credit2 = g->credit2_acc;
    g->credit2_acc = 0;
    write_result(g, mb, w, res + move);
    publish_final(g, hs, seq, res + move);
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

/* one step of a wave-wide argmax (larger score wins, the lower index among equals): every lane combines its pair with the one a
 * DPP move brings (row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143); a lane the move does not reach keeps its
 * own pair (the `old` operand), which the combination leaves as it is */
template <int DPP_CTRL, int ROW_MASK>
__device__ __forceinline__ void wave_argmax_step(double& v, int& i)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, DPP_CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, DPP_CTRL, ROW_MASK, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(i, i, DPP_CTRL, ROW_MASK, 0xf, false);
    const double ov = __hiloint2double(ohi, olo);
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

/* k_commit_batch: the sequential half of a batch, one workgroup.
 *
 * 1. DECIDE (wave 0, no barriers): for w = 0, 1, ...: stop if a contig of move w was modified by an earlier move of
 *    this batch (its scores were computed against a stale state) or if its slice did not fit the pool; otherwise score
 *    and argmax with the LIVE scalars (kept in registers) from the slot-major records of k_records, update the scalars,
 *    write the result record.  A winner whose slice was windowed and that changes the genome needs the exact k_delta
 *    pass: the batch stops BEFORE it (pending) and the host finishes that move with the one-move kernels.
 * 2. APPLY (whole workgroup): the committed moves touch pairwise disjoint contigs, so their winners are applied
 *    together: ownership marks, exact genome-distance deltas (each move's credits evaluated on the genome as of just
 *    before / just after that move, read through the marks), state + coordinate tables, the distance column of the
 *    results.  tab_prev receives every committed move but the last one (quirk Q12: tables before the last move). */
#define COMMIT_THREADS 1024
#ifndef COMMIT_GROUP
#define COMMIT_GROUP 512 /* threads that apply one move together */
#endif
/* step 1 of the batch commit: ONE wave (it may use the whole register file: the data of the next move is held in
 * registers while the current one is decided) */
/* FUSED (k_decide_commit): the wave hands every decision to the commit wave of its workgroup through LDS -- the record, the few
 * words of the control block the apply step reads, then prog[0] = slots decided so far; prog[1] = 1 + the final count once it is
 * through.  LDS operations of a wave complete in order: no fence (a release would wait for the prefetches of the next three
 * moves, i.e. one memory round trip per decision: measured 38 k instead of 43 k moves/s).  The record goes to memory from the
 * commit wave alone (two waves storing to the same words without a fence between them land in either order). */
#define FUSED_CHG_CAP 512
struct FusedLds {
    volatile int prog[2];
    volatile int bar[3 * IG_MAX_BATCH + 2]; /* arrivals at the commit waves' barriers (three per move, one behind their prologue) */
    long long delta[IG_MAX_BATCH];          /* a move's change of the genome-distance credits, summed over the commit waves */
    int fin_max_L, fin_max_SL;
    ig_move_result rec[IG_MAX_BATCH];
    int ch_c[IG_MAX_BATCH], ch_slot[IG_MAX_BATCH], ch_k[IG_MAX_BATCH], n_dirty[IG_MAX_BATCH], vmask[IG_MAX_BATCH];
    long long nzb_hi[IG_MAX_BATCH], nzb_lo[IG_MAX_BATCH];
    /* the statistics columns, summed over the candidates ahead of the decisions (they depend on the stale-flag mask of the decision
     * through candidate 0's list length alone: its terms apart) */
    long long st_sc[IG_MAX_BATCH], st_ev[IG_MAX_BATCH], st_by[IG_MAX_BATCH];
    long long s0_slice[IG_MAX_BATCH];
    int s0_mloc[IG_MAX_BATCH], s0_base[IG_MAX_BATCH];
    int n_chg[IG_MAX_BATCH]; /* per move: fragments whose prev / next / ori change, */
    int chg[FUSED_CHG_CAP];  /* ... and the local indices of the move being applied */
};
/* CHAIN (k_decide_chain): a segment of a chain of (move, nuisance step) pairs -- ig_common.cuh, ChainIn.  Behind every decision
 * the step's Metropolis test (CL:3026-3036: exp((L_test - L_move) / T) >= u) against the interval k_chain_hist_eval left for its
 * test set: L_test = the maintained exact sum as of before the move + D +- B, + the zero-pixel likelihood of the set; a step
 * whose whole interval lies below T ln u (with the margins nuis_end_body keeps on the host) is rejected and the wave goes on;
 * anything else stops the segment IN FRONT of the pair (stop code 4: nothing of it is committed, the host takes it through
 * ig_nuis_step_begin / ig_nuis_step_next).  A move that changes the genome is committed, its step tested -- the interval is that of
 * the state before the move --, and the segment ends behind it (stop code 5): the histogram follows the move first. */
struct ChainArgs {
    const ChainTest* tests;
    const ChainIn* in;
    int set0;
    double temperature;
};
template <bool FUSED, bool CHAIN = false>
__device__ __forceinline__ void decide_body(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out,
                                            volatile int* host_out, int seq, int resumed_plain, FusedLds* sh, int zcheck,
                                            const ChainArgs ca = ChainArgs{nullptr, nullptr, 0, 1.0}, double* host_scores = nullptr)
{
    /* host_scores (ig_step_draw: a launch that decides ONE move): the move's C x 24 scores as all_scores holds them (CL:1414, 1431),
     * into mapped host memory */
    /* zcheck bit 0: the records hold the CONTENDER columns only (two-tier scoring).  k_contend ruled the others out against lower
     * bounds of columns it took for scored -- under the batch-start scalars; a column whose score comes out as exactly 0.0 under the
     * LIVE ones counts as "not scored" in the reference's argmax (CL:1435-1440) and bounds nothing: the batch stops in front of such
     * a move (stop code 3) and the host scores it again with every column exact.  zcheck >> 8: fault injection for the tests
     * (ig_debug_set_zero_inject): every that-many-th move is treated as such a move. */
    /* w_start > 0: slot w_start - 1 was the pending move, meanwhile applied by the one-move kernels; the rest of the batch
     * is still valid wherever it does not touch a contig modified so far (dirty_buf carries the list across the calls).
     * resumed_plain: slot w_start - 1 was committed by an earlier launch of this kernel (a batch decided one move per call,
     * ig_nuis_step_begin): its contigs are on the list already */
    /* the contigs modified so far: entry q lives in lane q % 64 (register q / 64) -- the test of a move against the list is a
     * handful of compares and a ballot per candidate instead of a walk over an LDS array (it was a third of a decision) */
    /* one wave, often next to a pass that fills the machine (the nuisance step's, on its own stream): first in line for its
     * SIMD's issue slots (it took 70 us instead of 12 next to the persistent tiles kernel) */
    __builtin_amdgcn_s_setprio(3);
    constexpr int ND = (IG_MAX_BATCH * 2 + 2 + 63) / 64;
    int dirty[ND];
#pragma unroll
    for (int j = 0; j < ND; j++) dirty[j] = -2; /* no contig has this id */
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
        int n_dirty = 0, committed = w_start, pending = -1, n_cand = 0, n_predicted = 0, stop_overflow = 0;
        int max_L = g->max_L, max_SL = g->max_SL;
        const float n_frags_f = (float)g->N; /* not inside the loop: a load there waits for the prefetches issued before it */
        if (w_start > 0) {
            n_dirty = dirty_buf[0];
#pragma unroll
            for (int j = 0; j < ND; j++)
                if (lane + 64 * j < n_dirty) dirty[j] = dirty_buf[1 + lane + 64 * j];
            if (w_start > 0 && !(resumed_plain & 1)) {
                const MoveCtl pm = mb.ctl[PS(w_start - 1)];
                const CandMeta& m = mb.meta[CW(w_start - 1, pm.ch_c)];
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    if (lane + 64 * j == n_dirty) dirty[j] = m.ctgA;
                    if (lane + 64 * j == n_dirty + 1) dirty[j] = m.ctgB;
                }
                n_dirty += 2;
            }
        }
        /* Everything a decision reads is loaded THREE MOVES AHEAD (none of it depends on earlier decisions, only its
         * interpretation does): while move w is decided from registers with wave shuffles only, the loads of moves
         * w + 1 .. w + 3 are in flight.  Moves with more than 5 candidates (> 2 score records per lane) take the unpipelined path. */
        struct MoveData {
            int C, superset0;       /* uniform */
            CandPre cand;           /* lane c < C: candidate c */
            SlotPre rec[2];         /* score records lane, lane + 64 */
            double e_ext_d[2];      /* their candidates' slice sum under the current genome, */
            int e_r[2], e_base[2];  /* S_c mod 64, list entries before the block inserts */
        };
        /* the candidate counts of all slots up front (lane w: slot w): a prefetch must not wait for its own first load */
        const int all_C = (lane < W) ? mb.ctl[PS(lane)].C : 0;
        const int all_sup = (lane < W) ? mb.ctl[PS(lane)].superset0 : 0;
        /* CHAIN: the interval of step w_start + lane in lane `lane` */
        long long ct_s = 0, ct_b = 0, ct_f = 1;
        double ct_z = 0.0, ct_lnu = -IG_INF;
        bool halt_after = false;
        int chain_changed = 0;
        if (CHAIN && lane < W - w_start && lane < CHAIN_SEG) {
            const ChainTest t = ca.tests[lane];
            ct_s = t.s_fix;
            ct_b = t.b_fix;
            ct_f = t.flags;
            ct_z = t.z;
            ct_lnu = ca.in[ca.set0 + lane].ln_u;
        }
        auto load_move = [&](int w) {
            MoveData d;
            d.C = __builtin_amdgcn_readlane(all_C, w);
            d.superset0 = __builtin_amdgcn_readlane(all_sup, w);
            /* NOTHING here may touch a loaded value or load under a condition: either makes the wave wait for the data at
             * once (lanes past the end load a valid entry instead and are never read: every use below is under c < C or
             * i < C * IG_N_TMP_STRUCT) */
            d.cand = cpre_w(mb, w, lane < d.C ? lane : 0);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = max(min(lane + 64 * j, d.C * IG_N_TMP_STRUCT - 1), 0);
                d.rec[j] = pre_w(mb, w, i); /* (the records of a slot are one array, entry i = candidate * IG_N_TMP_STRUCT + column) */
                const CandPre& cp = cpre_w(mb, w, i / IG_N_TMP_STRUCT);
                d.e_ext_d[j] = cp.ext_d;
                d.e_r[j] = cp.r;
                d.e_base[j] = cp.base_cnt;
            }
            return d;
        };
        auto rl = [](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
        auto rl64 = [](long long v, int src) {
            const int lo = __builtin_amdgcn_readlane((int)(unsigned)v, src), hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
            return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
        };
        auto rld = [&](double v, int src) { return __longlong_as_double(rl64(__double_as_longlong(v), src)); };
        /* one decision: false = the batch stops before (conflict, pool) or at (pending) move w */
        auto decide_one = [&](const int w, const MoveData& d) -> bool {
            const int C = d.C;
            /* conflict with an earlier move of this batch?  slice pool overflow? */
            bool hitd = false;
            for (int cq = 0; cq < C; cq++) { /* candidate cq's two contigs against every lane's entries */
                const int qa = rl(d.cand.ctgA, cq), qb = rl(d.cand.ctgB, cq);
#pragma unroll
                for (int j = 0; j < ND; j++) hitd |= (dirty[j] == qa) | (dirty[j] == qb);
            }
            if (err0 || rl(d.cand.overflow, 0) || __any(hitd)) {
                stop_overflow = err0 ? 0 : rl(d.cand.overflow, 0); /* 1: the slice pool, 2: the exact kernel's grid */
                return false;
            }
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
            bool zhit = false; /* a scored column whose live score is exactly 0.0 */
            auto score_of = [&](int i, const SlotPre& r, double ext, int cr, int cbase) -> double {
                const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
                const bool sup = (c == 0) && d.superset0 && (slot >= 12);
                const bool scored = (i < n) && (r.k > 0) && !(sup && !((vmask >> (slot - 12)) & 1u));
                if (!scored) return 0.0;
                const int pos = sup ? cbase + __popc(vmask & ((1u << (slot - 12)) - 1u)) : r.k - 1;
                const double nzd = (cr > 0 && pos >= cr) ? r.nz_cut_d : r.nz_d; /* quirk Q5 */
                const double val_inter = -1.0 * log_e * (n_tot_pxl - (double)(n_intra + r.dni)) * p.v_inter;
                const double val_intra = ig_acc_to_double(z_hi + r.dz_hi, z_lo + r.dz_lo) * log_e;
                const double z = val_intra + val_inter;
                const double v = nzd + z + cur_nz - ext;
                zhit |= (v == 0.0);
                return v;
            };
#pragma unroll
            for (int j = 0; j < NJ; j++) sc[j] = 0.0;
#pragma unroll
            for (int j = 0; j < 2; j++) { /* the prefetched records: every move with <= 5 candidates */
                const int i = lane + 64 * j;
                const double v = score_of(i, d.rec[j], d.e_ext_d[j], d.e_r[j], d.e_base[j]);
                sc[j] = v;
                const double ok = (v == 0.0) ? -IG_INF : v;
                if (i < n && ok > bestv) { /* strictly greater: the lower index wins inside a lane */
                    bestv = ok;
                    best = i;
                }
            }
            if (n > 128) { /* more than 5 candidates: the rest straight from memory */
#pragma unroll
                for (int j = 2; j < NJ; j++) {
                    const int i = lane + 64 * j;
                    double v = 0.0;
                    if (i < n) {
                        const CandPre cp = cpre_w(mb, w, i / IG_N_TMP_STRUCT);
                        v = score_of(i, pre_w(mb, w, i), cp.ext_d, cp.r, cp.base_cnt);
                    }
                    sc[j] = v;
                    const double ok = (v == 0.0) ? -IG_INF : v;
                    if (i < n && ok > bestv) {
                        bestv = ok;
                        best = i;
                    }
                }
            }
            if (host_scores) {
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    if (lane + 64 * j < n) host_scores[lane + 64 * j] = sc[j];
                __threadfence_system(); /* (the record that says they are there is published by another wave, behind this wave's progress word) */
            }
            if (zcheck & 1) {
                const int inj = zcheck >> 8;
                if (__any(zhit) || (inj > 0 && (move0 + w) % inj == inj - 1)) {
                    n_cand -= C;
                    stop_overflow = 3;
                    return false;
                }
            }
            /* (score, index) reduced into lane 63 with DPP moves: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then lane 15 of
             * rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2, 3 -- six steps of three moves each instead of six rounds of three
             * ds_bpermute (a quarter of a decision's time: one wave cannot hide an LDS round trip) */
            wave_argmax_step<0x111, 0xf>(bestv, best);
            wave_argmax_step<0x112, 0xf>(bestv, best);
            wave_argmax_step<0x114, 0xf>(bestv, best);
            wave_argmax_step<0x118, 0xf>(bestv, best);
            wave_argmax_step<0x142, 0xa>(bestv, best);
            wave_argmax_step<0x143, 0xc>(bestv, best);
            best = rl(best, 63);
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
                br.info = (unsigned)rl((int)mine.info, owner);
                bests = rld((bj == 0) ? sc[0] : sc[1], owner);
            } else { /* through readlane as well: a load still pending at the join would make every move wait for all prefetches */
                const SlotPre ld = pre_w(mb, w, best);
                br.nz_hi = rl64(ld.nz_hi, 0);
                br.nz_lo = rl64(ld.nz_lo, 0);
                br.dz_hi = rl64(ld.dz_hi, 0);
                br.dz_lo = rl64(ld.dz_lo, 0);
                br.dni = rl64(ld.dni, 0);
                br.k = rl(ld.k, 0);
                br.info = (unsigned)rl((int)ld.info, 0);
                double sv = 0.0;
#pragma unroll
                for (int j = 2; j < NJ; j++) sv = (bj == j) ? sc[j] : sv;
                bests = rld(sv, owner);
            }
            if (n == 0) br.k = 0; /* a move without candidates: nothing was scored (error 3 below) */
            const int b_sw = rl(d.cand.same_windowed, bc), b_B = rl(d.cand.B, bc), b_nloc = rl(d.cand.n_loc, bc);
            const int windowed = (b_sw >> 1) & 1, b_same = b_sw & 1;
            const int b_cA = rl(d.cand.ctgA, bc), b_cB = rl(d.cand.ctgB, bc);
            const long long b_ext_hi = rl64(d.cand.ext_hi, bc), b_ext_lo = rl64(d.cand.ext_lo, bc);
            const int br_changed = (int)(br.info & 1u), br_heads = (int)(br.info >> 1);
            /* statistics of the move: off the critical path, k_commit_batch fills them in from the flag mask kept here
             * (a pending move needs them now: its record is written by the one-move kernels) */
            long long Sc = 0, ev = 0, by = 0;
            /* a windowed winner that changes the genome needs the exact full-contig delta: predicted and computed in advance
             * (k_predict), or the batch pauses here */
            const int pred = rl(d.cand.pred, 0);
            const bool have_delta = windowed && br_changed && (best == pred);
            const bool is_pending = windowed && br_changed && !have_delta;
            if (is_pending) {
                if (lane < C) {
                    int nu = d.cand.n_uniq;
                    if (lane == 0 && d.superset0) nu = d.cand.base_cnt + __popc(vmask); /* the list the reference would have scored */
                    Sc = d.cand.n_slice;
                    ev = Sc * (nu + 1);
                    by = 12 * Sc + 20LL * d.cand.m_loc * nu + 8LL * nu;
                }
                Sc = rl64(wave_sum_ll(Sc), 0);
                ev = rl64(wave_sum_ll(ev), 0);
                by = rl64(wave_sum_ll(by), 0);
            }
            if (CHAIN && !is_pending) { /* the step's test: a certain rejection, or the segment stops in front of this pair */
                const int kt = w - w_start;
                const long long t_b = rl64(ct_b, kt), t_f = rl64(ct_f, kt);
                const double mid = cur_nz + (double)rl64(ct_s, kt) * (1.0 / DIFF_FIX);
                const double B = (double)t_b * (1.0 / DIFF_FIX) + 1e-6 + 1e-14 * (__builtin_fabs(cur_nz) + __builtin_fabs(bests));
                const double x_hi = (((mid + B) + rld(ct_z, kt)) - bests) / ca.temperature;
                /* exp(x_hi) <= u (1 - 1e-9) on the host; here in the exponent, with twice the margin (ln u is the host's double) */
                const bool rej = (kt < CHAIN_SEG) && (t_f == 0) && (t_b > 0) && (x_hi <= rld(ct_lnu, kt) - 2e-9);
                if (!rej) {
                    n_cand -= C;
                    stop_overflow = 4;
                    return false;
                }
            }
            long long nzb_hi_w = 0, nzb_lo_w = 0;
            const unsigned vmask_w = vmask; /* the stale flags this move was scored under (vmask moves on below) */
            if (lane == 0) {
                MoveCtl& o = mb.ctl[PS(w)];
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
                o.nzb_hi = nz_hi; /* the state this move was scored against (the nuisance step's screened pass starts from it) */
                o.nzb_lo = nz_lo;
                nzb_hi_w = nz_hi;
                nzb_lo_w = nz_lo;
                o.n_dirty = br_changed;
                o.pad = (int)vmask; /* the stale flags this move was scored under */
                if (br.k <= 0) g->error = 3; /* an unscored slot won: cannot happen */
            }
            if (is_pending) { /* needs k_delta: hand this move to the one-move tail */
                pending = w;
                return false;
            }
            /* commit: scalars (exact), stale-flag state (quirk Q4), fresh ids */
            if (have_delta) {
                nz_hi += rl64(d.cand.pd_hi, 0);
                nz_lo += rl64(d.cand.pd_lo, 0);
                n_predicted++;
            } else {
                nz_hi += br.nz_hi - b_ext_hi;
                nz_lo += br.nz_lo - b_ext_lo;
            }
            ig_acc_normalize((int64_t*)&nz_hi, (int64_t*)&nz_lo);
            z_hi += br.dz_hi;
            z_lo += br.dz_lo;
            ig_acc_normalize((int64_t*)&z_hi, (int64_t*)&z_lo);
            n_intra += br.dni;
            n_contigs += br_heads - (b_same ? 1 : 2);
            next_cid += NFRESH;
            if (br_changed) { /* no contig made by this move is longer than its window */
                max_L = max(max_L, b_nloc);
                max_SL = max(max_SL, rl(d.cand.m_loc, bc));
            }
            {
                const int sel = (bslot >= 12) ? bc : C - 1; /* the family of the winner re-ran get_bounds (CL:2125-2126) */
                vmask = (unsigned)rl((int)d.cand.flag_mask, sel);
            }
            if (lane == 0) {
                ig_move_result r;
                r.o = bests;
                r.dist = 0.0; /* step 2 */
                r.mean_len = (double)(n_frags_f / (float)n_contigs);
                r.op_sampled = bslot;
                r.id_f_sampled = b_B;
                r.n_contigs = n_contigs;
                r.n_candidates = C;
                r.n_slice = 0; /* step 2 */
                r.n_evals = 0;
                r.bytes_min = 68LL * b_nloc;
                r.error = err0;
                r.pad = 0;
                if (FUSED) {
                    sh->rec[w] = r;
                    sh->ch_c[w] = bc;
                    sh->ch_slot[w] = bslot;
                    sh->ch_k[w] = br.k;
                    sh->n_dirty[w] = br_changed;
                    sh->vmask[w] = (int)vmask_w;
                    sh->nzb_hi[w] = nzb_hi_w;
                    sh->nzb_lo[w] = nzb_lo_w;
                } else {
                    res[move0 + w] = r;
                }
            }
            if (br_changed) {
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    if (lane + 64 * j == n_dirty) dirty[j] = b_cA;
                    if (lane + 64 * j == n_dirty + 1) dirty[j] = b_cB;
                }
                n_dirty += 2;
            }
            committed = w + 1;
            if (CHAIN) {
                chain_changed = br_changed;
                if (br_changed) {
                    halt_after = true;
                    stop_overflow = 5;
                }
            }
            if (FUSED) { /* the record of slot w is in LDS: the commit wave may apply it */
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) sh->prog[0] = w + 1;
            }
            return true;
        };
        /* three moves in flight, each in its OWN registers (a rotation `cur = nxt` made the compiler wait for every
         * outstanding load at the top of each iteration: one memory round trip per decision) */
        auto clampw = [&](int w) { return w < W ? w : W - 1; };
        MoveData d0 = load_move(clampw(w_start)), d1 = load_move(clampw(w_start + 1)), d2 = load_move(clampw(w_start + 2));
        for (int w = w_start; w < W; w += 3) {
            if (!decide_one(w, d0) || (CHAIN && halt_after)) break;
            d0 = load_move(clampw(w + 3));
            if (w + 1 >= W || !decide_one(w + 1, d1) || (CHAIN && halt_after)) break;
            d1 = load_move(clampw(w + 4));
            if (w + 2 >= W || !decide_one(w + 2, d2) || (CHAIN && halt_after)) break;
            d2 = load_move(clampw(w + 5));
        }
        /* the window rule (MoveBuf.ring): which of the positions this launch did NOT commit hold a slot that is stale now -- a contig it
         * reads (its focal bin's, its candidates') was written by a move of this chain (the list in the registers: every entry of it is
         * younger than every slot of the window, which the host keeps clean of everything older) -- or whose lists did not fit a pool.
         * Lane p: position p.  The host re-scores those with the next launch and keeps the rest (ig_host_batch.inc: run_moves_window). */
        unsigned long long stale_mask = 0ull;
        if (mb.ring) {
            const bool mine = lane >= committed && lane < W;
            bool st = false;
            int maxC = all_C;
            for (int o = 32; o > 0; o >>= 1) maxC = max(maxC, __shfl_xor(maxC, o, 64));
            for (int cq = 0; cq < maxC; cq++) {
                int qa = -3, qb = -3;
                if (mine && cq < all_C) {
                    const CandPre& cp = cpre_w(mb, lane, cq);
                    qa = cp.ctgA;
                    qb = cp.ctgB;
                    st |= (cq == 0 && cp.overflow != 0);
                }
#pragma unroll
                for (int j = 0; j < ND; j++) {
                    const int nj = min(max(n_dirty - 64 * j, 0), 64);
                    for (int q = 0; q < nj; q++) {
                        const int id = __builtin_amdgcn_readlane(dirty[j], q);
                        st |= (id == qa) | (id == qb);
                    }
                }
            }
            stale_mask = __ballot(mine && st);
        }
#pragma unroll
        for (int j = 0; j < ND; j++)
            if (lane + 64 * j < n_dirty) dirty_buf[1 + lane + 64 * j] = dirty[j];
        if (lane == 0) {
            g->nz_hi = nz_hi;
            g->nz_lo = nz_lo;
            g->z_hi = z_hi;
            g->z_lo = z_lo;
            g->n_intra = n_intra;
            g->n_contigs = n_contigs;
            g->next_cid = next_cid;
            g->max_L = max_L;
            g->max_SL = max_SL;
            dirty_buf[0] = n_dirty;
            batch_out[0] = committed;
            batch_out[1] = pending;
            batch_out[2] = (committed == w_start && pending < 0) ? stop_overflow : 0; /* nothing done because the first slot did not fit a pool */
            batch_out[8] = max_L;
            batch_out[9] = max_SL;
            batch_out[10] = stop_overflow; /* what the batch stopped at, if a pool: the host makes more room for the next ones */
            batch_out[3] = n_cand;
            batch_out[4] = n_predicted;
            batch_out[5] = n_contigs;
            int need = 0; /* two-tier scoring: work items the exact kernel's grid has to cover (8 x the longest sub-list, had every slot fitted) */
            if (mb.work)
                for (int x = 0; x < 8; x++) need = max(need, 8 * (int)mb.work[8 + x]);
            batch_out[6] = need;
            batch_out[11] = chain_changed; /* (a chain's segment: its last committed move changed the genome) */
            batch_out[12] = (int)(unsigned)stale_mask;
            batch_out[13] = (int)(unsigned)(stale_mask >> 32);
            /* the host polls this copy (mapped, coherent host memory): it learns the outcome while k_commit_batch is still
             * running and has the next launches queued behind it when it ends */
            if (host_out) {
                host_out[0] = committed;
                host_out[1] = pending;
                host_out[2] = (committed == w_start && pending < 0) ? stop_overflow : 0;
                host_out[8] = max_L;
                host_out[9] = max_SL;
                host_out[10] = stop_overflow;
                host_out[3] = n_cand;
                host_out[4] = n_predicted;
                host_out[5] = n_contigs;
                host_out[6] = need;
                host_out[11] = chain_changed;
                host_out[12] = (int)(unsigned)stale_mask;
                host_out[13] = (int)(unsigned)(stale_mask >> 32);
                __threadfence_system();
                host_out[7] = seq;
            }
        }
        if (lane < 12) g->valid_insert[lane] = ((vmask >> lane) & 1u) ? 1 : -1;
        if (FUSED) {
            if (lane == 0) {
                sh->fin_max_L = max_L;
                sh->fin_max_SL = max_SL;
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) sh->prog[1] = committed + 1;
        }
    }
}
__global__ void __launch_bounds__(64)
    k_decide_batch(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out,
                   volatile int* host_out, int seq, int resumed_plain, int zcheck, double* host_scores)
{
    decide_body<false>(g, mb, res, move0, W, w_start, dirty_buf, batch_out, host_out, seq, resumed_plain, nullptr, zcheck,
                       ChainArgs{nullptr, nullptr, 0, 1.0}, host_scores);
}

__global__ void __launch_bounds__(64)
    k_decide_chain(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out, volatile int* host_out, int seq,
                   int resumed_plain, int zcheck, ChainArgs ca)
{
    decide_body<false, true>(g, mb, res, move0, W, w_start, dirty_buf, batch_out, host_out, seq, resumed_plain, nullptr, zcheck, ca);
}

/* step 2 of the batch commit: one workgroup applies the moves [w_start, batch_out[0]) k_decide_batch committed */
__global__ void __launch_bounds__(COMMIT_THREADS)
    k_commit_batch(State st, Tables tab, Tables tab_prev, Glob* g, MoveBuf mb, const int* __restrict__ ip, const int* __restrict__ in,
                   const int* __restrict__ orientable, const unsigned char* __restrict__ black, int* own_tag, int* own_idx,
                   int* prev_touched, ig_move_result* res, int move0, int W, int w_start, const int* batch_out, NuisHost* hn, int hn_seq)
{
    __shared__ long long sh_delta[IG_MAX_BATCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int tag_base = g->stamp_ctr; /* tags/stamps of this batch: tag_base + w */
    if (tid < IG_MAX_BATCH) sh_delta[tid] = 0;
    const int committed = batch_out[0];
    __syncthreads();
    if (committed == w_start) return;
    /* threads that apply one move together: the whole workgroup when there is a single move to apply */
    const int gsz = (committed - w_start >= 2) ? COMMIT_GROUP : COMMIT_THREADS;
    const int grp = tid / gsz, gtid = tid % gsz, ngrp = COMMIT_THREADS / gsz;
    /* ---------------------------------------------------------------- 2. apply */
    const int N = mb.sN, M = mb.sM; /* strides of the window arrays */
    auto winner_loc = [&](int w) -> const int* {
        const MoveCtl& mc = mb.ctl[PS(w)];
        return mb.loc + ((size_t)(CW(w, mc.ch_c) * NSLOT + mc.ch_slot) * NDYN) * N;
    };
    /* The committed moves touch pairwise disjoint contigs: in every step below a group of COMMIT_GROUP threads takes a move
     * (moves grp, grp + ngrp, ...), so that the chains of dependent loads of several moves (control block -> window ->
     * fragment) overlap instead of being walked one move after the other by the whole workgroup.
     * 2a. ownership marks of the fragments whose state changes */
    for (int w = w_start + grp; w < committed; w += ngrp) {
        const MoveCtl& mc = mb.ctl[PS(w)];
        if (!mc.n_dirty) continue;
        const int cw = CW(w, mc.ch_c);
        const int n_loc = mb.meta[cw].n_loc;
        const int* gid = mb.Lloc + (size_t)cw * N;
        for (int x = gtid; x < n_loc; x += gsz) {
            const int f = gid[x];
            own_tag[f] = tag_base + w;
            own_idx[f] = x;
        }
    }
    __syncthreads();
    /* 2b. genome distance: credit(f) depends on prev/next/ori of f and on the orientation of its INITIAL neighbours
     * (CL:665-716), so move w can change the credits of its window and of the window's initial neighbours only; each is
     * evaluated on the genome as of move w-1 and as of move w (moves < t applied, read through the marks) */
    for (int w = w_start + grp; w < committed; w += ngrp) {
        const MoveCtl& mc = mb.ctl[PS(w)];
        if (!mc.n_dirty) continue;
        const int cw = CW(w, mc.ch_c);
        const int n_loc = mb.meta[cw].n_loc;
        const int* gid = mb.Lloc + (size_t)cw * N;
        long long d = 0;
        const int* wl = winner_loc(w);
        /* has fragment y one of its three fields changed by move w?  (then its own q = 0 item evaluates its credit) */
        auto changed_member = [&](int y) -> bool {
            if (y < 0 || own_tag[y] != tag_base + w) return false;
            const int xi = own_idx[y];
            return wl[(size_t)5 * N + xi] != st.prev[y] || wl[(size_t)6 * N + xi] != st.next[y] || wl[(size_t)10 * N + xi] != st.ori[y];
        };
        for (int item = gtid; item < 3 * n_loc; item += gsz) {
            const int x0 = item / 3;
            const int f0 = gid[x0];
            /* credit(f) reads prev / next / ori of f and the orientation of its initial neighbours: only a fragment whose own
             * three fields change can change a credit (its own, or those of its initial neighbours) */
            if (wl[(size_t)5 * N + x0] == st.prev[f0] && wl[(size_t)6 * N + x0] == st.next[f0] && wl[(size_t)10 * N + x0] == st.ori[f0])
                continue;
            const int q = item % 3;
            const int f = (q == 0) ? f0 : ((q == 1) ? ip[f0] : in[f0]);
            if (f < 0 || black[f]) continue;
            /* A fragment can be reached by up to three items of this move: as itself (q = 0), as the initial predecessor of
             * in[f] (q = 1) and as the initial successor of ip[f] (q = 2); exactly one of them evaluates it.  The rule reads
             * nothing another group writes (several moves are in flight side by side): the initial links are mutually
             * inverse (checked at upload: ig_set_initial_genome, else the batch path is not used), so the q = 1 item that
             * reaches f comes from in[f] and the q = 2 item from ip[f]. */
            if (q > 0) {
                if (changed_member(f)) continue;
                if (q == 2 && (ip[f0] == f || changed_member(in[f]))) continue;
            }
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
    for (int w = w_start + grp; w < committed; w += ngrp) {
        const MoveCtl& mc = mb.ctl[PS(w)];
        const int cw = CW(w, mc.ch_c);
        const CandMeta& m = mb.meta[cw];
        const bool last = (w == committed - 1);
        if (last && gtid == 0) g->n_prev_touched = mc.n_dirty ? m.m_loc : 0;
        if (!mc.n_dirty) continue;
        const int* base = winner_loc(w);
        const int* gid = mb.Lloc + (size_t)cw * N;
        for (int x = gtid; x < m.n_loc; x += gsz) {
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
        for (int ls = gtid; ls < m.m_loc; ls += gsz) {
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
        const MoveCtl& mc = mb.ctl[PS(w)];
        const unsigned vmask = (unsigned)mc.pad;
        long long Sc = 0, ev = 0, by = 0;
        for (int c = 0; c < mc.C; c++) {
            const CandMeta& m = mb.meta[CW(w, c)];
            const CandPre& cp = cpre_at(mb, CW(w, c));
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
    if (hn) { /* the last committed move's record straight to the (mapped) host memory, then the flag the host spins on */
        __syncthreads();
        if (tid == 0) {
            hn->res = res[move0 + committed - 1];
            hn->nzb[0] = mb.ctl[PS(committed - 1)].nzb_hi;
            hn->nzb[1] = mb.ctl[PS(committed - 1)].nzb_lo;
            hn->max_L = g->max_L;
            hn->max_SL = g->max_SL;
            hn->changed = mb.ctl[PS(committed - 1)].n_dirty;
            __threadfence_system();
            hn->res_seq = hn_seq;
        }
    }
}

/* k_decide_commit: both steps in ONE launch.  The decide wave is a chain of 24 decisions (2 us each) during which nothing else ran,
 * the apply step (one workgroup) a chain of dependent loads of its own that could not start before the last decision.  Here eight
 * more waves work behind the decide wave: as a decision arrives (through LDS) they write its record and, for a move that changes
 * the genome, its ownership marks and the credits of the genome distance it can change -- k_commit_batch's steps 2a / 2b / 2d,
 * which read the state as of the batch's start through the marks -- so that what is left behind the last decision is the catch-up
 * of tab_prev and the winners' copy into the live state (2c).  Same words in memory as the two kernels leave (every batch-path
 * golden and the W = 1 comparisons run through it). */
#ifndef FUSED_CW
#define FUSED_CW 7 /* commit waves of k_decide_commit (a move is applied by all of them: its loops are k_commit_batch's, 448 threads wide; 3 / 5 / 7 waves: 44.7 / 45.4 / 45.5 k moves/s) */
#endif
/* a barrier of the commit waves alone (the decide wave never waits): arrivals counted in LDS, one word per use */
template <int CW = FUSED_CW>
__device__ __forceinline__ void cw_barrier(FusedLds* sh, int id)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) atomicAdd((int*)&sh->bar[id], 1);
    while (sh->bar[id] < CW) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
/* NDW: decide waves in front of the commit waves (1: decide_body's one wave; DPAR: the decide waves of decide_rounds), CW: commit waves */
template <int NDW = 1, int CW = FUSED_CW>
__device__ __forceinline__ void commit_waves(State st, Tables tab, Tables tab_prev, Glob* g, MoveBuf mb, const int* __restrict__ ip,
                                             const int* __restrict__ in, const int* __restrict__ orientable, const unsigned char* __restrict__ black,
                                             int* own_tag, int* own_idx, int* prev_touched, ig_move_result* res, int move0, int W, int w_start,
                                             NuisHost* hn, int hn_seq, FusedLds* sh)
{
    const int lane = threadIdx.x & 63, ctid = (int)threadIdx.x - 64 * NDW, cwv = ctid >> 6;
    constexpr int NCT = CW * 64;
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 2
    const long long ct0 = wall_clock64();
    long long ct_spin = 0, ct_fin = 0;
#endif
    for (int i = ctid; i < IG_MAX_BATCH; i += NCT) sh->n_chg[i] = 0; /* (up before the statistics' barrier below) */
    /* the statistics columns of every slot, ahead of the decisions (k_commit_batch 2d): 16 lanes per slot */
    for (int w = w_start + (ctid >> 4); w < W; w += NCT / 16) {
        const int c = lane & 15;
        const int C = mb.ctl[PS(w)].C;
        long long Sc = 0, ev = 0, by = 0;
        if (c < C) {
            const CandMeta& m = mb.meta[CW(w, c)];
            const CandPre& cp = cpre_at(mb, CW(w, c));
            const bool apart = (c == 0) && mb.ctl[PS(w)].superset0;
            Sc = cp.n_slice;
            if (!apart) {
                ev = cp.n_slice * (m.n_uniq + 1);
                by = 12 * cp.n_slice + 20LL * m.m_loc * m.n_uniq + 8LL * m.n_uniq;
            }
            if (c == 0) {
                sh->s0_slice[w] = cp.n_slice;
                sh->s0_mloc[w] = m.m_loc;
                /* (-1: candidate 0 was scored with its own list.  The per-move section below read MoveCtl.superset0 from memory: one
                 * round trip per move in the commit waves' serial part -- hidden while one wave decided a move in 2.3 us, 50 us of a
                 * chain behind decide_rounds, tools/par_trace.py) */
                sh->s0_base[w] = apart ? cp.base_cnt : -1;
            }
        }
        for (int o = 8; o > 0; o >>= 1) { /* over the 16 lanes of the slot */
            Sc += __shfl_xor(Sc, o, 64);
            ev += __shfl_xor(ev, o, 64);
            by += __shfl_xor(by, o, 64);
        }
        if (c == 0) {
            sh->st_sc[w] = Sc;
            sh->st_ev[w] = ev;
            sh->st_by[w] = by;
        }
    }
    /* the coordinate tables of a move that changed the genome (k_commit_batch 2c); tab_prev receives every move but the last committed one */
    auto write_tables = [&](int w, bool last, int gtid, int GT) {
        const int cw = CW(w, sh->ch_c[w]);
        const CandMeta& m = mb.meta[cw];
        const int k = sh->ch_k[w];
        const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * mb.sM;
        const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
        const int* subs = mb.subs + (size_t)cw * mb.sM;
        const int fresh = mb.ctl[PS(w)].fresh;
        for (int ls = gtid; ls < m.m_loc; ls += GT) {
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
    };
    int pend_tab = -1; /* NDW > 1: a move whose tables are still to be written -- as soon as a later decision says it is not the last one */
    const int tag_base = g->stamp_ctr;
    long long c2 = g->credit2; /* (used by the first commit wave's lane 0) */
    const double norm = 3.0 * (double)(g->N - g->n_black);
    const int N = mb.sN, M = mb.sM;
    int processed = w_start;
    long long hn_last_sc = 0, hn_last_ev = 0, hn_last_by = 0; /* (lane 0 of the first commit wave: the last applied move's statistics) */
    cw_barrier<CW>(sh, 3 * IG_MAX_BATCH); /* (the statistics are in LDS) */
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 2
    if (NDW > 1 && ctid == 0) atomicAdd(&g->dbg[2], (int)(wall_clock64() - ct0));
#endif
    auto winner_loc = [&](int w) -> const int* { return mb.loc + ((size_t)(CW(w, sh->ch_c[w]) * NSLOT + sh->ch_slot[w]) * NDYN) * N; };
    /* While the decisions come in: a move that changes the genome gets its ownership marks and its credits -- evaluated through the
     * marks on the state as of the batch's start, as k_commit_batch does (nothing is applied yet: a later move's marks, written
     * while an earlier one's credits are still being read, say "later" to them either way) -- and every move its record. */
    for (;;) {
        int upto, fin;
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 2
        const long long cs0 = wall_clock64();
#endif
        for (;;) {
            fin = sh->prog[1]; /* (first: a final count read before the last progress word misses nothing, the other way round could) */
            upto = sh->prog[0];
            if (upto > processed || fin) break;
            __builtin_amdgcn_s_sleep(4);
        }
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 2
        ct_spin += wall_clock64() - cs0;
        if (fin && !ct_fin) ct_fin = wall_clock64();
#endif
        __asm__ volatile("" ::: "memory");
        if (fin) upto = fin - 1;
        if (processed == w_start && upto > w_start) { /* a move will be committed: tab_prev catches up with the move applied last before
                                                       * this call (k_commit_batch 2c) -- now, while there is little to do, not behind the
                                                       * last decision; the tables and the list are not touched before the end */
            const int n_prev0 = g->n_prev_touched;
            for (int i = ctid; i < n_prev0; i += NCT) {
                const int s2 = prev_touched[i];
                tab_prev.dist[s2] = tab.dist[s2];
                tab_prev.stot[s2] = tab.stot[s2];
                tab_prev.cp[s2] = tab.cp[s2];
                tab_prev.len[s2] = tab.len[s2];
            }
            cw_barrier<CW>(sh, 3 * IG_MAX_BATCH + 1);
        }
        for (int w = processed; w < upto; w++) {
            if (NDW > 1 && pend_tab >= 0) { /* (behind decide_rounds the decisions are in long before the commit waves are through: what used
                                             * to wait for the last decision -- 18 us behind it, tools/par_trace.py -- runs as the moves go by) */
                write_tables(pend_tab, false, ctid, NCT);
                pend_tab = -1;
            }
            if (sh->n_dirty[w]) {
                if (NDW > 1) pend_tab = w;
                const int cw = CW(w, sh->ch_c[w]);
                const int n_loc = mb.meta[cw].n_loc;
                const int* gid = mb.Lloc + (size_t)cw * N;
                const int* wl = winner_loc(w);
                if (mb.ring & 4) { /* IG_WINDOW_CHECK=1: the slot's window (the fragments in their order) must be what the live genome says */
                    const CandMeta& mm = mb.meta[cw];
                    for (int x = ctid; x < n_loc; x += NCT) {
                        const int f = gid[x];
                        const bool inA = mm.same || x < mm.LA;
                        const int want_c = inA ? mm.ctgA : mm.ctgB, want_p = inA ? x : x - mm.LA;
                        if (f < 0 || f >= g->N || st.cid[f] != want_c || st.pos[f] != want_p) {
                            g->error = 12;
                            g->dbg[0] = w;
                            g->dbg[1] = x;
                            g->dbg[2] = f;
                            g->dbg[3] = (f >= 0 && f < g->N) ? st.cid[f] : -1;
                            g->dbg[4] = want_c;
                            g->dbg[5] = (f >= 0 && f < g->N) ? st.pos[f] : -1;
                            g->dbg[6] = want_p;
                            g->dbg[7] = n_loc;
                        }
                    }
                }
                /* ... and, on the way, which fragments change one of the three fields a credit reads (a handful: the cut points, a flipped
                 * block): the credits below are evaluated for them alone, all at once -- walking all 3 n_loc items took the chain of a dozen
                 * dependent loads once per pass of the threads over the window (round 6: behind decide_rounds the apply step is what a
                 * chain's serial part waits for, ~17 us per genome-changing move; tools/par_trace.py) */
                for (int x = ctid; x < n_loc; x += NCT) {
                    const int f = gid[x];
                    own_tag[f] = tag_base + w;
                    own_idx[f] = x;
                    if (wl[(size_t)5 * N + x] != st.prev[f] || wl[(size_t)6 * N + x] != st.next[f] || wl[(size_t)10 * N + x] != st.ori[f]) {
                        const int at = atomicAdd(&sh->n_chg[w], 1);
                        if (at < FUSED_CHG_CAP) sh->chg[at] = x;
                    }
                }
                cw_barrier<CW>(sh, 3 * w);
                auto changed_member = [&](int y) -> bool {
                    if (y < 0 || own_tag[y] != tag_base + w) return false;
                    const int xi = own_idx[y];
                    return wl[(size_t)5 * N + xi] != st.prev[y] || wl[(size_t)6 * N + xi] != st.next[y] || wl[(size_t)10 * N + xi] != st.ori[y];
                };
                long long d = 0;
                const int n_chg = sh->n_chg[w];
                const bool listed = n_chg <= FUSED_CHG_CAP; /* (more than the list holds: the walk over every fragment, as before) */
                for (int item = ctid; item < 3 * (listed ? n_chg : n_loc); item += NCT) {
                    const int x0 = listed ? sh->chg[item / 3] : item / 3;
                    const int f0 = gid[x0];
                    if (!listed && wl[(size_t)5 * N + x0] == st.prev[f0] && wl[(size_t)6 * N + x0] == st.next[f0] && wl[(size_t)10 * N + x0] == st.ori[f0])
                        continue;
                    const int q = item % 3;
                    const int f = (q == 0) ? f0 : ((q == 1) ? ip[f0] : in[f0]);
                    if (f < 0 || black[f]) continue;
                    if (q > 0) {
                        if (changed_member(f)) continue;
                        if (q == 2 && (ip[f0] == f || changed_member(in[f]))) continue;
                    }
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
                if (lane == 0 && d) atomicAdd((unsigned long long*)&sh->delta[w], (unsigned long long)d);
                cw_barrier<CW>(sh, 3 * w + 1); /* (the move's sum is complete, its credits are read) */
                /* the winner becomes the live genome: nothing reads these fragments' state any more -- a later move's credits take
                 * them from the winner's buffers, through the marks */
                for (int x = ctid; x < n_loc; x += NCT) {
                    const int f = gid[x];
                    st.pos[f] = wl[x];
                    st.spos[f] = wl[(size_t)N + x];
                    st.cid[f] = wl[(size_t)2 * N + x];
                    st.sbp[f] = wl[(size_t)3 * N + x];
                    st.circ[f] = wl[(size_t)4 * N + x];
                    st.prev[f] = wl[(size_t)5 * N + x];
                    st.next[f] = wl[(size_t)6 * N + x];
                    st.L[f] = wl[(size_t)7 * N + x];
                    st.SL[f] = wl[(size_t)8 * N + x];
                    st.LB[f] = wl[(size_t)9 * N + x];
                    st.ori[f] = wl[(size_t)10 * N + x];
                }
            }
            if (ctid == 0) {
                const unsigned vmask = (unsigned)sh->vmask[w];
                long long Sc = sh->st_sc[w], ev = sh->st_ev[w], by = sh->st_by[w];
                if (sh->s0_base[w] >= 0) { /* candidate 0 as the reference would have scored it: its list under the stale flags of the decision */
                    const long long nu = sh->s0_base[w] + __popc(vmask);
                    ev += sh->s0_slice[w] * (nu + 1);
                    by += 12 * sh->s0_slice[w] + 20LL * sh->s0_mloc[w] * nu + 8LL * nu;
                }
                c2 += sh->delta[w];
                ig_move_result r = sh->rec[w];
                r.n_slice = Sc;
                r.n_evals = ev;
                r.bytes_min += by;
                r.dist = (norm - 0.5 * (double)c2) / norm;
                res[move0 + w] = r;
                hn_last_sc = Sc;
                hn_last_ev = ev;
                hn_last_by = r.bytes_min;
            }
        }
        processed = max(processed, upto);
        if (fin) break;
    }
    const int committed = processed;
    if (committed == w_start) return; /* nothing was committed: as k_commit_batch, nothing is touched */
    /* The decisions are in: the coordinate tables of the moves that changed the genome -- disjoint contigs, two at a time --; tab_prev
     * receives every move but the last one (k_commit_batch 2c; its catch-up ran when the first decision came) */
    if (NDW > 1) {
        if (pend_tab >= 0) write_tables(pend_tab, pend_tab == committed - 1, ctid, NCT);
    } else {
        constexpr int GT = NCT / 2; /* two groups of four waves, a move each */
        const int grp = ctid / GT, gtid = ctid % GT;
        int nth = 0; /* the moves that change the genome, dealt out in turn */
        for (int w = w_start; w < committed; w++) {
            if (!sh->n_dirty[w]) continue;
            if ((nth++ & 1) != grp) continue;
            write_tables(w, w == committed - 1, gtid, GT);
        }
    }
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 2
    if (NDW > 1 && ctid == 0) {
        atomicAdd(&g->dbg[3], (int)ct_spin);
        atomicAdd(&g->dbg[4], (int)(wall_clock64() - ct_fin));
    }
#endif
    if (ctid == 0) {
        g->credit2 = c2;
        g->stamp_ctr = tag_base + W + 2;
        g->n_prev_touched = sh->n_dirty[committed - 1] ? mb.meta[CW(committed - 1, sh->ch_c[committed - 1])].m_loc : 0;
        if (hn) { /* the last committed move's record straight to the (mapped) host memory, then the flag the host spins on */
            ig_move_result r = sh->rec[committed - 1];
            r.n_slice = hn_last_sc;
            r.n_evals = hn_last_ev;
            r.bytes_min = hn_last_by;
            r.dist = (norm - 0.5 * (double)c2) / norm;
            hn->res = r;
            hn->nzb[0] = sh->nzb_hi[committed - 1];
            hn->nzb[1] = sh->nzb_lo[committed - 1];
            hn->max_L = sh->fin_max_L;
            hn->max_SL = sh->fin_max_SL;
            hn->changed = sh->n_dirty[committed - 1];
            __threadfence_system();
            hn->res_seq = hn_seq;
        }
    }
}
__global__ void __launch_bounds__(64 + FUSED_CW * 64)
    k_decide_commit(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out, volatile int* host_out,
                    int seq, int resumed_plain, State st, Tables tab, Tables tab_prev, const int* __restrict__ ip, const int* __restrict__ in,
                    const int* __restrict__ orientable, const unsigned char* __restrict__ black, int* own_tag, int* own_idx, int* prev_touched,
                    NuisHost* hn, int hn_seq, int zcheck, double* host_scores)
{
    __shared__ FusedLds sh;
    for (int i = threadIdx.x; i < 3 * IG_MAX_BATCH + 2; i += blockDim.x) sh.bar[i] = 0;
    for (int i = threadIdx.x; i < IG_MAX_BATCH; i += blockDim.x) sh.delta[i] = 0;
    if (threadIdx.x == 0) {
        sh.prog[0] = w_start;
        sh.prog[1] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64)
        decide_body<true>(g, mb, res, move0, W, w_start, dirty_buf, batch_out, host_out, seq, resumed_plain, &sh, zcheck, ChainArgs{nullptr, nullptr, 0, 1.0},
                          host_scores);
    else commit_waves(st, tab, tab_prev, g, mb, ip, in, orientable, black, own_tag, own_idx, prev_touched, res, move0, W, w_start, hn, hn_seq, &sh);
}


/* ---- round 6: the decisions of a launch chain in ROUNDS (k_decide_commit was 105 us per chain of ~35 decisions,
 * one wave at ~3 us per decision next to an idle machine).
 *
 * decide_body walks the moves one after the other because each decision reads what its predecessor left: the live scalars (maintained
 * sums, pair count, contig count), the list of contigs written so far, the stale insert flags (quirk Q4).  But 85 % of the moves leave
 * all of that alone but for the flags -- their winner is a column whose genome IS the current genome -- and the flags a move leaves are
 * one of its candidates' precomputed masks, nearly always the last candidate's (CL:2125-2126: the family that re-ran get_bounds; a
 * block-insert winner's own otherwise).  So DPAR waves decide the next DPAR moves SIDE BY SIDE, each under the shared state as it
 * stands and under the flags its predecessor is expected to leave; then every wave reads the DPAR outcomes in order and finds the
 * prefix that is what decide_body would have decided: move l holds if every move before it held, left the state alone, and left the
 * flags move l assumed.  The prefix ends behind the first move that changes the state (its owner applies decide_body's update, word
 * for word), in front of the first move that assumed other flags (decided again next round, now under the right ones) and in front
 * of whatever stops a chain (a conflict, a pool, a score of exactly 0.0, a pending windowed winner).  The first move of a round
 * always holds -- its inputs are exact -- so a round decides 1 .. DPAR moves; at 15 % state-changing moves ~3 of 4.
 * One move's decision is decide_one's arithmetic on the same records in the same order of operations: same bits (the fused / unfused
 * comparison of tests/test_hip_fused_commit.py runs this against decide_body, the oracle suites against the reference's argmax).
 * Moves with more than 5 candidates, launches that publish their scores (ig_step_draw) and the nuisance chains keep decide_body. */
#ifndef DPAR
#define DPAR 4 /* decide waves */
#endif
#ifndef DPAR_CW
#define DPAR_CW 4 /* commit waves behind them (8 waves: two per SIMD, 256 registers each -- a decide wave holds two moves' records) */
#endif
struct ParLds {
    long long nz_hi, nz_lo, z_hi, z_lo, n_intra;
    int n_contigs, next_cid, max_L, max_SL;
    unsigned vmask;
    int n_dirty;
    int committed, pending, stop_overflow, n_cand, n_predicted;
    volatile int bar; /* arrivals at the decide waves' barriers (monotonic) */
    int dirty[2 * IG_MAX_BATCH + 4];
    unsigned vm_in[IG_MAX_BATCH], vm_out[IG_MAX_BATCH], vm_def[IG_MAX_BATCH];
    unsigned sens[IG_MAX_BATCH]; /* the flags a move's decision depends on (decide_rounds) */
    int kind[IG_MAX_BATCH]; /* 0 decided, 1 stops in front (conflict / pool / error: whatever the flags), 2 pending windowed winner, 3 a score of 0.0 */
    int changes[IG_MAX_BATCH], scode[IG_MAX_BATCH];
};
/* LDS operations of a wave complete in order: when a wave's arrival is visible, so is everything it wrote to LDS before it (no fence:
 * a release would wait for the wave's prefetch of its next move, decide_body's note) */
__device__ __forceinline__ void dw_barrier(ParLds* ps, int& phase)
{
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase += DPAR;
    if ((threadIdx.x & 63) == 0) atomicAdd((int*)&ps->bar, 1);
    while (ps->bar < phase) __builtin_amdgcn_s_sleep(1);
    __asm__ volatile("" ::: "memory");
}
__device__ __forceinline__ void decide_rounds(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out,
                                              volatile int* host_out, int seq, int resumed_plain, FusedLds* sh, ParLds* ps, int zcheck)
{
    __builtin_amdgcn_s_setprio(3);
    constexpr int ND = (IG_MAX_BATCH * 2 + 2 + 63) / 64;
    const int lane = threadIdx.x & 63, dwv = threadIdx.x >> 6; /* decide wave 0 .. DPAR - 1 */
    int phase = 0;
    const int err0 = g->error;
    const ig_params p = g->par[0];
    const double log_e = IG_LOG_E_F;
    const double n_tot_pxl = g->n_tot_pxl;
    const float n_frags_f = (float)g->N;
    if (dwv == 0) { /* the shared state: decide_body's registers */
        int n_dirty = 0;
        if (w_start > 0) {
            n_dirty = dirty_buf[0];
            for (int j = 0; j < ND; j++)
                if (lane + 64 * j < n_dirty) ps->dirty[lane + 64 * j] = dirty_buf[1 + lane + 64 * j];
            if (!(resumed_plain & 1)) {
                const MoveCtl pm = mb.ctl[PS(w_start - 1)];
                const CandMeta& m = mb.meta[CW(w_start - 1, pm.ch_c)];
                if (lane == 0) {
                    ps->dirty[n_dirty] = m.ctgA;
                    ps->dirty[n_dirty + 1] = m.ctgB;
                }
                n_dirty += 2;
            }
        }
        const int v = (lane < 12) ? g->valid_insert[lane] : -1;
        const unsigned vmask0 = (unsigned)__ballot(lane < 12 && v != -1);
        if (lane == 0) {
            ps->nz_hi = g->nz_hi;
            ps->nz_lo = g->nz_lo;
            ps->z_hi = g->z_hi;
            ps->z_lo = g->z_lo;
            ps->n_intra = g->n_intra;
            ps->n_contigs = g->n_contigs;
            ps->next_cid = g->next_cid;
            ps->max_L = g->max_L;
            ps->max_SL = g->max_SL;
            ps->vmask = vmask0;
            ps->n_dirty = n_dirty;
            ps->committed = w_start;
            ps->pending = -1;
            ps->stop_overflow = 0;
            ps->n_cand = 0;
            ps->n_predicted = 0;
        }
    }
    struct MoveData {
        int C, superset0;
        CandPre cand;
        SlotPre rec[2];
        double e_ext_d[2];
        int e_r[2], e_base[2];
    };
    const int all_C = (lane < W) ? mb.ctl[PS(lane)].C : 0;
    const int all_sup = (lane < W) ? mb.ctl[PS(lane)].superset0 : 0;
    auto load_move = [&](int w) { /* (decide_body's: nothing here touches a loaded value) */
        MoveData d;
        d.C = __builtin_amdgcn_readlane(all_C, w);
        d.superset0 = __builtin_amdgcn_readlane(all_sup, w);
        d.cand = cpre_w(mb, w, lane < d.C ? lane : 0);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int i = max(min(lane + 64 * j, d.C * IG_N_TMP_STRUCT - 1), 0);
            d.rec[j] = pre_w(mb, w, i);
            const CandPre& cp = cpre_w(mb, w, i / IG_N_TMP_STRUCT);
            d.e_ext_d[j] = cp.ext_d;
            d.e_r[j] = cp.r;
            d.e_base[j] = cp.base_cnt;
        }
        return d;
    };
    auto rl = [](int v, int src) { return __builtin_amdgcn_readlane(v, src); };
    auto rl64 = [](long long v, int src) {
        const int lo = __builtin_amdgcn_readlane((int)(unsigned)v, src), hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
        return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    };
    auto rld = [&](double v, int src) { return __longlong_as_double(rl64(__double_as_longlong(v), src)); };
    auto clampw = [&](int w) { return w < W ? w : W - 1; };
    bool fin = false;
    /* the rounds a wave spends on ONE move (its own registers: d), until the move is committed or the chain is over */
    auto rounds_for = [&](const int w, const MoveData& d, const MoveData& dn) {
        const bool have = w < W;
        const int C = d.C;
        unsigned my_out = 0; /* the flags this move left when it was last decided */
        for (;;) {
            /* (behind the barrier that closed the previous round, or the one in front of the first: the state of this round stands) */
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 3
            const long long q0 = wall_clock64();
#endif
            const int base = ps->committed;
            const bool active = have && w >= base && w < base + DPAR;
            /* what the decision is a function of */
            const long long nz_hi = ps->nz_hi, nz_lo = ps->nz_lo, z_hi = ps->z_hi, z_lo = ps->z_lo, n_intra = ps->n_intra;
            const int n_contigs = ps->n_contigs, n_dirty = ps->n_dirty;
            const unsigned vm0 = ps->vmask; /* (read here: the owner of the prefix's last move overwrites it while the others still scan) */
            unsigned vm = vm0;
            if (active && w > base) vm = ps->kind[w - 1] == 0 && ps->vm_in[w - 1] != 0xffffffffu ? ps->vm_out[w - 1] : ps->vm_def[w - 1];
            /* (vm_in = ~0: that move has not been decided yet -- its default) */
            /* this move's outcome, uniform */
            int kind = 0, scode = 0, changes = 0;
            unsigned vm_out = 0, sens = 0;
            int bc = 0, bslot = 0, br_k = 0, br_changed = 0, br_heads = 0, b_same = 0, b_nloc = 0, b_mloc = 0, b_cA = 0, b_cB = 0, windowed = 0;
            long long d_nz_hi = 0, d_nz_lo = 0, br_dz_hi = 0, br_dz_lo = 0, br_dni = 0;
            bool have_delta = false;
            double bests = 0.0;
            long long Sc = 0, ev = 0, by = 0;
            if (active) {
                int dirty[ND];
#pragma unroll
                for (int j = 0; j < ND; j++) dirty[j] = (lane + 64 * j < n_dirty) ? ps->dirty[lane + 64 * j] : -2;
                bool hitd = false;
                for (int cq = 0; cq < C; cq++) {
                    const int qa = rl(d.cand.ctgA, cq), qb = rl(d.cand.ctgB, cq);
#pragma unroll
                    for (int j = 0; j < ND; j++) hitd |= (dirty[j] == qa) | (dirty[j] == qb);
                }
                if (err0 || rl(d.cand.overflow, 0) || __any(hitd)) {
                    kind = 1;
                    scode = err0 ? 0 : rl(d.cand.overflow, 0);
                } else {
                    const double cur_nz = ig_acc_to_double(nz_hi, nz_lo);
                    const int n = C * IG_N_TMP_STRUCT;
                    double sc[2];
                    double bestv = -IG_INF;
                    int best = 0x7fffffff;
                    bool zhit = false;
                    auto score_of = [&](int i, const SlotPre& r, double ext, int cr, int cbase) -> double { /* decide_one's, with vm for the live flags */
                        const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
                        const bool sup = (c == 0) && d.superset0 && (slot >= 12);
                        const bool scored = (i < n) && (r.k > 0) && !(sup && !((vm >> (slot - 12)) & 1u));
                        if (!scored) return 0.0;
                        const int pos = sup ? cbase + __popc(vm & ((1u << (slot - 12)) - 1u)) : r.k - 1;
                        const double nzd = (cr > 0 && pos >= cr) ? r.nz_cut_d : r.nz_d; /* quirk Q5 */
                        const double val_inter = -1.0 * log_e * (n_tot_pxl - (double)(n_intra + r.dni)) * p.v_inter;
                        const double val_intra = ig_acc_to_double(z_hi + r.dz_hi, z_lo + r.dz_lo) * log_e;
                        const double z = val_intra + val_inter;
                        const double v = nzd + z + cur_nz - ext;
                        zhit |= (v == 0.0);
                        return v;
                    };
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const int i = lane + 64 * j;
                        const double v = score_of(i, d.rec[j], d.e_ext_d[j], d.e_r[j], d.e_base[j]);
                        sc[j] = v;
                        const double ok = (v == 0.0) ? -IG_INF : v;
                        if (i < n && ok > bestv) {
                            bestv = ok;
                            best = i;
                        }
                    }
                    /* which of the flags this decision is a function of: candidate 0's block-insert columns that hold a score (scored or
                     * not by their flag); and, where the Q5 tail applies to candidate 0 (r > 0), every flag below the highest of them --
                     * a column's list position counts the flags below it.  Under flags that differ elsewhere the decision is the same
                     * decision (the scan below holds a move to exactly that) */
                    {
                        const bool supk = d.superset0 && lane >= 12 && lane < IG_N_TMP_STRUCT && lane < n && d.rec[0].k > 0;
                        sens = ((unsigned)(__ballot(supk) >> 12)) & 0xfffu;
                        const int cr0 = rl(d.e_r[0], 0);
                        if (cr0 > 0 && sens) sens |= (1u << (31 - __clz((int)sens))) - 1u;
                    }
                    const int inj = zcheck >> 8;
                    if ((zcheck & 1) && (__any(zhit) || (inj > 0 && (move0 + w) % inj == inj - 1))) {
                        kind = 3;
                        scode = 3;
                    } else {
                        wave_argmax_step<0x111, 0xf>(bestv, best);
                        wave_argmax_step<0x112, 0xf>(bestv, best);
                        wave_argmax_step<0x114, 0xf>(bestv, best);
                        wave_argmax_step<0x118, 0xf>(bestv, best);
                        wave_argmax_step<0x142, 0xa>(bestv, best);
                        wave_argmax_step<0x143, 0xc>(bestv, best);
                        best = rl(best, 63);
                        if (best >= n) best = 0;
                        bc = best / IG_N_TMP_STRUCT;
                        bslot = best % IG_N_TMP_STRUCT;
                        const int owner = best & 63, bj = best >> 6;
                        const SlotPre mine = (bj == 0) ? d.rec[0] : d.rec[1];
                        const long long br_nz_hi = rl64(mine.nz_hi, owner), br_nz_lo = rl64(mine.nz_lo, owner);
                        br_dz_hi = rl64(mine.dz_hi, owner);
                        br_dz_lo = rl64(mine.dz_lo, owner);
                        br_dni = rl64(mine.dni, owner);
                        br_k = rl(mine.k, owner);
                        const unsigned br_info = (unsigned)rl((int)mine.info, owner);
                        bests = rld((bj == 0) ? sc[0] : sc[1], owner);
                        if (n == 0) br_k = 0;
                        const int b_sw = rl(d.cand.same_windowed, bc), b_B = rl(d.cand.B, bc);
                        b_nloc = rl(d.cand.n_loc, bc);
                        b_mloc = rl(d.cand.m_loc, bc);
                        windowed = (b_sw >> 1) & 1;
                        b_same = b_sw & 1;
                        b_cA = rl(d.cand.ctgA, bc);
                        b_cB = rl(d.cand.ctgB, bc);
                        const long long b_ext_hi = rl64(d.cand.ext_hi, bc), b_ext_lo = rl64(d.cand.ext_lo, bc);
                        br_changed = (int)(br_info & 1u);
                        br_heads = (int)(br_info >> 1);
                        const int pred = rl(d.cand.pred, 0);
                        have_delta = windowed && br_changed && (best == pred);
                        const bool is_pending = windowed && br_changed && !have_delta;
                        if (is_pending) {
                            kind = 2;
                            sens = 0xfffu; /* (its record's statistics count the flags: decided under the very flags, or again) */
                            if (lane < C) {
                                int nu = d.cand.n_uniq;
                                if (lane == 0 && d.superset0) nu = d.cand.base_cnt + __popc(vm);
                                Sc = d.cand.n_slice;
                                ev = Sc * (nu + 1);
                                by = 12 * Sc + 20LL * d.cand.m_loc * nu + 8LL * nu;
                            }
                            Sc = rl64(wave_sum_ll(Sc), 0);
                            ev = rl64(wave_sum_ll(ev), 0);
                            by = rl64(wave_sum_ll(by), 0);
                        }
                        if (have_delta) {
                            d_nz_hi = rl64(d.cand.pd_hi, 0);
                            d_nz_lo = rl64(d.cand.pd_lo, 0);
                        } else {
                            d_nz_hi = br_nz_hi - b_ext_hi;
                            d_nz_lo = br_nz_lo - b_ext_lo;
                        }
                        {
                            const int sel = (bslot >= 12) ? bc : C - 1; /* the family of the winner re-ran get_bounds (CL:2125-2126) */
                            vm_out = (unsigned)rl((int)d.cand.flag_mask, sel);
                        }
                        /* does committing this move change anything a later decision reads besides the flags? */
                        const int dheads = br_heads - (b_same ? 1 : 2);
                        changes = (br_changed || have_delta || d_nz_hi != 0 || d_nz_lo != 0 || br_dz_hi != 0 || br_dz_lo != 0 || br_dni != 0 || dheads != 0) ? 1 : 0;
                        if (kind == 0 && lane == 0) { /* the record, as decide_one leaves it for the commit waves: read by them once the move is committed */
                            ig_move_result r;
                            r.o = bests;
                            r.dist = 0.0;
                            r.mean_len = (double)(n_frags_f / (float)(n_contigs + dheads));
                            r.op_sampled = bslot;
                            r.id_f_sampled = b_B;
                            r.n_contigs = n_contigs + dheads;
                            r.n_candidates = C;
                            r.n_slice = 0;
                            r.n_evals = 0;
                            r.bytes_min = 68LL * b_nloc;
                            r.error = err0;
                            r.pad = 0;
                            sh->rec[w] = r;
                            sh->ch_c[w] = bc;
                            sh->ch_slot[w] = bslot;
                            sh->ch_k[w] = br_k;
                            sh->n_dirty[w] = br_changed;
                            sh->vmask[w] = (int)vm;
                            sh->nzb_hi[w] = nz_hi;
                            sh->nzb_lo[w] = nz_lo;
                        }
                    }
                }
                if (lane == 0) {
                    ps->kind[w] = kind;
                    ps->scode[w] = scode;
                    ps->changes[w] = changes;
                    ps->vm_in[w] = vm;
                    ps->vm_out[w] = vm_out;
                    ps->sens[w] = sens;
                }
                my_out = vm_out;
            }
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 3
            const long long q1 = wall_clock64();
#endif
            dw_barrier(ps, phase); /* the outcomes of this round are up */
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 3
            const long long q2 = wall_clock64();
#endif
            /* the prefix decide_body would have decided: the same scan in every wave */
            const int cnt = min(DPAR, W - base);
            int np = 0, stop_at = -1, f_code = 0, f_pend = -1;
            bool f = false;
            unsigned vm_act = vm; /* the flags this wave's move is decided under when it holds: what its predecessor left */
            {
                unsigned vm_prev = vm0;
                for (int l = 0; l < cnt; l++) {
                    const int wl = base + l, k = ps->kind[wl];
                    if (k == 1) { /* (a conflict, a pool, an error: whatever flags it assumed) */
                        f = true;
                        stop_at = wl;
                        f_code = ps->scode[wl];
                        break;
                    }
                    if ((ps->vm_in[wl] ^ vm_prev) & ps->sens[wl]) break; /* decided under flags that differ where it looks: again, next round */
                    if (wl == w) vm_act = vm_prev;
                    if (k == 3) {
                        f = true;
                        stop_at = wl;
                        f_code = 3;
                        break;
                    }
                    if (k == 2) {
                        f = true;
                        stop_at = wl;
                        f_pend = wl;
                        break;
                    }
                    np = l + 1;
                    vm_prev = ps->vm_out[wl];
                    if (ps->changes[wl]) break;
                }
                if (!f && base + np >= W) f = true;
            }
            const bool mine_committed = active && w < base + np;
            if (mine_committed) { /* this wave's NEXT move enters the window: the flags it leaves by default -- unless a block insert of another
                                   * candidate wins, its last candidate's (CL:2125-2126) -- for its successor's first decision (its records were
                                   * requested a move ago; up before the barrier that ends this round) */
                const unsigned vdef = (unsigned)rl((int)dn.cand.flag_mask, max(dn.C - 1, 0));
                if (w + DPAR < W && lane == 0) ps->vm_def[w + DPAR] = vdef;
            }
            if ((mine_committed || (active && w == f_pend)) && lane == 0) { /* decide_one's control block (a pending move's too: the one-move tail reads it) */
                MoveCtl& o = mb.ctl[PS(w)];
                o.ch_c = bc;
                o.ch_slot = bslot;
                o.ch_k = br_k;
                o.ch_windowed = windowed;
                o.ch_score = bests;
                o.n_slice_tot = Sc;
                o.n_eval_tot = ev;
                o.bytes_min = by;
                o.d_hi = 0;
                o.d_lo = 0;
                o.nzb_hi = nz_hi;
                o.nzb_lo = nz_lo;
                o.n_dirty = br_changed;
                o.pad = (int)vm_act;
                sh->vmask[w] = (int)vm_act; /* (the statistics columns count them: k_commit_batch 2d) */
                if (br_k <= 0) g->error = 3;
                atomicAdd(&ps->n_cand, C);
                if (mine_committed && have_delta) atomicAdd(&ps->n_predicted, 1);
            }
            if (mine_committed && w == base + np - 1 && lane == 0) { /* the last move of the prefix: the state behind it (decide_one's commit, the same operations) */
                if (changes) {
                    long long a = nz_hi + d_nz_hi, b = nz_lo + d_nz_lo;
                    ig_acc_normalize((int64_t*)&a, (int64_t*)&b);
                    ps->nz_hi = a;
                    ps->nz_lo = b;
                    a = z_hi + br_dz_hi;
                    b = z_lo + br_dz_lo;
                    ig_acc_normalize((int64_t*)&a, (int64_t*)&b);
                    ps->z_hi = a;
                    ps->z_lo = b;
                    ps->n_intra = n_intra + br_dni;
                    ps->n_contigs = n_contigs + (br_heads - (b_same ? 1 : 2));
                    if (br_changed) {
                        ps->max_L = max(ps->max_L, b_nloc);
                        ps->max_SL = max(ps->max_SL, b_mloc);
                        ps->dirty[n_dirty] = b_cA;
                        ps->dirty[n_dirty + 1] = b_cB;
                        ps->n_dirty = n_dirty + 2;
                    }
                }
                ps->next_cid += NFRESH * np;
                ps->vmask = my_out;
                ps->committed = base + np;
            }
            if (f && active && w == stop_at && lane == 0) {
                ps->stop_overflow = f_code;
                ps->pending = f_pend;
            }
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 3
            const long long q3 = wall_clock64();
#endif
            dw_barrier(ps, phase); /* the state behind this round is up (the last one's: the epilogue reads it) */
#if defined(IG_PAR_TRACE) && IG_PAR_TRACE == 3
            if (dwv == 0 && lane == 0) {
                atomicAdd(&g->dbg[2], (int)(q1 - q0));
                atomicAdd(&g->dbg[3], (int)(q2 - q1));
                atomicAdd(&g->dbg[4], (int)(q3 - q2));
                atomicAdd(&g->dbg[6], (int)(wall_clock64() - q3));
            }
#endif
            if (dwv == 0 && lane == 0) { /* the prefix goes to the commit waves (their records were in LDS before the barrier) */
                if (np > 0) sh->prog[0] = base + np;
                /* ... and, with the last one, the final count: the epilogue's scan (13 us) runs next to their work, and the last
                 * genome-changing move's tables -- which wait to hear whether it is the last committed one -- do not wait for it */
                if (f) {
                    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    sh->prog[1] = ps->committed + 1;
                }
            }
            if (f) {
                fin = true;
                return;
            }
            if (mine_committed) return; /* on to this wave's next move */
        }
    };
    /* vm_in = ~0 marks "not decided yet" for the moves ahead of the window */
    for (int i = threadIdx.x; i < IG_MAX_BATCH; i += DPAR * 64) {
        ps->vm_in[i] = 0xffffffffu;
        ps->kind[i] = 0;
        ps->changes[i] = 0;
    }
    /* a wave's moves: w_start + dwv, + DPAR, ...; two of them in registers (the next one is loaded while this one is decided) */
    int w = w_start + dwv;
#ifdef IG_PAR_TRACE
    const long long tr0 = wall_clock64();
    int tr_rounds = 0;
#endif
    MoveData d0 = load_move(clampw(w)), d1 = load_move(clampw(w + DPAR));
    {
        const unsigned vdef = (unsigned)rl((int)d0.cand.flag_mask, max(d0.C - 1, 0));
        if (w < W && lane == 0) ps->vm_def[w] = vdef;
    }
    dw_barrier(ps, phase); /* the shared state, the marks above and the first moves' default flags are up */
#ifdef IG_PAR_TRACE
    const long long tr1 = wall_clock64();
#endif
    for (;;) {
        rounds_for(w, d0, d1);
        if (fin) break;
        w += DPAR;
        d0 = load_move(clampw(w + DPAR));
        rounds_for(w, d1, d0);
        if (fin) break;
        w += DPAR;
        d1 = load_move(clampw(w + DPAR));
    }
    if (dwv != 0) return;
#ifdef IG_PAR_TRACE
    const long long tr2 = wall_clock64();
#endif
    /* ---- the epilogue of decide_body, from the shared state */
    const int committed = ps->committed, n_dirty = ps->n_dirty;
    unsigned long long stale_mask = 0ull;
    if (mb.ring) { /* the window rule: which of the positions this launch did not commit hold a slot that is stale now (decide_body) */
        const bool mine = lane >= committed && lane < W;
        bool st = false;
        int maxC = all_C;
        for (int o = 32; o > 0; o >>= 1) maxC = max(maxC, __shfl_xor(maxC, o, 64));
        for (int cq = 0; cq < maxC; cq++) {
            int qa = -3, qb = -3;
            if (mine && cq < all_C) {
                const CandPre& cp = cpre_w(mb, lane, cq);
                qa = cp.ctgA;
                qb = cp.ctgB;
                st |= (cq == 0 && cp.overflow != 0);
            }
            for (int q = 0; q < n_dirty; q++) {
                const int id = ps->dirty[q];
                st |= (id == qa) | (id == qb);
            }
        }
        stale_mask = __ballot(mine && st);
    }
    for (int j = 0; j < ND; j++)
        if (lane + 64 * j < n_dirty) dirty_buf[1 + lane + 64 * j] = ps->dirty[lane + 64 * j];
    const unsigned vmask = ps->vmask;
    if (lane == 0) {
        const int pending = ps->pending, stop_overflow = ps->stop_overflow, max_L = ps->max_L, max_SL = ps->max_SL, n_contigs = ps->n_contigs;
        g->nz_hi = ps->nz_hi;
        g->nz_lo = ps->nz_lo;
        g->z_hi = ps->z_hi;
        g->z_lo = ps->z_lo;
        g->n_intra = ps->n_intra;
        g->n_contigs = n_contigs;
        g->next_cid = ps->next_cid;
        g->max_L = max_L;
        g->max_SL = max_SL;
        dirty_buf[0] = n_dirty;
        int need = 0;
        if (mb.work)
            for (int x = 0; x < 8; x++) need = max(need, 8 * (int)mb.work[8 + x]);
        const int first_none = (committed == w_start && pending < 0) ? stop_overflow : 0;
        batch_out[0] = committed;
        batch_out[1] = pending;
        batch_out[2] = first_none;
        batch_out[8] = max_L;
        batch_out[9] = max_SL;
        batch_out[10] = stop_overflow;
        batch_out[3] = ps->n_cand;
        batch_out[4] = ps->n_predicted;
        batch_out[5] = n_contigs;
        batch_out[6] = need;
        batch_out[11] = 0;
        batch_out[12] = (int)(unsigned)stale_mask;
        batch_out[13] = (int)(unsigned)(stale_mask >> 32);
        if (host_out) {
            host_out[0] = committed;
            host_out[1] = pending;
            host_out[2] = first_none;
            host_out[8] = max_L;
            host_out[9] = max_SL;
            host_out[10] = stop_overflow;
            host_out[3] = ps->n_cand;
            host_out[4] = ps->n_predicted;
            host_out[5] = n_contigs;
            host_out[6] = need;
            host_out[11] = 0;
            host_out[12] = (int)(unsigned)stale_mask;
            host_out[13] = (int)(unsigned)(stale_mask >> 32);
            __threadfence_system();
            host_out[7] = seq;
        }
        sh->fin_max_L = max_L;
        sh->fin_max_SL = max_SL;
    }
    if (lane < 12) g->valid_insert[lane] = ((vmask >> lane) & 1u) ? 1 : -1;
#ifdef IG_PAR_TRACE
    if (lane == 0) { /* (tuning builds: 10 ns ticks, summed over the launches -- ig_debug_dbg) */
        atomicAdd(&g->dbg[0], 1);
        atomicAdd(&g->dbg[1], ps->bar / (2 * DPAR));
#if IG_PAR_TRACE == 1
        atomicAdd(&g->dbg[2], (int)(tr1 - tr0));
        atomicAdd(&g->dbg[3], (int)(tr2 - tr1));
        atomicAdd(&g->dbg[4], (int)(wall_clock64() - tr2));
#endif
        atomicAdd(&g->dbg[5], committed - w_start);
    }
#endif
}
__global__ void __launch_bounds__(DPAR * 64 + DPAR_CW * 64)
    k_decide_commit_par(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out, volatile int* host_out,
                        int seq, int resumed_plain, State st, Tables tab, Tables tab_prev, const int* __restrict__ ip, const int* __restrict__ in,
                        const int* __restrict__ orientable, const unsigned char* __restrict__ black, int* own_tag, int* own_idx, int* prev_touched,
                        int zcheck)
{
    __shared__ FusedLds sh;
    __shared__ ParLds ps;
    for (int i = threadIdx.x; i < 3 * IG_MAX_BATCH + 2; i += blockDim.x) sh.bar[i] = 0;
    for (int i = threadIdx.x; i < IG_MAX_BATCH; i += blockDim.x) sh.delta[i] = 0;
    if (threadIdx.x == 0) {
        sh.prog[0] = w_start;
        sh.prog[1] = 0;
        ps.bar = 0;
    }
    __syncthreads();
#ifdef IG_PAR_TRACE
    const long long tk0 = wall_clock64();
#endif
    if (threadIdx.x < DPAR * 64) decide_rounds(g, mb, res, move0, W, w_start, dirty_buf, batch_out, host_out, seq, resumed_plain, &sh, &ps, zcheck);
    else commit_waves<DPAR, DPAR_CW>(st, tab, tab_prev, g, mb, ip, in, orientable, black, own_tag, own_idx, prev_touched, res, move0, W, w_start, nullptr, 0, &sh);
#ifdef IG_PAR_TRACE
#if IG_PAR_TRACE != 3
    if (threadIdx.x == DPAR * 64) atomicAdd(&g->dbg[6], (int)(wall_clock64() - tk0)); /* the commit waves' end */
#endif
    if (threadIdx.x == 0) atomicAdd(&g->dbg[7], (int)(wall_clock64() - tk0));        /* the first decide wave's */
#endif
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

/* the same for a segment of a chain of (move, nuisance step) pairs (k_decide_chain's decisions, the apply step behind them) */
__global__ void __launch_bounds__(64 + FUSED_CW * 64)
    k_chain_decide_commit(Glob* g, MoveBuf mb, ig_move_result* res, int move0, int W, int w_start, int* dirty_buf, int* batch_out, volatile int* host_out,
                          int seq, int resumed_plain, State st, Tables tab, Tables tab_prev, const int* __restrict__ ip, const int* __restrict__ in,
                          const int* __restrict__ orientable, const unsigned char* __restrict__ black, int* own_tag, int* own_idx, int* prev_touched,
                          int zcheck, ChainArgs ca)
{
    __shared__ FusedLds sh;
    for (int i = threadIdx.x; i < 3 * IG_MAX_BATCH + 2; i += blockDim.x) sh.bar[i] = 0;
    for (int i = threadIdx.x; i < IG_MAX_BATCH; i += blockDim.x) sh.delta[i] = 0;
    if (threadIdx.x == 0) {
        sh.prog[0] = w_start;
        sh.prog[1] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) decide_body<true, true>(g, mb, res, move0, W, w_start, dirty_buf, batch_out, host_out, seq, resumed_plain, &sh, zcheck, ca);
    else commit_waves(st, tab, tab_prev, g, mb, ip, in, orientable, black, own_tag, own_idx, prev_touched, res, move0, W, w_start, nullptr, 0, &sh);
}
