/* ig_kernels_score.cuh -- scoring half of a (batch of) move(s): k_gather, k_mutate, k_offsets, k_slice,
 * k_score_list (the hot kernel), k_delta, k_tail, k_records. */
#pragma once

/* ------------------------------------------------------------------ the move(s)
 *
 * Candidate draws do not depend on the genome (CL:3103-3141 reads fixed distributions), so W consecutive
 * moves can be SCORED against the same base state in single launches (slot dimension w below) and then
 * COMMITTED in order by one workgroup (k_commit_batch) that stops at the first move whose contigs were
 * modified by an earlier move of the batch.  W = 1 is the plain one-move-at-a-time path.
 * Buffers of candidate c of slot w live at index cw = w * capC + c. */

#define CW(w, c) (PS(w) * mb.capC + (c)) /* (w: a position of the window -- ig_common.cuh, MoveBuf.rot) */

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
             int max_c, Tables tab, Tables tab_prev, const int* __restrict__ prev_touched, int force_slot, unsigned* touched_next,
             int n_touched_words)
{
    /* the sub-fragments of this batch's windows, one bit each (mb.touched: set below by the fragments that drop themselves into
     * a window, read by k_slice in front of its gather of the partner's (contig, rank)); two bitmaps take turns: the one the
     * NEXT batch sets is cleared here */
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_touched_words; i += gridDim.x * blockDim.x) touched_next[i] = 0u;
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
    const int N = mb.N, sN = mb.sN;
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
    /* which windows hold a contig: a hash of the launch's contig ids -> its entries (position w; candidate c, or "the focal contig of every
     * candidate of w").  Until round 6 every fragment walked all W x C windows for its contig id -- wave-uniform LDS loads, 8 per window,
     * one after the other: 1 400 dependent LDS round trips per wave for a launch of 36 slots, most of the kernel's 41 us, to find that 99 %
     * of the fragments are in none (tools note in DESIGN 4.2).  Every block builds the same table (300 inserts). */
    constexpr int GH = 1024, GE = IG_MAX_BATCH * (IG_MAX_CANDIDATES + 1);
    __shared__ int gh_head[GH];
    __shared__ int gh_cid[GE], gh_code[GE], gh_next[GE];
    __shared__ int gh_n;
    for (int i = threadIdx.x; i < GH; i += blockDim.x) gh_head[i] = -1;
    if (threadIdx.x == 0) gh_n = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < W * (max_c + 1); i += blockDim.x) {
        const int w = i / (max_c + 1), q = i % (max_c + 1); /* q == 0: the focal contig; q > 0: candidate q - 1's */
        if (KEPT(w)) continue; /* (a slot of the window scored by an earlier launch and still valid: its lists stand) */
        const int cA = sh_cA[w];
        int cid, code;
        if (q == 0) {
            cid = cA;
            code = (w << 8) | 0xff;
        } else {
            if (q - 1 >= sh_C[w]) continue;
            cid = sh_cB[w * IG_MAX_CANDIDATES + q - 1];
            if (cid == cA) continue; /* (a candidate in the focal contig: that contig's entry fills its window) */
            code = (w << 8) | (q - 1);
        }
        const int e = atomicAdd(&gh_n, 1);
        gh_cid[e] = cid;
        gh_code[e] = code;
        gh_next[e] = atomicExch(&gh_head[(unsigned)cid * 2654435761u >> 22], e);
    }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < N) {
        const int cf = st.cid[f], pf = st.pos[f];
        const int lb = st.lb[f], sl = st.sl[f];
        const int sf = st.sub_first[f]; /* (with the others: not a round trip of its own for the fragments that need it) */
        bool in_window = false;
        for (int e = gh_head[(unsigned)cf * 2654435761u >> 22]; e >= 0; e = gh_next[e]) {
            if (gh_cid[e] != cf) continue;
            const int code = gh_code[e], w = code >> 8, cq = code & 0xff;
            const bool focal = cq == 0xff;
            const int slot = focal ? pf : sh_LA[w] + pf;
            const int c_lo = focal ? 0 : cq, c_hi = focal ? sh_C[w] : cq + 1;
            for (int c = c_lo; c < c_hi; c++) {
                if (slot >= sN) g->error = 9; /* a window beyond the buffers' stride: the host's bound on the contig lengths failed */
                else {
                    const size_t o = (size_t)CW(w, c) * sN + slot;
                    mb.Lloc[o] = f;
                    mb.lbloc[o] = lb;
                    mb.slloc[o] = sl;
                    in_window = true;
                }
            }
        }
        if (in_window) { /* its sub-fragments (consecutive ids): one or two words */
            for (int s0 = sf; s0 < sf + sl;) {
                const int wd = s0 >> 5, hi = min(sf + sl, (wd + 1) << 5); /* [s0, hi) lies in word wd */
                const unsigned bits = (hi - s0 >= 32) ? 0xffffffffu : (((1u << (hi - s0)) - 1u) << (s0 & 31));
                atomicOr(&mb.touched[wd], bits);
                s0 = hi;
            }
        }
    }
    const int w = blockIdx.x;
    if (w >= W) return;
    const int t = threadIdx.x;
    if (w == 0 && t < 16 && mb.work) mb.work[t] = 0; /* the exact kernel's work list: lengths and needs of the sub-lists (k_worklist) */
    if (KEPT(w)) {
        if ((mb.ring & 4) && t < sh_C[w]) { /* IG_WINDOW_CHECK=1: a kept slot's candidates and contigs must be what the live state says */
            const CandMeta& m = mb.meta[CW(w, t)];
            const int A = frags_all[move0 + w], B = cands_all[(size_t)(move0 + w) * max_c + t];
            const bool ok = mb.ctl[PS(w)].A == A && m.B == B && st.cid[A] == m.ctgA && st.cid[B] == m.ctgB && st.L[A] == m.LA && st.L[B] == m.LB &&
                            st.SL[A] == m.SLA && st.SL[B] == m.SLB && st.pos[A] == m.lA;
            if (!ok) {
                g->error = 11;
                g->dbg[0] = w;
                g->dbg[1] = t;
                g->dbg[2] = st.cid[A];
                g->dbg[3] = m.ctgA;
                g->dbg[4] = st.cid[B];
                g->dbg[5] = m.ctgB;
                g->dbg[6] = st.L[A] * 1000 + m.LA;
                g->dbg[7] = st.L[B] * 1000 + m.LB;
            }
        }
        return;
    }
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
        mc.superset0 = ((w > 0 || mb.ring) && force_slot < 0) ? 1 : 0; /* (the window rule: a slot may be decided by a later launch, behind other moves) */
        mc.n_dirty = 0;
        mc.pred = -1;
        mc.pred_c = mc.pred_k = 0;
        mc.pred_pad = -1;
        mc.pd_hi = mc.pd_lo = 0;
        mc.overflow = 0;
        mc.pad = 0;
        mc.exact_chunk = 0;
        mc.pad2 = 0;
        mb.ctl[PS(w)] = mc;
    }
    for (int i = t; i < C * P_STRIDE; i += blockDim.x) mb.part[(size_t)CW(w, 0) * P_STRIDE + i] = 0;
    for (int i = t; i < C * Q_STRIDE; i += blockDim.x) mb.qpart[(size_t)CW(w, 0) * Q_STRIDE + i] = 0;
    for (int i = t; i < C * IG_N_TMP_STRUCT; i += blockDim.x) mb.scores[(size_t)CW(w, 0) * IG_N_TMP_STRUCT + i] = 0.0;
    for (int i = t; i < C * NSLOT * 2; i += blockDim.x) ((long long*)mb.scr)[(size_t)CW(w, 0) * NSLOT * 2 + i] = 0;
    if (t < C) {
        mb.scr_void[CW(w, t)] = 0;
        mb.scr_ub[CW(w, t)] = 0;
        mb.cont[CW(w, t)] = 0xffffffffu;
        mb.ident[CW(w, t)] = 0;
        mb.livecol[CW(w, t)] = 0;
        mb.tail_n[CW(w, t)] = -1; /* new windows: the Q5 tail has to be found again */
    }
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
        if (m.n_loc > mb.sN || m.m_loc > mb.sM) g->error = 9; /* a window beyond the buffers' strides */
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
            n = build_uniq(u, true, m->LA, m->LB, (w == 0 && !mb.ring) ? g->valid_insert : nullptr);
        } else {
            n = build_uniq(u, false, m->LA, m->LB, sh_flags[t - 1]);
        }
        m->n_uniq = n;
        m->kidx[IG_N_TMP_STRUCT] = 0; /* current genome = column 0 */
        for (int k = 0; k < n; k++) m->kidx[u[k]] = k + 1;
    }
}

#ifndef MUT_LDS_N
#define MUT_LDS_N 384 /* fragments of a window mutated in LDS (14 arrays x 4 bytes each: 21 KB) */
#endif
/* one workgroup = one candidate genome (slot `slot` of candidate c of move slot w) on the local window */
__device__ __forceinline__ void mutate_one(const State& st, const Tables& tab, const SubTab* __restrict__ sub,
                                           const long long* __restrict__ rowptr, Glob* g, const MoveBuf& mb, const PzTab& pz, int slot,
                                           int c, int w)
{
    const MoveCtl& mc = mb.ctl[PS(w)];
    if (c >= mc.C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int k = m.kidx[slot];
    if (k < 0) return;
    const int N = mb.sN, M = mb.sM, n = m.n_loc; /* strides of the window arrays */
    if (n > N || m.m_loc > M) return; /* k_gather flagged it */
    int* base = mb.loc + ((size_t)(cw * NSLOT + slot) * NDYN) * N;
    /* Windows of up to MUT_LDS_N fragments are mutated in LDS and written out once: the operator chain is a dozen dependent
     * phases over the 13 arrays, each a global-memory round trip when they live in the window buffers. */
    __shared__ int s_loc[(NDYN + 3) * MUT_LDS_N];
    const bool in_lds = n <= MUT_LDS_N;
    int* wb = in_lds ? s_loc : base;
    const size_t ws = in_lds ? (size_t)MUT_LDS_N : (size_t)N;
    igd::Loc S;
    S.pos = wb;
    S.spos = wb + ws;
    S.cid = wb + 2 * ws;
    S.sbp = wb + 3 * ws;
    S.circ = wb + 4 * ws;
    S.prev = wb + 5 * ws;
    S.next = wb + 6 * ws;
    S.L = wb + 7 * ws;
    S.SL = wb + 8 * ws;
    S.LB = wb + 9 * ws;
    S.ori = wb + 10 * ws;
    const int* g_gid = mb.Lloc + (size_t)cw * N;
    S.gid = in_lds ? (const int*)(s_loc + 11 * MUT_LDS_N) : g_gid;
    S.lb = in_lds ? (const int*)(s_loc + 12 * MUT_LDS_N) : (mb.lbloc + (size_t)cw * N);
    S.sl = in_lds ? (const int*)(s_loc + 13 * MUT_LDS_N) : (mb.slloc + (size_t)cw * N);
    S.n = n;
    for (int x = threadIdx.x; x < n; x += blockDim.x) {
        const int f = g_gid[x];
        if (in_lds) {
            s_loc[11 * MUT_LDS_N + x] = f;
            s_loc[12 * MUT_LDS_N + x] = mb.lbloc[(size_t)cw * N + x];
            s_loc[13 * MUT_LDS_N + x] = mb.slloc[(size_t)cw * N + x];
        }
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
            if (in_lds) { /* the candidate genome to where the commit step (and ig_debug_candidate_state) read it */
#pragma unroll
                for (int a = 0; a < NDYN; a++) base[(size_t)a * N + x] = s_loc[a * MUT_LDS_N + x];
            }
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
        if (threadIdx.x == 0) {
            mb.sinfo[cw * NSLOT + slot] = make_int2(sh_ch, sh_hd);
            if (sh_ch && k > 0) atomicOr(&mb.livecol[cw], 1u << k); /* (by column: what the screening kernel pairs) */
        }
    }
    /* ---- coordinate column k (fill_vect_dist, KA:3699-3760) + zero-pixel sums on the window */
    const ig_params p = g->par[0];
    const float mean = g->mean_kb;
    uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
    ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
    int* subs = mb.subs + (size_t)cw * M;
    long long hi = 0, lo = 0, ni = 0;
    __shared__ long long seg_bound[SLICE_SEG];
    if (threadIdx.x < SLICE_SEG) seg_bound[threadIdx.x] = 0;
    __syncthreads();
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
                mb.rowcnt[(size_t)cw * M + ls] = 0; /* k_slice adds the kept contacts of the row (several waves per row) */
                /* upper bound of the slice segment this row appends to; the row's range for k_slice */
                const long long rb = rowptr[s], re = rowptr[s + 1];
                atomicAdd((unsigned long long*)&seg_bound[ls & (mb.nseg - 1)], (unsigned long long)(re - rb));
                mb.rowbe[(size_t)cw * M + ls] = make_int4((int)(unsigned)rb, (int)(rb >> 32), (int)(re - rb), s);
            }
            if (npos == 0) ni += ((long long)SLc * (long long)(SLc - 1)) / 2;
            if (npos > 0) {
                const long long q2 = zero_q(p, npos, SLc, stot, mean, pz.v, pz.n);
                hi += q2 >> 32;
                lo += (long long)(unsigned int)q2;
            }
        }
    }
    __shared__ long long red[4][16];
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    ni = wave_sum_ll(ni);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
        red[2][wv] = ni;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long* q = mb.qpart + (size_t)cw * Q_STRIDE;
        long long s0 = 0, s1 = 0, s2 = 0;
        for (int v = 0; v < (int)(blockDim.x >> 6); v++) { /* 4 or 16 waves (k_mutate's two launch shapes) */
            s0 += red[0][v];
            s1 += red[1][v];
            s2 += red[2][v];
        }
        q[Q_Z + 2 * k] = s0;
        q[Q_Z + 2 * k + 1] = s1;
        q[Q_NI + k] = s2;
    }
    if (k == 0 && threadIdx.x < SLICE_SEG) mb.slbound[(size_t)cw * SLICE_SEG + threadIdx.x] = seg_bound[threadIdx.x];
}

/* k_mutate: the 25 genomes of every candidate of the move slots [w_begin, w_begin + gridDim.z) */
__global__ void __launch_bounds__(1024) k_mutate(State st, Tables tab, const SubTab* __restrict__ sub, const long long* __restrict__ rowptr,
                                                Glob* g, MoveBuf mb, PzTab pz, int w_begin)
{
    if (KEPT(w_begin + (int)blockIdx.z)) return; /* (a slot of the window scored by an earlier launch and still valid) */
    mutate_one(st, tab, sub, rowptr, g, mb, pz, blockIdx.x, blockIdx.y, w_begin + blockIdx.z);
}

/* k_mutate_winners (slots split over GPUs): a rank builds the candidate genomes of its own slots only; once the batch is
 * decided it rebuilds, for the slots of the OTHER ranks that the commit step is about to apply (and for a pending one),
 * just what that step reads -- the current genome (column 0: local row -> sub-fragment map) and the winner. */
__global__ void __launch_bounds__(256) k_mutate_winners(State st, Tables tab, const SubTab* __restrict__ sub,
                                                        const long long* __restrict__ rowptr, Glob* g, MoveBuf mb, PzTab pz, int w_start,
                                                        int own_begin, int own_end, const int* __restrict__ batch_out)
{
    const int w = blockIdx.y;
    const int committed = batch_out[0], pending = batch_out[1];
    if (w >= own_begin && w < own_end) return; /* built before scoring */
    if (!((w >= w_start && w < committed) || w == pending)) return;
    const MoveCtl& mc = mb.ctl[PS(w)];
    if (blockIdx.x == 1 && mc.ch_slot == IG_N_TMP_STRUCT) return;
    mutate_one(st, tab, sub, rowptr, g, mb, pz, blockIdx.x == 0 ? IG_N_TMP_STRUCT : mc.ch_slot, mc.ch_c, w);
}

/* k_offsets: where each candidate's slice list starts in the pool = exclusive prefix sum of the upper bounds
 * (one small workgroup; a slot whose lists do not fit is flagged and re-run at the head of the next batch) */
#define OFFSETS_THREADS 1024
__global__ void __launch_bounds__(OFFSETS_THREADS) k_offsets(MoveBuf mb, int W, int w_begin, int w_end, int max_c)
{
    /* entries = (slot, candidate, segment) in this order; thread t owns `per` consecutive entries; exclusive scan of the
     * per-thread sums across the workgroup (wave scans + one LDS step) */
    __shared__ long long wave_tot[OFFSETS_THREADS / 64];
    __shared__ long long s_tot[IG_MAX_BATCH * IG_MAX_CANDIDATES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    /* (for the order of the screening launch, below) */
    const int n_idx = (w_end - w_begin) * max_c;
    const bool can_order = n_idx <= IG_MAX_BATCH * IG_MAX_CANDIDATES && n_idx <= (int)OFFSETS_THREADS;
    long long my_tot = 0;
    if (can_order && tid < n_idx) {
        const int w = w_begin + tid / max_c, c = tid % max_c;
        const bool used = c < mb.ctl[PS(w)].C && !KEPT(w);
        for (int sg = 0; sg < SLICE_SEG; sg++) {
            const long long b = mb.slbound[(size_t)CW(w, c) * SLICE_SEG + sg]; /* (requested whether used or not: no wait in front of the scan) */
            my_tot += (used && b > 0) ? b : 0;
        }
    }
    const int n = W * mb.capC * SLICE_SEG;
    const int per = (n + OFFSETS_THREADS - 1) / OFFSETS_THREADS;
    /* entry i = (position w, candidate c, segment): its arrays live at the PHYSICAL index phys(i) (MoveBuf.rot); the pool is dealt
     * out in the order of the positions -- the window's front first */
    auto phys = [&](int i) -> size_t {
        const int cwl = i / SLICE_SEG;
        return (size_t)CW(cwl / mb.capC, cwl % mb.capC) * SLICE_SEG + (size_t)(i % SLICE_SEG);
    };
    auto bound_of = [&](int i) -> long long {
        const int cwl = i / SLICE_SEG;
        const int w = cwl / mb.capC, c = cwl % mb.capC;
        return (w >= w_begin && w < w_end && !KEPT(w) && c < mb.ctl[PS(w)].C) ? mb.slbound[phys(i)] : -1; /* -1: not an entry of this launch */
    };
    /* a thread's entries stay in registers over the three passes (round 6: each pass used to load them again -- control block, then
     * bound: two dependent round trips per entry and pass, a dozen of the kernel's 20 us); n <= 64 x 16 x 16: at most 16 per thread */
    constexpr int PER_MAX = (IG_MAX_BATCH * IG_MAX_CANDIDATES * SLICE_SEG + OFFSETS_THREADS - 1) / OFFSETS_THREADS;
    long long bq[PER_MAX];
    __shared__ int s_over[IG_MAX_BATCH]; /* slot (position) w has a list that does not fit the pool */
    if (tid < IG_MAX_BATCH) s_over[tid] = 0;
    long long sum = 0;
#pragma unroll
    for (int q = 0; q < PER_MAX; q++) {
        const int i = tid * per + q;
        bq[q] = (q < per && i < n) ? bound_of(i) : -1;
        sum += bq[q] > 0 ? bq[q] : 0;
    }
    long long incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    long long run = incl - sum;
    for (int q = 0; q < wv; q++) run += wave_tot[q];
#pragma unroll
    for (int q = 0; q < PER_MAX; q++) {
        const int i = tid * per + q;
        const long long b = bq[q];
        if (b >= 0) {
            if (run + b > mb.pool_cap) {
                mb.sloff[phys(i)] = -1;
                mb.ctl[PS(i / SLICE_SEG / mb.capC)].overflow = 1;
                s_over[i / SLICE_SEG / mb.capC] = 1;
            } else {
                mb.sloff[phys(i)] = run;
            }
            run += b;
        }
    }
    /* a slot that does not fit is re-run as a whole: all its segments are marked */
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER_MAX; q++) {
        const int i = tid * per + q;
        if (bq[q] >= 0 && s_over[i / SLICE_SEG / mb.capC]) mb.sloff[phys(i)] = -1;
    }
    /* the (slot, candidate) pairs of this launch by falling list size: the screening kernel's workgroups are as long as their
     * candidate's lists (two grown contigs: ten times the median), and the long ones handed out last were the launch's tail
     * (the totals were requested at the top: their round trips ran next to the scan's) */
    if (can_order) {
        if (tid < n_idx) s_tot[tid] = my_tot;
        __syncthreads();
        if (tid < n_idx) {
            int rank = 0;
            for (int u = 0; u < n_idx; u++) rank += (s_tot[u] > my_tot) || (s_tot[u] == my_tot && u < tid);
            mb.order[rank] = ((w_begin + tid / max_c) << 8) | (tid % max_c);
        }
        /* ... and the slots by falling size of all their lists (k_slice's grid: a slot per z) */
        const int n_slots = w_end - w_begin;
        long long st = 0;
        if (tid < n_slots)
            for (int cc2 = 0; cc2 < max_c; cc2++) st += s_tot[tid * max_c + cc2];
        __syncthreads();
        if (tid < n_slots) s_tot[tid] = st;
        __syncthreads();
        if (tid < n_slots) {
            int rank = 0;
            for (int u = 0; u < n_slots; u++) rank += (s_tot[u] > st) || (s_tot[u] == st && u < tid);
            mb.order[mb.capC * mb.capW + rank] = w_begin + tid;
        }
    } else if (tid < w_end - w_begin) {
        mb.order[mb.capC * mb.capW + tid] = w_begin + tid;
    }
}

#define LDS_COL_CAP 4096  /* local sub-fragments whose column fits the 32 KB LDS stage */
#define DELTA_RB 128

/* k_slice: slice_sp_mat (KA:485-607) restricted to the CSR rows of the touched contigs (instead of a scan of
 * all Z contacts).  One wave per row: up to SLICE_UNROLL x 64 contacts are loaded back to back (coalesced
 * 8-byte loads, then one packed (contig, rank) gather each), the predicate is evaluated, and the kept ones
 * are appended to the candidate's list with ONE wave-aggregated atomic per batch (ballot + popcount ranks).
 * No sort afterwards: the reference sorted by row only to feed its shared-memory row cache (CL:1045-1050). */
#ifndef SLICE_RB
#define SLICE_RB 96 /* workgroups (of 4 rows at a time) per candidate plane (with the one-bit partner filter: 64 .. 128 within 3 %, 32: +15 %, 256: +20 %) */
#endif
#ifndef SLICE_UNROLL
#define SLICE_UNROLL 4
#endif
#ifndef SLICE_MIN_WAVES
#define SLICE_MIN_WAVES 8 /* eight workgroups per CU need <= 80 SGPRs (81 admit seven: MI355X_MICROARCH.md, residency) */
#endif
/* grid: (workgroups, candidates + 1, slots).  The rows of the focal contig A are the same for every candidate of a move whose
 * partner lies in another contig, and so are the contacts read from them and the partners' records gathered: the first plane
 * walks A's rows ONCE for all those candidates (a contact inside A goes to every list, one into B_c to candidate c's), the
 * candidates' planes walk the rows of B_c only (all rows where A and B_c are one contig: the windowed predicate of
 * KA:565-586 is the candidate's own).  40 % fewer (row, contact chunk) chains per move at five candidates.
 * Measured and dropped: several shared planes (slower), the cursors one per 128-byte line (no change), write-through / non-temporal
 * list stores (no change / slower), one workgroup per (segment, slot) with the cursors in LDS (190 us: too few chains in
 * flight), counting pass + writing pass without atomics (178 us: the rows are read twice). */
template <bool PACKED>
__global__ void __launch_bounds__(256, SLICE_MIN_WAVES) k_slice(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab,
                                                                 Glob* g, MoveBuf mb, int rank, int world, int w_begin, int share_rows, int max_j)
{
    const int w = mb.order[mb.capC * mb.capW + blockIdx.z]; /* (the slots with the longest rows first: k_offsets; w_begin + z without it) */
    if (KEPT(w)) return;
    const bool shared_plane = (blockIdx.y == 0); /* first: its waves write every kept contact once per candidate */
    const int cand_plane = (int)blockIdx.y - 1;
    __shared__ long long seg_off[IG_MAX_CANDIDATES][SLICE_SEG];
    __shared__ int s_cw[IG_MAX_CANDIDATES], s_ctgB[IG_MAX_CANDIDATES], s_idx[IG_MAX_CANDIDATES], s_nc;
    __shared__ int a_same[IG_MAX_CANDIDATES], a_ctgB[IG_MAX_CANDIDATES], a_SLA, a_ctgA, a_mloc[IG_MAX_CANDIDATES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    /* ONE round trip for everything the prologue needs (this launch is tens of thousands of short waves: every dependent load in
     * front of the rows counts): the slot's candidate count, the few fields of every candidate's window, every list start */
    const int C = mb.ctl[PS(w)].C;
    const int capC = min(mb.capC, IG_MAX_CANDIDATES);
    if ((int)threadIdx.x < capC) {
        const CandMeta& mt = mb.meta[CW(w, threadIdx.x)]; /* (entries behind the slot's last candidate: stale but in bounds, not used) */
        a_same[threadIdx.x] = mt.same;
        a_ctgB[threadIdx.x] = mt.ctgB;
        a_mloc[threadIdx.x] = mt.m_loc;
        if (threadIdx.x == 0) {
            a_SLA = mt.SLA; /* the focal contig is every candidate's */
            a_ctgA = mt.ctgA;
        }
    }
    for (int i = threadIdx.x; i < capC * SLICE_SEG; i += blockDim.x) seg_off[i / SLICE_SEG][i % SLICE_SEG] = mb.sloff[(size_t)CW(w, i / SLICE_SEG) * SLICE_SEG + i % SLICE_SEG];
    __syncthreads();
    if (threadIdx.x == 0) {
        int nc = 0;
        if (shared_plane) {
            if (share_rows)
                for (int c = 0; c < C; c++)
                    if (!a_same[c]) {
                        s_idx[nc] = c;
                        s_cw[nc] = CW(w, c);
                        s_ctgB[nc++] = a_ctgB[c];
                    }
        } else if (cand_plane < C) {
            s_idx[0] = cand_plane;
            s_cw[0] = CW(w, cand_plane);
            s_ctgB[0] = a_ctgB[cand_plane];
            nc = 1;
        }
        s_nc = nc;
    }
    __syncthreads();
    const int nc = s_nc;
    if (nc == 0) return;
    if (seg_off[0][0] < 0) return; /* slice pool exhausted (k_offsets flags all segments of a slot together) */
    const int cw0 = s_cw[0], c0 = s_idx[0];
    const CandMeta& m = mb.meta[cw0]; /* only the windowed predicate of a candidate in A's own contig reads it */
    const int M = mb.sM;
    const bool m_same = a_same[c0] != 0;
    /* the rows this plane walks: A's (shared plane), B's (a candidate in another contig, A's being walked by the shared plane), all */
    const bool own_all = !shared_plane && (m_same || !share_rows);
    const int SLA = a_SLA, ctgA = a_ctgA;
    const int row_lo = (shared_plane || own_all) ? 0 : SLA;
    const int n_rows = shared_plane ? SLA : (a_mloc[c0] - row_lo);
    const int4* rowbe = mb.rowbe + (size_t)cw0 * M;
    const int ctgB0 = s_ctgB[0];
    /* work items = (row, j): wave j of a row takes the row's contact chunks j, j + J, ... (J waves per row: a row of thousands
     * of contacts is a chain of dependent round trips per chunk, and the launch waits for the longest chain) */
    const int nrw = gridDim.x * 4;
    const int J = min(max_j, max(1, nrw / max(n_rows, 1)));
    for (int item = blockIdx.x * 4 + wv; item < n_rows * J; item += nrw) {
        const int r = row_lo + item % n_rows, j = item / n_rows;
        const int seg = r & (mb.nseg - 1);
        /* the row's range, written by k_mutate next to the window's sub-fragment list: ONE round trip in front of the contacts;
         * its (contig, rank) follows from its place in the window (A's sub-fragments first, by rank, then B's) */
        const int4 be = rowbe[r];
        const long long b = (long long)(unsigned)be.x | ((long long)be.y << 32), e = b + be.z;
        const bool in_a = m_same || r < SLA;
        const int2 cp1 = make_int2(in_a ? ctgA : ctgB0, in_a ? r : r - SLA);
        const bool mine = (world <= 1) || ((r % world) == rank);
        int rc = 0; /* lane k: contacts of this row kept for candidate k */
        if (b != e) {
            for (long long q0 = b + (long long)j * 64 * SLICE_UNROLL; q0 < e; q0 += (long long)J * 64 * SLICE_UNROLL) {
                int2 v[SLICE_UNROLL], cp2[SLICE_UNROLL];
#pragma unroll
                for (int u = 0; u < SLICE_UNROLL; u++) {
                    const long long qi = q0 + u * 64 + lane;
                    v[u] = (qi < e) ? cc[qi] : make_int2(-1, 0);
                }
#pragma unroll
                for (int u = 0; u < SLICE_UNROLL; u++) { /* a kept partner lies in a window of the batch: one bit (75 KB at 600 k sub-fragments) in front of the 8-byte gather */
                    const bool hit = (v[u].x >= 0) && ((mb.touched[v[u].x >> 5] >> (v[u].x & 31)) & 1u);
                    cp2[u] = hit ? tab.cp[v[u].x] : make_int2(-1, -1);
                }
                for (int k = 0; k < nc; k++) {
                    bool keep[SLICE_UNROLL];
                    unsigned long long mask[SLICE_UNROLL];
                    int add = 0;
                    const int ctgB = s_ctgB[k];
#pragma unroll
                    for (int u = 0; u < SLICE_UNROLL; u++) {
                        if (own_all) keep[u] = (v[u].x >= 0) && slice_keep(m, cp1.x, cp2[u].x, cp1.y, cp2[u].y, v[u].y, false);
                        else keep[u] = (v[u].x >= 0) && (v[u].y > 0) && ((cp2[u].x == ctgA) || (cp2[u].x == ctgB)); /* slice_keep, two contigs */
                        mask[u] = __ballot(keep[u]);
                        add += __popcll(mask[u]);
                    }
                    if (!add) continue;
                    rc += (lane == k) ? add : 0;
                    if (!mine) continue;
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd((unsigned long long*)(mb.part + (size_t)s_cw[k] * P_STRIDE + P_CNT + seg), (unsigned long long)add);
                    base = __shfl(base, 0, 64);
                    const long long off = seg_off[s_idx[k]][seg];
                    int o2 = 0;
#pragma unroll
                    for (int u = 0; u < SLICE_UNROLL; u++) {
                        if (keep[u]) {
                            const long long at = off + (long long)base + o2 + __popcll(mask[u] & lt_mask);
                            const int lj = ((m_same || cp2[u].x == ctgA) ? 0 : SLA) + cp2[u].y;
                            if (PACKED) {
                                mb.sl_pk[at] = (unsigned long long)r | ((unsigned long long)lj << 20) | ((unsigned long long)v[u].y << 40);
                            } else {
                                mb.sl_li[at] = r;
                                mb.sl_lj[at] = lj;
                                mb.sl_ob[at] = v[u].y;
                            }
                        }
                        o2 += __popcll(mask[u]);
                    }
                }
            }
        }
        /* every rank knows every row's count: the tail walk needs them */
        if (lane < nc && rc) atomicAdd(&mb.rowcnt[(size_t)s_cw[lane] * M + r], rc);
    }
}

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
 * term_checked: the long way, for the workgroups of score_loop_general (a circular contig on the window, or parameters
 * outside the one-log domain of the contract) */
#define DKEY_CIRC 0x40000000
__device__ __forceinline__ void term_checked(const ig_hot& h, const ig_params& p, float mean_kb, float s, int dkey, int ob,
                                             const PzTab& pz, const double* lgf_s, const double* __restrict__ lgf_tab,
                                             const ColMeta* cm_s, const double* T, int& qh, unsigned& ql)
{
    const bool inter = dkey < 0;
    const double lgf = (ob < LDS_LGF) ? lgf_s[max(ob, 0)] : lgfact_dev(ob, lgf_tab);
    double t;
    if (!inter && (dkey & DKEY_CIRC)) {
        const int code = (dkey >> 27) & 7;
        t = term_circ(p, mean_kb, s, cm_s[code].stot, dkey & 0x07ffffff, cm_s[code].len, ob, lgf);
    } else {
        t = term_general(p, mean_kb, s, dkey, ob, lgf, pz);
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
    const int *sli, *slj, *slo;    /* the segment of the slice list (< 2^31 entries, 32-bit offsets from a uniform base) */
    const unsigned long long* slp; /* or its packed form */
    unsigned n;
    const uint2* gcol; /* column k in global memory */
    const uint2* lcol; /* and its LDS copy */
    const double *pzc_s, *lgf_s, *mt_s; /* LDS tables: P_z * log10(e) per rank distance, log10(ob!), log2 / exp2 */
    const ColMeta* cm_s;
    const double* lgf_tab;
    PzTab pz;
    unsigned circ_mask;
    float mean;
};
/* tuning (measured on cfg3): terms interleaved per lane, waves per SIMD the register budget is cut for */
#ifndef SCORE_BATCH
#define SCORE_BATCH 2
#endif
#ifndef SCORE_WAVES
#define SCORE_WAVES 6
#endif
__device__ __forceinline__ unsigned abs_diff_u32(unsigned x, unsigned y)
{
    unsigned r;
    __asm__("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
/* one entry of the list: (local row, local column, count) */
template <bool PACKED>
__device__ __forceinline__ void load_entry(const ScoreArgs& a, unsigned e, unsigned& li, unsigned& lj, unsigned& ob)
{
    if (PACKED) {
        const unsigned long long pk = a.slp[e];
        li = (unsigned)pk & 0xfffffu;
        lj = (unsigned)(pk >> 20) & 0xfffffu;
        ob = (unsigned)(pk >> 40);
    } else {
        li = (unsigned)a.sli[e];
        lj = (unsigned)a.slj[e];
        ob = (unsigned)a.slo[e];
    }
}
/* the streaming loop for the rare workgroups (see term_checked), out of line.  Arguments by value: a reference would put
 * the caller's argument block in scratch memory for every thread of the kernel. */
template <bool STAGED, bool PACKED>
__device__ __noinline__ longlong2 score_loop_general(const ScoreArgs a, const ig_params p)
{
    const ig_hot hp = ig_hot_make(p, ig_tab());
    long long hi = 0, lo = 0;
    for (unsigned e = threadIdx.x; e < a.n; e += SCORE_THREADS) {
        unsigned l_i, l_j, o_b;
        load_entry<PACKED>(a, e, l_i, l_j, o_b);
        const uint2 ai = STAGED ? a.lcol[l_i] : a.gcol[l_i];
        const uint2 bj = STAGED ? a.lcol[l_j] : a.gcol[l_j];
        float sv;
        int dkey, qh;
        unsigned ql;
        classify_pair(ai, bj, a.circ_mask, sv, dkey);
        term_checked(hp, p, a.mean, sv, dkey, (int)o_b, a.pz, a.lgf_s, a.lgf_tab, a.cm_s, a.mt_s, qh, ql);
        hi += qh;
        lo += (long long)ql;
    }
    return make_longlong2(hi, lo);
}
/* ... and for everything else: linear contigs only, parameters in the one-log domain (ig_hot.fast).  The term is the
 * contract's ig_term_hot, spelled out with the count's log-factorial and P_z * log10(e) read from the LDS tables
 * (entries from the table's end on hold the trans level: a trans pair, and a cis pair beyond the table, read that).
 * Counts >= LDS_LGF, zero counts and rank distances beyond a table longer than the LDS copy are fixed up behind a
 * wave-uniform branch (term_of_entry).  Quantisation: t * 2^32 + 1.5 * 2^52 rounds to the integer (half-even, as
 * ig_quantize) and leaves it in the low 52 bits of the sum; |t| >= 2^19 (or NaN) takes ig_quantize itself.
 * STAGED: the column is in LDS (ds_read), else 8-byte gathers from L2. */
#define IG_QMAGIC 6755399441055744.0
#define IG_QMAGIC_BITS 0x4338000000000000ULL
#define SCORE_FLUSH (1u << 20) /* entries between two folds of the lane sums into the limbs */
/* the checked term of one entry, from scratch (coordinates from the global column): the fix-up of score_loop */
__device__ __noinline__ double term_of_entry(const uint2* gcol, float mean, const double* lgf_tab, PzTab pz, const ig_params p, unsigned l_i,
                                             unsigned l_j, unsigned o_b)
{
    const uint2 ai = gcol[l_i], bj = gcol[l_j];
    const bool cis = (ai.y ^ bj.y) < 0x10000000u;
    const unsigned d = abs_diff_u32(ai.y, bj.y);
    const float sv = fabsf(__uint_as_float(ai.x) - __uint_as_float(bj.x));
    return term_general(p, mean, cis ? sv : 0.0f, cis ? (int)d : -1, (int)o_b, lgfact_dev((int)o_b, lgf_tab), pz);
}
template <bool STAGED, bool PACKED>
__device__ __forceinline__ void score_loop(const ScoreArgs& a, const ig_hot& h, const ig_params& p, long long& hi, long long& lo)
{
    const double* T = a.mt_s;
    const unsigned cut = a.pz.n > LDS_PZ ? (unsigned)LDS_PZ : 0xffffffffu;
    const double lv = h.log2_v_inter, slope = h.slope, la = h.log2_amp;
    const float d_max = h.d_max;
    for (unsigned base = 0; base < a.n; base += SCORE_FLUSH) {
        const unsigned end = min(a.n, base + SCORE_FLUSH);
        unsigned long long acc = 0, accl = 0;
        /* a wave takes steps of 64 x SCORE_BATCH consecutive entries (steps wave, wave + 4, ...: at most one partly filled
         * step per wave); the entries of the next step are loaded before this step's terms are evaluated (the loads' L2
         * latency is several times a step's arithmetic) */
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned step = 64 * SCORE_BATCH, stride = step * (SCORE_THREADS / 64);
        unsigned nli[SCORE_BATCH], nlj[SCORE_BATCH], nob[SCORE_BATCH];
#pragma unroll
        for (int u = 0; u < SCORE_BATCH; u++) load_entry<PACKED>(a, min(base + wave * step + u * 64 + lane, end - 1), nli[u], nlj[u], nob[u]);
        for (unsigned e0 = base + wave * step + lane; e0 - lane < end; e0 += stride) {
            unsigned li[SCORE_BATCH], lj[SCORE_BATCH], ob[SCORE_BATCH];
#pragma unroll
            for (int u = 0; u < SCORE_BATCH; u++) {
                li[u] = nli[u];
                lj[u] = nlj[u];
                ob[u] = nob[u];
            }
#pragma unroll
            for (int u = 0; u < SCORE_BATCH; u++) load_entry<PACKED>(a, min(e0 + stride + u * 64, end - 1), nli[u], nlj[u], nob[u]);
            double t[SCORE_BATCH];
            bool rare[SCORE_BATCH], any_rare = false;
#pragma unroll
            for (int u = 0; u < SCORE_BATCH; u++) {
                const uint2 ai = STAGED ? a.lcol[li[u]] : a.gcol[li[u]];
                const uint2 bj = STAGED ? a.lcol[lj[u]] : a.gcol[lj[u]];
                const unsigned o_b = ob[u];
                const bool cis = (ai.y ^ bj.y) < 0x10000000u; /* same contig code */
                const unsigned d = abs_diff_u32(ai.y, bj.y);   /* then: the rank distance */
                const float sv = fabsf(__uint_as_float(ai.x) - __uint_as_float(bj.x));
                const bool in = cis && (sv > 0.0f) && (sv < d_max);
                const double pzc = a.pzc_s[cis ? min(d, (unsigned)LDS_PZ) : (unsigned)LDS_PZ];
                const double lgf = a.lgf_s[min(o_b, (unsigned)(LDS_LGF - 1))];
                const double y = ig_fma(slope, ig_log2_pos((double)sv, T), la);
                const double yy = in ? __builtin_fmax(y, lv) : lv; /* in: y is a number */
                const double ex = ig_exp2_core(yy, T);
                const double lg = yy * IG_LOG2_10_INV;
                t[u] = (ig_fma((double)o_b, lg, -ex) - lgf) + pzc;
                rare[u] = (o_b - 1u >= (unsigned)(LDS_LGF - 1)) || (cis && d >= cut);
                any_rare |= rare[u];
            }
            if (__any(any_rare)) {
#pragma unroll
                for (int u = 0; u < SCORE_BATCH; u++)
                    if (rare[u]) t[u] = term_of_entry(a.gcol, a.mean, a.lgf_tab, a.pz, p, li[u], lj[u], ob[u]);
            }
            unsigned long long bits[SCORE_BATCH];
            bool any_big = false;
#pragma unroll
            for (int u = 0; u < SCORE_BATCH; u++) {
                bits[u] = ig_d2u(ig_fma(t[u], IG_QSCALE, IG_QMAGIC));
                any_big |= !(__builtin_fabs(t[u]) < 524288.0);
            }
            if (__any(any_big)) {
#pragma unroll
                for (int u = 0; u < SCORE_BATCH; u++)
                    if (!(__builtin_fabs(t[u]) < 524288.0)) bits[u] = (unsigned long long)ig_quantize(t[u]) + IG_QMAGIC_BITS;
            }
#pragma unroll
            for (int u = 0; u < SCORE_BATCH; u++) {
                if (e0 + u * 64 < end) { /* the constant of the rounding trick goes out with the entry */
                    acc += bits[u] - IG_QMAGIC_BITS;
                    accl += (unsigned)bits[u];
                }
            }
        }
        const long long q = (long long)acc;
        hi += (q - (long long)accl) >> 32; /* exact: the sum of the q >> 32 */
        lo += (long long)accl;              /* the sum of the (uint32) q */
    }
}
/* k_score_list: the hot kernel.  One workgroup = (segment of the slice list, coordinate column k, candidate cw).
 * Staged in LDS: the tables (ScoreTables, one 13 KB copy) and the column (8 B per local sub-fragment).  Waves stream the
 * slice list (coalesced 8-byte entries), read both endpoints' coordinates from LDS, evaluate the Rippe / Poisson term
 * and add it as an exact integer.  Wave shuffles, one LDS step, two atomics per workgroup.
 * Windows above LDS_COL_SMALL sub-fragments are not staged: 8-byte gathers of the coordinates from L2 (a second instance of
 * the loop behind a uniform branch).  A 32 KB-column instance for them was measured slower (3 waves / SIMD) even on windows
 * of thousands of sub-fragments and is gone. */
#ifndef LDS_COL_SMALL
#define LDS_COL_SMALL 1024
#endif
/* what every workgroup of k_score_list stages: built once per parameter set (k_build_score_const), copied to LDS as is */
struct ScoreTables {
    double mt[IG_TAB_SIZE];  /* the log2 / exp2 tables of the arithmetic contract; at LDS offset 0: the table pair is read without address arithmetic */
    double pzc[LDS_PZ + 2];  /* P_z * log10(e) per rank distance; from the table's end on (and for trans pairs): the trans level */
    double lgf[LDS_LGF];     /* log10(ob!) */
};
#define TILE_HB 64 /* bins of a tile's histogram of counts (k_full_nz_tiled) */
struct ScoreConst {
    ScoreTables tab;
    ig_hot hot;
    ig_params par;
    float mean_kb;
    unsigned long long qtrans[TILE_HB]; /* the quantised term of a trans pair with count ob (+ the rounding magic): k_tile_trans */
};
static_assert(sizeof(ScoreTables) % 16 == 0, "copied as 16-byte vectors");
/* the quantised trans term of count o_b (+ the rounding magic): the straight-line term of the kernels below with the trans
 * level for P and P_z -- same expression, same bits */
__device__ __forceinline__ unsigned long long trans_term_bits(unsigned o_b, double lv, const double* T, const double* lgf, double pzc_trans)
{
    const double ex = ig_exp2_core(lv, T), lg = lv * IG_LOG2_10_INV;
    const double t = (ig_fma((double)o_b, lg, -ex) - lgf[o_b]) + pzc_trans;
    return !(__builtin_fabs(t) < 524288.0) ? (unsigned long long)ig_quantize(t) + IG_QMAGIC_BITS : ig_d2u(ig_fma(t, IG_QSCALE, IG_QMAGIC));
}

/* one wave per off-diagonal tile with a histogram (the last n_trans_blocks blocks); the blocks in front of them, if any, are the
 * zero-pixel pass over all sub-fragments (k_full_zero's job: it needs nothing from the tiles and would otherwise be one more
 * launch behind k_full_nz_tiled) */
__device__ __forceinline__ void build_score_const_block(const Glob* g, PzTab pz, const double* __restrict__ lgf_tab, ScoreConst* out, int which)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const ig_params p = g->par[which];
    const int pzn = min(pz.n, LDS_PZ);
    if (i < IG_TAB_SIZE) out->tab.mt[i] = ig_tab()[i];
    if (i < LDS_PZ + 2) out->tab.pzc[i] = (double)(i < pzn ? pz.v[i] : p.v_inter) * IG_LOG_E_F;
    if (i < LDS_LGF) out->tab.lgf[i] = lgf_tab[i];
    if (i == 0) {
        out->hot = ig_hot_make(p, ig_tab());
        out->par = p;
        out->mean_kb = g->mean_kb;
    }
    if (i < TILE_HB) { /* from the sources of the table entries it reads (they are being written by other threads) */
        const ig_hot h = ig_hot_make(p, ig_tab());
        out->qtrans[i] = i ? trans_term_bits((unsigned)i, h.log2_v_inter, ig_tab(), lgf_tab, (double)p.v_inter * IG_LOG_E_F) : IG_QMAGIC_BITS;
    }
}
__global__ void k_build_score_const(const Glob* g, PzTab pz, const double* __restrict__ lgf_tab, ScoreConst* out, int which)
{
    build_score_const_block(g, pz, lgf_tab, out, which);
}

template <int CAP>
struct ScoreLds {
    ScoreTables tab;
    uint2 col[CAP];
    ColMeta cm[NCODE];
    long long red[2][SCORE_THREADS / 64];
};
/* one workgroup: segment blockIdx.x of the slice list of candidate (w, c), column k */
/* one workgroup: entries [first, first + count) of segment `seg` of the slice list of candidate (w, c), under column k */
template <int CAP>
__device__ __forceinline__ void score_workgroup(ScoreLds<CAP>& L, const ScoreConst* __restrict__ sc, const MoveBuf& mb,
                                                const double* __restrict__ lgf_tab, const PzTab& pz, int ablate, int w, int c, int k,
                                                int seg, long long first, long long count)
{
    /* everything the early exits and the set-up need is loaded before the first branch: one round trip, not five */
    const int cw = CW(w, c);
    const int C = mb.ctl[PS(w)].C;
    const int n_uniq = mb.meta[cw].n_uniq, m_loc = mb.meta[cw].m_loc;
    const long long n_seg = mb.part[(size_t)cw * P_STRIDE + P_CNT + seg];
    const long long off = mb.sloff[(size_t)cw * SLICE_SEG + seg] + first;
    const long long n = n_seg - first < count ? n_seg - first : count;
    const ig_params p = sc->par;
    const ig_hot hp = sc->hot;
    const float mean = sc->mean_kb;
    if (c >= C || k > n_uniq || n <= 0 || off < first) return;
    const int M = mb.sM;
    const uint2* gcol = mb.coords + (size_t)(cw * NSLOT + k) * M;
    const bool staged = m_loc <= CAP;
    {
        const float4* src = (const float4*)&sc->tab;
        float4* dst = (float4*)&L.tab;
        for (int i = threadIdx.x; i < (int)(sizeof(ScoreTables) / 16); i += SCORE_THREADS) dst[i] = src[i];
    }
    if (staged)
        for (int i = threadIdx.x; i < m_loc; i += SCORE_THREADS) L.col[i] = gcol[i];
    if (threadIdx.x < NCODE) L.cm[threadIdx.x] = mb.cmeta[(size_t)(cw * NSLOT + k) * NCODE + threadIdx.x];
    __syncthreads();
    unsigned circ_mask = 0;
#pragma unroll
    for (int q = 0; q < NCODE; q++) circ_mask |= (L.cm[q].stot != 0) ? (1u << q) : 0u;
    long long hi = 0, lo = 0;
    const ScoreArgs sa{mb.packed ? nullptr : mb.sl_li + off, mb.packed ? nullptr : mb.sl_lj + off, mb.packed ? nullptr : mb.sl_ob + off,
                       mb.packed ? mb.sl_pk + off : nullptr, (unsigned)n, gcol, L.col, L.tab.pzc, L.tab.lgf, L.tab.mt, L.cm, lgf_tab, pz, circ_mask, mean};
    const bool general = circ_mask || !hp.fast || (ablate & 2);
    if (general) {
        longlong2 r;
        if (mb.packed) r = staged ? score_loop_general<true, true>(sa, p) : score_loop_general<false, true>(sa, p);
        else r = staged ? score_loop_general<true, false>(sa, p) : score_loop_general<false, false>(sa, p);
        hi = r.x;
        lo = r.y;
    } else if (mb.packed) {
        if (staged) score_loop<true, true>(sa, hp, p, hi, lo);
        else score_loop<false, true>(sa, hp, p, hi, lo);
    } else {
        if (staged) score_loop<true, false>(sa, hp, p, hi, lo);
        else score_loop<false, false>(sa, hp, p, hi, lo);
    }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        L.red[0][wv] = hi;
        L.red[1][wv] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hi = L.red[0][0] + L.red[0][1] + L.red[0][2] + L.red[0][3];
        lo = L.red[1][0] + L.red[1][1] + L.red[1][2] + L.red[1][3];
        if (hi | lo) {
            long long* part = mb.part + (size_t)cw * P_STRIDE;
            atomic_add_ll(&part[P_NZ + 2 * k], hi);
            atomic_add_ll(&part[P_NZ + 2 * k + 1], lo);
        }
    }
}

template <int CAP>
__global__ void __launch_bounds__(SCORE_THREADS) __attribute__((amdgpu_waves_per_eu(SCORE_WAVES)))
    k_score_list(const ScoreConst* __restrict__ sc, MoveBuf mb, const double* __restrict__ lgf_tab, PzTab pz, int ablate, int max_c,
                 int w_begin, int contenders_only)
{
    __shared__ ScoreLds<CAP> L;
    if (!contenders_only) { /* every column: the grid is (segment, column, candidate), a workgroup streams a whole segment */
        if (KEPT(w_begin + (int)blockIdx.z / max_c)) return;
        score_workgroup<CAP>(L, sc, mb, lgf_tab, pz, ablate, w_begin + blockIdx.z / max_c, blockIdx.z % max_c, blockIdx.y, blockIdx.x, 0,
                             0x7fffffffffffffffLL);
        return;
    }
    /* two-tier scoring: the block index is an index into the work list k_contend left (MoveBuf.work: eight interleaved
     * sub-lists, their lengths in work[0..8), the items from work[16] on); a workgroup past the end of its sub-list leaves on its first (scalar) load */
    if ((blockIdx.x >> 3) >= mb.work[blockIdx.x & 7]) return;
    const unsigned long long e = mb.work[16 + blockIdx.x];
    const int cw = (int)((e >> 12) & 0xfffffu), k = (int)((e >> 4) & 0xffu), seg = (int)(e & 0xfu);
    const int w = LW(cw / mb.capC); /* (the items carry the physical cw) */
    const long long ch = mb.ctl[PS(w)].exact_chunk;
    score_workgroup<CAP>(L, sc, mb, lgf_tab, pz, ablate, w, cw % mb.capC, k, seg, (long long)(e >> 32) * ch, ch);
}

/* k_full_nz: evaluate_likelihood_sparse (KA:4374-4488) over all contacts, exact sums -> out[0..1] (the from-scratch
 * likelihood: set-up, every nuisance step).  Contact-parallel (the reference's own decomposition, KA:4426-4462): a wave
 * takes steps of FULL_UNROLL x 64 consecutive contacts -- coalesced loads of (column, count) and of the row index, one
 * 16-byte gather per endpoint (k_pack_tab; the row's record is the same line for most of a wave), the next step's
 * contacts loaded before this step's terms -- so the work is balanced whatever the row lengths.  The term is
 * score_loop's: tables of the parameter set in LDS (ScoreTables), one fma to quantise; circular contigs, counts >=
 * LDS_LGF, rank distances the LDS table does not cover and parameters outside the one-log domain go through
 * ig_pair_term behind a wave-uniform branch. */
#ifndef FULL_UNROLL
#define FULL_UNROLL 2
#endif
__device__ __noinline__ long long full_q_general(const ig_params p, float mean, int cis, float s, int d, float sti, int len_i, int ob,
                                                 const double* lgf_tab)
{
    const ig_hot h = ig_hot_make(p, ig_tab());
    return ig_quantize(ig_pair_term(p, &h, cis, s, (float)d * mean, sti, (float)len_i * mean, ob, lgfact_dev(ob, lgf_tab), ig_tab()));
}
/* streamed once: non-temporal, so that the contact arrays do not evict the endpoint records from L2 */
__device__ __forceinline__ int2 ld_stream(const int2* p)
{
    const long long v = __builtin_nontemporal_load((const long long*)p);
    return make_int2((int)(unsigned)v, (int)(v >> 32));
}
__device__ __forceinline__ int ld_stream(const int* p) { return __builtin_nontemporal_load(p); }
__global__ void __launch_bounds__(256) k_full_nz(const int* __restrict__ crow, const int2* __restrict__ cc, const int4* __restrict__ rec,
                                                 const int* __restrict__ len, const ScoreConst* __restrict__ sc,
                                                 const double* __restrict__ lgf_tab, long long Z, int pz_n, long long* out)
{
    __shared__ ScoreTables L;
    __shared__ long long red[2][4];
    {
        const float4* src = (const float4*)&sc->tab;
        float4* dst = (float4*)&L;
        for (int i = threadIdx.x; i < (int)(sizeof(ScoreTables) / 16); i += blockDim.x) dst[i] = src[i];
    }
    const ig_params p = sc->par;
    const ig_hot hot = sc->hot;
    const float mean = sc->mean_kb, d_max = hot.d_max;
    const double lv = hot.log2_v_inter, slope = hot.slope, la = hot.log2_amp;
    /* rank distances from `cut` on are not in the LDS table (no table at all: every cis pair the long way) */
    const unsigned cut = pz_n <= 0 ? 0u : (pz_n > LDS_PZ ? (unsigned)LDS_PZ : 0xffffffffu);
    const bool checked = !hot.fast;
    __syncthreads();
    const double* T = L.mt;
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long step = 64 * FULL_UNROLL;
    unsigned long long acc = 0, accl = 0;
    /* three stages in flight: the contacts of step i + 2 are loaded and the endpoint records of step i + 1 gathered while
     * the terms of step i are evaluated (HBM and L2 latencies are several times a step's arithmetic) */
    int2 nv[FULL_UNROLL], gv[FULL_UNROLL];
    int nr[FULL_UNROLL], grow[FULL_UNROLL];
    int4 gri[FULL_UNROLL], grj[FULL_UNROLL];
#pragma unroll
    for (int u = 0; u < FULL_UNROLL; u++) {
        const long long k = min(wave * step + u * 64 + lane, Z - 1);
        gv[u] = ld_stream(cc + k);
        grow[u] = ld_stream(crow + k);
    }
#pragma unroll
    for (int u = 0; u < FULL_UNROLL; u++) {
        const long long k = min((wave + nwaves) * step + u * 64 + lane, Z - 1);
        nv[u] = ld_stream(cc + k);
        nr[u] = ld_stream(crow + k);
    }
#pragma unroll
    for (int u = 0; u < FULL_UNROLL; u++) {
        gri[u] = rec[grow[u]];
        grj[u] = rec[gv[u].x];
    }
    for (long long k0 = wave * step; k0 < Z; k0 += nwaves * step) {
        int2 v[FULL_UNROLL];
        int4 ri[FULL_UNROLL], rj[FULL_UNROLL];
        int row[FULL_UNROLL];
#pragma unroll
        for (int u = 0; u < FULL_UNROLL; u++) {
            v[u] = gv[u];
            row[u] = grow[u];
            ri[u] = gri[u];
            rj[u] = grj[u];
        }
#pragma unroll
        for (int u = 0; u < FULL_UNROLL; u++) {
            gv[u] = nv[u];
            grow[u] = nr[u];
            gri[u] = rec[grow[u]];
            grj[u] = rec[gv[u].x];
        }
#pragma unroll
        for (int u = 0; u < FULL_UNROLL; u++) {
            const long long k = min(k0 + 2 * nwaves * step + u * 64 + lane, Z - 1);
            nv[u] = ld_stream(cc + k);
            nr[u] = ld_stream(crow + k);
        }
        double t[FULL_UNROLL];
        bool rare[FULL_UNROLL], any_rare = false, any_big = false;
        unsigned long long bits[FULL_UNROLL];
#pragma unroll
        for (int u = 0; u < FULL_UNROLL; u++) {
            const unsigned o_b = (unsigned)v[u].y;
            const bool cis = ri[u].z == rj[u].z;
            const unsigned d = abs_diff_u32((unsigned)ri[u].w, (unsigned)rj[u].w);
            const float sv = fabsf(__int_as_float(ri[u].x) - __int_as_float(rj[u].x));
            const bool in = cis && (sv > 0.0f) && (sv < d_max);
            const double pzc = L.pzc[cis ? min(d, (unsigned)LDS_PZ) : (unsigned)LDS_PZ];
            const double lgf = L.lgf[min(o_b, (unsigned)(LDS_LGF - 1))];
            const double y = ig_fma(slope, ig_log2_pos((double)sv, T), la);
            const double yy = in ? __builtin_fmax(y, lv) : lv; /* in: y is a number */
            const double ex = ig_exp2_core(yy, T);
            const double lg = yy * IG_LOG2_10_INV;
            t[u] = (ig_fma((double)o_b, lg, -ex) - lgf) + pzc;
            rare[u] = (o_b - 1u >= (unsigned)(LDS_LGF - 1)) || (cis && (d >= cut || __int_as_float(ri[u].y) != 0.0f)) || checked;
            bits[u] = ig_d2u(ig_fma(t[u], IG_QSCALE, IG_QMAGIC));
            any_rare |= rare[u];
            any_big |= !(__builtin_fabs(t[u]) < 524288.0);
        }
        if (__any(any_big)) {
#pragma unroll
            for (int u = 0; u < FULL_UNROLL; u++)
                if (!(__builtin_fabs(t[u]) < 524288.0)) bits[u] = (unsigned long long)ig_quantize(t[u]) + IG_QMAGIC_BITS;
        }
        if (__any(any_rare)) {
#pragma unroll
            for (int u = 0; u < FULL_UNROLL; u++)
                if (rare[u]) {
                    const int cis = ri[u].z == rj[u].z;
                    const float sv = fabsf(__int_as_float(ri[u].x) - __int_as_float(rj[u].x));
                    const int d = (int)abs_diff_u32((unsigned)ri[u].w, (unsigned)rj[u].w);
                    bits[u] = (unsigned long long)full_q_general(p, mean, cis, sv, d, __int_as_float(ri[u].y), len[row[u]], v[u].y, lgf_tab) +
                              IG_QMAGIC_BITS;
                }
        }
#pragma unroll
        for (int u = 0; u < FULL_UNROLL; u++) {
            if (k0 + u * 64 + lane < Z) {
                acc += bits[u] - IG_QMAGIC_BITS;
                accl += (unsigned)bits[u];
            }
        }
    }
    /* lane sums < 2^63 (|q| < 2^52, far fewer than 2^11 contacts per lane per launch is not guaranteed: fold as limbs) */
    long long hi = ((long long)acc - (long long)accl) >> 32, lo = (long long)accl;
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    if (lane == 0) {
        red[0][threadIdx.x >> 6] = hi;
        red[1][threadIdx.x >> 6] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) { /* one pair of atomics per workgroup (thousands of waves on two addresses serialise) */
        atomic_add_ll(&out[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomic_add_ll(&out[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

/* k_full_nz_tiled: the same sum from a second, TILED copy of the contacts (built once at upload): the sub-fragments are cut
 * into blocks of FULL_TB, tile (bi, bj) holds the contacts with row in block bi and column in block bj, a work item = up to
 * FULL_CHUNK contacts of one tile.  A workgroup stages (dist, rank, contig) of the two blocks' sub-fragments in LDS
 * (2 x 24 KB) and streams its contacts -- 8 bytes each: (row | column << 11) inside the blocks, count -- so the two
 * endpoint gathers per contact that bound k_full_nz (0.33 T random 16-byte gathers/s into L2) become LDS reads.  Same term,
 * same exact integer sums (any order of the contacts gives the same total).
 *
 * Most contacts are never read: a trans pair's term is a function of its count alone, so the sum of a tile between two
 * blocks that share no contig is its (static) histogram of counts times the table of those terms.  k_tile_trans decides
 * that per off-diagonal tile from the blocks' contig signatures (k_pack_tab_sig) and either adds the histogram's sum or
 * puts the tile's work items on a list; k_full_nz_tiled is launched over the STATIC items (diagonal tiles, tiles with a
 * count beyond the histogram) and every workgroup goes on with items from that list until it is empty. */
#define FULL_TB 2048
#ifndef FULL_CHUNK
#define FULL_CHUNK 4096 /* contacts per work item: the granularity of the even split of a pass over its workgroups (tile_runs) */
#endif
#ifndef FULL_TILED_THREADS
#define FULL_TILED_THREADS 1024
#endif
struct TileWork {
    long long off; /* first contact of the item in the tiled array */
    int n, bi, bj;
    int pad; /* items of the same tile behind this one (they follow it in the array, and their contacts its contacts in memory) */
};
struct TileInfo { /* an off-diagonal tile whose counts are all in 1 .. TILE_HB-1 */
    int bi, bj, first_item, n_items;
};
struct TileDyn {
    int count, next, done, pad; /* items on the list (k_tile_trans), items taken, workgroups through (k_full_nz_tiled); zeroed by k_pack_tab_sig */
};
/* The work of a pass = the sequence of items [static ones: work[0 .. n_static)] [the list k_tile_trans left: dyn_list[0 .. count)];
 * workgroup b of G takes positions [b T / G, (b + 1) T / G) of it.  The items of a tile are consecutive in the sequence AND in
 * memory, so a workgroup's share is a handful of RUNS -- contiguous contact ranges of one tile each: it stages a tile's blocks
 * once and streams the run.  (Until round 3: items of 16 384 contacts handed out one at a time to persistent workgroups
 * through an atomic cursor, the blocks staged again for every item: 10 us of staging per workgroup and a launch that ended
 * 25 us after its median workgroup, tools/tile_trace.py.) */
struct TileRun {
    long long off;
    int n, bi, bj;
};
struct TileRuns {
    const TileWork* work;
    const int* dyn_list;
    int n_static, p, pe;
    __device__ __forceinline__ TileRuns(const TileWork* w, const int* dl, int ns, int n_dyn) : work(w), dyn_list(dl), n_static(ns)
    {
        const long long T = (long long)ns + n_dyn;
        p = (int)((long long)blockIdx.x * T / gridDim.x);
        pe = (int)((long long)(blockIdx.x + 1) * T / gridDim.x);
    }
    __device__ __forceinline__ int item(int q) const { return q < n_static ? q : dyn_list[q - n_static]; }
    __device__ __forceinline__ bool next(TileRun& r)
    {
        if (p >= pe) return false;
        /* two dependent loads per run, whatever its length: the first item says how many items of its tile follow it (they sit
         * behind it in `work`, and -- static part or list -- at the positions behind p), the last one where the run ends */
        const int it0 = item(p);
        const TileWork w0 = work[it0];
        int k = min(pe - p, w0.pad + 1);
        if (p < n_static) k = min(k, n_static - p);
        const TileWork wl = work[it0 + k - 1];
        r.off = w0.off;
        r.n = (int)(wl.off - w0.off) + wl.n;
        r.bi = w0.bi;
        r.bj = w0.bj;
        p += k;
        return true;
    }
};

struct FullTiledLds {
    ScoreTables tab;
    unsigned long long qtrans[LDS_LGF]; /* the quantised term of a trans pair with count ob (it depends on nothing else), + the rounding magic */
    /* per sub-fragment of the row / column block: (dist, rank | circular contig << 31) and the contig id -- 12 bytes, so that
     * two workgroups fit the LDS of a CU */
    uint2 rrec[FULL_TB], crec[FULL_TB];
    int rctg[FULL_TB], cctg[FULL_TB];
    long long red[2][FULL_TILED_THREADS / 64];
    int next_item;
};

#define TILE_TRANS_THREADS 1024 /* a tile per wave; few, large workgroups: every workgroup ends in a pair of atomics on the same two words */
__global__ void __launch_bounds__(TILE_TRANS_THREADS) k_tile_trans(const TileInfo* __restrict__ tiles, int n_tiles, const unsigned* __restrict__ sig,
                                                    const unsigned* __restrict__ hist, const ScoreConst* __restrict__ sc, TileDyn* dyn,
                                                    int* __restrict__ dyn_list, int use_hist, long long* partial, int n_trans_blocks, Tables zt,
                                                    const Glob* g, int which, int M, long long* zero_out,
                                                    const ScoreConst* __restrict__ sc0 = nullptr, long long* partial0 = nullptr)
{
    /* sc0 / partial0 (the screened nuisance pass, ig_kernels_nuis.cuh): the same histogram sums under a second parameter set
     * (the model's current one) -> partial0: the difference of the two is the exact change of the all-trans tiles' share */
    const unsigned* fold = sig + (size_t)((M + FULL_TB - 1) / FULL_TB) * SIG_WORDS; /* the folded signatures (k_pack_tab_sig) */
    /* the zero-pixel blocks first: they are the longer ones (dispatched first, they run next to the tiles' instead of behind them) */
    const int n_zero_blocks = (int)gridDim.x - n_trans_blocks;
    if ((int)blockIdx.x < n_zero_blocks) {
        full_zero_block(zt, g, which, M, zero_out, (int)blockIdx.x, n_zero_blocks);
        return;
    }
    const int tb = (int)blockIdx.x - n_zero_blocks; /* this workgroup's number among the tiles' */
    __shared__ long long red[2][TILE_TRANS_THREADS / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = tb * (TILE_TRANS_THREADS / 64) + wv;
    /* everything is requested before anything is looked at: tile, signatures, histogram bin; the table of trans terms by
     * the first wave meanwhile */
    TileInfo ti = {0, 0, 0, 0};
    unsigned both = 0;
    unsigned long long n = 0;
    if (t < n_tiles) {
        ti = tiles[t];
        n = hist[(size_t)t * TILE_HB + lane];
        if (lane < SIG_FOLD) both = fold[(size_t)ti.bi * SIG_FOLD + lane] & fold[(size_t)ti.bj * SIG_FOLD + lane];
        if (__any(both != 0u)) { /* the folded signatures intersect: the full ones decide */
            both = 0;
            for (int i = lane; i < SIG_WORDS; i += 64) both |= sig[(size_t)ti.bi * SIG_WORDS + i] & sig[(size_t)ti.bj * SIG_WORDS + i];
        }
    }
    const bool fast = sc->hot.fast;
    const unsigned long long qt_lane = sc->qtrans[lane]; /* the table of trans terms of this parameter set (k_build_score_const) */
    const unsigned long long q0_lane = sc0 ? sc0->qtrans[lane] : IG_QMAGIC_BITS;
    long long hi = 0, lo = 0, hi0 = 0, lo0 = 0;
    if (t < n_tiles) {
        if (__any(both != 0u) || !use_hist || !fast) { /* a contig in both blocks (or its alias): read the contacts */
            int base = 0;
            if (lane == 0) base = atomicAdd(&dyn->count, ti.n_items);
            base = __shfl(base, 0, 64);
            for (int i = lane; i < ti.n_items; i += 64) dyn_list[base + i] = ti.first_item + i;
        } else if (lane >= 1) {
            const long long q = (long long)(qt_lane - IG_QMAGIC_BITS);
            const unsigned long long ql = (unsigned)q;
            const long long qh = (q - (long long)ql) >> 32;
            const unsigned long long pl = n * ql; /* < 2^63: a tile holds fewer than 2^31 contacts */
            hi = (long long)n * qh + (long long)(pl >> 32);
            lo = (long long)(pl & 0xffffffffull);
            if (sc0) {
                const long long q0 = (long long)(q0_lane - IG_QMAGIC_BITS);
                const unsigned long long ql0 = (unsigned)q0;
                const long long qh0 = (q0 - (long long)ql0) >> 32;
                const unsigned long long pl0 = n * ql0;
                hi0 = (long long)n * qh0 + (long long)(pl0 >> 32);
                lo0 = (long long)(pl0 & 0xffffffffull);
            }
        }
    }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    __shared__ long long red0[2][TILE_TRANS_THREADS / 64];
    if (sc0) {
        hi0 = wave_sum_ll(hi0);
        lo0 = wave_sum_ll(lo0);
    }
    if (lane == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
        red0[0][wv] = hi0;
        red0[1][wv] = lo0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        hi = lo = hi0 = lo0 = 0;
        for (int q = 0; q < TILE_TRANS_THREADS / 64; q++) {
            hi += red[0][q];
            lo += red[1][q];
            hi0 += red0[0][q];
            lo0 += red0[1][q];
        }
        /* one pair of words per workgroup, summed by the first workgroup of k_full_nz_tiled: thousands of workgroups adding to
         * the same two words with atomics took 36 ns each, one after the other (97 us at 43 k tiles) */
        partial[2 * (size_t)tb] = hi;
        partial[2 * (size_t)tb + 1] = lo;
        if (partial0) {
            partial0[2 * (size_t)tb] = hi0;
            partial0[2 * (size_t)tb + 1] = lo0;
        }
    }
}

/* everything a nuisance step's pass needs before its tiles, in ONE launch (k_set_par + k_build_pz + the scratch memset +
 * k_pack_tab_sig + k_build_score_const were five, each a few microseconds of work behind a launch gap): the test parameters
 * come from the host, so nothing here waits for anything else.  Blocks [0, n_pack): the records and signatures of a block
 * of sub-fragments; the blocks behind: P_z table and score constants of the set (same expressions as k_build_pz and
 * k_build_score_const). */
struct DiffConst;
struct ScreenConst;
__device__ __forceinline__ void build_diff_const(int i, const Glob* g, const ig_params pt, float mean_kb, float pzv_t, int pz_n_t,
                                                 const ScoreConst* __restrict__ sc0, int pz_n_c, DiffConst* out, const ScreenConst* __restrict__ scr0);
__global__ void __launch_bounds__(256) k_nuis_prepare(Glob* g, int which, ig_params p, float mean_kb, float* __restrict__ pz, int pz_n,
                                                      const double* __restrict__ lgf_tab, ScoreConst* out, long long* scratch8, Tables t,
                                                      int M, int4* __restrict__ rec, unsigned* __restrict__ sig, int tb, int* dyn2, int n_pack,
                                                      const ScoreConst* __restrict__ sc0 = nullptr, int pz_n0 = 0, DiffConst* dc = nullptr,
                                                      long long* diff8 = nullptr, const ScreenConst* __restrict__ scr0 = nullptr,
                                                      Tables live = Tables{nullptr, nullptr, nullptr, nullptr}, const int* __restrict__ prev_touched = nullptr)
{
    if ((int)blockIdx.x < n_pack) {
        if (live.dist) { /* tab_prev := the state before the move about to be decided (k_catch_up's job, one launch and one stream
                          * event less at the head of every step): each block of sub-fragments takes the touched entries of its own
                          * range before it packs them */
            const int s0 = (int)blockIdx.x * tb, s1 = s0 + tb, n = g->n_prev_touched;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int s = prev_touched[i];
                if (s >= s0 && s < s1) {
                    t.dist[s] = live.dist[s];
                    t.stot[s] = live.stot[s];
                    t.cp[s] = live.cp[s];
                    t.len[s] = live.len[s];
                }
            }
            __syncthreads();
        }
        pack_tab_sig_block(t, M, rec, sig, tb, dyn2);
        return;
    }
    const int i = ((int)blockIdx.x - n_pack) * blockDim.x + threadIdx.x;
    const float s_z = (float)i * mean_kb;
    const float pzv = (i < pz_n && s_z < p.d_max) ? ig_rippe(s_z, p, ig_tab()) : p.v_inter;
    if (i < pz_n) pz[i] = pzv;
    if (i < IG_TAB_SIZE) out->tab.mt[i] = ig_tab()[i];
    if (i < LDS_PZ + 2) out->tab.pzc[i] = (double)(i < min(pz_n, LDS_PZ) ? pzv : p.v_inter) * IG_LOG_E_F;
    if (i < LDS_LGF) out->tab.lgf[i] = lgf_tab[i];
    if (i == 0) {
        out->hot = ig_hot_make(p, ig_tab());
        out->par = p;
        out->mean_kb = mean_kb;
        g->par[which] = p;
        g->mean_kb = mean_kb;
        for (int q = 0; q < 8; q++) scratch8[q] = 0;
        if (diff8)
            for (int q = 0; q < 8; q++) diff8[q] = 0;
    }
    if (i < TILE_HB) {
        const ig_hot h = ig_hot_make(p, ig_tab());
        out->qtrans[i] = i ? trans_term_bits((unsigned)i, h.log2_v_inter, ig_tab(), lgf_tab, (double)p.v_inter * IG_LOG_E_F) : IG_QMAGIC_BITS;
    }
    if (dc) build_diff_const(i, g, p, mean_kb, pzv, pz_n, sc0, pz_n0, dc, scr0); /* the screened pass (ig_kernels_nuis.cuh) */
}

__global__ void __launch_bounds__(FULL_TILED_THREADS, 8) /* 8 waves per SIMD: two workgroups per CU need <= 80 SGPRs (112 admit 6 waves) */
    k_full_nz_tiled(const TileWork* __restrict__ work, const uint2* __restrict__ tc, const int4* __restrict__ rec, const int* __restrict__ len,
                    const ScoreConst* __restrict__ sc, const double* __restrict__ lgf_tab, int M, int pz_n, long long* out, int n_static,
                    TileDyn* dyn, const int* __restrict__ dyn_list, long long* trace, NuisHost* hn, int hn_seq,
                    const long long* __restrict__ partial, int n_partial)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    FullTiledLds& L = *(FullTiledLds*)lds_raw;
    long long t_start = 0;
    if (trace && threadIdx.x == 0) t_start = (long long)wall_clock64(); /* ig_debug_tile_trace */
    {
        const float4* src = (const float4*)&sc->tab;
        float4* dst = (float4*)&L.tab;
        for (int i = threadIdx.x; i < (int)(sizeof(ScoreTables) / 16); i += blockDim.x) dst[i] = src[i];
    }
    const ig_params p = sc->par;
    const ig_hot hot = sc->hot;
    const float mean = sc->mean_kb, d_max = hot.d_max;
    const double lv = hot.log2_v_inter, slope = hot.slope, la = hot.log2_amp;
    const unsigned cut = pz_n <= 0 ? 0u : (pz_n > LDS_PZ ? (unsigned)LDS_PZ : 0xffffffffu);
    const bool checked = !hot.fast;
    __syncthreads();
    const double* T = L.tab.mt;
    if (threadIdx.x < LDS_LGF) L.qtrans[threadIdx.x] = trans_term_bits(threadIdx.x, lv, T, L.tab.lgf, L.tab.pzc[LDS_PZ]);
    unsigned long long acc = 0, accl = 0;
    int n_items = 0, n_contacts = 0;
    const int nth = blockDim.x;
    /* this workgroup's share of the pass: runs of contacts of one tile each (tile_runs) */
    TileRuns runs(work, dyn_list, n_static, dyn->count);
    TileRun wk;
    int st_bi = -1, st_bj = -1;
    while (runs.next(wk)) {
        const bool diag = wk.bi == wk.bj;
        n_items++;
        n_contacts += wk.n;
        /* the first two contacts of a lane are on their way while the blocks are staged */
        const uint2* src = tc + wk.off;
        const int n = wk.n;
        uint2 nx0 = src[min((int)threadIdx.x, n - 1)], nx1 = src[min((int)threadIdx.x + nth, n - 1)];
        if (st_bi != wk.bi || st_bj != wk.bj) {
            __syncthreads(); /* everybody is through with the blocks of the run before */
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
            __syncthreads();
            st_bi = wk.bi;
            st_bj = wk.bj;
        }
        const uint2* cre = diag ? L.rrec : L.crec;
        const int* cct = diag ? L.rctg : L.cctg;
        for (int e0 = threadIdx.x; e0 < n; e0 += 2 * nth) {
            const uint2 v0 = nx0, v1 = nx1;
            nx0 = src[min(e0 + 2 * nth, n - 1)];
            nx1 = src[min(e0 + 3 * nth, n - 1)];
            /* .x dist, .y s_tot != 0 (a circular contig: the long way), .z contig, .w rank */
            auto record = [](uint2 r, int ctg) { return make_int4((int)r.x, (int)(r.y >> 31), ctg, (int)(r.y & 0x7fffffffu)); };
            const int4 ri0 = record(L.rrec[v0.x & (FULL_TB - 1)], L.rctg[v0.x & (FULL_TB - 1)]);
            const int4 rj0 = record(cre[(v0.x >> 11) & (FULL_TB - 1)], cct[(v0.x >> 11) & (FULL_TB - 1)]);
            const int4 ri1 = record(L.rrec[v1.x & (FULL_TB - 1)], L.rctg[v1.x & (FULL_TB - 1)]);
            const int4 rj1 = record(cre[(v1.x >> 11) & (FULL_TB - 1)], cct[(v1.x >> 11) & (FULL_TB - 1)]);
            /* the whole wave on trans pairs with tabulated counts: no arithmetic at all */
            const bool easy = (ri0.z != rj0.z) && (ri1.z != rj1.z) && (v0.y - 1u < (unsigned)(LDS_LGF - 1)) && (v1.y - 1u < (unsigned)(LDS_LGF - 1));
            if (!checked && __all(easy)) {
                const unsigned long long b0 = L.qtrans[v0.y], b1 = L.qtrans[v1.y];
                if (e0 < n) {
                    acc += b0 - IG_QMAGIC_BITS;
                    accl += (unsigned)b0;
                }
                if (e0 + nth < n) {
                    acc += b1 - IG_QMAGIC_BITS;
                    accl += (unsigned)b1;
                }
                continue;
            }
            /* both terms in one straight line (the two dependent chains interleave), ONE branch for whatever needs the long way */
            double t[2];
            unsigned long long bits[2];
            bool rare[2], cisv[2];
            unsigned dv[2];
            float svv[2];
            bool fix = false;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint2 v = u ? v1 : v0;
                const int4 ri = u ? ri1 : ri0, rj = u ? rj1 : rj0;
                const unsigned o_b = v.y;
                const bool cis = ri.z == rj.z;
                const unsigned d = abs_diff_u32((unsigned)ri.w, (unsigned)rj.w);
                const float sv = fabsf(__int_as_float(ri.x) - __int_as_float(rj.x));
                const bool in = cis & (sv > 0.0f) & (sv < d_max);
                const double pzc = L.tab.pzc[cis ? min(d, (unsigned)LDS_PZ) : (unsigned)LDS_PZ];
                const double lgf = L.tab.lgf[min(o_b, (unsigned)(LDS_LGF - 1))];
                const double y = ig_fma(slope, ig_log2_pos((double)sv, T), la);
                const double yy = in ? __builtin_fmax(y, lv) : lv;
                const double ex = ig_exp2_core(yy, T);
                const double lg = yy * IG_LOG2_10_INV;
                t[u] = (ig_fma((double)o_b, lg, -ex) - lgf) + pzc;
                rare[u] = (o_b - 1u >= (unsigned)(LDS_LGF - 1)) | (cis & ((d >= cut) | (ri.y != 0))) | checked;
                bits[u] = ig_d2u(ig_fma(t[u], IG_QSCALE, IG_QMAGIC));
                fix |= rare[u] | !(__builtin_fabs(t[u]) < 524288.0);
                cisv[u] = cis;
                dv[u] = d;
                svv[u] = sv;
            }
            if (__any(fix)) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const uint2 v = u ? v1 : v0;
                    const int4 ri = u ? ri1 : ri0;
                    if (rare[u]) {
                        const int gi = min(wk.bi * FULL_TB + (int)(v.x & (FULL_TB - 1)), M - 1);
                        bits[u] = (unsigned long long)full_q_general(p, mean, cisv[u], svv[u], (int)dv[u], ri.y ? __int_as_float(rec[gi].y) : 0.0f,
                                                                     len[gi], (int)v.y, lgf_tab) +
                                  IG_QMAGIC_BITS;
                    } else if (!(__builtin_fabs(t[u]) < 524288.0)) {
                        bits[u] = (unsigned long long)ig_quantize(t[u]) + IG_QMAGIC_BITS;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (e0 + u * nth < n) {
                    acc += bits[u] - IG_QMAGIC_BITS;
                    accl += (unsigned)bits[u];
                }
        }
    }
    long long hi = ((long long)acc - (long long)accl) >> 32, lo = (long long)accl;
    if (blockIdx.x == 0) /* the histogram sums of the tiles that were not read (k_tile_trans, one pair per workgroup) */
        for (int i = threadIdx.x; i < n_partial; i += blockDim.x) {
            hi += partial[2 * (size_t)i];
            lo += partial[2 * (size_t)i + 1];
        }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
        L.red[0][threadIdx.x >> 6] = hi;
        L.red[1][threadIdx.x >> 6] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long h = 0, l = 0;
        for (int v = 0; v < (int)(blockDim.x >> 6); v++) {
            h += L.red[0][v];
            l += L.red[1][v];
        }
        atomic_add_ll(&out[0], h);
        atomic_add_ll(&out[1], l);
        if (hn) { /* the last workgroup through: all eight sums (out[2..] are k_tile_trans's) straight to the mapped host memory */
            __threadfence();
            if (atomicAdd(&dyn->done, 1) == (int)gridDim.x - 1) {
                __threadfence();
                for (int q = 0; q < 8; q++) hn->sums[q] = __hip_atomic_load(&out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                hn->sums_seq = hn_seq;
            }
        }
        if (trace) { /* start, end (100 MHz clock), XCC_ID << 32 | HW_ID, items << 32 | contacts */
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            trace[4 * (size_t)blockIdx.x] = t_start;
            trace[4 * (size_t)blockIdx.x + 1] = (long long)wall_clock64();
            trace[4 * (size_t)blockIdx.x + 2] = (long long)(((unsigned long long)xcc << 32) | hw);
            trace[4 * (size_t)blockIdx.x + 3] = ((long long)n_items << 32) | (long long)n_contacts;
        }
    }
}

/* k_delta: exact update of the full likelihood when the winner's slice was windowed (KA:565-586 keeps only
 * pairs near A and B): sum over ALL pairs of the contig of (term under the winner - term under the current
 * genome).  Row-parallel with a per-wave compaction queue; two columns (current, winner). */
__global__ void __launch_bounds__(SCORE_THREADS)
    k_delta(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, Tables tab_prev,
            const int* __restrict__ prev_touched, Glob* g, MoveBuf mb, const double* __restrict__ lgf_tab, PzTab pz, int w, int predicted,
            int which, long long* acc2 = nullptr)
{
    /* predicted = 0: the chosen winner of slot w (one-move tail); 1: the predicted winners of slots w + blockIdx.z;
     * 2: the winner of slot w AFTER it was applied, under parameter set `which` (an accepted nuisance step: the maintained
     * sum under the new parameters = their full pass on the state before the move + this delta); the caller passes the
     * tables of the state before the move as `tab` */
    if (predicted == 1) {
        if (mb.pred_list) { /* the positions k_predict listed (at most gridDim.z of them: it listed no more) */
            if ((int)blockIdx.z >= min(mb.pred_list[0], (int)gridDim.z)) return;
            w = mb.pred_list[1 + blockIdx.z];
        } else {
            w += blockIdx.z;
        }
        if (KEPT(w)) return;
    }
    /* tab_prev catches up with the last applied move before k_apply replaces the touched list (quirk Q12) */
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < (predicted ? 0 : g->n_prev_touched);
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
    MoveCtl& mc = mb.ctl[PS(w)];
    if (g->error || (predicted == 0 && g->retry_pool) || (predicted == 1 ? mc.pred < 0 : (predicted == 0 && !mc.ch_windowed))) return;
    const int c = predicted == 1 ? mc.pred_c : mc.ch_c;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    const int kk = blockIdx.y;
    const int k = (kk == 0) ? 0 : (predicted == 1 ? mc.pred_k : mc.ch_k);
    const int M = mb.sM, m_loc = m.m_loc;
    const ig_params p = g->par[which];
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
            /* a predicted winner's delta goes straight into the records (candidate 0 of the slot): the decide step reads it there */
            CandPre& pc = cpre_at(mb, CW(w, 0));
            /* (an accepted nuisance step's delta, predicted == 2: into the caller's accumulators) */
            atomic_add_ll(acc2 ? &acc2[0] : (predicted == 1 ? &pc.pd_hi : &mc.d_hi), kk == 0 ? -hi : hi);
            atomic_add_ll(acc2 ? &acc2[1] : (predicted == 1 ? &pc.pd_lo : &mc.d_lo), kk == 0 ? -lo : lo);
        }
    }
}

/* k_predict: one workgroup per own slot.  The scores of a move depend on its predecessors only through scalars that are
 * added to every candidate alike (up to rounding) and through the stale insert flags of candidate 0, so the winner can be
 * predicted from the batch-start state: if it is a windowed candidate that changes the genome, its exact full-contig delta
 * is computed NOW (k_delta, predicted) and the decide step does not have to pause the batch for it.  A wrong prediction
 * costs nothing but the pause it failed to avoid. */
/* (until round 5 two launches: pass 0 left every slot's winner under the flags of its predecessor's last candidate, pass 1 read the
 * predecessor's.  A block now redoes its predecessor's first pass itself -- one argmax more per block, one launch and one dependent
 * dispatch less per chain) */
__global__ void __launch_bounds__(256) k_predict(Glob* g, MoveBuf mb, int w_begin, int zcap)
{
    __shared__ double sc[IG_MAX_CANDIDATES * IG_N_TMP_STRUCT];
    __shared__ int s_best;
    const int w = w_begin + blockIdx.x, tid = threadIdx.x;
    if (KEPT(w)) return;
    MoveCtl& mc = mb.ctl[PS(w)];
    if (cpre_at(mb, CW(w, 0)).overflow) return;
    const ig_params p = g->par[0];
    const double log_e = IG_LOG_E_F, n_tot_pxl = g->n_tot_pxl;
    const long long z_hi = g->z_hi, z_lo = g->z_lo, n_intra = g->n_intra;
    const double cur_nz = ig_acc_to_double(g->nz_hi, g->nz_lo);
    auto live_flags = [&]() -> unsigned {
        unsigned v = 0;
        for (int q = 0; q < 12; q++) v |= (g->valid_insert[q] != -1) ? (1u << q) : 0u;
        return v;
    };
    auto flags_of = [&](int ws, int sel) -> unsigned {
        const CandMeta& pm = mb.meta[CW(ws, sel)];
        unsigned v = 0;
        for (int q = 0; q < 12; q++) v |= (pm.flags[q] != -1) ? (1u << q) : 0u;
        return v;
    };
    /* the winner of slot ws under the batch-start scalars and the stale insert flags `vmask` (the whole workgroup; -> every thread) */
    auto argmax_of = [&](int ws, unsigned vmask) -> int {
        const MoveCtl& wc = mb.ctl[PS(ws)];
        const int n = min(wc.C, IG_MAX_CANDIDATES) * IG_N_TMP_STRUCT;
        for (int i = tid; i < n; i += blockDim.x) {
            const int c = i / IG_N_TMP_STRUCT, slot = i % IG_N_TMP_STRUCT;
            const SlotPre r = pre_at(mb, CW(ws, c), slot);
            const CandPre& cp = cpre_at(mb, CW(ws, c));
            const bool sup = (c == 0) && wc.superset0 && (slot >= 12);
            double v = 0.0;
            if ((r.k > 0) && !(sup && !((vmask >> (slot - 12)) & 1u))) {
                const int pos = sup ? cp.base_cnt + __popc(vmask & ((1u << (slot - 12)) - 1u)) : r.k - 1;
                const double nzd = (cp.r > 0 && pos >= cp.r) ? r.nz_cut_d : r.nz_d;
                const double val_inter = -1.0 * log_e * (n_tot_pxl - (double)(n_intra + r.dni)) * p.v_inter;
                const double val_intra = ig_acc_to_double(z_hi + r.dz_hi, z_lo + r.dz_lo) * log_e;
                v = nzd + (val_intra + val_inter) + cur_nz - cp.ext_d;
            }
            sc[i] = v;
        }
        __syncthreads();
        if (tid < 64) {
            double bestv = -IG_INF;
            int best = 0x7fffffff;
            for (int i = tid; i < n; i += 64) {
                const double ok = (sc[i] == 0.0) ? -IG_INF : sc[i];
                if (ok > bestv) {
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
            if (tid == 0) s_best = best >= n ? 0 : best;
        }
        __syncthreads();
        const int b = s_best;
        __syncthreads(); /* (sc and s_best are used again) */
        return b;
    };
    /* the flags candidate 0 most likely sees: the live ones for the first slot; else those of the previous slot's last candidate, or of
     * the family the previous slot's own likely winner would leave the flags to (CL:2125-2126: a block-insert winner re-ran get_bounds
     * for its own candidate) -- that winner under the flags ITS predecessor's last candidate leaves */
    unsigned vmask;
    if (w == 0) {
        vmask = live_flags();
    } else {
        const MoveCtl& pc = mb.ctl[PS(w - 1)];
        int sel = max(pc.C - 1, 0);
        const unsigned vprev = (w == 1) ? live_flags() : flags_of(w - 2, max(mb.ctl[PS(w - 2)].C - 1, 0));
        const int bp = argmax_of(w - 1, vprev);
        if ((bp % IG_N_TMP_STRUCT) >= 12 && bp / IG_N_TMP_STRUCT < pc.C) sel = bp / IG_N_TMP_STRUCT;
        vmask = flags_of(w - 1, sel);
    }
    const int best = argmax_of(w, vmask);
    if (tid == 0) {
        const int c = best / IG_N_TMP_STRUCT, slot = best % IG_N_TMP_STRUCT;
        const SlotPre r = pre_at(mb, CW(w, c), slot);
        const bool windowed = (cpre_at(mb, CW(w, c)).same_windowed >> 1) & 1;
        if (windowed && (r.info & 1u) && r.k > 0) {
            /* onto k_delta's list -- the launch behind this one covers zcap positions; one more than that stays unpredicted (the decide
             * step then pauses for it as for any wrong prediction: the one-move tail) */
            const int at = mb.pred_list ? atomicAdd(&mb.pred_list[0], 1) : 0;
            if (at < zcap) {
                if (mb.pred_list) mb.pred_list[1 + at] = w;
                mc.pred = best;
                cpre_at(mb, CW(w, 0)).pred = best; /* travels with the records; pd_hi / pd_lo (zeroed by k_records) are k_delta's */
                mc.pred_c = c;
                mc.pred_k = r.k;
            }
        }
    }
}

/* prefinal_tail (k_tail): one workgroup per (candidate, slot).  Computes, for
 * every column, T[k] = sum of the terms of the LAST r = S_c mod 64 sliced contacts (canonical order = COO
 * order, so "last" = highest rows, found by bisection on the row id).  Quirk Q5 (KA:4362, block 64 CL:200):
 * a column at list position >= r never receives those contacts; which columns that applies to is decided
 * when the uniq list is known (k_scores / k_commit_batch). */
/* the tail walk's LDS (a kernel that hosts the walk next to other workgroups overlays it with theirs: k_screen_tail) */
struct TailLds {
    int t_li[64], t_lj[64], t_ob[64], t_rows[64];
    int sh_n_rows, sh_n_tail, sh_cnt;
    long long sh_red[4];
    int hist[256];
};
__device__ void prefinal_tail(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables tab, Glob* g, MoveBuf mb,
                              const double* __restrict__ lgf_tab, int tail_quirk, PzTab pz, int w_begin, int c, int w_rel, TailLds& TL, int tpart = 0,
                              int tparts = 1)
{
    /* tpart / tparts (round 6): a launch of ONE slot deals a candidate's columns to several workgroups -- each finds the same tail entries
     * (the same words into the slot's cache) and evaluates the columns k = tpart (mod tparts): the walk's six serial column passes per
     * wave, three or four dependent round trips each, were the longest chain of a one-move call */
    const int w = w_begin + w_rel;
    if (KEPT(w) || c >= mb.ctl[PS(w)].C) return;
    const int cw = CW(w, c);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = mb.sM; /* stride; the bisection below runs over the global sub-fragment ids [0, mb.M) */
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
        if (k % tparts != tpart) continue; /* (its own columns only: another part may be through with its own already) */
        qp[Q_TAIL + 2 * k] = 0;
        qp[Q_TAIL + 2 * k + 1] = 0;
    }
    __syncthreads();
    if (mb.stale) { /* scored earlier, re-evaluated now: not if a move committed meanwhile modified one of its contigs (MoveBuf.stale) */
        const int nd = mb.stale[0];
        int hit = 0;
        for (int q = tid; q < nd; q += blockDim.x) hit |= (mb.stale[1 + q] == m.ctgA) | (mb.stale[1 + q] == m.ctgB);
        if (__syncthreads_or(hit)) return;
    }
    const long long Sc = slice_total(part);
    const int r = (int)(Sc % 64);
    if (!(tail_quirk && r > 0)) return;
    int n_tail;
    /* (several parts: a fresh slot, every part walks -- part 0's words for the cache may be on their way while another part looks) */
    const int cached = tparts > 1 ? -1 : mb.tail_n[cw];
    if (cached >= 0) { /* found when this slot was scored before (under other parameters): only the terms are new */
        n_tail = min(cached, 64);
        if (tid < n_tail) {
            TL.t_li[tid] = mb.tail_ent[(size_t)cw * 192 + tid];
            TL.t_lj[tid] = mb.tail_ent[(size_t)cw * 192 + 64 + tid];
            TL.t_ob[tid] = mb.tail_ent[(size_t)cw * 192 + 128 + tid];
        }
        __syncthreads();
    } else {
    /* T = the largest sub-fragment id with (kept contacts in rows of id >= T) >= r.  Radix descent, 8 bits of the id per pass over
     * the window's rows (histogram of the kept contacts by id, suffix sums from the top): 3 passes at M = 150 k where a
     * bisection on the id took 18 -- on windows of thousands of rows this walk is the longest chain of the launch. */
    int lo_t = 0, bits = 0; /* invariant: count(id >= lo_t) >= r > count(id >= lo_t + 2^bits) =: n_above */
    while ((1 << bits) < mb.M) bits++;
    int n_above = 0;
    while (bits > 0) {
        const int sh = max(bits - 8, 0), nb = 1 << (bits - sh);
        for (int b = tid; b < nb; b += blockDim.x) TL.hist[b] = 0;
        __syncthreads();
        for (int ls = tid; ls < m.m_loc; ls += blockDim.x) {
            const int d = subs[ls] - lo_t, n = rowcnt[ls];
            if (n > 0 && d >= 0 && (d >> bits) == 0) atomicAdd(&TL.hist[d >> sh], n);
        }
        __syncthreads();
        /* the highest bin b with n_above + (bins b .. nb-1) >= r: wave 0, four bins per lane, suffix sums across the lanes */
        if (wv == 0) {
            int h[4], mine = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                h[q] = (4 * lane + q < nb) ? TL.hist[4 * lane + q] : 0;
                mine += h[q];
            }
            int suf = mine; /* inclusive suffix sum over lanes >= lane */
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off, 64);
                if (lane + off < 64) suf += o;
            }
            const unsigned long long ok = __ballot(n_above + suf >= r);
            const int top = 63 - __builtin_clzll(ok); /* ok != 0: lane 0's suffix is the whole range, count(id >= lo_t) >= r */
            if (lane == top) {
                int acc = n_above + suf - mine, b = 3; /* contacts above this lane's bins */
                for (; b > 0; b--) {
                    if (acc + h[b] >= r) break;
                    acc += h[b];
                }
                TL.hist[0] = 4 * lane + b; /* the bin, and the contacts above it */
                TL.hist[1] = acc;
            }
        }
        __syncthreads();
        const int b = TL.hist[0];
        n_above = TL.hist[1];
        __syncthreads();
        lo_t += b << sh;
        bits = sh;
    }
    const int T = lo_t;
    if (tid == 0) {
        TL.sh_n_rows = 0;
        TL.sh_n_tail = 0;
        TL.sh_cnt = 0;
    }
    __syncthreads();
    int above = 0; /* kept contacts in rows > T */
    for (int ls = tid; ls < m.m_loc; ls += blockDim.x) {
        const int s = subs[ls];
        if (s >= T && rowcnt[ls] > 0) {
            const int slot = atomicAdd(&TL.sh_n_rows, 1);
            if (slot < 64) TL.t_rows[slot] = ls;
            if (s > T) above += rowcnt[ls];
        }
    }
    above = wave_sum_i(above);
    if (lane == 0 && above) atomicAdd(&TL.sh_cnt, above);
    __syncthreads();
    const int n_rows = min(TL.sh_n_rows, 64);
    const int need_T = r - TL.sh_cnt; /* contacts to take from the END of row T */
    for (int ri = wv; ri < n_rows; ri += 4) {
        const int ls = TL.t_rows[ri];
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
            if (lane == 0 && took) basei = atomicAdd(&TL.sh_n_tail, took);
            basei = __shfl(basei, 0, 64);
            if (keep && rank < remaining && basei + rank < 64) {
                TL.t_li[basei + rank] = ls;
                TL.t_lj[basei + rank] = lj;
                TL.t_ob[basei + rank] = v.y;
            }
            remaining -= took;
        }
    }
    __syncthreads();
    n_tail = min(TL.sh_n_tail, 64);
    if (tid == 0 && n_tail != r) { /* the walk must find exactly r contacts */
        g->error = 5;
        g->dbg[0] = w;
        g->dbg[1] = c;
        g->dbg[2] = r;
        g->dbg[3] = n_tail;
        g->dbg[4] = (int)Sc;
        g->dbg[5] = KEPT(w);
        g->dbg[6] = mb.rot;
        g->dbg[7] = m.m_loc;
    }
    if (tid < n_tail && tpart == 0) {
        mb.tail_ent[(size_t)cw * 192 + tid] = TL.t_li[tid];
        mb.tail_ent[(size_t)cw * 192 + 64 + tid] = TL.t_lj[tid];
        mb.tail_ent[(size_t)cw * 192 + 128 + tid] = TL.t_ob[tid];
    }
    if (tid == 0 && tpart == 0) mb.tail_n[cw] = n_tail;
    }
    for (int k = (tpart == 0 ? tparts : tpart) + tparts * wv; k < ncol; k += 4 * tparts) { /* (tparts = 1: k = 1 + wv, 5 + wv, ...) */
        const uint2* col = mb.coords + (size_t)(cw * NSLOT + k) * M;
        const ColMeta* cm = mb.cmeta + (size_t)(cw * NSLOT + k) * NCODE;
        long long hi = 0, lo = 0;
        if (lane < n_tail) {
            const long long q = eval_q(p, hot, mean, col[TL.t_li[lane]], col[TL.t_lj[lane]], cm, TL.t_ob[lane], lgfact_dev(TL.t_ob[lane], lgf_tab), pz,
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
    __shared__ TailLds T;
    prefinal_tail(rowptr, cc, tab, g, mb, lgf_tab, tail_quirk, pz, w_begin, blockIdx.x, blockIdx.y, T);
}

/* k_records: after k_score_list and k_tail: the slot-major records of the commit step */
__global__ void __launch_bounds__(64) k_records(MoveBuf mb, int w_begin, int contenders_only)
{
    const int c = blockIdx.x, w = w_begin + blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && mb.pred_list) mb.pred_list[0] = 0; /* k_predict, the next launch, lists afresh */
    if (KEPT(w) || c >= mb.ctl[PS(w)].C) return;
    const int cw = CW(w, c);
    const CandMeta& m = mb.meta[cw];
    long long* qp = mb.qpart + (size_t)cw * Q_STRIDE;
    const int t = threadIdx.x;
    {
        const long long* part = mb.part + (size_t)cw * P_STRIDE;
        if (t <= m.n_uniq) { /* two-tier scoring: a column whose genome is the current one was not scored: its sums are column 0's */
            const int src = (contenders_only && ((mb.ident[cw] >> t) & 1u)) ? 0 : t;
            qp[Q_NZFULL + 2 * t] = part[P_NZ + 2 * src];
            qp[Q_NZFULL + 2 * t + 1] = part[P_NZ + 2 * src + 1];
        }
    }
    __syncthreads();
    if (t < IG_N_TMP_STRUCT) {
        SlotPre r;
        int k = m.kidx[t];
        /* two-tier scoring: a column that cannot win (k_contend) was not scored exactly: the decide step must not see it */
        if (contenders_only && k > 0 && !((mb.cont[cw] >> k) & 1u)) k = -1;
        r.k = k > 0 ? k : 0;
        r.nz_hi = r.nz_lo = r.dz_hi = r.dz_lo = r.dni = 0;
        r.nz_d = r.nz_cut_d = 0.0;
        r.info = 0;
        if (k > 0) {
            r.nz_hi = qp[Q_NZFULL + 2 * k];
            r.nz_lo = qp[Q_NZFULL + 2 * k + 1];
            r.dz_hi = qp[Q_Z + 2 * k] - qp[Q_Z];
            r.dz_lo = qp[Q_Z + 2 * k + 1] - qp[Q_Z + 1];
            r.dni = qp[Q_NI + k] - qp[Q_NI];
            r.nz_d = ig_acc_to_double(r.nz_hi, r.nz_lo);
            r.nz_cut_d = ig_acc_to_double(r.nz_hi - qp[Q_TAIL + 2 * k], r.nz_lo - qp[Q_TAIL + 2 * k + 1]);
            const int2 si = mb.sinfo[cw * NSLOT + t];
            r.info = (si.x ? 1u : 0u) | ((unsigned)si.y << 1);
        }
        pre_at(mb, cw, t) = r;
    }
    if (t == 0) {
        CandPre cp;
        cp.ext_hi = qp[Q_NZFULL];
        cp.ext_lo = qp[Q_NZFULL + 1];
        cp.ext_d = ig_acc_to_double(cp.ext_hi, cp.ext_lo);
        cp.n_slice = slice_total(mb.part + (size_t)cw * P_STRIDE);
        cp.r = (int)(cp.n_slice % 64);
        int nb = 0;
        for (int q = 0; q < m.n_uniq; q++) nb += (m.uniq[q] < 12);
        cp.base_cnt = nb;
        cp.ctgA = m.ctgA;
        cp.ctgB = m.ctgB;
        cp.m_loc = m.m_loc;
        cp.n_loc = m.n_loc;
        cp.n_uniq = m.n_uniq;
        cp.B = m.B;
        cp.same_windowed = (m.same ? 1 : 0) | (m.windowed ? 2 : 0);
        unsigned fm = 0;
        for (int q = 0; q < 12; q++) fm |= (m.flags[q] != -1) ? (1u << q) : 0u;
        cp.flag_mask = fm;
        cp.overflow = mb.ctl[PS(w)].overflow; /* travels with the records: the slot must be re-run */
        cp.pred = -1; /* k_predict, k_delta */
        cp.pd_hi = cp.pd_lo = 0;
        cpre_at(mb, cw) = cp;
    }
}
