/* ig_kernels_setup.cuh -- set-up and whole-genome kernels: tables, from-scratch likelihood (KA:4374-4488,
 * 3850-3917), explode, genome distance (CL:665-716). */
#pragma once

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

#define FULL_UNROLL 4
/* evaluate_likelihood_sparse (KA:4374-4488) over the whole CSR, exact sums -> out[0..1].  One wave per CSR row; the
 * log2/exp2 tables of the contract in LDS, the zero-pixel companion P_z from the table of parameter set `which`. */
__global__ void __launch_bounds__(256) k_full_nz(const long long* __restrict__ rowptr, const int2* __restrict__ cc, Tables t, const Glob* g,
                                                 int which, const double* __restrict__ lgf_tab, int M, int rank, int world,
                                                 long long* out, PzTab pz)
{
    __shared__ double mt_s[IG_TAB_SIZE];
    {
        const double* T0 = ig_tab();
        for (int i = threadIdx.x; i < IG_TAB_SIZE; i += blockDim.x) mt_s[i] = T0[i];
    }
    __syncthreads();
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
        /* FULL_UNROLL x 64 contacts of the row in flight: coalesced 8-byte loads, then both gathers of every contact */
        for (long long k0 = b; k0 < e; k0 += 64 * FULL_UNROLL) {
            int2 v[FULL_UNROLL], cpj[FULL_UNROLL];
            float dj[FULL_UNROLL];
#pragma unroll
            for (int u = 0; u < FULL_UNROLL; u++) {
                const long long k = k0 + u * 64 + lane;
                v[u] = (k < e) ? cc[k] : make_int2(-1, 0);
            }
#pragma unroll
            for (int u = 0; u < FULL_UNROLL; u++) {
                const int j = v[u].x >= 0 ? v[u].x : i;
                cpj[u] = t.cp[j];
                dj[u] = t.dist[j];
            }
#pragma unroll
            for (int u = 0; u < FULL_UNROLL; u++) {
                if (v[u].x < 0) continue;
                const int ob = v[u].y;
                const float s = fabsf(di - dj[u]);
                const int dp = pi - cpj[u].y;
                const int d = dp < 0 ? -dp : dp;
                const double lgf = lgfact_dev(ob, lgf_tab);
                double term;
                if (ci == cpj[u].x && sti == 0 && hot.fast && ob > 0) /* the hot case of ig_pair_term with P_z from its table */
                    term = ig_term_hot(s, 0, ob, lgf, pz_lookup(pz, p, mean, d), &hot, mt_s);
                else
                    term = ig_pair_term(p, &hot, ci == cpj[u].x, s, (float)d * mean, sti, (float)li * mean, ob, lgf, mt_s);
                const long long q = ig_quantize(term);
                hi += q >> 32;
                lo += (long long)(unsigned int)q;
            }
        }
    }
    /* one pair of atomics per workgroup (thousands of waves on two addresses serialise) */
    __shared__ long long red[2][4];
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    if (lane == 0) {
        red[0][threadIdx.x >> 6] = hi;
        red[1][threadIdx.x >> 6] = lo;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomic_add_ll(&out[0], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomic_add_ll(&out[1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
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
