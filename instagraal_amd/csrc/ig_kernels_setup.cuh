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

/* (dist, s_tot, contig, rank) of every sub-fragment in one 16-byte record: one gather per contact endpoint in k_full_nz */
__global__ void k_pack_tab(Tables t, int M, int4* __restrict__ rec)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= M) return;
    const int2 cp = t.cp[s];
    rec[s] = make_int4(__float_as_int(t.dist[s]), __float_as_int(t.stot[s]), cp.x, cp.y);
}

/* the same, one workgroup per block of FULL_TB sub-fragments (the tiles of k_full_nz_tiled), which also leaves the block's
 * SIGNATURE: bit (contig id mod 32 SIG_WORDS) of every contig with a sub-fragment in the block.  Two blocks whose signatures
 * do not intersect share no contig: every contact between them is a trans pair. */
#define SIG_WORDS 256
#define SIG_FOLD 32
__device__ __forceinline__ void pack_tab_sig_block(const Tables& t, int M, int4* __restrict__ rec, unsigned* __restrict__ sig, int tb, int* dyn2)
{
    const int gridDim_blocks = (M + tb - 1) / tb; /* blocks of sub-fragments (k_nuis_prepare's grid holds other blocks too) */
    __shared__ unsigned lsig[SIG_WORDS];
    if (blockIdx.x == 0 && threadIdx.x == 0) dyn2[0] = dyn2[1] = dyn2[2] = 0; /* the list of tiles to read (k_tile_trans), its cursor, the workgroups through */
    for (int i = threadIdx.x; i < SIG_WORDS; i += blockDim.x) lsig[i] = 0;
    __syncthreads();
    const int s0 = blockIdx.x * tb;
    for (int i = threadIdx.x; i < tb && s0 + i < M; i += blockDim.x) {
        const int s = s0 + i;
        const int2 cp = t.cp[s];
        rec[s] = make_int4(__float_as_int(t.dist[s]), __float_as_int(t.stot[s]), cp.x, cp.y);
        const unsigned h = (unsigned)cp.x & (32u * SIG_WORDS - 1u);
        atomicOr(&lsig[h >> 5], 1u << (h & 31u));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SIG_WORDS; i += blockDim.x) sig[(size_t)blockIdx.x * SIG_WORDS + i] = lsig[i];
    /* behind the signatures of all blocks: the same folded to SIG_FOLD words (bit h mod 32 SIG_FOLD) -- two blocks whose folded
     * signatures do not intersect do not intersect at all, and most pairs are told apart by these 128 bytes */
    if (threadIdx.x < SIG_FOLD) {
        unsigned f = 0;
        for (int j = threadIdx.x; j < SIG_WORDS; j += SIG_FOLD) f |= lsig[j];
        sig[(size_t)gridDim_blocks * SIG_WORDS + (size_t)blockIdx.x * SIG_FOLD + threadIdx.x] = f;
    }
}
__global__ void __launch_bounds__(256) k_pack_tab_sig(Tables t, int M, int4* __restrict__ rec, unsigned* __restrict__ sig, int tb, int* dyn2)
{
    pack_tab_sig_block(t, M, rec, sig, tb, dyn2);
}

/* eval_likelihood_on_zero (KA:3850-3917) over all sub-fragments -> out[0..2] = hi, lo, n_intra */
/* (the parameter set and the scalars as arguments: the chain's segments evaluate sets that never become Glob.par[1]) */
/* pz / pz_n: the set's P_z table where the caller has it (pz[pos] is the direct evaluation's value: k_build_pz / k_nuis_prepare build it
 * with the same functions -- a lookup instead of a power function per sub-fragment); nullptr: evaluated */
__device__ __forceinline__ void full_zero_block_p(const Tables& t, const ig_params p, const float mean, const double n_tot_pxl, int M, long long* out,
                                                  int block, int n_blocks, const float* __restrict__ pz = nullptr, int pz_n = 0)
{
    long long hi = 0, lo = 0, ni = 0;
    for (int s = block * blockDim.x + threadIdx.x; s < M; s += n_blocks * blockDim.x) {
        const int pos = t.cp[s].y, len = t.len[s];
        if (pos == 0) ni += ((long long)len * (long long)(len - 1)) / 2;
        if (pos > 0) {
            const long long q = zero_q(p, pos, len, t.stot[s], mean, pz, pz_n);
            hi += q >> 32;
            lo += (long long)(unsigned int)q;
        }
    }
    hi = wave_sum_ll(hi);
    lo = wave_sum_ll(lo);
    ni = wave_sum_ll(ni);
    /* one triple of atomics per workgroup (a thousand waves on three addresses take longer than the sums) */
    __shared__ long long red[3][16];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = hi;
        red[1][wv] = lo;
        red[2][wv] = ni;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        long long v = 0;
        for (int q = 0; q < (int)(blockDim.x >> 6); q++) v += red[threadIdx.x][q];
        if (v) atomic_add_ll(&out[threadIdx.x], v);
    }
    /* what the host needs next to the sums to form the zero-pixel likelihood: one copy back instead of two */
    if (block == 0 && threadIdx.x == 0) { /* (atomic stores: a launch whose last workgroup reads them needs no fence for these two) */
        __hip_atomic_store(&out[3], __double_as_longlong(n_tot_pxl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[5], (long long)__float_as_int(p.v_inter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void full_zero_block(const Tables& t, const Glob* g, int which, int M, long long* out, int block, int n_blocks,
                                                const float* __restrict__ pz = nullptr, int pz_n = 0)
{
    full_zero_block_p(t, g->par[which], g->mean_kb, g->n_tot_pxl, M, out, block, n_blocks, pz, pz_n);
}
__global__ void k_full_zero(Tables t, const Glob* g, int which, int M, long long* out, const float* __restrict__ pz = nullptr, int pz_n = 0)
{
    full_zero_block(t, g, which, M, out, blockIdx.x, gridDim.x, pz, pz_n);
}

__global__ void k_count_heads(State st, int N, int* out, Glob* g)
{
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    int h = (f < N && st.pos[f] == 0) ? 1 : 0;
    int mL = (f < N) ? st.L[f] : 0, mS = (f < N) ? st.SL[f] : 0; /* longest contig, in fragments and in sub-fragments */
    h = wave_sum_i(h);
    for (int o = 32; o > 0; o >>= 1) {
        mL = max(mL, __shfl_down(mL, o, 64));
        mS = max(mS, __shfl_down(mS, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (h) atomicAdd(out, h);
        atomicMax(&g->max_L, mL);
        atomicMax(&g->max_SL, mS);
    }
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
