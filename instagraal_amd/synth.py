"""Seeded synthetic Hi-C problems at sampler level.

Emits exactly the objects the reference's ``sampler`` constructor consumes
(``/root/reference/src/instagraal/simu_single.py:120-153``): the level-L
fragment SoA (``pyramid_sparse.py:1836-1849`` conventions: contig ids start
at 1, ``prev``/``next`` = -1 at contig ends), the sub-fragment table of
``simu_single.py:674-723``, the level-(L-1) contact matrix (upper triangular
scipy CSR) and its 3->1 aggregation at level L.

Deviation from the written plan, stated once: "80 % cis" is infeasible at the
headline shape (50 k bins in ~1000 contigs hold only ~22 M distinct cis
sub-fragment pairs, fewer than 0.8 x 50 M), so the cis share is capped at
``cis_fill`` (default 0.85) of the cis pairs that exist; the remainder of the
``Z`` contacts is uniform trans.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import scipy.sparse as sp

DEFAULT_SEED = 20260529

FLOAT4 = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("w", np.float32)], align=True)
INT4 = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32), ("w", np.int32)], align=True)
INT2 = np.dtype([("x", np.int32), ("y", np.int32)], align=True)
FLOAT3 = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32)], align=True)
INT3 = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32)], align=True)


@dataclasses.dataclass
class SynthProblem:
    n_frags: int
    n_sub_frags: int
    n_contacts: int
    S_o_A_frags: dict
    S_o_A_sub_frags: dict
    np_sub_frags_2_frags: np.ndarray  # (M,) FLOAT4: parent bin, watson kb, crick kb, index in bin
    np_sub_frags_id: np.ndarray  # (N,) INT4: ids of the <=3 sub-frags, count in .w
    np_sub_frags_len_bp: np.ndarray
    np_sub_frags_accu: np.ndarray
    sub_csr: sp.csr_matrix  # level L-1, strict upper triangle, int32
    level_csr: sp.csr_matrix  # level L, upper triangle (diagonal dropped), int32
    coo_row: np.ndarray  # the same contacts as sorted COO (what CL:592-615 uploads)
    coo_col: np.ndarray
    coo_cnt: np.ndarray
    mean_value_trans: float
    params: dict  # fixed P(s) parameters used for timing
    mean_subfrag_kb: float
    seed: int

    def sampler_kwargs(self):
        """Positional arguments of the reference sampler constructor (CL:92-125), in order."""
        N, M = self.n_frags, self.n_sub_frags
        ident = np.arange(N, dtype=np.int32)
        disp = np.zeros(N, dtype=INT2)
        disp["x"] = ident
        disp["y"] = ident + 1
        sident = np.arange(M, dtype=np.int32)
        sdisp = np.zeros(M, dtype=INT2)
        sdisp["x"] = sident
        sdisp["y"] = sident + 1
        rep_sub = self.np_sub_frags_id.copy()
        rng = np.random.RandomState(self.seed & 0x7FFFFFFF)
        pos = np.zeros((N, 4), dtype=np.float32)
        vel = np.zeros((N, 4), dtype=np.float32)
        pos[:, 0] = rng.rand(N)
        return dict(
            use_rippe=True,
            S_o_A_frags=self.S_o_A_frags,
            collector_id_repeats=ident,
            frag_dispatcher=disp,
            id_frag_duplicated=[],
            id_frags_blacklisted=[],
            n_frags=N,
            n_new_frags=N,
            init_n_sub_frags=M,
            n_new_sub_frags=M,
            np_rep_sub_frags_id=rep_sub,
            sub_sampled_sparse_matrix=self.level_csr,
            np_sub_frags_len_bp=self.np_sub_frags_len_bp,
            np_sub_frags_id=self.np_sub_frags_id,
            np_sub_frags_accu=self.np_sub_frags_accu,
            np_sub_frags_2_frags=self.np_sub_frags_2_frags,
            mean_squared_frags_per_bin=np.float32(1.0),
            norm_vect_accu=np.full((1, N), 3),
            sub_candidates_dup=[],
            sub_candidates_output_data=[],
            S_o_A_sub_frags=self.S_o_A_sub_frags,
            sub_collector_id_repeats=sident,
            sub_frag_dispatcher=sdisp,
            sparse_matrix=self.sub_csr,
            mean_value_trans=self.mean_value_trans,
            n_iterations=1,
            is_simu=False,
            vel=vel,
            pos=pos,
        )


def _contig_sizes(rng, n_frags, mean_len):
    k = max(1, int(round(n_frags / mean_len)))
    sizes = rng.geometric(1.0 / mean_len, size=k).astype(np.int64)
    # fix the total to exactly n_frags, keeping every contig >= 1
    diff = int(n_frags - sizes.sum())
    while diff != 0:
        idx = rng.integers(0, k, size=abs(diff))
        if diff > 0:
            np.add.at(sizes, idx, 1)
        else:
            ok = idx[sizes[idx] > 1]
            np.subtract.at(sizes, ok, 1)
            sizes = np.maximum(sizes, 1)
        diff = int(n_frags - sizes.sum())
    return sizes


def rippe_params(mean_subfrag_kb, v_inter=5e-3, p_at_mean=20.0):
    """Fixed P(s) parameters of SURVEY 8(d): kuhn 50, lm 9.6, slope -1.5, d 2, P(1.8 kb)=20."""
    from scipy.optimize import brentq

    kuhn, lm, slope, d = 50.0, 9.6, -1.5, 2.0
    c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))

    def raw(s):
        return 0.53 * kuhn**-3.0 * np.power(lm * s / kuhn, slope)

    fact = p_at_mean / raw(1.8)
    d_max = brentq(lambda s: fact * raw(s) - v_inter, 1.0, 1e9)
    return dict(kuhn=kuhn, lm=lm, c1=float(c1), slope=slope, d=d, d_max=float(d_max), fact=float(fact), v_inter=v_inter)


def settled_params(params):
    """P(s) parameters as a nuisance chain has them once it has settled on the synthetic data (tools/soak.py at cfg3 after two whole
    nuisance cycles, profiles/r03y_soak_cfg3.txt: slope -0.53, fact 7.8e5, d_max 2.9e6 kb, trans level 2.9e-3) -- where a default run
    spends 95 of its 100 cycles.  The P_z table is longer than its staged copy there and 4 x as many columns reach the exact tier."""
    kuhn, lm, slope = float(params["kuhn"]), float(params["lm"]), -0.5303
    c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
    return dict(params, slope=slope, c1=float(c1), fact=7.76e5, d_max=2.93e6, v_inter=2.936e-3)


def make_problem(n_frags, n_contacts, seed=DEFAULT_SEED, mean_contig_len=50, cis_frac=0.8, cis_fill=0.85,
                 max_cis_kb=2000.0, with_level_csr=True) -> SynthProblem:
    rng = np.random.default_rng(seed)
    N = int(n_frags)
    sizes = _contig_sizes(rng, N, mean_contig_len)
    K = sizes.size
    contig_of_frag = np.repeat(np.arange(1, K + 1, dtype=np.int64), sizes)
    contig_start = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    pos = (np.arange(N) - np.repeat(contig_start, sizes)).astype(np.int64)
    last_in_contig = pos == (np.repeat(sizes, sizes) - 1)
    sub_len = np.full(N, 3, dtype=np.int64)
    sub_len[last_in_contig] = rng.integers(1, 4, size=int(last_in_contig.sum()))
    M = int(sub_len.sum())
    sub_first = np.concatenate([[0], np.cumsum(sub_len)[:-1]])
    sub2frag = np.repeat(np.arange(N, dtype=np.int64), sub_len)
    sub_w = (np.arange(M) - np.repeat(sub_first, sub_len)).astype(np.int64)
    sub_len_bp = np.clip(np.round(rng.lognormal(np.log(1800.0), 0.6, size=M)), 50, 20000).astype(np.int64)
    len_bp = np.bincount(sub2frag, weights=sub_len_bp, minlength=N).astype(np.int64)
    # per-contig cumulative coordinates
    cum_bp = np.cumsum(len_bp) - len_bp
    start_bp = cum_bp - np.repeat(cum_bp[contig_start], sizes)
    l_cont_bp = np.repeat(np.bincount(contig_of_frag - 1, weights=len_bp, minlength=K).astype(np.int64), sizes)
    cum_sub = np.cumsum(sub_len) - sub_len
    sub_pos = cum_sub - np.repeat(cum_sub[contig_start], sizes)
    sub_l_cont = np.repeat(np.bincount(contig_of_frag - 1, weights=sub_len, minlength=K).astype(np.int64), sizes)
    ident = np.arange(N, dtype=np.int64)
    prev = np.where(pos == 0, -1, ident - 1)
    nxt = np.where(last_in_contig, -1, ident + 1)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    soa = dict(pos=i32(pos), sub_pos=i32(sub_pos), id_c=i32(contig_of_frag), start_bp=i32(start_bp), len_bp=i32(len_bp),
               sub_len=i32(sub_len), circ=i32(np.zeros(N)), id=i32(ident), prev=i32(prev), next=i32(nxt),
               l_cont=i32(np.repeat(sizes, sizes)), sub_l_cont=i32(sub_l_cont), l_cont_bp=i32(l_cont_bp),
               n_accu=i32(np.full(N, 3)), rep=i32(np.zeros(N)), activ=i32(np.ones(N)), id_d=i32(ident))

    # sub-fragment table, arithmetic of simu_single.py:696-707 (float32 throughout)
    v_len = sub_len_bp.astype(np.float32) / np.float32(1000.0)
    csum = np.cumsum(v_len.astype(np.float64))  # only used to locate; the table itself is float32 sums below
    tab = np.zeros(M, dtype=FLOAT4)
    tab["x"] = sub2frag.astype(np.float32)
    tab["w"] = sub_w.astype(np.float32)
    wd = np.zeros(M, dtype=np.float32)
    cd = np.zeros(M, dtype=np.float32)
    # bins hold <= 3 sub-frags: write the float32 sums explicitly so they equal np.sum on float32 slices
    l0 = v_len
    for n_sub in (1, 2, 3):
        first = sub_first[sub_len == n_sub]
        if first.size == 0:
            continue
        ls = [l0[first + j] for j in range(n_sub)]
        half = np.float32(2.0)
        for j in range(n_sub):
            before = np.float32(0.0) * ls[0]
            for t in range(0, j):
                before = (before + ls[t]).astype(np.float32)
            after = np.float32(0.0) * ls[0]
            for t in range(n_sub - 1, j, -1):
                after = (after + ls[t]).astype(np.float32)
            wd[first + j] = (before + ls[j] / half).astype(np.float32)
            cd[first + j] = (after + ls[j] / half).astype(np.float32)
    tab["y"] = wd
    tab["z"] = cd
    del csum
    sub_id = np.zeros(N, dtype=INT4)
    sub_id["x"] = sub_first
    sub_id["y"] = np.where(sub_len > 1, sub_first + 1, 0)
    sub_id["z"] = np.where(sub_len > 2, sub_first + 2, 0)
    sub_id["w"] = sub_len
    sub_len_tab = np.zeros(N, dtype=FLOAT3)
    sub_len_tab["x"] = v_len[sub_first]
    sub_len_tab["y"] = np.where(sub_len > 1, v_len[np.minimum(sub_first + 1, M - 1)], 0)
    sub_len_tab["z"] = np.where(sub_len > 2, v_len[np.minimum(sub_first + 2, M - 1)], 0)
    sub_accu = np.zeros(N, dtype=INT3)
    sub_accu["x"] = 1
    sub_accu["y"] = (sub_len > 1).astype(np.int32)
    sub_accu["z"] = (sub_len > 2).astype(np.int32)

    # centre coordinate (kb) of each sub-frag in its contig, for the contact model only
    sub_contig = contig_of_frag[sub2frag]
    sub_cum = np.cumsum(sub_len_bp) - sub_len_bp
    first_sub_of_contig = sub_first[contig_start]
    n_sub_of_contig = np.bincount(sub_contig - 1, minlength=K)
    centre_kb = (sub_cum - np.repeat(sub_cum[first_sub_of_contig], n_sub_of_contig) + sub_len_bp / 2.0) / 1000.0

    # ---- cis candidates: all (i, i+k) inside one contig within max_cis_kb
    Z = int(n_contacts)
    max_k = int(n_sub_of_contig.max()) - 1
    ci, cj = [], []
    for k in range(1, max_k + 1):
        i = np.nonzero(sub_contig[:-k] == sub_contig[k:])[0]
        if i.size == 0:
            break  # (no pair k apart inside a contig: none further apart either)
        s = centre_kb[i + k] - centre_kb[i]
        i = i[s <= max_cis_kb]
        if i.size == 0:
            break  # (s(i, i + k) grows with k: the same lists as without the break, minus thousands of empty passes on long contigs)
        ci.append(i)
        cj.append(i + k)
    ci = np.concatenate(ci) if ci else np.zeros(0, np.int64)
    cj = np.concatenate(cj) if cj else np.zeros(0, np.int64)
    n_avail = ci.size
    n_cis = int(min(cis_frac * Z, cis_fill * n_avail))
    s_all = (centre_kb[cj] - centre_kb[ci]).astype(np.float64)
    w = np.power(np.maximum(s_all, 0.05), -1.5)
    # weighted sampling without replacement (Efraimidis-Spirakis keys), exact count
    keys = np.log(rng.random(n_avail)) / w
    if n_cis < n_avail:
        sel = np.argpartition(keys, n_avail - n_cis)[n_avail - n_cis:]
    else:
        sel = np.arange(n_avail)
    ci, cj, s_cis = ci[sel], cj[sel], s_all[sel]
    lam = np.minimum(20.0 * np.power(np.maximum(s_cis, 0.05) / 1.8, -1.0), 200.0)
    cnt_cis = 1 + rng.poisson(lam)
    del keys, w, s_all

    # ---- trans: uniform distinct pairs in different contigs
    n_trans = Z - n_cis
    n_trans_avail = (M * (M - 1) - int((n_sub_of_contig.astype(np.int64) * (n_sub_of_contig - 1)).sum())) // 2  # pairs across contigs
    if n_trans > 0.6 * n_trans_avail:  # (the rejection loop below would crawl, or never end)
        raise ValueError("make_problem: %d contacts asked for, %d cis pairs and %d trans pairs exist (at most 60 %% of the trans pairs are drawn)"
                         % (Z, n_avail, n_trans_avail))
    tkeys = np.zeros(0, dtype=np.int64)
    while tkeys.size < n_trans:
        need = int((n_trans - tkeys.size) * 1.1) + 1024
        a = rng.integers(0, M, size=need)
        b = rng.integers(0, M, size=need)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        ok = sub_contig[lo] != sub_contig[hi]
        tkeys = np.unique(np.concatenate([tkeys, lo[ok] * M + hi[ok]]))
    if tkeys.size > n_trans:
        tkeys = tkeys[np.sort(rng.choice(tkeys.size, size=n_trans, replace=False))]
    cnt_trans = 1 + rng.poisson(0.05, size=n_trans)

    allk = np.concatenate([ci * M + cj, tkeys])
    allc = np.concatenate([cnt_cis, cnt_trans]).astype(np.int32)
    order = np.argsort(allk, kind="stable")
    allk, allc = allk[order], allc[order]
    row = (allk // M).astype(np.int32)
    col = (allk % M).astype(np.int32)
    assert row.size == Z and np.all(row < col)
    sub_csr = sp.csr_matrix((allc, (row, col)), shape=(M, M), dtype=np.int32)
    sub_csr.sort_indices()

    level_csr = None
    if with_level_csr:
        fi, fj = sub2frag[row], sub2frag[col]
        het = fi != fj
        level_csr = sp.coo_matrix((allc[het], (fi[het], fj[het])), shape=(N, N), dtype=np.int32).tocsr()
        level_csr.sort_indices()

    # mean trans value as pyramid_sparse.py:1880-1896 computes it at level L-1
    tot_trans = float(cnt_trans.sum())
    n_tot = M * (M - 1) / 2 - float((n_sub_of_contig * (n_sub_of_contig - 1) / 2).sum())
    mean_value_trans = tot_trans / np.float32(n_tot) if n_tot > 0 else 1e-3

    sub_soa = dict(len_bp=i32(sub_len_bp), id=i32(np.arange(M)), id_c=i32(sub_contig))
    mean_kb = float(sub_len_bp.mean() / 1000.0)
    return SynthProblem(
        n_frags=N, n_sub_frags=M, n_contacts=Z, S_o_A_frags=soa, S_o_A_sub_frags=sub_soa,
        np_sub_frags_2_frags=tab, np_sub_frags_id=sub_id, np_sub_frags_len_bp=sub_len_tab, np_sub_frags_accu=sub_accu,
        sub_csr=sub_csr, level_csr=level_csr, coo_row=row, coo_col=col, coo_cnt=allc,
        mean_value_trans=float(mean_value_trans), params=rippe_params(mean_kb), mean_subfrag_kb=mean_kb, seed=int(seed),
    )


CONFIGS = {
    # name: (n_frags, n_contacts) -- BASELINE.json configs[1..4]
    "tiny": (300, 20_000),
    "small": (1_000, 150_000),
    "cfg2": (5_000, 2_000_000),
    "cfg3": (50_000, 50_000_000),
    "cfg5": (200_000, 500_000_000),
    # few long contigs: windows of 1 000 .. 7 000 sub-fragments (the 32 KB LDS stage and the unstaged path of k_score_list)
    "bigctg": (4_000, 400_000, DEFAULT_SEED, 1_000),
    # the headline's size where an assembly ENDS: the reference merges contigs (paste_contigs KA:3367-3693, insert_block KA:2724-2975; its own
    # GPU test asserts 15 - 45 contigs at the end, tests/test_instagraal_gpu.py:126-340) -- 50 k bins / 50 M contacts in ~20 contigs of
    # ~2 500 bins (7 500 sub-fragments; windows of two contigs: 5 000 - 30 000 sub-fragments, far beyond every LDS stage)
    "cfg3_late": (50_000, 50_000_000, DEFAULT_SEED, 2_500),
}


def write_text_dataset(folder, n_contigs=10, mean_frags=110, seed=7, contacts_per_frag=40):
    """A small synthetic instaGRAAL INPUT folder (the three text files `run_instagraal` reads, instagraal.py:502-517,
    pyramid_sparse.py:198-200, plus the FASTA): restriction fragments of log-normal size on `n_contigs` contigs, a few
    fragments shorter than 50 bp and a few without any contact (both are removed by the builder's filter), P(s)-like
    cis contacts and uniform trans contacts.  Formats: fragments_list.txt `id chrom start_pos end_pos size gc_content`
    (id 1-based per contig), info_contigs.txt `contig length n_frags cumul_length`, abs_fragments_contacts_weighted.txt
    `id_frag_a id_frag_b n_contact` with 0-based absolute ids.  Returns (n_frags, n_contact_lines)."""
    import os

    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)
    sizes = np.maximum(8, rng.poisson(mean_frags, n_contigs))
    names = ["ctg%02d" % (i + 1) for i in range(n_contigs)]
    frag_rows, contig_rows, seqs = [], [], []
    starts_abs, mids_kb, contig_of = [], [], []
    cumul = 0
    for ci, (name, nf) in enumerate(zip(names, sizes)):
        lens = np.clip(np.round(rng.lognormal(np.log(900.0), 0.8, size=nf)), 20, 12000).astype(np.int64)
        lens[rng.random(nf) < 0.02] = rng.integers(20, 50)  # a few fragments below the 50 bp filter
        ends = np.cumsum(lens)
        begins = ends - lens
        for k in range(nf):
            frag_rows.append("%d\t%s\t%d\t%d\t%d\t%.4f\n" % (k + 1, name, begins[k], ends[k], lens[k], rng.uniform(0.3, 0.6)))
            mids_kb.append((begins[k] + ends[k]) / 2000.0)
            contig_of.append(ci)
        contig_rows.append("%s\t%d\t%d\t%d\n" % (name, ends[-1], nf, cumul))
        cumul += nf
        seqs.append("".join(rng.choice(list("ACGT"), size=int(ends[-1]))))
    n = cumul
    mids_kb, contig_of = np.array(mids_kb), np.array(contig_of)
    dead = rng.random(n) < 0.015  # fragments that never appear in a contact
    pairs = {}
    n_draw = n * contacts_per_frag
    a = rng.integers(0, n, size=n_draw)
    cis = rng.random(n_draw) < 0.85
    # cis partner: a log-uniform genomic offset (~ s^-1 decay), trans partner: uniform
    off = np.exp(rng.uniform(np.log(0.3), np.log(300.0), size=n_draw)) * rng.choice([-1.0, 1.0], size=n_draw)
    for i in range(n_draw):
        fa = int(a[i])
        if cis[i]:
            members = np.nonzero(contig_of == contig_of[fa])[0]
            fb = int(members[np.argmin(np.abs(mids_kb[members] - (mids_kb[fa] + off[i])))])
        else:
            fb = int(rng.integers(0, n))
        if fa == fb or dead[fa] or dead[fb]:
            continue
        key = (min(fa, fb), max(fa, fb))
        pairs[key] = pairs.get(key, 0) + 1
    with open(os.path.join(folder, "fragments_list.txt"), "w") as f:
        f.write("id\tchrom\tstart_pos\tend_pos\tsize\tgc_content\n")
        f.writelines(frag_rows)
    with open(os.path.join(folder, "info_contigs.txt"), "w") as f:
        f.write("contig\tlength\tn_frags\tcumul_length\n")
        f.writelines(contig_rows)
    with open(os.path.join(folder, "abs_fragments_contacts_weighted.txt"), "w") as f:
        f.write("id_frag_a\tid_frag_b\tn_contact\n")
        for (fa, fb) in sorted(pairs):
            f.write("%d\t%d\t%d\n" % (fa, fb, pairs[(fa, fb)]))
    with open(os.path.join(folder, "genome.fa"), "w") as f:
        for name, seq in zip(names, seqs):
            f.write(">%s\n" % name)
            for i in range(0, len(seq), 70):
                f.write(seq[i:i + 70] + "\n")
    return n, len(pairs)


def write_text_dataset_large(folder, n_frags0=85_000, n_contigs=140, seed=11, contacts_per_frag=30, median_bp=300.0):
    """The same three text files + FASTA as ``write_text_dataset`` at the scale of BASELINE.json's config 1 (a
    yeast-like folder, ~85 k restriction fragments on ~140 contigs; level 4 of its pyramid has ~1 000 bins), generated with
    array operations (the small generator's per-contact Python loop would take hours here).  Cis partners: the fragment of
    the same contig nearest to a log-uniform genomic offset (~ 1/s decay); trans partners uniform; a few fragments below the
    builder's 50 bp filter and a few without any contact.  Returns (n_frags, n_contact_lines)."""
    import os

    rng = np.random.default_rng(seed)
    os.makedirs(folder, exist_ok=True)
    w = rng.dirichlet(np.full(n_contigs, 3.0))
    sizes = np.maximum(20, np.round(w * n_frags0)).astype(np.int64)
    n = int(sizes.sum())
    names = ["ctg%03d" % (i + 1) for i in range(n_contigs)]
    contig_of = np.repeat(np.arange(n_contigs), sizes)
    first = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    lens = np.clip(np.round(rng.lognormal(np.log(median_bp), 0.6, size=n)), 20, 4000).astype(np.int64)
    short = rng.random(n) < 0.02
    lens[short] = rng.integers(20, 50, size=int(short.sum()))
    ends_glob = np.cumsum(lens)
    base = np.concatenate([[0], ends_glob])[first][contig_of]  # bp before the fragment's contig
    ends = ends_glob - base
    begins = ends - lens
    mids_kb = (begins + ends) / 2000.0
    idx_in = np.arange(n) - first[contig_of] + 1
    gc = rng.uniform(0.3, 0.6, size=n)
    with open(os.path.join(folder, "fragments_list.txt"), "w") as f:
        f.write("id\tchrom\tstart_pos\tend_pos\tsize\tgc_content\n")
        f.write("".join("%d\t%s\t%d\t%d\t%d\t%.4f\n" % (idx_in[k], names[contig_of[k]], begins[k], ends[k], lens[k], gc[k]) for k in range(n)))
    clen = ends[first + sizes - 1]
    with open(os.path.join(folder, "info_contigs.txt"), "w") as f:
        f.write("contig\tlength\tn_frags\tcumul_length\n")
        for ci in range(n_contigs):
            f.write("%s\t%d\t%d\t%d\n" % (names[ci], clen[ci], sizes[ci], first[ci]))
    # contacts
    dead = rng.random(n) < 0.015
    n_draw = n * contacts_per_frag
    a = rng.integers(0, n, size=n_draw)
    cis = rng.random(n_draw) < 0.85
    off = np.exp(rng.uniform(np.log(0.3), np.log(300.0), size=n_draw)) * rng.choice([-1.0, 1.0], size=n_draw)
    # nearest fragment of the same contig to mids[a] + off: search in a key that orders contigs apart
    span = float(mids_kb.max()) + 1e4
    key = contig_of * (4.0 * span) + mids_kb
    want = contig_of[a] * (4.0 * span) + np.clip(mids_kb[a] + off, 0.0, span)
    pos = np.searchsorted(key, want)
    lo, hi = first[contig_of[a]], first[contig_of[a]] + sizes[contig_of[a]] - 1
    p0 = np.clip(pos - 1, lo, hi)
    p1 = np.clip(pos, lo, hi)
    b_cis = np.where(np.abs(key[p0] - want) <= np.abs(key[p1] - want), p0, p1)
    b = np.where(cis, b_cis, rng.integers(0, n, size=n_draw))
    ok = (a != b) & ~dead[a] & ~dead[b]
    fa, fb = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
    pair, cnt = np.unique(fa.astype(np.int64) * n + fb, return_counts=True)
    with open(os.path.join(folder, "abs_fragments_contacts_weighted.txt"), "w") as f:
        f.write("id_frag_a\tid_frag_b\tn_contact\n")
        f.write("".join("%d\t%d\t%d\n" % (q // n, q % n, k) for q, k in zip(pair.tolist(), cnt.tolist())))
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(os.path.join(folder, "genome.fa"), "wb") as f:
        for ci in range(n_contigs):
            seq = letters[rng.integers(0, 4, size=int(clen[ci]))]
            f.write((">%s\n" % names[ci]).encode())
            full = (len(seq) // 70) * 70
            if full:
                body = np.empty((full // 70, 71), np.uint8)
                body[:, :70] = seq[:full].reshape(-1, 70)
                body[:, 70] = 10
                f.write(body.tobytes())
            if len(seq) > full:
                f.write(seq[full:].tobytes() + b"\n")
    return n, int(pair.size)
