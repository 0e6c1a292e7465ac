"""Host-side mirror of the reference's ``sampler`` over the HIP C ABI (drop-in for the scoring path).

Mirrors ``/root/reference/src/instagraal/cuda_lib_gl_single.py`` ("CL"): same constructor
arguments (CL:92-125), same public methods and attributes that ``simulation`` and
``instagraal_class.full_em`` touch:

    step_sampler(id_frag, n_neighbours, dt) -> 6-tuple            CL:1401-1465
    step_nuisance_parameters(dt, t, n_step) -> 8-tuple            CL:2961-3051
    estimate_parameters_rippe / eval_likelihood_init              CL:2239-2372, 1193-1243
    bomb_the_genome, modify_gl_cuda_buffer, dist_inter_genome     CL:1925-1948, 2715-2881, 665-716
    apply_replay_simu, free_gpu                                   CL:2546-2553, 3167-3177
    gpu_vect_frags.copy_from_gpu() + numpy attributes             gpustruct.py:162-186

What the reference does with ~450 synchronous pycuda launches and 5 PCIe sorts per move is ONE
call here (``ig_step``: eight launches, one 80-byte D2H).  ``step_sampler_batch`` enqueues many
moves without any host round trip; candidate draws do not depend on the genome state
(CL:3103-3141 reads fixed distributions), so they can be drawn ahead with the same RNG stream.
"""
from __future__ import annotations

import numpy as np

from . import hip_lib
from . import optim_rippe_curve_update as opti

LIST_SIZE = np.array([1, 3, 5, 10, 20, 50, 200, 200], dtype=np.int32)  # CL:417
N_INSERT_BLOCKS = 6  # CL:192
N_TMP_STRUCT = 24  # CL:194
PARAM_NAMES = ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")  # KA:91-100
PARAM_DTYPE = np.dtype([(k, np.float32) for k in PARAM_NAMES], align=True)  # CL:235-247


def soa17_from_dict(S_o_A_frags, n):
    """17 x N int32 in KA:40-58 order; ``ori`` starts at +1 like create_gpu_struct (CL:537-541)."""
    out = np.zeros((17, n), np.int32)
    for k, name in enumerate(hip_lib.FRAG_FIELDS):
        if name == "ori":
            out[k] = 1
        elif name == "id":
            out[k] = np.arange(n)
        else:
            out[k] = S_o_A_frags[name]
    return out


def estimate_rippe_host(sparse_matrix, np_sub_frags_2_frags, S_o_A_frags, n_frags, mean_value_trans, max_dist_kb, size_bin_kb):
    """Host part of estimate_parameters_rippe (CL:2239-2341), vectorised: for the first n_frags // 10 sub-fragment rows
    whose contig is longer than one bin, the cis contacts are binned by genomic distance (zeros included: every such
    row contributes one value per bin, CL:2289-2292); the per-bin means (+ the trans level) are fitted with
    optim_rippe_curve_update.estimate_param_rippe; the trans level is then lowered tenfold (CL:2338) before the
    cis/trans cut-off is solved.  -> (bins_upd, mean_contacts_upd, p, y_estim, mean_value_trans / 10, d_max)"""
    bins = np.arange(size_bin_kb, max_dist_kb + size_bin_kb, size_bin_kb)
    nb = len(bins)
    sm = sparse_matrix.tocsr()
    parent = np_sub_frags_2_frags["x"].astype(np.int64)
    id_c = S_o_A_frags["id_c"][parent]
    s_all = S_o_A_frags["start_bp"][parent] / 1000.0 + np_sub_frags_2_frags["y"]
    len_kb = S_o_A_frags["l_cont_bp"][parent] / 1000
    acc = np.zeros(nb, np.float64)
    rows = 0
    for i in range(0, int(n_frags) // 10):
        if not (size_bin_kb < len_kb[i]):
            continue
        s, e = sm.indptr[i], sm.indptr[i + 1]
        j, dat = sm.indices[s:e], sm.data[s:e]
        keep = id_c[j] == id_c[i]
        d = np.abs(s_all[i] - s_all[j[keep]])
        ok = d < max_dist_kb
        acc += np.bincount((d[ok] / size_bin_kb).astype(np.int64), weights=dat[keep][ok], minlength=nb)[:nb]
        rows += 1
    mean = acc / max(rows, 1)
    mean_contacts = np.where((rows == 0) | (mean == 0), np.nan, mean + mean_value_trans).astype(np.float32)
    good = ~np.isnan(mean_contacts)
    bins_upd = np.array(bins[good])
    mean_contacts_upd = np.array(mean_contacts[good])
    p, y_estim = opti.estimate_param_rippe(mean_contacts_upd, bins_upd)
    mvt = mean_value_trans / 10.0  # CL:2338
    return bins_upd, mean_contacts_upd, p, y_estim, mvt, opti.estimate_max_dist_intra(p, mvt)


def problem_to_context(prob, params=None, device_id=0, rank=0, world=1):
    """Upload a synth.SynthProblem (contacts, sub-fragment table, state, fixed parameters)."""
    ctx = hip_lib.Context(device_id)
    ctx.upload_subfrag_table(prob.np_sub_frags_2_frags)
    ctx.upload_contacts(prob.coo_row, prob.coo_col, prob.coo_cnt, prob.n_sub_frags, rank, world)
    max_bounds = LIST_SIZE[:N_INSERT_BLOCKS].max() * np.int32(np.round(prob.S_o_A_frags["sub_len"].mean()) + 1)  # CL:418-420
    ctx.set_insert_config(LIST_SIZE[:N_INSERT_BLOCKS], int(max_bounds))
    ctx.upload_state(soa17_from_dict(prob.S_o_A_frags, prob.n_frags))
    p = prob.params if params is None else params
    mean_kb = np.float32(prob.S_o_A_sub_frags["len_bp"].mean() / 1000.0)  # CL:231, 1120
    ctx.set_params([np.float32(p[k]) for k in PARAM_NAMES], mean_kb, 0)
    ctx.set_params([np.float32(p[k]) for k in PARAM_NAMES], mean_kb, 1)
    return ctx


class DeviceFrags:
    """Stand-in for GPUStruct (gpustruct.py): ``copy_from_gpu()`` refreshes numpy attributes
    ``pos, sub_pos, id_c, ...`` from the device (contig ids canonically renumbered)."""

    def __init__(self, ctx):
        self._ctx = ctx
        self.copy_from_gpu()

    def copy_from_gpu(self, skip=None):
        soa = self._ctx.download_state()
        for k, name in enumerate(hip_lib.FRAG_FIELDS):
            setattr(self, name, soa[k].copy())
        return self

    def soa17(self):
        return np.stack([getattr(self, k) for k in hip_lib.FRAG_FIELDS]).astype(np.int32)


class sampler:  # noqa: N801 - the reference's class name
    def __init__(self, use_rippe, S_o_A_frags, collector_id_repeats, frag_dispatcher, id_frag_duplicated,
                 id_frags_blacklisted, n_frags, n_new_frags, init_n_sub_frags, n_new_sub_frags, np_rep_sub_frags_id,
                 sub_sampled_sparse_matrix, np_sub_frags_len_bp, np_sub_frags_id, np_sub_frags_accu, np_sub_frags_2_frags,
                 mean_squared_frags_per_bin, norm_vect_accu, sub_candidates_dup, sub_candidates_output_data,
                 S_o_A_sub_frags, sub_collector_id_repeats, sub_frag_dispatcher, sparse_matrix, mean_value_trans,
                 n_iterations, is_simu, vel=None, pos=None, device_id=0, coo=None, keep_all_scores=True):
        if not use_rippe:
            raise NotImplementedError("use_rippe=False is unreachable from the reference CLI (instagraal.py:564)")
        if len(sub_candidates_dup) or len(id_frag_duplicated):
            raise NotImplementedError("repeated fragments are dead in the reference (simu_single.py:513)")
        self.o = 0
        self.log_e = 0.43429448190325182  # CL:128
        self.n_frags = np.int32(n_frags)
        self.n_new_frags = np.int32(n_new_frags)
        self.init_n_sub_frags = np.int32(init_n_sub_frags)
        self.n_new_sub_frags = np.int32(n_new_sub_frags)
        self.id_frags_blacklisted = list(id_frags_blacklisted)
        self.S_o_A_frags = S_o_A_frags
        self.S_o_A_sub_frags = S_o_A_sub_frags
        self.np_sub_frags_id = np_sub_frags_id
        self.np_sub_frags_2_frags = np_sub_frags_2_frags
        self.sub_sampled_sparse_matrix = sub_sampled_sparse_matrix
        self.mean_len_bp_frags = S_o_A_sub_frags["len_bp"].mean()  # CL:231
        self.mean_value_trans = mean_value_trans
        self.n_iterations = n_iterations
        self.is_simu = is_simu
        self.dt = np.float32(0.01)
        self.n_insert_blocks = N_INSERT_BLOCKS
        self.n_tmp_struct = N_TMP_STRUCT
        N, M = int(n_new_frags), int(n_new_sub_frags)

        # contacts: CL:129, 564-615 (symmetrise, strict upper triangle, COO row-major)
        if coo is None:
            import scipy.sparse as sp

            self.sparse_matrix = sparse_matrix + sparse_matrix.transpose()
            c = sp.triu(self.sparse_matrix.tocoo(), k=1, format="coo")
            order = np.lexsort((c.col, c.row))
            row, col, dat = c.row[order], c.col[order], c.data[order]
        else:
            row, col, dat = coo
            self.sparse_matrix = None
        self.n_non_zero = int(len(dat))

        self.ctx = hip_lib.Context(device_id)
        self.ctx.upload_subfrag_table(np_sub_frags_2_frags)
        self.ctx.upload_contacts(row, col, dat, M)
        self.max_bounds_insert = LIST_SIZE[:N_INSERT_BLOCKS].max() * np.int32(np.round(S_o_A_frags["sub_len"].mean()) + 1)
        self.ctx.set_insert_config(LIST_SIZE[:N_INSERT_BLOCKS], int(self.max_bounds_insert))
        self.ctx.upload_state(soa17_from_dict(S_o_A_frags, N))
        # CL:269-276
        self.np_init_prev = np.copy(S_o_A_frags["prev"])
        self.np_init_next = np.copy(S_o_A_frags["next"])
        self.np_init_orientable = np.array([np_sub_frags_id[S_o_A_frags["id_d"][i]]["w"] > 1 for i in range(N)],
                                           dtype=np.int32)
        self.np_init_ori = np.ones(N, dtype=np.int32)
        self.ctx.set_initial_genome(self.np_init_prev, self.np_init_next, self.np_init_orientable, self.id_frags_blacklisted)
        self.gpu_vect_frags = DeviceFrags(self.ctx)
        self.setup_distri_frags()
        self.param_simu = None
        self.param_simu_test = None
        self.likelihood_t = 0.0
        n, m, _ = self.ctx.renumber_contigs()
        self.n_contigs, self.mean_length_contigs = n, m
        self.candidates = []
        self.all_scores = np.zeros(0)
        # step_sampler leaves the 24 x C scores of its move in ``all_scores`` as the reference does (CL:1414-1431) -- which nothing
        # outside step_sampler ever reads (CL:1435-1454).  False (what ``simulation`` constructs: the reference's loop, IG:221-228):
        # ``all_scores`` stays None and the move is scored in two tiers, the exact kernel for the columns that can still win only.
        self.keep_all_scores = bool(keep_all_scores)

    # ------------------------------------------------------------ parameters
    def mean_kb(self):
        return np.float32(self.mean_len_bp_frags / 1000.0)  # CL:1120

    def setup_rippe_parameters(self, param, d_max):  # CL:2206-2221
        kuhn, lm, slope, d, fact = param
        fact = np.float32(np.abs(fact))
        kuhn = np.float32(np.abs(kuhn))
        lm = np.float32(np.abs(lm))
        c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
        return np.array([(kuhn, lm, c1, np.float32(slope), np.float32(d), np.float32(d_max), np.float32(fact),
                          self.mean_value_trans)], dtype=PARAM_DTYPE)

    def set_param_simu(self, p, which=0):
        """p: dict or 1-element structured array (KA:91-100 fields)."""
        if isinstance(p, dict):
            arr = np.zeros(1, PARAM_DTYPE)
            for k in PARAM_NAMES:
                arr[k] = np.float32(p[k])
            p = arr
        p = np.array(p, dtype=PARAM_DTYPE).reshape(1)
        vals = [p[k][0] for k in PARAM_NAMES]
        if which == 0:
            self.param_simu = p
            self.ctx.set_params(vals, self.mean_kb(), 0)
            if self.param_simu_test is None:
                self.param_simu_test = p.copy()
                self.ctx.set_params(vals, self.mean_kb(), 1)
        else:
            self.param_simu_test = p
            self.ctx.set_params(vals, self.mean_kb(), 1)

    def estimate_parameters_rippe(self, max_dist_kb, size_bin_kb, display_graph=False):
        """CL:2239-2372: binned mean cis contacts of the first n_frags/10 sub-fragment rows, leastsq fit,
        cis/trans cut-off; then the initial likelihood."""
        self.bins = np.arange(size_bin_kb, max_dist_kb + size_bin_kb, size_bin_kb)
        self.bins_upd, self.mean_contacts_upd, p, self.y_estim, self.mean_value_trans, estim_max_dist = estimate_rippe_host(
            self.sparse_matrix, self.np_sub_frags_2_frags, self.S_o_A_frags, self.n_frags, self.mean_value_trans, max_dist_kb,
            size_bin_kb)
        self.set_param_simu(self.setup_rippe_parameters(p, estim_max_dist), 0)
        self.set_param_simu(self.param_simu, 1)
        self.eval_likelihood_init()

    # ------------------------------------------------------------- neighbours
    def setup_distri_frags(self):  # CL:3053-3101
        m = (self.sub_sampled_sparse_matrix + self.sub_sampled_sparse_matrix.T).tocsr()
        self.sym_sub_sampled_sparse_matrix = m
        self.distri_frags = {}
        fact = 3.0
        indptr, indices, data = m.indptr, m.indices, m.data
        for i in range(int(self.n_frags)):
            s, e = indptr[i], indptr[i + 1]
            yk, vk = indices[s:e], data[s:e]
            het = yk != i
            xk = yk[het]
            dat = np.float32(vk[het]) * fact
            if dat.sum() > 0:
                pk = dat / np.linalg.norm(dat, 1)
            else:
                tmp = np.ones_like(dat, dtype=np.float32)
                pk = tmp / tmp.sum()
            if len(xk) > 0:
                self.distri_frags[i] = dict(distri="ok", xk=np.array(xk), pk=pk)
            else:
                self.distri_frags[i] = dict(distri=None)
        # the same distributions as one CSR in the library's host memory: the draw of whole runs of moves happens there
        # (hip_lib.Neighbours / csrc/ig_draw.cpp), on numpy's generator state, off the interpreter
        n = int(self.n_frags)
        lens = np.array([len(self.distri_frags[i]["xk"]) if self.distri_frags[i]["distri"] is not None else 0 for i in range(n)],
                        dtype=np.int64)
        ok = [i for i in range(n) if lens[i]]
        nb_indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nb_xk = np.concatenate([self.distri_frags[i]["xk"] for i in ok]).astype(np.int32) if ok else np.zeros(0, np.int32)
        nb_pk = np.concatenate([self.distri_frags[i]["pk"] for i in ok]).astype(np.float32) if ok else np.zeros(0, np.float32)
        self.neighbours = hip_lib.Neighbours(nb_indptr, nb_xk, nb_pk, n, self.id_frags_blacklisted)

    def return_neighbours(self, id_fA, delta0):  # CL:3103-3141
        ori_id = int(id_fA)  # id_d is the identity without repeats
        d = self.distri_frags[ori_id]
        if d["distri"] is not None:
            distri = d["pk"]
            n_max = min(delta0, np.nonzero(distri != 0)[0].shape[0])
            init_id = np.random.choice(d["xk"], n_max, p=distri, replace=False)
        else:
            init_id = np.random.choice(self.n_frags, delta0, replace=False)
        black = self.id_frags_blacklisted
        return [int(e) for e in init_id if e not in black]

    # ------------------------------------------------------------ likelihood
    def eval_likelihood_init(self):  # CL:1193-1243
        nz, z, _ = self.ctx.full_likelihood(0)
        self.curr_likelihood_on_nz = np.array([nz])
        # the reference's initial zero-pixel scalar is garbage (-inf): an int is passed where the kernel
        # takes a float (CL:743, quirk Q8).  It never reaches a score and likelihood_t is overwritten by
        # the first step_sampler (CL:1457); the well-defined value is reported instead.
        self.curr_likelihood_on_z = z
        self.likelihood_t = self.curr_likelihood_on_nz + z

    def eval_likelihood(self):  # CL:1245-1292
        nz, z, _ = self.ctx.full_likelihood(0)
        return nz

    def eval_likelihood_4_nuisance(self):  # CL:1296-1344: test parameters, coordinates of the state BEFORE the last move
        nz, z, _ = self.ctx.full_likelihood(1, use_prev_tables=True)
        self.curr_likelihood_nuis = np.array([nz]) + z
        return self.curr_likelihood_nuis

    # ------------------------------------------------------------------ moves
    def _clean(self, id_frag, candidates):
        c = sorted(int(x) for x in candidates)
        if int(id_frag) in c:
            # reachable only through the uniform fallback draw (CL:3124); the reference then scores
            # stale collector buffers (quirk Q13) -- no defined result to reproduce
            c = [x for x in c if x != int(id_frag)]
        return c

    def step_sampler(self, id_frag, n_neighbours, dt=None, candidates=None):  # CL:1401-1465
        # one library call (ig_step_draw): the draw of return_neighbours (CL:3103-3141) on numpy's generator state in place -- the same
        # list and the same state afterwards (tests/test_cpu_abi_and_host.py) -- the move, and the record as soon as it is complete
        res, sc, self.candidates = self.ctx.step_draw(self.neighbours, id_frag, max(1, int(n_neighbours)),
                                                      None if candidates is None else self._clean(id_frag, candidates), self.keep_all_scores)
        self.all_scores = sc
        self.o = res.o
        self.likelihood_t = res.o
        self.n_contigs = np.int32(res.n_contigs)
        self.mean_length_contigs = np.float32(res.mean_len)
        self.last_result = res
        return (res.o, res.dist, res.op_sampled, res.id_f_sampled, self.mean_length_contigs, self.n_contigs)

    def draw_candidates(self, frags, n_neighbours):
        """Candidates of consecutive moves, consuming numpy's global RNG exactly as successive
        step_sampler calls would (state-independent: CL:3103-3141)."""
        return self.neighbours.draw(frags, max(1, n_neighbours))

    def draw_candidates_python(self, frags, n_neighbours):
        """the same through return_neighbours, one numpy call per move (what draw_candidates is checked against)"""
        out = np.full((len(frags), max(1, n_neighbours)), -1, np.int32)
        for i, f in enumerate(frags):
            c = self._clean(f, self.return_neighbours(int(f), n_neighbours))
            out[i, : len(c)] = c
        return out

    def step_sampler_batch(self, frags, n_neighbours, candidates=None):
        """len(frags) consecutive step_sampler calls -- candidate draw included -- in ONE library call: a host thread draws
        the lists of the moves ahead on numpy's generator state while the launches of the moves in flight run; returns the
        structured result array (fields o, dist, op_sampled, id_f_sampled, mean_len, n_contigs, ...)."""
        frags = np.ascontiguousarray(frags, np.int32)
        if candidates is None:
            res, self.last_candidates = self.ctx.step_batch_draw(self.neighbours, frags, max(1, n_neighbours))
        else:
            res = self.ctx.step_batch(frags, candidates)
            self.last_candidates = candidates
        last = res[-1]
        self.o = self.likelihood_t = float(last["o"])
        self.n_contigs = np.int32(last["n_contigs"])
        self.mean_length_contigs = np.float32(last["mean_len"])
        return res

    def apply_replay_simu(self, id_fA, id_fB, op_sampled, dt=None):  # CL:2546-2553
        self.ctx.apply(int(id_fA), int(id_fB), int(op_sampled))
        self.gpu_vect_frags.copy_from_gpu()

    def test_copy_struct(self, id_fA, id_f_sampled, mode, max_id=None):  # CL:2094-2151
        self.ctx.apply(int(id_fA), int(id_f_sampled), int(mode))

    def modify_gl_cuda_buffer(self, id_fi=0, dt=None):  # CL:2715-2881
        n, m, max_id = self.ctx.renumber_contigs()
        self.n_contigs, self.mean_length_contigs = np.int32(n), m
        return np.int32(max_id)

    def bomb_the_genome(self):  # CL:1925-1948
        a = np.arange(0, self.n_new_frags, dtype=np.int32)
        np.random.shuffle(a)
        self.ctx.bomb(a)
        self.modify_gl_cuda_buffer(0, self.dt)

    def dist_inter_genome(self, tmp_gpu_vect_frags=None):  # CL:665-716
        return self.ctx.genome_distance()

    # -------------------------------------------------------------- nuisance
    def temperature(self, t, n_step):  # CL:3163-3165
        return 1.0

    def _sigmas(self, curr_param):  # CL:2970-2974
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = curr_param[0]
        self.sigma_fact = 10 ** (np.log10(fact) - 2)
        self.sigma_slope = 0.005
        self.sigma_d_max = 100
        self.sigma_d_nuc = 10 ** (np.log10(d_nuc) - 2)
        self.sigma_d = 10

    def _propose(self, curr_param, id_modif, normal):
        """the test parameters of one nuisance step (CL:2979-3017).  ``normal(sigma)`` -> the value numpy's
        ``np.random.normal(loc=0.0, scale=sigma)`` returns at this point of the stream (a Python float)."""
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = curr_param[0]
        if id_modif == 0:
            new_fact = fact + normal(self.sigma_fact)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, new_fact], d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, new_fact, d_nuc)]
        elif id_modif == 1:
            new_slope = slope + normal(self.sigma_slope)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, new_slope, d, fact], d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, new_slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, new_slope, d, new_d_max, fact, d_nuc)]
        elif id_modif == 2:
            new_d_max = d_max + normal(self.sigma_d_max)
            new_d_nuc = opti.peval(new_d_max, [kuhn, lm, slope, d, fact])  # 5 values where 4 are read (quirk Q12)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        else:
            if self.sigma_d_nuc <= 0:
                new_d_nuc = d_nuc
            else:
                new_d_nuc = d_nuc + normal(self.sigma_d_nuc)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, fact], new_d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        return np.array(out, dtype=PARAM_DTYPE)

    def _epoch8(self, curr_param):
        """what the proposals of the steps between two accepted ones share: the current parameters as eight float32 scalars, c1
        recomputed from them the way every branch of ``_propose`` recomputes it (CL:2985-3013)"""
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = curr_param[0]
        c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
        return (kuhn, lm, c1, slope, d, d_max, fact, d_nuc)

    def _propose8(self, e8, id_modif, g):
        """``_propose`` on ``_epoch8``'s tuple with the standard normal ``g`` of the step, -> the eight float32 values of the
        structured array ``_propose`` builds (same expressions, same dtypes, same bits: tests/test_cpu_abi_and_host.py): the run loop
        computes one proposal per step on the host's critical path and needs the array only for the steps that are accepted"""
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = e8
        f32 = np.float32
        if id_modif == 0:
            new_fact = fact + (0.0 + float(self.sigma_fact) * g)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, new_fact], d_nuc, d_max)
            return (kuhn, lm, c1, slope, d, f32(new_d_max), f32(new_fact), d_nuc)
        if id_modif == 1:
            new_slope = slope + (0.0 + float(self.sigma_slope) * g)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, new_slope, d, fact], d_nuc, d_max)
            c1 = f32((0.53 * np.power(lm / kuhn, new_slope)) * np.power(kuhn, -3))
            return (kuhn, lm, c1, f32(new_slope), d, f32(new_d_max), fact, d_nuc)
        if id_modif == 2:
            new_d_max = d_max + (0.0 + float(self.sigma_d_max) * g)
            new_d_nuc = opti.peval(new_d_max, [kuhn, lm, slope, d, fact])  # 5 values where 4 are read (quirk Q12)
            return (kuhn, lm, c1, slope, d, f32(new_d_max), fact, f32(new_d_nuc))
        new_d_nuc = d_nuc if self.sigma_d_nuc <= 0 else d_nuc + (0.0 + float(self.sigma_d_nuc) * g)
        new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, fact], new_d_nuc, d_max)
        return (kuhn, lm, c1, slope, d, f32(new_d_max), fact, f32(new_d_nuc))

    def step_nuisance_parameters(self, dt, t, n_step):  # CL:2961-3051
        curr_param = np.copy(self.param_simu)
        self._sigmas(curr_param)
        id_modif = np.random.choice(4)
        out = self._propose(curr_param, id_modif, lambda sigma: np.random.normal(loc=0.0, scale=sigma))
        self.set_param_simu(out, 1)
        self.likelihood_nuis = self.eval_likelihood_4_nuisance()
        F_t = self.temperature(t, n_step)
        with np.errstate(over="ignore"):  # a much better likelihood at a low temperature: inf >= u, accepted
            ratio = np.exp((self.likelihood_nuis - self.likelihood_t) / F_t)
        u = np.random.rand()
        success = 0
        if ratio >= u:
            success = 1
            self.set_param_simu(out, 0)
            self.likelihood_t = self.likelihood_nuis
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = self.param_simu[0]
        y_rippe = opti.peval(self.bins, [kuhn, lm, slope, d, fact]) if hasattr(self, "bins") else None
        return (fact, d, d_max, d_nuc, slope, self.likelihood_t, success, y_rippe)

    def step_sampler_nuisance_batch(self, frags, n_neighbours, dt=None, t0=0, n_step=0):
        """``for f in frags: step_sampler(f, n, dt); step_nuisance_parameters(dt, t, n_step)`` -- the loop of
        instagraal.py:217-262 for cycles > 4 -- with the same results and the same generator stream, arranged for the GPU:

        * the stream of the whole run is drawn up front in the library (candidate lists, choice(4), the standard normal
          behind normal(0, sigma), the acceptance uniform: its consumption does not depend on the parameters);
        * the nuisance step's pass over all contacts reads the state BEFORE the move (quirk Q12) and only needs the move's
          score besides: it runs next to the move (``ig_nuis_step_begin`` / ``ig_nuis_end``);
        * a rejected step changes nothing a move reads: the moves are scored ahead in batches and decided one per step
          (``ig_nuis_run_begin``); an accepted step discards what was scored ahead;
        * while both run, the host prepares the next step's proposal (the root finding for d_max, CL:2983) for both
          outcomes of this one; the acceptance test itself and the first launches of the next step are one library call
          (``ig_nuis_step_next``), so that no Python runs between two steps' kernels.

        ``self.likelihood_nuis`` (the test likelihood of the last step: read nowhere outside the reference's method, CL:3023-3036) is
        set by the steps that go the plain way only -- a step rejected inside a chain is decided from an interval that
        lies below the threshold, and no value of it exists.

        -> (structured move results, list of the 8-tuples of step_nuisance_parameters with y_rippe = None)"""
        frags = np.ascontiguousarray(frags, np.int32)
        n = frags.size
        res = np.zeros(n, hip_lib.MOVE_RESULT_DTYPE)
        self._sigmas(np.copy(self.param_simu))
        # one step at a time: the one case where the stream depends on the parameters; an initial genome whose prev / next
        # links are not mutually inverse (the batch commit's bookkeeping needs that)
        if n == 0 or self.sigma_d_nuc <= 0 or not self.ctx.links_inverse():
            tuples = []
            for i, f in enumerate(frags):
                r = self.step_sampler(int(f), n_neighbours, dt)
                for k in res.dtype.names:
                    res[k][i] = getattr(self.last_result, k)
                tuples.append(self.step_nuisance_parameters(dt, t0 + i, n_step))
            return res, tuples
        with opti.quiet_runs():
            tuples, n_done = self._nuisance_run(frags, n_neighbours, t0, n_step, res)
        if n_done < n:
            # an accepted step took d_nuc to 0 (sigma_d_nuc with it: CL:2973): from the next step on the reference draws no normal for
            # id_modif == 3 (CL:3007-3010) and the stream drawn up front is not the reference's any more -- the rest one step at a time
            rest, more = self.step_sampler_nuisance_batch(frags[n_done:], n_neighbours, dt, t0 + n_done, n_step)
            res[n_done:] = rest
            tuples += more
        return res, tuples

    def _nuisance_run(self, frags, n_neighbours, t0, n_step, res):
        """the run of (move, nuisance step) pairs.  Two kinds of library calls: a CHAIN (``ig_nuis_chain_begin``: the pairs ahead as far
        as the device decides them alone -- moves from the batch's score records, steps rejected from the histogram tier's interval --
        while this loop prepares the proposals of the steps behind), and ONE pair the plain way for whatever a chain stops in front of
        (an accepted or undecided step, a conflict, a batch used up: ``ig_nuis_step_begin`` + ``ig_nuis_step_next``).  A rejected step
        changes no parameter, so the proposals of the steps ahead are a pure function of the current parameters and the pre-drawn
        stream: they are computed once per accepted step's epoch, ahead of their use."""
        import time as _t

        n = frags.size
        stream0 = np.random.get_state()  # (a run that ends early rewinds to its last step: `stop`)
        cands, id_modif, gauss, unif = self.neighbours.draw_nuisance(frags, max(1, n_neighbours))
        mean_kb = self.mean_kb()
        gauss_l, id_modif_l, unif_l = gauss.tolist(), id_modif.tolist(), unif.tolist()
        # (the base method returns 1.0: not called n times; one overridden in a subclass OR assigned on the instance is)
        plain_T = getattr(self.temperature, "__func__", None) is sampler.temperature
        temps = np.ones(n) if plain_T else np.array([float(self.temperature(t0 + i, n_step)) for i in range(n)])
        prof = self.nuis_profile = dict(propose=0.0, step=0.0, book=0.0, chain=0.0)
        trace = getattr(self, "nuis_step_trace", None)  # a list: (seconds, accepted) per plain step (tools/nuisance_rate.py)
        use_chain = hip_lib.nuis_chain_wanted()
        if not hasattr(self, "_nuis_acc_ema"):
            self._nuis_acc_ema = 0.0  # share of the recent steps that were accepted (kept from run to run)
        names = PARAM_NAMES
        curr = np.copy(self.param_simu)
        self._sigmas(curr)
        e8 = self._epoch8(curr)
        props = {}  # step -> the eight float32 test parameters of its proposal: valid while `curr` stands
        P = np.empty((n, 8), np.float32)  # ... and row i of it: a chain call hands over rows i .. i + K - 1 as they lie

        def prop(i):
            q = props.get(i)
            if q is None:
                q = props[i] = self._propose8(e8, id_modif_l[i], gauss_l[i])
                P[i] = q
            return q

        def as_array(p8):
            return np.array([p8], dtype=PARAM_DTYPE)

        last_test = [None]  # the test parameters of the last step that went the plain way

        # per step: the parameters it ends with (index into `epochs`), success; likelihood_t of an accepted step; the exact likelihood
        # of a step accepted ahead of its exact pass is filled in when the next plain step (or the end of the run) fetches it
        epochs = [self.param_simu]
        ep_of = np.zeros(n, np.int64)
        success_of = np.zeros(n, np.int8)
        lik_acc = {}
        patch = None  # (step, z)
        stop = [None]  # the run ends behind this many pairs: an accepted step left sigma_d_nuc <= 0 (step_sampler_nuisance_batch)

        def fill_in():
            j, zj = patch
            lik_acc[j] = np.array([self.ctx.nuis_exact_result()]) + zj

        def accepted(i, p8, lik_nuis, deferred, z):
            nonlocal curr, props, patch, e8
            out = as_array(p8)
            curr = np.copy(out)
            self.param_simu = out
            epochs.append(out)
            props = {}
            self._sigmas(curr)
            e8 = self._epoch8(curr)
            lik_acc[i] = lik_nuis
            if deferred:
                patch = (i, z)
            if self.sigma_d_nuc <= 0:
                stop[0] = i + 1

        def pairs_pipelined(i0, i1):
            """steps i0 .. i1 - 1, one pair per library call with the next step begun inside the call that ends this one (the loop of
            rounds 2 - 3: where chains are off, the histogram tier is not in use, or every third step is accepted -- a chain only ever
            rejects): while the GPU works on step i the proposal of step i + 1 for the case that this one is rejected is worked out;
            the one for the other case only if it comes to that"""
            nonlocal patch
            p8 = prop(i0)
            self.ctx.nuis_step_begin(i0, p8, mean_kb)
            for i in range(i0, i1):
                ta = _t.perf_counter()
                has_next = i + 1 < i1
                nxt = prop(i + 1) if has_next else None
                if patch is not None:  # (the exact pass of the step accepted last ran behind its decision: done by now)
                    fill_in()
                    patch = None
                t1 = _t.perf_counter()
                T = float(temps[i])
                r, nz, z, success = self.ctx.nuis_step_next(T, unif_l[i], nxt if has_next else None, None, mean_kb, has_next)
                t2 = _t.perf_counter()
                deferred = success == 3
                if deferred:
                    success = 1
                lik_nuis = np.array([nz]) + z
                began = success == 0  # (rejected: the library has begun step i + 1 with the parameters handed over)
                if success == 2:  # exp() within 1e-9 of u: the reference's own arithmetic decides
                    with np.errstate(over="ignore"):
                        ratio = np.exp((lik_nuis - r.o) / T)
                    success = 1 if ratio >= unif_l[i] else 0
                    if success:
                        self.ctx.nuis_accept()
                if success:
                    accepted(i, p8, lik_nuis, deferred, z)
                    has_next = has_next and stop[0] is None
                    nxt = prop(i + 1) if has_next else None  # (under the promoted parameters)
                ep_of[i] = len(epochs) - 1
                success_of[i] = success
                self.likelihood_nuis = lik_nuis
                last_test[0] = p8
                self._nuis_acc_ema = 0.9 * self._nuis_acc_ema + 0.1 * float(success)
                if has_next:
                    if not began:
                        self.ctx.nuis_step_begin(i + 1, nxt, mean_kb)
                    p8 = nxt
                t3 = _t.perf_counter()
                prof["propose"] += t1 - ta
                prof["step"] += t2 - t1
                prof["book"] += t3 - t2
                if trace is not None:
                    trace.append((t3 - ta, int(success)))
                if stop[0] is not None:
                    return

        self.ctx.nuis_run_begin(frags, cands)
        i = 0
        try_chain = False  # (the first pair scores the first batch: the plain way)
        empty = 0  # chains in a row that decided no pair (intervals that decide nothing: a void histogram tier, tests at the margin)
        LOOK = 2 * hip_lib.CHAIN_MAX
        _KCAP = 32  # sets of a chain call at most (16 / 24 / 32 / 40 / 48: 17.5 / 18.5 / 18.5 / 18.5 / 17.4 k pairs/s once the chain has settled)
        try:
            if not use_chain:
                pairs_pipelined(0, n)
                i = n
            while i < n and stop[0] is None:
                if self._nuis_acc_ema > 0.3 or empty >= 4:
                    # (the first steps of a run's first nuisance cycle: a third of the proposals accepted; or chains that get nowhere:
                    # a chain only ever rejects, and only from a certain interval -- one pair per call for a while, then another try)
                    i1 = min(n, i + (16 if empty < 4 else 32))
                    pairs_pipelined(i, i1)
                    i = i1
                    try_chain = False
                    empty = min(empty, 3)
                    continue
                if use_chain and try_chain:
                    ta = _t.perf_counter()
                    ready = 0  # proposals in hand from step i on
                    while ready < _KCAP and (i + ready) in props:
                        ready += 1
                    K = min(n - i, max(8, ready))
                    for t in range(i + ready, i + K):
                        prop(t)
                    self.ctx.nuis_chain_begin(i, P[i:i + K], unif[i:i + K], temps[i:i + K], mean_kb)
                    t = i + K
                    lim = min(n, i + K + LOOK)
                    while t < lim and not self.ctx.nuis_chain_done():  # the proposals of the steps behind, while the device works
                        if t not in props:
                            prop(t)
                        t += 1
                    j, reason = self.ctx.nuis_chain_end()
                    empty = 0 if j else empty + 1
                    self._nuis_acc_ema *= 0.9 ** j
                    ep_of[i:i + j] = len(epochs) - 1
                    i += j
                    prof["chain"] += _t.perf_counter() - ta
                    if reason == 0 and i < n:
                        continue  # every set was used: the next chain
                    if i >= n:
                        break
                    if reason == 6:  # the histogram tier is not in use: one pair per call from here on
                        pairs_pipelined(i, n)
                        i = n
                        break
                # ---- one pair the plain way
                ta = _t.perf_counter()
                p8 = prop(i)
                if patch is not None:  # (before the next step can be accepted ahead of its exact pass)
                    fill_in()
                    patch = None
                t1 = _t.perf_counter()
                T = float(temps[i])
                self.ctx.nuis_step_begin(i, p8, mean_kb)
                # (has_next without parameters: behind an accepted step the library re-scores the slots ahead under the promoted
                # parameters at once -- that needs nothing from here and runs while the proposals of the new epoch are worked out)
                r, nz, z, success = self.ctx.nuis_step_next(T, unif_l[i], None, None, mean_kb, i + 1 < n)
                t2 = _t.perf_counter()
                deferred = success == 3  # accepted from the screened interval: nz is its midpoint until the exact pass is through
                if deferred:
                    success = 1
                lik_nuis = np.array([nz]) + z
                rescored = success == 1
                if success == 2:  # exp() within 1e-9 of u: the reference's own arithmetic decides
                    with np.errstate(over="ignore"):
                        ratio = np.exp((lik_nuis - r.o) / T)
                    success = 1 if ratio >= unif_l[i] else 0
                    if success:
                        self.ctx.nuis_accept()
                if success:
                    accepted(i, p8, lik_nuis, deferred, z)
                ep_of[i] = len(epochs) - 1
                success_of[i] = success
                self.likelihood_nuis = lik_nuis
                last_test[0] = p8
                self._nuis_acc_ema = 0.9 * self._nuis_acc_ema + 0.1 * float(success)
                try_chain = rescored or not success  # (accepted by the arithmetic here: the slots ahead are void, a plain pair scores them)
                i += 1
                t3 = _t.perf_counter()
                prof["propose"] += t1 - ta
                prof["step"] += t2 - t1
                prof["book"] += t3 - t2
                if trace is not None:
                    trace.append((t3 - ta, int(success)))
        except BaseException:
            if patch is not None:
                try:  # (the run failed: whatever fetching the pending exact sum raises on the failed handle must not replace the cause)
                    fill_in()
                except Exception:
                    pass
            raise
        if patch is not None:
            fill_in()
        # the records of all moves at once; the 8-tuples from them
        n_done = n if stop[0] is None else stop[0]
        if n_done < n:  # the generator where the reference's is behind pair n_done - 1 (the same draws again: their number per pair is fixed while sigma_d_nuc > 0)
            np.random.set_state(stream0)
            self.neighbours.draw_nuisance(frags[:n_done], max(1, n_neighbours))
        res[:n_done] = self.ctx.batch_results(n_done)
        o_col = res["o"]
        tuples = []
        for k in range(n_done):
            kuhn, lm, c1, slope, d, d_max, fact, d_nuc = epochs[ep_of[k]][0]
            lik = lik_acc[k] if success_of[k] else o_col[k]
            tuples.append((fact, d, d_max, d_nuc, slope, lik, int(success_of[k]), None))
        self.param_simu = epochs[-1]
        if last_test[0] is not None:
            self.param_simu_test = as_array(last_test[0])
        self.likelihood_t = tuples[-1][5]
        last = res[n_done - 1]
        self.o = float(last["o"])
        self.n_contigs = np.int32(last["n_contigs"])
        self.mean_length_contigs = np.float32(last["mean_len"])
        self.candidates = [int(x) for x in cands[n_done - 1] if x >= 0]
        return tuples, n_done

    def free_gpu(self):  # CL:3167-3177
        self.ctx.close()
