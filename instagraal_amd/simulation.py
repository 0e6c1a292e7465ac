"""From an instaGRAAL input folder to a running sampler and back to `info_frags.txt` / `genome.fasta`: the callers on
either side of the scoring path (reference: simu_single.py "SS" and instagraal.py "IG").

* ``simulation`` mirrors SS:27-175: pyramid (``pyramid.build_and_filter`` with 9 levels of factor 3, SS:539-550), the
  two levels in use, the sub-fragment tables (``create_sub_frags`` SS:674-723, ``create_new_sub_frags`` SS:725-739),
  the structure-of-arrays states with the duplicate-fragment columns (``modify_vect_frags`` SS:221-347: the reference
  hard-wires an empty duplicate list, SS:512, so these only add ``rep / activ / id_d``), the 29 constructor arguments
  of the sampler (SS:120-153) and the initial P(s) estimation (SS:157-171).
* ``instagraal_class.full_em`` mirrors IG:196-291: cycles over shuffled bins, per-cycle outputs
  (``save_simu_step_<j>.txt``, ``info_frags.txt``, ``genome.fasta``, the ``list_*.txt`` traces).  Cycles without
  nuisance sampling run through ``step_sampler_batch`` (the results are identical).
* ``run_instagraal`` mirrors IG:502-581.

``assemble_sampler_args`` needs no GPU and is what the CPU tests pin against the reference's own values.
"""
from __future__ import annotations

import os

import numpy as np

from . import pyramid as pyr

INT2 = np.dtype([("x", np.int32), ("y", np.int32)], align=True)
INT3 = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32)], align=True)
INT4 = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32), ("w", np.int32)], align=True)
FLOAT3 = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32)], align=True)
FLOAT4 = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("w", np.float32)], align=True)

SIZE_PYRAMID, FACTOR = 9, 3  # SS:541-542


def _with_duplicate_columns(soa):
    """SS:221-347 / 349-457 without duplicated fragments: the loader's arrays + rep = 0, activ = 1, id_d = id"""
    out = {k: np.array(v, dtype=np.int32) for k, v in soa.items()}
    n = len(out["id"])
    out["rep"] = np.zeros(n, np.int32)
    out["activ"] = np.ones(n, np.int32)
    out["id_d"] = out["id"].copy()
    return out


def _dispatcher(n):
    """SS:312-328 without duplicates: every initial fragment owns one slot"""
    d = np.zeros(n, INT2)
    d["x"] = np.arange(n)
    d["y"] = np.arange(n) + 1
    return np.arange(n, dtype=np.int32), d


def create_sub_frags(spec_level, sub_level_soa):
    """SS:674-723: per bin its <= 3 sub-fragments (ids, lengths in kb, accumulated restriction fragments) and, per
    sub-fragment, (parent bin, Watson offset, Crick offset, index in bin) = distance from either end of the bin to the
    middle of the sub-fragment, accumulated in float32 exactly as the reference does."""
    lo = spec_level["sub_low_index"] - 1
    hi = spec_level["sub_high_index"] - 1
    n = len(lo)
    unkb = np.float32(1000.0)
    len_bp = np.zeros(n, FLOAT3)
    ids = np.zeros(n, INT4)
    accu = np.zeros(n, INT3)
    sub2frag, collect_accu, norm = [], [], []
    for i in range(n):
        n_sub = int(hi[i] - lo[i] + 1)
        v_len = [np.float32(sub_level_soa["len_bp"][lo[i] + j]) / unkb for j in range(n_sub)]
        v_acc = [np.int32(sub_level_soa["n_accu"][lo[i] + j]) for j in range(n_sub)]
        for j, key in zip(range(n_sub), ("x", "y", "z")):
            len_bp[key][i] = v_len[j]
            ids[key][i] = lo[i] + j
            accu[key][i] = v_acc[j]
        ids["w"][i] = n_sub
        collect_accu.extend(v_acc)
        tmp_len = np.array(v_len, dtype=np.float32)
        for j in range(n_sub):
            w_d = np.sum(tmp_len[0:j]) + tmp_len[j] / 2.0
            c_d = np.sum(tmp_len[list(range(n_sub - 1, j, -1))]) + tmp_len[j] / 2.0
            sub2frag.append((i, w_d, c_d, j))
        norm.append(np.sum(v_acc))
    return dict(np_sub_frags_2_frags=np.array(sub2frag, dtype=FLOAT4), np_sub_frags_len_bp=len_bp, np_sub_frags_accu=accu,
                np_sub_frags_id=ids, collect_accu_frags=np.array(collect_accu, dtype=np.float32), norm_vect=np.asmatrix(norm),
                init_n_sub_frags=int(np.sum(hi - lo + 1)))


def assemble_sampler_args(hic_pyr, level, n_iterations=10, is_simu=False, use_rippe=True):
    """The 29 constructor arguments of the sampler (CL:92-125) for resolution `level` of a built pyramid, in order,
    as a dict (vel / pos, OpenGL leftovers, are None).  No GPU involved."""
    lev = hic_pyr.get_level(level)
    sub = hic_pyr.get_level(level - 1)
    spec = hic_pyr.spec_level[str(level)]
    t = create_sub_frags(spec, sub.S_o_A_frags)
    n_frags = lev.n_frags
    new_soa = _with_duplicate_columns(lev.S_o_A_frags)
    new_sub_soa = _with_duplicate_columns(sub.S_o_A_frags)
    collector, dispatcher = _dispatcher(n_frags)
    sub_collector, sub_dispatcher = _dispatcher(len(sub.S_o_A_frags["id"]))
    # create_new_sub_frags SS:725-739: consecutive ids for the sub-fragments of the (possibly duplicated) bins
    rep_ids = np.zeros(n_frags, INT4)
    n_sub = t["np_sub_frags_id"]["w"][new_soa["id_d"]]
    first = np.cumsum(n_sub) - n_sub
    for j, key in enumerate(("x", "y", "z")):
        rep_ids[key] = np.where(n_sub > j, first + j, 0)
    rep_ids["w"] = n_sub
    args = dict(
        use_rippe=use_rippe, S_o_A_frags=new_soa, collector_id_repeats=collector, frag_dispatcher=dispatcher, id_frag_duplicated=[],
        id_frags_blacklisted=[], n_frags=n_frags, n_new_frags=len(new_soa["id"]), init_n_sub_frags=t["init_n_sub_frags"],
        n_new_sub_frags=int(n_sub.sum()), np_rep_sub_frags_id=rep_ids, sub_sampled_sparse_matrix=lev.sparse_mat_csr,
        np_sub_frags_len_bp=t["np_sub_frags_len_bp"], np_sub_frags_id=t["np_sub_frags_id"], np_sub_frags_accu=t["np_sub_frags_accu"],
        np_sub_frags_2_frags=t["np_sub_frags_2_frags"],
        mean_squared_frags_per_bin=np.float32((t["collect_accu_frags"].mean()) ** 2), norm_vect_accu=t["norm_vect"],
        sub_candidates_dup=[], sub_candidates_output_data=[], S_o_A_sub_frags=new_sub_soa, sub_collector_id_repeats=sub_collector,
        sub_frag_dispatcher=sub_dispatcher, sparse_matrix=sub.sparse_mat_csr, mean_value_trans=sub.mean_value_trans,
        n_iterations=n_iterations, is_simu=is_simu, vel=None, pos=None)
    return args, lev, sub


class simulation:
    """SS:27-175 with the MI355X sampler underneath."""

    def __init__(self, name, folder_path, fasta, level, n_iterations, is_simu, use_rippe, thresh_factor=1, output_folder=None,
                 device_id=0):
        from .sampler import sampler as sampler_lib

        self.name = self.data_set = name
        self.use_rippe = use_rippe
        self.str_level, self.str_sub_level = str(level), str(level - 1)
        self.thresh_factor = thresh_factor
        self.fasta = fasta
        self.base_folder = folder_path
        self.output_folder = output_folder if output_folder is not None else os.path.join(os.getcwd(), "results")
        self.n_iterations = n_iterations
        self.select_data_set(name)
        args, self.level, self.sub_level = assemble_sampler_args(self.hic_pyr, level, n_iterations, is_simu, use_rippe)
        self.level.build_seq_per_bin(genome_fasta=self.fasta)
        self.n_frags = args["n_new_frags"]
        self.new_S_o_A_frags, self.new_sub_S_o_A_frags = args["S_o_A_frags"], args["S_o_A_sub_frags"]
        self.sampler = sampler_lib(**args, device_id=device_id, keep_all_scores=False)  # (nobody reads them: CL:1414-1454)
        # SS:157-171
        g = self.sampler.gpu_vect_frags
        g.copy_from_gpu()
        id_start = np.nonzero(g.start_bp == 0)[0]
        max_dist_kb = g.l_cont_bp[id_start].max() / 1000.0
        mean_size_bin_kb = self.new_sub_S_o_A_frags["len_bp"].mean() / 1000.0
        if self.use_rippe and not is_simu:
            self.sampler.estimate_parameters_rippe(max_dist_kb, mean_size_bin_kb / 2.0, False)
        else:
            raise NotImplementedError("only the Rippe model on real data is supported (use_rippe=False is unreachable in the "
                                      "reference as well, SURVEY section 2a)")

    def select_data_set(self, name):  # SS:539-570
        os.makedirs(self.output_folder, exist_ok=True)
        self.hic_pyr = pyr.build_and_filter(self.base_folder, SIZE_PYRAMID, FACTOR, thresh_factor=self.thresh_factor,
                                            output_folder=self.output_folder)
        self.output_folder = os.path.join(self.output_folder, self.data_set, "test_mcmc_" + self.str_level)
        os.makedirs(self.output_folder, exist_ok=True)
        self.new_fasta = os.path.join(self.output_folder, "genome.fasta")
        self.info_frags = os.path.join(self.output_folder, "info_frags.txt")

    def export_new_fasta(self):  # SS:780-782
        self.sampler.gpu_vect_frags.copy_from_gpu()
        self.level.generate_new_fasta(self.sampler.gpu_vect_frags, self.new_fasta, self.info_frags)

    def release(self):
        self.sampler.free_gpu()


class instagraal_class:
    """IG:76-330: the cycle loop and the run's text outputs."""

    TRACES = ("mean_len", "n_contigs", "dist_init_genome", "likelihood", "fact", "slope", "d_max", "d_nuc", "d", "success")

    def __init__(self, name, folder_path, fasta, device, level, n_iterations_em, n_iterations_mcmc, is_simu, scrambled, perform_em,
                 use_rippe, sample_param, thresh_factor, output_folder):
        self.device = device
        self.sample_param = sample_param
        self.simulation = simulation(name, folder_path, fasta, level, n_iterations_em, is_simu, use_rippe, thresh_factor,
                                     output_folder=output_folder, device_id=device)
        self.dt = np.float32(0.01)
        self.collect = {k: [] for k in self.TRACES}
        self.collect_op_sampled, self.collect_id_fA_sampled, self.collect_id_fB_sampled = [], [], []
        self.collect_likelihood_nuisance = []

    def _out(self, name):
        return os.path.join(self.simulation.output_folder, name)

    def _record(self, id_frag, o, dist, op_sampled, id_f_sampled, mean_len, n_contigs):
        c = self.collect
        c["likelihood"].append(o)
        c["n_contigs"].append(n_contigs)
        c["mean_len"].append(mean_len)
        c["dist_init_genome"].append(dist)
        self.collect_op_sampled.append(op_sampled)
        self.collect_id_fB_sampled.append(id_f_sampled)
        self.collect_id_fA_sampled.append(id_frag)

    def full_em(self, n_cycles, n_neighbours, bomb, id_start_sample_param, save_matrix=False):  # IG:196-291
        sampler = self.simulation.sampler
        if bomb:
            sampler.bomb_the_genome()
        list_frags = np.arange(0, sampler.n_new_frags)
        t = 0
        n_iter = n_cycles * sampler.n_new_frags
        for j in range(n_cycles):
            np.random.shuffle(list_frags)
            if self.sample_param and j > id_start_sample_param:
                # step_sampler + step_nuisance_parameters per bin (IG:221-252), the two in flight together on the GPU
                res, tuples = sampler.step_sampler_nuisance_batch(list_frags, n_neighbours, self.dt, t, n_iter)
                for id_frag, r, (fact, d, d_max, d_nuc, slope, lik, success, _) in zip(list_frags, res, tuples):
                    self._record(id_frag, float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]),
                                 np.float32(r["mean_len"]), int(r["n_contigs"]))
                    for k, v in (("fact", fact), ("d", d), ("d_max", d_max), ("d_nuc", d_nuc), ("slope", slope), ("success", success)):
                        self.collect[k].append(v)
                    self.collect_likelihood_nuisance.append(lik)
                t += len(list_frags)
            else:  # no nuisance step between the moves: the whole cycle is one batch (same RNG stream, same results)
                res = sampler.step_sampler_batch(list_frags, n_neighbours)
                for id_frag, r in zip(list_frags, res):
                    self._record(id_frag, float(r["o"]), float(r["dist"]), int(r["op_sampled"]), int(r["id_f_sampled"]),
                                 np.float32(r["mean_len"]), int(r["n_contigs"]))
                t += len(list_frags)
            c = sampler.gpu_vect_frags
            c.copy_from_gpu()
            with open(self._out("save_simu_step_%d.txt" % j), "w") as h:
                for pos, start_bp, id_c, ori in zip(c.pos, c.start_bp, c.id_c, c.ori):
                    h.write(str(pos) + "\t" + str(start_bp) + "\t" + str(id_c) + "\t" + str(ori) + "\n")
            self.simulation.export_new_fasta()
            self.save_behaviour_to_txt()
        self.save_behaviour_to_txt()

    def save_behaviour_to_txt(self):  # IG:293-330
        for k in self.TRACES:
            with open(self._out("list_%s.txt" % k), "w") as h:
                for item in self.collect[k]:
                    h.write("%s\n" % item)
        with open(self._out("list_mutations.txt"), "w") as h:
            h.write("id_fA\tid_fB\tid_mutation\n")
            for a, b, m in zip(self.collect_id_fA_sampled, self.collect_id_fB_sampled, self.collect_op_sampled):
                h.write("%s\t%s\t%s\n" % (a, b, m))


def run_instagraal(hic_folder, reference_fa, output_folder=None, level=4, cycles=100, coverage_std=1, neighborhood=5, device=0,
                   circular=False, bomb=False, pyramid_only=False, save_pickle=False, save_matrix=False, simple=False):
    """IG:502-581 (defaults of cli/main.py: level 4, 100 cycles, 5 neighbours, 1 std).  The three trailing switches of the
    reference's signature (IG:512-514) are accepted: ``save_pickle`` dumps the run object to ``graal.pkl`` as the reference
    tries to (IG:589-594: a warning when it cannot be pickled -- device handles here, h5py handles there); ``save_matrix``
    asks for the matplotlib contact-map previews of ``display_current_matrix`` (IG:279-284), which are outside this path
    (SURVEY 2a): ignored with a warning; ``simple`` calls ``instagraal_class.simple_start``, a method the reference does not
    define (IG:582 raises AttributeError): refused."""
    import warnings

    if simple and not pyramid_only:
        raise NotImplementedError("simple=True: the reference calls instagraal_class.simple_start (instagraal.py:582), which it does not define")
    if save_matrix:
        warnings.warn("save_matrix: the per-cycle contact-map previews (instagraal.py:279-284, matplotlib) are not part of this path; ignored")
    name = os.path.basename(os.path.normpath(str(hic_folder)))
    if pyramid_only:
        root = str(output_folder) if output_folder is not None else os.path.join(os.getcwd(), "results")
        os.makedirs(root, exist_ok=True)
        return pyr.build_and_filter(str(hic_folder), SIZE_PYRAMID, FACTOR, thresh_factor=coverage_std, output_folder=root)
    p2 = instagraal_class(name=name, folder_path=str(hic_folder), fasta=str(reference_fa), device=device, level=level,
                          n_iterations_em=30, n_iterations_mcmc=100, is_simu=False, scrambled=False, perform_em=False, use_rippe=True,
                          sample_param=True, thresh_factor=coverage_std, output_folder=str(output_folder) if output_folder else None)
    if circular:
        # IG:569-570: the flag is applied to the loader's arrays AFTER the sampler copied them to the device, i.e. it
        # has no effect on the run in the reference either (quirk Q14).  Accepted and applied the same (ineffective) way.
        warnings.warn("--circular has no effect on the assembly (as in the reference: instagraal.py:569-570 sets the flag "
                      "after the sampler copied the fragment arrays)")
        p2.simulation.level.S_o_A_frags["circ"] += 1
    p2.full_em(n_cycles=cycles, n_neighbours=neighborhood, bomb=bomb, id_start_sample_param=4)
    if save_pickle:  # IG:589-594
        import pickle

        try:
            with open("graal.pkl", "wb") as h:
                pickle.dump(p2, h)
        except Exception as e:  # ctypes handles do not pickle (the reference: h5py handles)
            warnings.warn("could not pickle the run state: %s" % (e,))
            try:
                os.remove("graal.pkl")
            except OSError:
                pass
    return p2
