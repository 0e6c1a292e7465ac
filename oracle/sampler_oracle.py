"""CPU restatement of the reference sampler's per-move host logic.

TEST INFRASTRUCTURE ONLY.  Follows, call for call and buffer for buffer,
``/root/reference/src/instagraal/cuda_lib_gl_single.py`` ("CL") over the C
kernels of oracle/ig_oracle_*.c:

    step_sampler            CL:1401-1465      modify_gl_cuda_buffer  CL:2715-2881
    perform_mutations       CL:1918-1923      pop_out_pop_in         CL:1642-1778
    transloc                CL:1780-1841      insert_blocks          CL:1843-1916
    slice_sparse_mat        CL:1009-1069      eval_all_sub_likelihood CL:1092-1154
    extract_current_sub_likelihood CL:1156-1191   eval_likelihood    CL:1245-1292
    test_copy_struct        CL:2094-2151      dist_inter_genome      CL:665-716
    setup_distri_frags / return_neighbours    CL:3053-3141
    step_nuisance_parameters CL:2961-3051     bomb_the_genome        CL:1925-1948

State that persists between candidates and moves in the reference (the 24
collector structs, pop/trans scratch structs, ``list_valid_insert``) persists
here too, because stale content is observable (Q4, Q13).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import oracle_lib as ol
from . import rippe_fit as opti
from .oracle_lib import FragStruct, f32, i32, ptr

N_TMP = 24
BLOCK_SUB = 64
N_INSERT_BLOCKS = 6
LIST_SIZE = np.array([1, 3, 5, 10, 20, 50, 200, 200], dtype=np.int32)  # CL:417


class OracleSampler:
    def __init__(self, use_rippe, S_o_A_frags, collector_id_repeats, frag_dispatcher, id_frag_duplicated,
                 id_frags_blacklisted, n_frags, n_new_frags, init_n_sub_frags, n_new_sub_frags, np_rep_sub_frags_id,
                 sub_sampled_sparse_matrix, np_sub_frags_len_bp, np_sub_frags_id, np_sub_frags_accu, np_sub_frags_2_frags,
                 mean_squared_frags_per_bin, norm_vect_accu, sub_candidates_dup, sub_candidates_output_data,
                 S_o_A_sub_frags, sub_collector_id_repeats, sub_frag_dispatcher, sparse_matrix, mean_value_trans,
                 n_iterations, is_simu, vel, pos, mode=ol.MODE_LIBM):
        assert len(sub_candidates_dup) == 0 and len(id_frag_duplicated) == 0, "repeats are dead in the reference"
        self.lib = ol.lib()
        self.mode = mode
        ol.set_mode(mode)
        self.log_e = 0.43429448190325182  # CL:128
        self.n_frags = np.int32(n_frags)
        self.n_new_frags = np.int32(n_new_frags)
        self.init_n_sub_frags = np.int32(init_n_sub_frags)
        self.n_new_sub_frags = np.int32(n_new_sub_frags)
        self.id_frags_blacklisted = id_frags_blacklisted
        self.S_o_A_frags = S_o_A_frags
        self.np_sub_frags_id = np_sub_frags_id
        self.np_sub_frags_2_frags = np.ascontiguousarray(np_sub_frags_2_frags)
        self.mean_len_bp_frags = S_o_A_sub_frags["len_bp"].mean()  # CL:231
        self.mean_value_trans = mean_value_trans
        self.sub_sampled_sparse_matrix = sub_sampled_sparse_matrix
        N, M = int(self.n_new_frags), int(self.n_new_sub_frags)
        self.N, self.M = N, M

        # CL:129, 564-615 -- symmetrise, strict upper triangle, COO row-major
        sym = (sparse_matrix + sparse_matrix.transpose()).tocsr()
        coo = sp.triu(sym.tocoo(), k=1, format="coo")
        order = np.lexsort((coo.col, coo.row))
        self.sp_row = np.ascontiguousarray(coo.row[order], np.int32)
        self.sp_col = np.ascontiguousarray(coo.col[order], np.int32)
        self.sp_dat = np.ascontiguousarray(coo.data[order], np.int32)
        self.n_non_zero = int(self.sp_dat.shape[0])
        self.sub_row = np.zeros(self.n_non_zero, np.int32)
        self.sub_col = np.zeros(self.n_non_zero, np.int32)
        self.sub_dat = np.zeros(self.n_non_zero, np.int32)

        # CL:321-424
        data = dict(S_o_A_frags)
        data["ori"] = np.ones(N, dtype=np.int32)  # CL:537-541
        self.gpu_vect_frags = FragStruct(N, data)
        self.gpu_id_contigs = np.array(S_o_A_frags["id_c"], dtype=np.int32)
        self.collector_gpu_vect_frags = [FragStruct(N) for _ in range(N_TMP)]
        self.pop_gpu_vect_frags = FragStruct(N)
        self.pop_gpu_id_contigs = self.gpu_id_contigs.copy()
        self.trans1_gpu_vect_frags = FragStruct(N)
        self.trans1_gpu_id_contigs = self.gpu_id_contigs.copy()
        self.trans2_gpu_vect_frags = FragStruct(N)
        self.trans2_gpu_id_contigs = self.gpu_id_contigs.copy()
        MT = M * N_TMP
        self.collect_dist = np.ones(MT, np.float32)
        self.collect_id_c = np.ones(MT, np.int32)
        self.collect_s_tot = np.ones(MT, np.float32)
        self.collect_pos = np.ones(MT, np.int32)
        self.collect_len = np.ones(MT, np.int32)
        self.vect_dist = np.ones(M, np.float32)
        self.vect_id_c = np.ones(M, np.int32)
        self.vect_s_tot = np.ones(M, np.float32)
        self.vect_pos = np.ones(M, np.int32)
        self.vect_len = np.ones(M, np.int32)
        self.gpu_counter_select = np.zeros(1, np.int32)
        self.gpu_counter_select_gl = np.zeros(1, np.int32)
        self.gpu_likelihood_on_zeros = np.zeros(1, np.float64)
        self.gpu_likelihood_on_zeros_nuis = np.zeros(1, np.float64)
        self.gpu_vect_likelihood_z = np.zeros(N_TMP, np.float64)
        self.gpu_n_vals_intra = np.zeros(1, np.int32)
        self.gpu_all_n_vals_intra = np.zeros(N_TMP, np.int32)
        self.n_pixl_sub_mat = int(self.init_n_sub_frags) * (int(self.init_n_sub_frags) - 1) / 2  # CL:366
        self.gpu_n_pixl_sub_mat = np.array([self.n_pixl_sub_mat], np.float64)
        self.gpu_list_uniq_mutations = np.zeros(N_TMP, np.int32)
        self.gpu_n_uniq = np.zeros(1, np.int32)
        self.gpu_sub_sp_block_indptr = np.zeros(max(self.n_non_zero, 1), np.int32)
        self.gpu_info_blocks = np.zeros(self.n_non_zero // BLOCK_SUB + 1, ol.INT3)
        self.gpu_sub_vect_likelihood_nz = np.zeros(N_TMP, np.float64)
        self.gpu_curr_likelihood_nz_extract = np.zeros(1, np.float64)
        self.gpu_all_scores = np.zeros(N_TMP, np.float64)
        self.gpu_curr_likelihood_nz = np.zeros(1, np.float64)
        self.gpu_curr_likelihood_nz_nuis = np.zeros(1, np.float64)
        self.gpu_uniq_id_c = np.zeros(M, np.int32)
        self.gpu_uniq_len = np.zeros(M, np.int32)
        self.gpu_old_2_new_id_c = np.zeros(int(M + M / 10), np.int32)
        self.max_bounds_insert = LIST_SIZE[:N_INSERT_BLOCKS].max() * np.int32(
            np.round(S_o_A_frags["sub_len"].mean()) + 1)  # CL:418-420
        self.gpu_list_valid_insert = np.zeros(N_INSERT_BLOCKS * 2, np.int32)
        self.gpu_list_bounds = np.array(LIST_SIZE[:N_INSERT_BLOCKS], np.int32)
        self.gpu_list_f_upstream = np.zeros(N_INSERT_BLOCKS, np.int32)
        self.gpu_list_f_downstream = np.zeros(N_INSERT_BLOCKS, np.int32)

        # CL:269-276
        self.np_init_prev = np.copy(S_o_A_frags["prev"])
        self.np_init_next = np.copy(S_o_A_frags["next"])
        self.np_init_orientable = np.array(
            [np_sub_frags_id[S_o_A_frags["id_d"][i]]["w"] > 1 for i in range(N)], dtype=np.int32)
        self.np_init_ori = np.ones(N, dtype=np.int32)
        self.dt = np.float32(0.01)
        self.setup_distri_frags()
        self.param_simu = None
        self.param_simu_test = None
        self.likelihood_t = 0.0
        self.n_contigs = 0
        self.mean_length_contigs = 0.0
        self.trace = None  # optional list collecting per-move records

    # ------------------------------------------------------------------ params
    def set_param_simu(self, p):
        """p: dict or structured array with the 8 fields of KA:91-100"""
        if isinstance(p, dict):
            arr = np.zeros(1, ol.PARAM_DTYPE)
            for k in arr.dtype.names:
                arr[k] = np.float32(p[k])
            p = arr
        self.param_simu = np.array(p, dtype=ol.PARAM_DTYPE).reshape(1)
        self.param_simu_test = self.param_simu.copy()

    def mean_kb(self):
        return np.float32(self.mean_len_bp_frags / 1000.0)

    # ------------------------------------------------------------- neighbours
    def setup_distri_frags(self):  # CL:3053-3101
        self.sym_sub_sampled_sparse_matrix = (self.sub_sampled_sparse_matrix + self.sub_sampled_sparse_matrix.T).tocsr()
        m = self.sym_sub_sampled_sparse_matrix
        self.distri_frags = {}
        fact = 3.0
        for i in range(int(self.n_frags)):
            s, e = m.indptr[i], m.indptr[i + 1]
            vk, yk = m.data[s:e], m.indices[s:e]
            het = np.nonzero(yk != i)[0]
            xk = np.copy(yk)[het]
            dat = np.float32(np.copy(vk)[het]) * fact
            if dat.sum() > 0:
                pk = dat / np.linalg.norm(dat, 1)
            else:
                tmp = np.ones_like(dat, dtype=np.float32)
                pk = tmp / tmp.sum()
            if len(xk) > 0:
                self.distri_frags[i] = dict(distri="ok", xk=xk, pk=pk)
            else:
                self.distri_frags[i] = dict(distri=None)

    def return_neighbours(self, id_fA, delta0):  # CL:3103-3141 (repeat branches are dead)
        ori_id = self.gpu_vect_frags.id_d[id_fA]
        d = self.distri_frags[ori_id]
        if d["distri"] is not None:
            distri = d["pk"]
            n_max = min(delta0, np.nonzero(distri != 0)[0].shape[0])
            init_id = np.random.choice(d["xk"], n_max, p=distri, replace=False)
        else:
            init_id = np.random.choice(self.n_frags, delta0, replace=False)
        return [int(e) for e in init_id if e not in self.id_frags_blacklisted]

    # ---------------------------------------------------------- coordinates
    def fill_dist_single(self):  # CL:936-970
        self.lib.igo_uni_fill_vect_dist(ptr(self.np_sub_frags_2_frags), ptr(self.gpu_vect_frags), ptr(self.vect_dist),
                                        ptr(self.vect_id_c), ptr(self.vect_s_tot), ptr(self.vect_pos), ptr(self.vect_len),
                                        i32(self.init_n_sub_frags))

    def fill_dist_all_mut(self):  # CL:898-934
        for id_mut in range(N_TMP):
            self.lib.igo_fill_vect_dist(ptr(self.np_sub_frags_2_frags), ptr(self.collector_gpu_vect_frags[id_mut]),
                                        ptr(self.collect_dist), ptr(self.collect_id_c), ptr(self.collect_s_tot),
                                        ptr(self.collect_pos), ptr(self.collect_len), i32(self.init_n_sub_frags), i32(id_mut))

    # ------------------------------------------------------------ likelihood
    def approx_single_likelihood_on_zeros(self, params=None, nuis=False):  # CL:718-802
        out = self.gpu_likelihood_on_zeros_nuis if nuis else self.gpu_likelihood_on_zeros
        out.fill(0)
        self.gpu_n_vals_intra.fill(0)
        p = self.param_simu_test if nuis else self.param_simu
        if nuis:
            mean = np.float32(self.mean_len_bp_frags / 1000.0)
        else:
            # CL:743 passes np.int32(mean) where the kernel takes a float (quirk Q8): the int bit
            # pattern is read as a float.  The value never reaches a score.
            mean = np.array([np.int32(self.mean_len_bp_frags / 1000.0)], np.int32).view(np.float32)[0]
        self.lib.igo_eval_likelihood_on_zero(ptr(self.vect_id_c), ptr(self.vect_s_tot), ptr(self.vect_pos),
                                             ptr(self.vect_len), ptr(p), f32(mean), ptr(out), ptr(self.gpu_n_vals_intra),
                                             i32(self.init_n_sub_frags))
        val_intra = out[0] * self.log_e
        n_vals_intra = self.gpu_n_vals_intra[0]
        val_inter = self.log_e * (self.n_pixl_sub_mat - n_vals_intra) * -1.0 * p["v_inter"][0]
        return val_intra + val_inter

    def _full_nz(self, out, params):
        out.fill(0.0)
        self.lib.igo_evaluate_likelihood_sparse(ptr(self.sp_dat), ptr(self.sp_row), ptr(self.sp_col), ptr(params),
                                                f32(self.mean_kb()), ptr(self.vect_dist), ptr(self.vect_id_c),
                                                ptr(self.vect_s_tot), ptr(self.vect_pos), ptr(self.vect_len), ptr(out),
                                                C.c_int64(self.n_non_zero))

    def eval_likelihood_init(self):  # CL:1193-1243
        self.fill_dist_single()
        self.curr_likelihood_on_z = self.approx_single_likelihood_on_zeros()
        self._full_nz(self.gpu_curr_likelihood_nz, self.param_simu)
        self.likelihood_t = self.gpu_curr_likelihood_nz.copy() + self.curr_likelihood_on_z

    def eval_likelihood(self):  # CL:1245-1292
        self.fill_dist_single()
        self.curr_likelihood_on_z = self.approx_single_likelihood_on_zeros()
        self._full_nz(self.gpu_curr_likelihood_nz, self.param_simu)

    def eval_likelihood_4_nuisance(self):  # CL:1296-1344
        self.curr_likelihood_on_z_nuis = self.approx_single_likelihood_on_zeros(nuis=True)
        self._full_nz(self.gpu_curr_likelihood_nz_nuis, self.param_simu_test)
        self.curr_likelihood_nuis = self.gpu_curr_likelihood_nz_nuis.copy() + self.curr_likelihood_on_z_nuis
        return self.curr_likelihood_nuis

    def slice_sparse_mat(self, id_ctg1, id_ctg2, id_fragA, id_fragB):  # CL:1009-1069
        self.gpu_counter_select.fill(0)
        self.lib.igo_slice_sp_mat(ptr(self.sp_dat), ptr(self.sp_row), ptr(self.sp_col), ptr(self.gpu_vect_frags),
                                  ptr(self.vect_id_c), ptr(self.vect_pos), ptr(self.sub_row), ptr(self.sub_col),
                                  ptr(self.sub_dat), i32(id_ctg1), i32(id_ctg2), i32(id_fragA), i32(id_fragB),
                                  i32(self.max_bounds_insert), ptr(self.gpu_counter_select), C.c_int64(self.n_non_zero))
        self.n_sub_vals = int(self.gpu_counter_select[0])
        n = self.n_sub_vals
        idx = np.argsort(self.sub_row[:n], kind="stable")  # CL:45-55
        self.sub_row[:n] = self.sub_row[:n][idx]
        self.sub_col[:n] = self.sub_col[:n][idx]
        self.sub_dat[:n] = self.sub_dat[:n][idx]
        self.gpu_counter_select.fill(0)
        self.lib.igo_prepare_sparse_call(ptr(self.sub_row), ptr(self.gpu_info_blocks), ptr(self.gpu_sub_sp_block_indptr),
                                         ptr(self.gpu_counter_select), i32(n))

    def extract_current_sub_likelihood(self):  # CL:1156-1191
        self.gpu_curr_likelihood_nz_extract.fill(0.0)
        self.lib.igo_extract_sub_likelihood(ptr(self.sub_dat), ptr(self.gpu_info_blocks), ptr(self.gpu_sub_sp_block_indptr),
                                            ptr(self.sub_row), ptr(self.sub_col), ptr(self.param_simu), f32(self.mean_kb()),
                                            ptr(self.vect_dist), ptr(self.vect_id_c), ptr(self.vect_s_tot),
                                            ptr(self.vect_pos), ptr(self.vect_len), ptr(self.gpu_curr_likelihood_nz_extract),
                                            i32(self.n_sub_vals), i32(self.init_n_sub_frags))
        if self.mode == ol.MODE_DET:
            hi, lo = ol.last_limbs()
            self.last_extract_limbs = (int(hi[N_TMP]), int(lo[N_TMP]))

    def approx_all_likelihood_on_zeros(self):  # CL:848-896
        self.gpu_vect_likelihood_z.fill(0)
        self.gpu_all_n_vals_intra.fill(0)
        self.lib.igo_eval_all_likelihood_on_zero_1st(
            ptr(self.collect_id_c), ptr(self.collect_s_tot), ptr(self.collect_pos), ptr(self.collect_len),
            ptr(self.param_simu), f32(self.mean_kb()), ptr(self.gpu_list_uniq_mutations), ptr(self.gpu_n_uniq),
            ptr(self.gpu_vect_likelihood_z), ptr(self.gpu_all_n_vals_intra), i32(self.init_n_sub_frags))
        if self.mode == ol.MODE_DET:
            self.last_z_limbs = ol.last_limbs()
        self.lib.igo_eval_all_likelihood_on_zero_2nd(ptr(self.gpu_list_uniq_mutations), ptr(self.gpu_n_uniq),
                                                     ptr(self.param_simu), ptr(self.gpu_vect_likelihood_z),
                                                     ptr(self.gpu_all_n_vals_intra), ptr(self.gpu_n_pixl_sub_mat))

    def eval_all_sub_likelihood(self):  # CL:1092-1154
        self.fill_dist_all_mut()
        self.approx_all_likelihood_on_zeros()
        self.gpu_sub_vect_likelihood_nz.fill(0.0)
        self.gpu_all_scores.fill(0.0)
        self.lib.igo_eval_sub_likelihood(
            ptr(self.sub_dat), ptr(self.gpu_info_blocks), ptr(self.gpu_sub_sp_block_indptr), ptr(self.sub_row),
            ptr(self.sub_col), ptr(self.param_simu), f32(self.mean_kb()), ptr(self.collect_dist), ptr(self.collect_id_c),
            ptr(self.collect_s_tot), ptr(self.collect_pos), ptr(self.collect_len), ptr(self.gpu_list_uniq_mutations),
            ptr(self.gpu_n_uniq), ptr(self.gpu_sub_vect_likelihood_nz), i32(self.n_sub_vals), i32(self.init_n_sub_frags))
        if self.mode == ol.MODE_DET:
            self.last_nz_limbs = ol.last_limbs()
        self.lib.igo_eval_all_scores(ptr(self.gpu_list_uniq_mutations), ptr(self.gpu_n_uniq), ptr(self.gpu_vect_likelihood_z),
                                     ptr(self.gpu_sub_vect_likelihood_nz), ptr(self.gpu_curr_likelihood_nz_extract),
                                     ptr(self.gpu_curr_likelihood_nz), ptr(self.gpu_all_scores))
        return self.gpu_all_scores.copy()

    # -------------------------------------------------------------- mutations
    def extract_uniq_mutations(self, id_fi, id_fj, flip_eject):  # CL:1499-1519
        self.gpu_list_uniq_mutations.fill(0)
        self.lib.igo_extract_uniq_mutations(ptr(self.gpu_vect_frags), i32(id_fi), i32(id_fj),
                                            ptr(self.gpu_list_uniq_mutations), ptr(self.gpu_list_valid_insert),
                                            ptr(self.gpu_n_uniq), i32(flip_eject))

    def pop_out_pop_in(self, id_f_pop, id_f_ins, mode, max_id):  # CL:1642-1778
        N = self.n_new_frags
        self.lib.igo_pop_out_frag(ptr(self.pop_gpu_vect_frags), ptr(self.gpu_vect_frags), ptr(self.pop_gpu_id_contigs),
                                  i32(id_f_pop), i32(max_id), i32(N))
        max_id2 = np.int32(self.pop_gpu_id_contigs.max())
        out = self.collector_gpu_vect_frags[mode]
        if mode == 0:
            self.lib.igo_simple_copy(ptr(out), ptr(self.pop_gpu_vect_frags), i32(N))
        elif mode == 1:
            self.lib.igo_flip_frag(ptr(out), ptr(self.gpu_vect_frags), i32(id_f_pop), i32(N))
        else:
            fn = {2: self.lib.igo_pop_in_frag_1, 3: self.lib.igo_pop_in_frag_1, 4: self.lib.igo_pop_in_frag_2,
                  5: self.lib.igo_pop_in_frag_2, 6: self.lib.igo_pop_in_frag_3, 7: self.lib.igo_pop_in_frag_3}[mode]
            ori = 1 if mode % 2 == 0 else -1
            fn(ptr(out), ptr(self.pop_gpu_vect_frags), i32(id_f_pop), i32(id_f_ins), i32(max_id2), i32(ori), i32(N))

    def transloc(self, id_fA, id_fB, max_id):  # CL:1780-1841
        N = self.n_new_frags
        mode = 0
        for upA in range(2):
            self.lib.igo_split_contig(ptr(self.trans1_gpu_vect_frags), ptr(self.gpu_vect_frags),
                                      ptr(self.trans1_gpu_id_contigs), i32(id_fA), i32(upA), i32(max_id), i32(N))
            for upB in range(2):
                max_id1 = np.int32(self.trans1_gpu_id_contigs.max())
                self.lib.igo_split_contig(ptr(self.trans2_gpu_vect_frags), ptr(self.trans1_gpu_vect_frags),
                                          ptr(self.trans2_gpu_id_contigs), i32(id_fB), i32(upB), i32(max_id1), i32(N))
                max_id2 = np.int32(self.trans2_gpu_id_contigs.max())
                self.lib.igo_paste_contigs(ptr(self.collector_gpu_vect_frags[8 + mode]), ptr(self.trans2_gpu_vect_frags),
                                           i32(id_fA), i32(id_fB), i32(max_id2), i32(N))
                mode += 1

    def insert_blocks(self, id_fA, id_fB, max_id):  # CL:1843-1916
        N = self.n_new_frags
        self.gpu_list_valid_insert.fill(-1)
        self.gpu_list_f_upstream.fill(-1)
        self.gpu_list_f_downstream.fill(-1)
        self.lib.igo_get_bounds(ptr(self.gpu_vect_frags), i32(id_fA), i32(id_fB), ptr(self.gpu_list_valid_insert),
                                ptr(self.gpu_list_bounds), ptr(self.gpu_list_f_upstream), ptr(self.gpu_list_f_downstream),
                                i32(N_INSERT_BLOCKS), i32(N))
        idm = 0
        for i in range(N_INSERT_BLOCKS):
            for j in (1, 0):
                lb = self.gpu_list_f_upstream if j == 1 else self.gpu_list_f_downstream
                self.lib.igo_extract_block(ptr(self.trans1_gpu_vect_frags), ptr(self.gpu_vect_frags),
                                           ptr(self.trans1_gpu_id_contigs), i32(id_fA), ptr(lb), i32(i), i32(j), i32(max_id),
                                           i32(N))
                self.lib.igo_insert_block(ptr(self.collector_gpu_vect_frags[12 + idm]), ptr(self.trans1_gpu_vect_frags),
                                          ptr(self.gpu_vect_frags), i32(id_fA), i32(id_fB), ptr(lb),
                                          ptr(self.gpu_list_valid_insert), i32(idm), i32(i), i32(j), i32(N))
                idm += 1

    def perform_mutations(self, id_fA, id_fB, max_id):  # CL:1918-1923
        for mode in range(8):
            self.pop_out_pop_in(id_fA, id_fB, mode, max_id)
        self.transloc(id_fA, id_fB, max_id)
        self.insert_blocks(id_fA, id_fB, max_id)

    def test_copy_struct(self, id_fA, id_f_sampled, mode, max_id):  # CL:2094-2151
        if mode < 8:
            self.pop_out_pop_in(id_fA, id_f_sampled, mode, max_id)
        elif mode < 12:
            self.transloc(id_fA, id_f_sampled, max_id)
        else:
            self.insert_blocks(id_fA, id_f_sampled, max_id)
        self.lib.igo_copy_struct(ptr(self.gpu_vect_frags), ptr(self.collector_gpu_vect_frags[mode]),
                                 ptr(self.gpu_id_contigs), i32(self.n_new_frags))

    # ---------------------------------------------------- contig bookkeeping
    def modify_gl_cuda_buffer(self, id_fi, dt):  # CL:2715-2881 (viewer parts dropped)
        N = self.n_new_frags
        self.gpu_counter_select_gl.fill(0)
        self.lib.igo_select_uniq_id_c(ptr(self.gpu_vect_frags), ptr(self.gpu_uniq_id_c), ptr(self.gpu_uniq_len),
                                      ptr(self.gpu_counter_select_gl), i32(N))
        self.n_contigs = self.gpu_counter_select_gl[0]
        n = int(self.n_contigs)
        idx = np.argsort(-self.gpu_uniq_len[:n].astype(np.int64), kind="stable")  # CL:69-77
        self.gpu_uniq_len[:n] = self.gpu_uniq_len[:n][idx]
        self.gpu_uniq_id_c[:n] = self.gpu_uniq_id_c[:n][idx]
        self.cpu_length_contigs = np.float32(self.gpu_uniq_len)
        self.mean_length_contigs = self.cpu_length_contigs[:n].mean()
        self.lib.igo_make_old_2_new_id_c(ptr(self.gpu_uniq_id_c), ptr(self.gpu_old_2_new_id_c), i32(n))
        max_id = np.float32(self.n_contigs - 1)
        self.lib.igo_renumber_id_c(ptr(self.gpu_vect_frags), ptr(self.gpu_old_2_new_id_c), ptr(self.gpu_id_contigs),
                                   f32(max_id), i32(N))
        # prefix_sum over gpu_uniq_len (CL:2811) only feeds the dead viewer
        return np.int32(max_id)

    def bomb_the_genome(self):  # CL:1925-1948
        a = np.arange(0, self.n_new_frags, dtype=np.int32)
        np.random.shuffle(a)
        self.lib.igo_explode_genome(ptr(self.gpu_vect_frags), ptr(a), i32(self.n_new_frags))
        self.modify_gl_cuda_buffer(0, self.dt)

    def dist_inter_genome(self, g1):  # CL:665-716, vectorised without changing a single comparison
        N = int(self.n_new_frags)
        black = np.zeros(N, bool)
        black[list(self.id_frags_blacklisted)] = True
        p0, n0, o0 = self.np_init_prev, self.np_init_next, self.np_init_ori
        p1, n1, o1 = g1.prev.copy(), g1.next.copy(), g1.ori
        orientable = self.np_init_orientable.astype(bool)
        d = 3.0 * (N - int(black.sum()))
        norm = d
        credit = np.zeros(N, np.float64)
        credit += ((p1 == p0) & (n1 == n0)) | ((p1 == n0) & (n1 == p0))
        flip = orientable & (o0 != o1)
        swap = np.where(flip, -1, 1)
        p1s = np.where(flip, n1, p1)
        n1s = np.where(flip, p1, n1)
        for t0, t1 in ((p0, p1s), (n0, n1s)):
            same = orientable & (t0 == t1)
            end = same & (t0 == -1)
            safe = np.where(t1 >= 0, t1, 0)
            non_or = same & (t0 != -1) & ~orientable[safe]
            both = same & (t0 != -1) & orientable[safe]
            credit += end * 1.0 + non_or * 1.0 + both * 0.5
            credit += (both & (o0[np.where(t0 >= 0, t0, 0)] == swap * o1[safe])) * 0.5
        no = ~orientable
        credit += no & ((p1 == p0) | (p1 == n0))
        credit += no & ((n1 == n0) | (n1 == p0))
        d -= credit[~black].sum()
        return d / norm

    # ------------------------------------------------------------------ moves
    def step_sampler(self, id_frag, n_neighbours, dt, candidates=None):  # CL:1401-1465
        if candidates is None:
            self.candidates = self.return_neighbours(id_frag, n_neighbours)
        else:
            self.candidates = list(candidates)
        self.candidates.sort()
        n = len(self.candidates)
        self.fill_dist_single()
        self.eval_likelihood()
        host = self.gpu_vect_frags.as_dict()  # copy_from_gpu(): host mirror taken BEFORE renumbering
        id_ctg_a = host["id_c"][id_frag]
        self.all_scores = np.zeros(N_TMP * n, dtype=np.float64)
        max_id = self.modify_gl_cuda_buffer(id_frag, dt)
        flip_eject = 1
        rec = None
        if self.trace is not None:
            rec = dict(id_frag=int(id_frag), candidates=list(self.candidates), uniq=[], n_sub_vals=[],
                       curr_nz=float(self.gpu_curr_likelihood_nz[0]), extract=[], nz=[], z=[])
        for i, id_cand in enumerate(self.candidates):
            self.extract_uniq_mutations(id_frag, id_cand, flip_eject)
            self.perform_mutations(id_frag, id_cand, max_id)
            id_ctg_b = host["id_c"][id_cand]
            self.slice_sparse_mat(id_ctg_a, id_ctg_b, id_frag, id_cand)
            self.extract_current_sub_likelihood()
            self.all_scores[i * N_TMP:(i + 1) * N_TMP] = self.eval_all_sub_likelihood()
            flip_eject = 0
            if rec is not None:
                nu = int(self.gpu_n_uniq[0])
                rec["uniq"].append([int(v) for v in self.gpu_list_uniq_mutations[:nu]])
                rec["n_sub_vals"].append(int(self.n_sub_vals))
                rec["extract"].append(float(self.gpu_curr_likelihood_nz_extract[0]))
                rec["nz"].append(self.gpu_sub_vect_likelihood_nz.copy())
                rec["z"].append(self.gpu_vect_likelihood_z.copy())
        scores_ok = np.copy(self.all_scores)
        scores_ok[scores_ok == 0] = -np.inf
        max_score = scores_ok.max()
        thresh_overflow = 30
        filtered = scores_ok - (max_score - thresh_overflow)
        filtered[filtered < 0] = 0
        global_id = int(np.argmax(filtered))
        id_f_sampled = self.candidates[int(global_id / N_TMP)]
        op_sampled = global_id % N_TMP
        self.test_copy_struct(id_frag, id_f_sampled, op_sampled, max_id)
        self.modify_gl_cuda_buffer(id_frag, dt)
        o = self.all_scores[global_id]
        self.o = o
        dist = self.dist_inter_genome(self.gpu_vect_frags)
        self.likelihood_t = o
        if rec is not None:
            rec.update(scores=self.all_scores.copy(), op=int(op_sampled), id_f_sampled=int(id_f_sampled), o=float(o),
                       dist=float(dist), n_contigs=int(self.n_contigs), mean_len=float(self.mean_length_contigs),
                       valid_after=self.gpu_list_valid_insert.copy())
            self.trace.append(rec)
        return (o, dist, op_sampled, id_f_sampled, self.mean_length_contigs, self.n_contigs)

    # -------------------------------------------------------------- nuisance
    def step_nuisance_parameters(self, dt, t, n_step):  # CL:2961-3051
        curr_param = np.copy(self.param_simu)
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = curr_param[0]
        self.sigma_fact = 10 ** (np.log10(fact) - 2)
        self.sigma_slope = 0.005
        self.sigma_d_max = 100
        self.sigma_d_nuc = 10 ** (np.log10(d_nuc) - 2)
        id_modif = np.random.choice(4)
        if id_modif == 0:
            new_fact = fact + np.random.normal(loc=0.0, scale=self.sigma_fact)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, new_fact], d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, new_fact, d_nuc)]
        elif id_modif == 1:
            new_slope = slope + np.random.normal(loc=0.0, scale=self.sigma_slope)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, new_slope, d, fact], d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, new_slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, new_slope, d, new_d_max, fact, d_nuc)]
        elif id_modif == 2:
            new_d_max = d_max + np.random.normal(loc=0.0, scale=self.sigma_d_max)
            new_d_nuc = opti.peval(new_d_max, [kuhn, lm, slope, d, fact])  # 5 params where 4 are read: quirk Q12
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        else:
            if self.sigma_d_nuc <= 0:
                new_d_nuc = d_nuc
            else:
                new_d_nuc = d_nuc + np.random.normal(loc=0.0, scale=self.sigma_d_nuc)
            new_d_max = opti.estimate_max_dist_intra_nuis([kuhn, lm, slope, d, fact], new_d_nuc, d_max)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        out = np.array(out, dtype=ol.PARAM_DTYPE)
        self.param_simu_test = out
        self.likelihood_nuis = self.eval_likelihood_4_nuisance()
        ratio = np.exp((self.likelihood_nuis - self.likelihood_t) / 1.0)
        u = np.random.rand()
        success = 0
        if ratio >= u:
            success = 1
            self.param_simu = out
            self.likelihood_t = self.likelihood_nuis
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = self.param_simu[0]
        y_rippe = opti.peval(self.bins, [kuhn, lm, slope, d, fact]) if hasattr(self, "bins") else None
        return (fact, d, d_max, d_nuc, slope, self.likelihood_t, success, y_rippe)
