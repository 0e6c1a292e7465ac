#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f) rows f1/f4: the reference's OWN pyramid builder, loader, sampler-argument assembly and
FASTA/info_frags writer (pyramid_sparse.py, simu_single.py), run unmodified in the authoring container over
tools/fake_h5py (h5py is absent) and tools/fake_pycuda, on the seeded synthetic input folder of
instagraal_amd.synth.write_text_dataset.  Stored: every text file of the two pyramids (bytes), the sparse level
matrices, the loader's structures, the 29 constructor arguments `simulation` hands to the sampler, and the writer's
outputs for a scrambled genome.  Only data is stored; no reference source enters the repo.

usage: python tools/gen_golden_pyramid.py [--out tests/golden/pyramid_small.npz]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools", "fake_pycuda"))
sys.path.insert(0, os.path.join(ROOT, "tools", "fake_h5py"))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference/src")

LEVEL = 2
DATASET = dict(n_contigs=10, mean_frags=110, seed=7, contacts_per_frag=40)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "pyramid_small.npz"))
    a = ap.parse_args()
    import matplotlib

    matplotlib.use("Agg")
    from instagraal_amd import synth
    from oracle import oracle_lib as ol

    ol.set_mode(1)
    work = tempfile.mkdtemp()
    os.chdir(work)  # sparsity plots and the log file land in the CWD
    data = os.path.join(work, "data")
    out = os.path.join(work, "out")
    synth.write_text_dataset(data, **DATASET)

    from instagraal import simu_single as SS

    captured = {}
    ref_sampler = SS.sampler_lib

    class Recorder(ref_sampler):
        def __init__(self, *args):
            captured["args"] = args
            super().__init__(*args)

    SS.sampler_lib = Recorder
    np.random.seed(3)
    sim = SS.simulation("synth", data, os.path.join(data, "genome.fa"), LEVEL, 10, False, True, 1, out)

    blob = {}
    # ---- every text file of both pyramids
    pyr_root = os.path.join(out, "pyramids")
    names = []
    for dirpath, _, files in sorted(os.walk(pyr_root)):
        for f in sorted(files):
            if f.endswith(".txt"):
                rel = os.path.relpath(os.path.join(dirpath, f), pyr_root)
                names.append(rel)
                blob["txt/" + rel] = np.frombuffer(open(os.path.join(dirpath, f), "rb").read(), dtype=np.uint8)
    blob["txt_names"] = np.array(names)
    # ---- sparse level matrices (HDF5 payload)
    for lv in range(9):
        blob["h5/%d/data" % lv] = np.array(sim.hic_pyr.data[str(lv)]["data"][:, :], dtype=np.int32)
        blob["h5/%d/nfrags" % lv] = np.array(sim.hic_pyr.data[str(lv)]["nfrags"][:, :], dtype=np.int32)
    # ---- loader structures of the two levels in use
    for tag, lev in (("level", sim.level), ("sub_level", sim.sub_level)):
        for k, v in lev.S_o_A_frags.items():
            blob["%s/soa/%s" % (tag, k)] = np.asarray(v)
        blob["%s/n_frags" % tag] = np.int64(lev.n_frags)
        blob["%s/mean_value_trans" % tag] = np.float64(lev.mean_value_trans)
        blob["%s/frags_init_contigs" % tag] = np.array(lev.frags_init_contigs)
        csr = lev.sparse_mat_csr
        blob["%s/csr_indptr" % tag], blob["%s/csr_indices" % tag], blob["%s/csr_data" % tag] = csr.indptr, csr.indices, csr.data
    # ---- the sampler's constructor arguments (CL:92-125 order)
    arg_names = ["use_rippe", "S_o_A_frags", "collector_id_repeats", "frag_dispatcher", "id_frag_duplicated", "id_frags_blacklisted",
                 "n_frags", "n_new_frags", "init_n_sub_frags", "n_new_sub_frags", "np_rep_sub_frags_id", "sub_sampled_sparse_matrix",
                 "np_sub_frags_len_bp", "np_sub_frags_id", "np_sub_frags_accu", "np_sub_frags_2_frags", "mean_squared_frags_per_bin",
                 "norm_vect_accu", "sub_candidates_dup", "sub_candidates_output_data", "S_o_A_sub_frags", "sub_collector_id_repeats",
                 "sub_frag_dispatcher", "sparse_matrix", "mean_value_trans", "n_iterations", "is_simu", "vel", "pos"]
    args = captured["args"]
    assert len(args) == len(arg_names), (len(args), len(arg_names))
    for n, v in zip(arg_names, args):
        if n in ("vel", "pos"):
            continue  # OpenGL leftovers (random)
        if isinstance(v, dict):
            for k, x in v.items():
                blob["arg/%s/%s" % (n, k)] = np.asarray(x)
        elif hasattr(v, "indptr"):
            c = v.tocsr()
            blob["arg/%s/indptr" % n], blob["arg/%s/indices" % n], blob["arg/%s/data" % n] = c.indptr, c.indices, c.data
            blob["arg/%s/shape" % n] = np.array(c.shape)
        else:
            blob["arg/%s" % n] = np.asarray(v)
    # ---- what simulation.__init__ derives for estimate_parameters_rippe (SS:157-171)
    g = sim.sampler.gpu_vect_frags
    g.copy_from_gpu()
    id_start = np.nonzero(g.start_bp == 0)[0]
    blob["max_dist_kb"] = np.float64(g.l_cont_bp[id_start].max() / 1000.0)
    blob["mean_size_bin_kb"] = np.float64(sim.new_sub_S_o_A_frags["len_bp"].mean() / 1000.0) if hasattr(sim, "new_sub_S_o_A_frags") else np.float64(0)
    # ---- the writers on a scrambled genome: reverse every second contig's order, flip a third of the bins
    rng = np.random.default_rng(11)
    n = len(g.id_c)
    pos, ori, id_c = g.pos.copy(), g.ori.copy(), g.id_c.copy()
    for c in np.unique(id_c)[::2]:
        m = np.nonzero(id_c == c)[0]
        pos[m] = pos[m].max() - pos[m]
    ori[rng.random(n) < 0.33] = -1
    # merge the two last contigs into one (ids as modify_gl_cuda_buffer would leave them are arbitrary here)
    cs = np.unique(id_c)
    a_, b_ = cs[-2], cs[-1]
    ma, mb_ = np.nonzero(id_c == a_)[0], np.nonzero(id_c == b_)[0]
    pos[mb_] += len(ma)
    id_c[mb_] = a_
    g.pos, g.ori, g.id_c = pos, ori, id_c
    blob["scr/pos"], blob["scr/ori"], blob["scr/id_c"] = pos, ori, id_c
    blob["scr/id_d"], blob["scr/activ"] = np.asarray(g.id_d), np.asarray(g.activ)
    fa, info = os.path.join(work, "new.fa"), os.path.join(work, "info_frags.txt")
    sim.level.generate_new_fasta(g, fa, info)
    blob["scr/fasta"] = np.frombuffer(open(fa, "rb").read(), dtype=np.uint8)
    blob["scr/info_frags"] = np.frombuffer(open(info, "rb").read(), dtype=np.uint8)
    blob["level"] = np.int64(LEVEL)
    blob["dataset"] = np.array([DATASET[k] for k in ("n_contigs", "mean_frags", "seed", "contacts_per_frag")])
    np.savez_compressed(a.out, **blob)
    print("wrote", a.out, "%.1f KB" % (os.path.getsize(a.out) / 1024.0), "level n_frags", sim.level.n_frags, "sub", sim.sub_level.n_frags,
          "files", len(names))


if __name__ == "__main__":
    main()
