"""the pyramid builder, the loader and the FASTA / info_frags writers against goldens made
by the reference's own pyramid_sparse.py / simu_single.py (tools/gen_golden_pyramid.py) on the seeded synthetic input
folder of synth.write_text_dataset: every text file byte for byte, every sparse level matrix, every loader array."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    from instagraal_amd import pyramid, synth

    g = np.load(os.path.join(GOLDEN, "pyramid_small.npz"))
    work = tmp_path_factory.mktemp("pyr")
    data, out = str(work / "data"), str(work / "out")
    nc, mf, seed, cpf = [int(x) for x in g["dataset"]]
    synth.write_text_dataset(data, n_contigs=nc, mean_frags=mf, seed=seed, contacts_per_frag=cpf)
    os.makedirs(out)
    pyr = pyramid.build_and_filter(data, 9, 3, thresh_factor=1, output_folder=out)
    return g, data, out, pyr


def test_builder_text_files_byte_for_byte(built):
    g, data, out, pyr = built
    root = os.path.join(out, "pyramids")
    for rel in g["txt_names"]:
        rel = str(rel)
        want = g["txt/" + rel].tobytes()
        got = open(os.path.join(root, rel), "rb").read()
        assert got == want, rel


def test_sparse_level_matrices(built):
    g, data, out, pyr = built
    for lv in range(9):
        d, n = pyr.store.get(lv)
        assert n == int(g["h5/%d/nfrags" % lv][0, 0]), lv
        assert np.array_equal(d, g["h5/%d/data" % lv]), lv


@pytest.mark.parametrize("tag,delta", [("level", 0), ("sub_level", -1)])
def test_loader_structures(built, tag, delta):
    g, data, out, pyr = built
    lev = pyr.get_level(int(g["level"]) + delta)
    assert lev.n_frags == int(g[tag + "/n_frags"])
    for k in lev.SOA_KEYS:
        assert np.array_equal(lev.S_o_A_frags[k], g["%s/soa/%s" % (tag, k)]), k
        assert lev.S_o_A_frags[k].dtype == np.int32
    assert list(lev.frags_init_contigs) == [str(x) for x in g[tag + "/frags_init_contigs"]]
    assert float(lev.mean_value_trans) == float(g[tag + "/mean_value_trans"])
    c = lev.sparse_mat_csr
    assert np.array_equal(c.indptr, g[tag + "/csr_indptr"]) and np.array_equal(c.indices, g[tag + "/csr_indices"])
    assert np.array_equal(c.data, g[tag + "/csr_data"])


def test_writers_byte_for_byte(built, tmp_path):
    g, data, out, pyr = built
    lev = pyr.get_level(int(g["level"]))
    lev.build_seq_per_bin(os.path.join(data, "genome.fa"))

    class V:
        pass

    v = V()
    v.id_c, v.pos, v.ori, v.id_d, v.activ = g["scr/id_c"], g["scr/pos"], g["scr/ori"], g["scr/id_d"], g["scr/activ"]
    fa, info = str(tmp_path / "genome.fasta"), str(tmp_path / "info_frags.txt")
    lev.generate_new_fasta(v, fa, info)
    assert open(info, "rb").read() == g["scr/info_frags"].tobytes()
    assert open(fa, "rb").read() == g["scr/fasta"].tobytes()


def test_sampler_arguments_assembled_like_simu_single(built):
    """the 29 constructor arguments `simulation` hands to the sampler (SS:120-153), against the reference's own"""
    from instagraal_amd.simulation import assemble_sampler_args

    g, data, out, pyr = built
    args, lev, sub = assemble_sampler_args(pyr, int(g["level"]), n_iterations=10, is_simu=False, use_rippe=True)
    for name, v in args.items():
        if name in ("vel", "pos"):
            continue
        if isinstance(v, dict):
            keys = [k[len("arg/%s/" % name):] for k in g.files if k.startswith("arg/%s/" % name)]
            assert sorted(keys) == sorted(v.keys()), name
            for k in keys:
                want = g["arg/%s/%s" % (name, k)]
                assert np.array_equal(v[k], want) and v[k].dtype == want.dtype, (name, k)
        elif hasattr(v, "indptr"):
            c = v.tocsr()
            assert np.array_equal(c.indptr, g["arg/%s/indptr" % name]) and np.array_equal(c.indices, g["arg/%s/indices" % name])
            assert np.array_equal(c.data, g["arg/%s/data" % name]) and list(c.shape) == list(g["arg/%s/shape" % name])
        else:
            want = g["arg/" + name]
            got = np.asarray(v)
            assert got.shape == want.shape, (name, got.shape, want.shape)
            if got.dtype.names:
                assert got.dtype == want.dtype, name
                assert got.tobytes() == want.tobytes(), name
            else:
                assert np.array_equal(got, want), name
                if got.dtype.kind == "f":
                    assert got.dtype == want.dtype, name
    soa = args["S_o_A_sub_frags"]
    assert float(soa["len_bp"].mean() / 1000.0) == float(g["mean_size_bin_kb"])
