"""Pyramid builder and loader: the data formats on the input side of the scoring path.

Restates what the reference's ``pyramid_sparse.py`` ("PS") does between the three text files of an instaGRAAL input
folder and the arrays the sampler is built from:

* ``build`` (PS:178-277) / ``build_and_filter`` (PS:30-175): level 0 = the input, optionally filtered
  (``remove_problematic_fragments``, PS:731-1029: fragments with too few / far too many contact partners or shorter
  than 50 bp are merged into the next kept fragment of their contig), then ``size_pyramid - 1`` successive binnings
  by ``factor`` (``subsample_data_set``, PS:468-724).  Every level is a folder of three text files plus the
  sub->super index of the level below, byte-identical to the reference's, and a sparse matrix
  ``{data: (3, nnz) int32 = rows, cols, counts; nfrags}``.
* ``Pyramid`` (PS:1351-1500) and ``Level.load_data`` (PS:1713-1906): fragment tables, the structure-of-arrays genome
  state, the CSR level matrix, the mean trans contact level.

The sparse matrices live in ``pyramid.hdf5`` with the reference's layout when h5py is importable, else in
``pyramid.npz`` next to the level folders (same arrays, same ``done`` flags).  Host-only numpy code, as in the
reference; the quirks that change the files are kept and marked (Q-P1 .. Q-P4).
"""
from __future__ import annotations

import json
import os
import shutil

import numpy as np
import scipy.sparse as sp

try:  # h5py-optional
    import h5py  # type: ignore
except ImportError:  # pragma: no cover - the container has no h5py
    h5py = None

FRAG_HEADER_L0 = ("id", "chrom", "start_pos", "end_pos", "size", "gc_content", "accu_frag", "frag_start", "frag_end")
FRAG_HEADER_LN = ("id", "chrom", "start_pos", "end_pos", "size", "gc_content", "accu_frag", "init_frag_start", "init_frag_end",
                  "sub_frag_start", "sub_frag_end")
CONTIG_HEADER = ("contig", "length_kb", "n_frags", "cumul_length")
CONTACT_HEADER = ("id_frag_a", "id_frag_b", "n_contact")


# ------------------------------------------------------------------------------------------------ sparse store


class SparseStore:
    """The per-level sparse matrices of one pyramid folder: HDF5 (reference layout, PS:386-396) or .npz."""

    def __init__(self, folder):
        self.folder = folder
        self.h5_path = os.path.join(folder, "pyramid.hdf5")
        self.npz_path = os.path.join(folder, "pyramid.npz")
        self.use_h5 = h5py is not None
        self._levels = {}
        self._done = {}
        if self.use_h5:
            self._h = h5py.File(self.h5_path, "a")
        elif os.path.exists(self.npz_path):
            with np.load(self.npz_path, allow_pickle=False) as z:
                self._done = json.loads(str(z["attrs"]))
                for k in z.files:
                    if k.endswith("/data"):
                        lv = k.split("/")[0]
                        self._levels[lv] = (np.array(z[k]), int(z[lv + "/nfrags"][0, 0]))

    def done(self, level):
        if self.use_h5:
            return self._h.attrs.get(str(level), None) == "done"
        return self._done.get(str(level)) == "done"

    def put(self, level, data3, nfrags):
        lv = str(level)
        data3 = np.ascontiguousarray(data3, np.int32)
        if self.use_h5:
            grp = self._h.create_group(lv)
            grp.create_dataset("data", data=data3)
            grp.create_dataset("nfrags", data=np.array([[nfrags]], np.int32))
            self._h.attrs[lv] = "done"
        else:
            self._levels[lv] = (data3, int(nfrags))
            self._done[lv] = "done"

    def get(self, level):
        lv = str(level)
        if self.use_h5:
            return np.array(self._h[lv]["data"], np.int32), int(self._h[lv]["nfrags"][0, 0])
        return self._levels[lv]

    def close(self):
        if self.use_h5:
            self._h.close()
        else:
            blob = {"attrs": np.array(json.dumps(self._done))}
            for lv, (d, n) in self._levels.items():
                blob[lv + "/data"] = d
                blob[lv + "/nfrags"] = np.array([[n]], np.int32)
            np.savez(self.npz_path, **blob)


# ------------------------------------------------------------------------------------------------ text helpers


def _rows(path):
    """data rows of a tab-separated file with one header line"""
    with open(path, "r") as f:
        lines = f.readlines()
    return [ln.split("\t") for ln in lines[1:]]


def _write_table(path, header, rows):
    with open(path, "w") as f:
        f.write("\t".join(header) + "\n")
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")


def file_len(path):
    with open(path) as f:
        return sum(1 for _ in f)


def _accumulate_contacts(fa, fb, nc):
    """sum the counts of equal unordered pairs; -> (f1, f2, count) with f1 <= f2, sorted by (f1, f2)"""
    f1, f2 = np.minimum(fa, fb), np.maximum(fa, fb)
    if f1.size == 0:
        return f1, f2, nc
    width = int(f2.max()) + 1
    key = f1.astype(np.int64) * width + f2
    uniq, inv = np.unique(key, return_inverse=True)
    tot = np.bincount(inv, weights=nc.astype(np.float64)).astype(np.int64)
    return uniq // width, uniq % width, tot


def _read_contacts(path, skip_first_data_line=False):
    rows = _rows(path)
    if skip_first_data_line:
        rows = rows[1:]
    if not rows:
        z = np.zeros(0, np.int64)
        return z, z, z
    a = np.array([(int(r[0]), int(r[1]), int(r[2])) for r in rows], dtype=np.int64)
    return a[:, 0], a[:, 1], a[:, 2]


# ------------------------------------------------------------------------------------------------ level 0


def init_frag_list(fragment_list, new_frag_list):
    """PS:399-465: the input fragment list with the three bookkeeping columns of level 0 appended"""
    out = []
    for d in _rows(fragment_list):
        out.append((d[0], d[1], d[2], d[3], d[4], str(float(d[5])), "1", d[0], d[0]))
    _write_table(new_frag_list, FRAG_HEADER_L0, out)
    return len(out)


def fill_sparse_pyramid_level(store, level, contact_file, nfrags):
    """PS:331-396: text contact list -> (3, nnz) int32.  Rows ascending; inside a row the columns keep the order of
    their first appearance in the file (the reference iterates a dict it filled in file order)."""
    fa, fb, nc = _read_contacts(contact_file)
    f1, f2 = np.minimum(fa, fb), np.maximum(fa, fb)
    if f1.size:
        width = int(f2.max()) + 1
        key = f1 * width + f2
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        tot = np.bincount(inv, weights=nc.astype(np.float64)).astype(np.int64)
        order = np.lexsort((first, uniq // width))
        data3 = np.stack([(uniq // width)[order], (uniq % width)[order], tot[order]]).astype(np.int32)
    else:
        data3 = np.zeros((3, 0), np.int32)
    store.put(level, data3, nfrags)


# ------------------------------------------------------------------------------------------------ binning


def subsample_data_set(contig_info, fragments_list, fact_sub_sample, abs_fragments_contacts, new_abs_fragments_contacts_file,
                       min_bin_per_contig, new_contig_list_file, new_fragments_list_file, old_2_new_file):
    """PS:468-724: bins of `fact_sub_sample` consecutive fragments inside every contig that has at least
    `min_bin_per_contig` such bins (float32 test, PS:516); other contigs keep their fragments."""
    if fact_sub_sample <= 1:
        shutil.copy(fragments_list, new_fragments_list_file)
        shutil.copy(contig_info, new_contig_list_file)
        shutil.copy(abs_fragments_contacts, new_abs_fragments_contacts_file)
        nfrags = file_len(fragments_list) - 1
        _write_table(old_2_new_file, ("current_id", "super_id"), [(i + 1, i + 1) for i in range(nfrags)])
        return nfrags

    # -- which new bin every old fragment falls in, contig by contig
    old2new = []           # 1-based new absolute id per old absolute id (0-based list)
    new_frags = []         # per new bin: dict(first, last (old abs ids, 1-based), id_rel, contig)
    contig_rows = []
    for d in _rows(contig_info):
        name, length_kb, n_in = d[0], d[1], int(d[2])
        binned = (n_in / np.float32(fact_sub_sample)) >= min_bin_per_contig
        first_new = len(new_frags)
        for k in range(n_in):
            old_abs = len(old2new) + 1
            if (not binned) or (k % fact_sub_sample == 0):
                new_frags.append(dict(first=old_abs, last=old_abs, id_rel=len(new_frags) - first_new + 1, contig=name, gc=[]))
            new_frags[-1]["last"] = old_abs
            old2new.append(len(new_frags))
        n_new = len(new_frags) - first_new
        contig_rows.append((name, length_kb, n_new, len(new_frags) - n_new))
    _write_table(new_contig_list_file, CONTIG_HEADER, contig_rows)

    # -- bin attributes from the fragment list of the level below
    for i, d in enumerate(_rows(fragments_list)):
        old_abs = i + 1
        b = new_frags[old2new[i] - 1]
        b["gc"].append(float(d[5]))
        if old_abs == b["first"]:
            b["start_pos"] = int(d[2])
            b["init_first"] = int(d[7])
        if old_abs == b["last"]:
            b["end_pos"] = int(d[3])
            b["init_last"] = int(d[8])
    frag_rows = []
    for b in new_frags:
        frag_rows.append((b["id_rel"], b["contig"], b["start_pos"], b["end_pos"], b["end_pos"] - b["start_pos"], np.array(b["gc"]).mean(),
                          b["init_last"] - b["init_first"] + 1, b["init_first"], b["init_last"], b["first"], b["last"]))
    _write_table(new_fragments_list_file, FRAG_HEADER_LN, frag_rows)

    # -- contacts.  Q-P1: the reference consumes the header with readline() and then starts its loop at index 1 of
    #    the remaining lines, so the FIRST contact of every level's list is dropped when binning (PS:672-677).
    if abs_fragments_contacts != "SIMU":
        fa, fb, nc = _read_contacts(abs_fragments_contacts, skip_first_data_line=True)
        o2n = np.array(old2new, np.int64)
        f1, f2, tot = _accumulate_contacts(o2n[fa] - 1, o2n[fb] - 1, nc)
        _write_table(new_abs_fragments_contacts_file, CONTACT_HEADER, zip(f1.tolist(), f2.tolist(), tot.tolist()))
    _write_table(old_2_new_file, ("current_id", "super_id"), [(i + 1, v) for i, v in enumerate(old2new)])
    return len(new_frags)


# ------------------------------------------------------------------------------------------------ filter


def remove_problematic_fragments(contig_info, fragments_list, abs_fragments_contacts, new_contig_list_file, new_fragments_list_file,
                                 new_abs_fragments_contacts_file, level0, thresh_factor=1):
    """PS:731-1029.  `level0` = (data3, nfrags) of the unfiltered level 0.  A fragment is "problematic" when the share of
    fragments it has a contact with is <= mean - thresh_factor * std or > mean + 50 * std (float32 statistics), or when it
    is shorter than 50 bp.  Walking every contig, a problematic fragment is not emitted: it is glued to the following
    fragments until a good one closes the run (the emitted fragment spans the whole run); a run still open at the end of
    a contig is destroyed together with its contacts.  Returns the lower threshold."""
    data3, nfrags = level0
    csr = sp.csr_matrix((data3[2, :], data3[0:2, :]), shape=(nfrags, nfrags))
    full = csr + csr.transpose()
    sparsity = np.float32(np.diff(full.indptr)) / np.float32(nfrags)
    mean_s, std_s = sparsity.mean(), sparsity.std()
    thresh = mean_s - thresh_factor * std_s
    thresh_max = mean_s + 50 * std_s
    frows = _rows(fragments_list)
    bad = set(np.nonzero(sparsity <= thresh)[0].tolist()) | set(np.nonzero(sparsity > thresh_max)[0].tolist())
    bad |= {i for i, d in enumerate(frows) if int(d[3]) - int(d[2]) < 50}
    bad_tags = {frows[i][0] + "-" + frows[i][1] for i in bad}  # PS:838-845: matched by "<id in contig>-<contig>"

    contig_stats = {d[0]: dict(n_new=0, length=0) for d in _rows(contig_info)}
    old2new = {}
    out_rows = []
    new_abs, new_rel = 1, 0
    run = None  # the open run of glued fragments

    def open_run():
        return dict(start_pos=0, size=0, accu=0, gc=[], members=[], lock=False)

    run = open_run()
    for i, d in enumerate(frows):
        old_abs = i + 1
        fid = int(d[0])
        if fid == 1:  # a new contig starts: an open run of the previous contig dies
            new_rel = 1
            if run["lock"]:
                for m in run["members"]:
                    old2new[m] = "destroyed"
            carried = run["accu"]  # Q-P4: the reference forgets to reset this counter here (PS:886-899)
            run = open_run()
            run["accu"] = carried
        chrom, start_pos, end_pos, size = d[1], d[2], d[3], int(d[4])
        locked = (d[0] + "-" + chrom) in bad_tags
        run["size"] += size
        run["accu"] += int(d[6])
        run["gc"].append(float(d[5]))
        run["members"].append(old_abs)
        run["lock"] = locked or size <= 1
        old2new[old_abs] = new_abs
        if not locked:  # Q-P2: the emission test uses the tag only; a 1 bp fragment locks the run but is still emitted
            contig_stats[chrom]["n_new"] += 1
            contig_stats[chrom]["length"] += run["size"]
            out_rows.append((new_rel, chrom, run["start_pos"], end_pos, run["size"], np.array(run["gc"]).mean(), run["accu"], new_rel,
                             new_rel))
            run = open_run()
            run["start_pos"] = end_pos
            new_rel += 1
            new_abs += 1
    if run["lock"]:
        for m in run["members"]:
            old2new[m] = "destroyed"
    _write_table(new_fragments_list_file, FRAG_HEADER_L0, out_rows)

    rows, cumul = [], 0
    for d in _rows(contig_info):
        st = contig_stats[d[0]]
        if st["n_new"] > 0:
            rows.append((d[0], st["length"], st["n_new"], cumul))
            cumul += st["n_new"]
    _write_table(new_contig_list_file, CONTIG_HEADER, rows)

    fa, fb, nc = _read_contacts(abs_fragments_contacts)
    lut = np.full(len(frows) + 2, -1, np.int64)
    for k, v in old2new.items():
        if v != "destroyed":
            lut[k] = v - 1
    na, nb = lut[fa + 1], lut[fb + 1]
    keep = (na >= 0) & (nb >= 0)
    f1, f2, tot = _accumulate_contacts(na[keep], nb[keep], nc[keep])
    _write_table(new_abs_fragments_contacts_file, CONTACT_HEADER, zip(f1.tolist(), f2.tolist(), tot.tolist()))
    return thresh


# ------------------------------------------------------------------------------------------------ builders


def _level_paths(pyramid_folder, level):
    d = os.path.join(pyramid_folder, "level_%d" % level)
    p = "%d_" % level
    return d, os.path.join(d, p + "contig_info.txt"), os.path.join(d, p + "fragments_list.txt"), \
        os.path.join(d, p + "abs_frag_contacts.txt"), os.path.join(d, p + "sub_2_super_index_frag.txt")


def _bin_levels(store, pyramid_folder, size_pyramid, factor, min_bin_per_contig):
    """levels 1 .. size_pyramid-1 from level 0 (shared tail of PS:118-171 and PS:222-276) + the sparse matrices"""
    _, cur_contigs, cur_frags, cur_contacts, cur_index = _level_paths(pyramid_folder, 0)
    nfrags = file_len(cur_frags) - 1
    for level in range(size_pyramid):
        d, contigs, frags, contacts, index = _level_paths(pyramid_folder, level)
        os.makedirs(d, exist_ok=True)
        if level > 0:
            if all(os.path.exists(x) for x in (contigs, frags, contacts, cur_index)):
                nfrags = file_len(frags) - 1
            else:
                nfrags = subsample_data_set(cur_contigs, cur_frags, factor, cur_contacts, contacts, min_bin_per_contig, contigs, frags,
                                            cur_index)
        if not store.done(level):
            fill_sparse_pyramid_level(store, level, contacts, nfrags)
        cur_contigs, cur_frags, cur_contacts, cur_index = contigs, frags, contacts, index


def build(base_folder, size_pyramid, factor, min_bin_per_contig, output_folder=None):
    """PS:178-277: unfiltered pyramid `pyramids/pyramid_<n>_no_thresh`"""
    root = output_folder if output_folder is not None else base_folder
    folder = os.path.join(root, "pyramids", "pyramid_%d_no_thresh" % size_pyramid)
    d0, contigs0, frags0, contacts0, _ = _level_paths(folder, 0)
    os.makedirs(d0, exist_ok=True)
    shutil.copyfile(os.path.join(base_folder, "info_contigs.txt"), contigs0)
    shutil.copyfile(os.path.join(base_folder, "abs_fragments_contacts_weighted.txt"), contacts0)
    init_frag_list(os.path.join(base_folder, "fragments_list.txt"), frags0)
    store = SparseStore(folder)
    _bin_levels(store, folder, size_pyramid, factor, min_bin_per_contig)
    store.close()
    return folder


def build_and_filter(base_folder, size_pyramid, factor, thresh_factor=1, output_folder=None):
    """PS:30-175: `pyramids/pyramid_1_no_thresh` (level 0 as given), its filtered copy as level 0 of
    `pyramids/pyramid_<n>_thresh_auto`, then the binned levels.  Returns the loaded Pyramid."""
    root = output_folder if output_folder is not None else base_folder
    os.makedirs(os.path.join(root, "pyramids"), exist_ok=True)
    init_folder = os.path.join(root, "pyramids", "pyramid_1_no_thresh")
    if not os.path.exists(init_folder):
        build(base_folder, 1, factor, 1, output_folder=root)
    _, contigs_i, frags_i, contacts_i, _ = _level_paths(init_folder, 0)
    folder = os.path.join(root, "pyramids", "pyramid_%d_thresh_auto" % size_pyramid)
    d0, contigs0, frags0, contacts0, _ = _level_paths(folder, 0)
    os.makedirs(d0, exist_ok=True)
    if not all(os.path.exists(x) for x in (contigs0, frags0, contacts0)):
        s0 = SparseStore(init_folder)
        remove_problematic_fragments(contigs_i, frags_i, contacts_i, contigs0, frags0, contacts0, s0.get(0), thresh_factor=thresh_factor)
        if s0.use_h5:
            s0._h.close()
    store = SparseStore(folder)
    _bin_levels(store, folder, size_pyramid, factor, 1)
    store.close()
    return Pyramid(folder, size_pyramid)


# ------------------------------------------------------------------------------------------------ loader


class Pyramid:
    """PS:1351-1500: the fragment tables of every level (1-based fragment ids as in the reference's dictionaries)."""

    def __init__(self, pyramid_folder, n_levels):
        self.pyramid_folder = pyramid_folder
        self.n_levels = n_levels
        self.store = SparseStore(pyramid_folder)
        self.spec_level = {}
        for i in range(n_levels):
            _, contigs, frags, _, index = _level_paths(pyramid_folder, i)
            spec = self._read_fragments(frags, i)
            spec["level_folder"], spec["fragments_list_file"], spec["contig_info_file"] = os.path.dirname(frags), frags, contigs
            if i < n_levels - 1:  # PS:1388-1395
                sup = np.array([int(r[1]) for r in _rows(index)], np.int64)
                spec["super_index"] = sup
            self.spec_level[str(i)] = spec
        self.list_contigs_name = self.spec_level["0"]["contig_names"]
        self.list_contigs_id = list(range(1, len(self.list_contigs_name) + 1))

    @staticmethod
    def _read_fragments(path, level):
        """PS:1409-1482, as columns: row r = the reference's fragments_dict[r + 1]"""
        rows = _rows(path)
        n = len(rows)
        spec = dict(init_contig=[r[1] for r in rows], index=np.array([int(r[0]) for r in rows], np.int64),
                    start_pos=np.array([int(r[2]) for r in rows], np.int64), end_pos=np.array([int(r[3]) for r in rows], np.int64),
                    size=np.array([int(r[4]) for r in rows], np.int64), gc_content=np.array([float(r[5]) for r in rows]),
                    n_accu_frags=np.array([int(r[6]) for r in rows], np.int64))
        if level > 0:
            spec["sub_low_index"] = np.array([int(r[9]) for r in rows], np.int64)
            spec["sub_high_index"] = np.array([int(r[10]) for r in rows], np.int64)
        else:
            spec["sub_low_index"] = spec["index"].copy()
            spec["sub_high_index"] = spec["index"].copy()
        names, cid = [], np.zeros(n, np.int64)
        seen = {}
        for r, name in enumerate(spec["init_contig"]):
            if name not in seen:
                seen[name] = len(names) + 1
                names.append(name)
            cid[r] = seen[name]
        spec["contig_names"], spec["contig_id"] = names, cid
        spec["super_index"] = spec["index"].copy()
        return spec

    def get_level(self, level_id):
        return Level(self, level_id)

    def load_reference_sequence(self, genome_fasta):
        """PS:1630-1660.  Q-P3: the reference joins `all_lines[start:-1]` for the LAST record, i.e. the last line of the
        file is not part of the last sequence."""
        import gzip

        opener = gzip.open if str(genome_fasta).endswith(".gz") else open
        with opener(genome_fasta, "rt") as f:
            lines = f.readlines()
        seqs, name, start = {}, lines[0][1:].split()[0].strip(), 1
        for i in range(1, len(lines)):
            if lines[i][0] == ">":
                seqs[name] = "".join(lines[start:i])
                start = i + 1
                name = lines[i][1:].split()[0].strip()
        seqs[name] = "".join(lines[start:-1])
        self.dict_sequence_contigs = {k: v.replace("\n", "").replace("\r", "") for k, v in seqs.items()}

    def close(self):
        if self.store.use_h5:
            self.store._h.close()


class Level:
    """PS:1663-1906 `level`: what the sampler is built from at one resolution."""

    SOA_KEYS = ("pos", "sub_pos", "id_c", "start_bp", "len_bp", "sub_len", "circ", "id", "prev", "next", "l_cont", "sub_l_cont",
                "l_cont_bp", "n_accu")

    def __init__(self, pyramid, level):
        self.pyramid, self.level = pyramid, level
        self.load_data(pyramid)

    def load_data(self, pyramid):
        data3, self.n_frags = pyramid.store.get(self.level)
        self.np_2_scipy_sparse = data3
        self.sparse_mat_csr = sp.csr_matrix((data3[2, :], data3[0:2, :]), shape=(self.n_frags, self.n_frags))
        spec = pyramid.spec_level[str(self.level)]
        sub = pyramid.spec_level[str(self.level - 1)] if str(self.level - 1) in pyramid.spec_level else spec
        n = self.n_frags
        cid = spec["contig_id"]
        n_contigs = len(spec["contig_names"])
        l_cont = np.bincount(cid, minlength=n_contigs + 1)
        sub_l_cont = np.bincount(sub["contig_id"], minlength=n_contigs + 1)
        l_cont_bp = np.bincount(cid, weights=spec["size"].astype(np.float64), minlength=n_contigs + 1).astype(np.int64)
        n_sub = spec["sub_high_index"] - spec["sub_low_index"] + 1
        first = np.concatenate([[True], cid[1:] != cid[:-1]]) if n else np.zeros(0, bool)
        last = np.concatenate([cid[1:] != cid[:-1], [True]]) if n else np.zeros(0, bool)
        cum_sub = np.cumsum(n_sub) - n_sub
        start_of_contig = np.maximum.accumulate(np.where(first, np.arange(n), 0))
        ident = np.arange(n)
        soa = dict(pos=spec["index"] - 1, sub_pos=cum_sub - cum_sub[start_of_contig], id_c=cid, start_bp=spec["start_pos"],
                   len_bp=spec["size"], sub_len=n_sub, circ=np.zeros(n), id=ident, prev=np.where(first, -1, ident - 1),
                   next=np.where(last, -1, ident + 1), l_cont=l_cont[cid], sub_l_cont=sub_l_cont[cid], l_cont_bp=l_cont_bp[cid],
                   n_accu=spec["n_accu_frags"])
        self.S_o_A_frags = {k: np.array(soa[k], dtype=np.int32) for k in self.SOA_KEYS}
        self.frags_init_contigs = list(spec["init_contig"])
        self.n_contigs = n_contigs
        # mean trans contact level (PS:1876-1898): contacts of the upper-triangular level matrix whose row lies in a contig
        # and whose column does not, over the number of trans pairs
        total_trans = 0
        n_tot_intra = 0
        for c in range(1, n_contigs + 1):
            members = np.nonzero(cid == c)[0]
            rows_c = self.sparse_mat_csr[members, :]
            intra = rows_c.tocsc()[:, members]
            total_trans += rows_c.sum()
            total_trans -= intra.sum()
            n_tot_intra += len(members) * (len(members) - 1) / 2
        n_tot = n * (n - 1) / 2 - n_tot_intra
        with np.errstate(invalid="ignore", divide="ignore"):
            self.mean_value_trans = total_trans / np.float32(n_tot)
        if np.isnan(self.mean_value_trans):
            self.mean_value_trans = np.amin(self.sparse_mat_csr.data) / 10.0
        self.distri_frag = spec["size"].copy()

    def build_seq_per_bin(self, genome_fasta):
        self.pyramid.load_reference_sequence(genome_fasta)

    def generate_new_fasta(self, vect_frags, new_fasta, info_frags):
        from .io_frags import write_assembly

        spec = self.pyramid.spec_level[str(self.level)]
        write_assembly(vect_frags, self.frags_init_contigs, spec["start_pos"], spec["end_pos"], self.pyramid.dict_sequence_contigs,
                       new_fasta, info_frags)
