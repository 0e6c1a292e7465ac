"""Writers of the two user-visible outputs of a run, `info_frags.txt` and `genome.fasta` (SURVEY 8(f) row f1;
reference: pyramid_sparse.py:1963-2033 `level.generate_new_fasta`, called once per cycle from instagraal.py:263-277)."""
from __future__ import annotations

import numpy as np

_COMPLEMENT = str.maketrans("TAGCtagc", "ATCGATCG")  # PS:1997: lower case comes out upper case
FASTA_WIDTH = 61  # PS:2020


def write_assembly(vect_frags, frags_init_contigs, start_bp, end_bp, sequences, new_fasta, info_frags):
    """vect_frags: object with numpy attributes id_c, pos, ori, activ, id_d (the downloaded genome state);
    frags_init_contigs[i], start_bp[i], end_bp[i]: original contig and coordinates of initial bin i;
    sequences: {contig name: str}.

    info_frags.txt: one block per scaffold in ascending contig-id order, bins in scaffold order:
    `init_contig  id_frag  orientation  start  end`.  Scaffolds holding an inactive bin are skipped.
    genome.fasta: the same scaffolds, longest first (ties: ascending id), 61 columns."""
    id_c, pos, ori, activ, id_d = (np.asarray(getattr(vect_frags, k)) for k in ("id_c", "pos", "ori", "activ", "id_d"))
    scaffolds = {}
    with open(info_frags, "w") as h_info:
        for cid in np.unique(id_c):
            members = np.nonzero(id_c == cid)[0]
            if not np.all(activ[members] == 1):
                continue
            h_info.write(">3C-assembly|contig_%s\n" % cid)
            h_info.write("init_contig\tid_frag\torientation\tstart\tend\n")
            parts = []
            for f in members[np.argsort(pos[members])]:
                init = id_d[f]
                name, s, e = frags_init_contigs[init], start_bp[init], end_bp[init]
                seq = sequences[name][s:e]
                if ori[f] == -1:
                    seq = seq[::-1].translate(_COMPLEMENT)
                h_info.write("%s\t%s\t%s\t%s\t%s\n" % (name, init, ori[f], s, e))
                parts.append(seq)
            scaffolds[cid] = "".join(parts)
    with open(new_fasta, "w") as h_fa:
        for cid in sorted(scaffolds, key=lambda c: len(scaffolds[c]), reverse=True):
            seq = scaffolds[cid]
            h_fa.write(">3C-assembly-contig_%s\n" % cid)
            n = len(seq)
            if n > 0:
                cuts = list(range(0, n, FASTA_WIDTH))
                for k in range(1, len(cuts)):
                    h_fa.write(seq[cuts[k - 1]:cuts[k]] + "\n")
                if cuts[-1] != n - 1:  # PS:2027: a final line of exactly one base is not written
                    h_fa.write(seq[cuts[-1]:] + "\n")
