"""BASELINE.json config 1 at its stated scale: a yeast-like Hi-C folder -- ~85 k restriction fragments on 140
contigs, generated here (the checkout holds no yeast pairs file) -- through ``run_instagraal(level=4, cycles=5, bomb=True)``:
the text pyramid (level 4: ~1 000 bins), the MI355X sampler, five cycles over all bins, the run's output files.

* the reference's own file-shape contract for such a run (/root/reference/tests/test_instagraal_gpu.py:126-340: which files,
  FASTA headers and alphabet, ``info_frags.txt`` blocks, ``save_simu_step_<j>.txt`` rows, ``list_*.txt`` lengths and
  domains, ``list_mutations.txt`` columns and id ranges) -- except the matrix previews (matplotlib, outside this path);
* run-to-run determinism under ``np.random.seed(0)``: two runs, every output file byte for byte;
* the first moves of the first cycle against the oracle (the CPU restatement of the reference's algorithm) replaying them
  from the same arguments, fitted parameters and generator state."""
import math
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEVEL, CYCLES = 4, 5
ORACLE_MOVES = 160


def test_cfg1_yeast_scale_folder_level4_five_cycles(tmp_path):
    from instagraal_amd import synth
    from instagraal_amd.simulation import assemble_sampler_args, run_instagraal
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    data, out = str(tmp_path / "data"), str(tmp_path / "out")
    n0, n_lines = synth.write_text_dataset_large(data, n_frags0=85_000, n_contigs=140, seed=11)
    assert 80_000 < n0 < 90_000 and n_lines > 1_000_000
    fasta = os.path.join(data, "genome.fa")

    np.random.seed(0)
    p2 = run_instagraal(data, fasta, out, level=LEVEL, cycles=CYCLES, bomb=True)
    s = p2.simulation.sampler
    N = int(s.n_new_frags)
    assert 900 < N < 1300  # level 4 of the pyramid: ~1 000 bins
    n_iters = CYCLES * N
    folder = p2.simulation.output_folder
    fitted = {k: float(s.param_simu[k][0]) for k in s.param_simu.dtype.names}
    hic_pyr = p2.simulation.hic_pyr
    p2.simulation.release()

    # ---- the reference's file-shape contract
    for f in ["genome.fasta", "info_frags.txt", "list_likelihood.txt", "list_n_contigs.txt", "list_mean_len.txt", "list_dist_init_genome.txt",
              "list_mutations.txt", "save_simu_step_0.txt", "save_simu_step_%d.txt" % (CYCLES - 1)]:
        assert os.path.exists(os.path.join(folder, f)), f
    fa_lines = open(os.path.join(folder, "genome.fasta")).read().splitlines()
    headers = [x for x in fa_lines if x.startswith(">")]
    assert len(headers) >= 1
    valid = set("ACGTNacgtn")
    current, started = None, False
    for line in fa_lines:
        if line.startswith(">"):
            name = line.lstrip(">")
            assert name.startswith("3C-assembly-contig_") and name.split("3C-assembly-contig_")[1].isdigit(), line
            if current is not None:
                assert started, "empty sequence for " + current
            current, started = line, False
        else:
            assert set(line) <= valid
            started = started or bool(line.strip())
    assert started
    blocks = [b.strip() for b in open(os.path.join(folder, "info_frags.txt")).read().split(">") if b.strip()]
    assert len(blocks) == len(headers)
    n_rows = 0
    for block in blocks:
        lines = block.splitlines()
        assert len(lines) >= 2 and set(lines[1].split()) == {"init_contig", "id_frag", "orientation", "start", "end"}
        for row in lines[2:]:
            fields = row.split()
            assert len(fields) == 5 and int(fields[2]) in (1, -1)
        n_rows += len(lines) - 2
    assert n_rows == N  # every bin of the level is placed exactly once
    steps = sorted(f for f in os.listdir(folder) if f.startswith("save_simu_step_"))
    assert len(steps) == CYCLES
    for f in steps:
        rows = open(os.path.join(folder, f)).read().splitlines()
        assert len(rows) == N
        for line in rows:
            fields = line.split()
            assert len(fields) == 4 and int(fields[3]) in (1, -1)
            int(fields[0]), int(fields[1]), int(fields[2])
    lik = [v for v in open(os.path.join(folder, "list_likelihood.txt")).read().splitlines() if v.strip()]
    assert len(lik) == n_iters and all(math.isfinite(float(v)) for v in lik)
    ncont = [int(v) for v in open(os.path.join(folder, "list_n_contigs.txt")).read().split()]
    assert len(ncont) == n_iters and min(ncont) > 0
    for f in ("list_mean_len.txt", "list_dist_init_genome.txt"):
        assert len([v for v in open(os.path.join(folder, f)).read().splitlines() if v.strip()]) == n_iters
    mut = open(os.path.join(folder, "list_mutations.txt")).read().splitlines()
    assert mut[0].split("\t") == ["id_fA", "id_fB", "id_mutation"] and len(mut) == n_iters + 1
    m = np.array([[int(x) for x in r.split("\t")] for r in mut[1:]])
    assert m[:, 0].min() >= 0 and m[:, 0].max() < N and m[:, 1].min() >= 0 and m[:, 1].max() < N and m[:, 2].min() >= 0 and m[:, 2].max() < 24
    # the assembly went somewhere: after --bomb every bin is its own contig; five cycles later there are far fewer
    assert ncont[-1] < N // 4

    # ---- determinism: the same run again (the pyramid is found on disk), byte for byte
    first = str(tmp_path / "first")
    shutil.copytree(folder, first)
    np.random.seed(0)
    p3 = run_instagraal(data, fasta, out, level=LEVEL, cycles=CYCLES, bomb=True)
    p3.simulation.release()
    for f in sorted(os.listdir(first)):
        a, b = os.path.join(first, f), os.path.join(folder, f)
        if os.path.isfile(a):
            assert open(a, "rb").read() == open(b, "rb").read(), f

    # ---- the first moves against the oracle replaying them
    ol.build()
    ol.set_threads(min(16, os.cpu_count() or 1))
    try:
        args, _, _ = assemble_sampler_args(hic_pyr, LEVEL, 30, False, True)
        args.pop("vel"), args.pop("pos")
        o = OracleSampler(**args, vel=None, pos=None, mode=ol.MODE_DET)
        o.set_param_simu(fitted)
        o.eval_likelihood_init()
        np.random.seed(0)
        o.bomb_the_genome()
        list_frags = np.arange(0, o.n_new_frags)
        np.random.shuffle(list_frags)
        for t, id_frag in enumerate(list_frags[:ORACLE_MOVES]):
            r = o.step_sampler(int(id_frag), 5, o.dt)
            assert [int(id_frag), int(r[3]), int(r[2])] == list(m[t]), t
            assert float(lik[t]) == float(r[0]) and ncont[t] == int(r[5]), t
    finally:
        ol.set_threads(1)
