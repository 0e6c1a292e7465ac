"""GPU end-to-end: an instaGRAAL input folder (text files + FASTA) -> pyramid -> MI355X sampler -> cycles -> info_frags.txt /
genome.fasta (instagraal_amd.simulation), checked against the oracle sampler replaying the same run on the
CPU with the same fitted parameters: every move, the final genome, the output files byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LEVEL = 2


@pytest.mark.parametrize("id_start_sample_param", [4, 0])
def test_text_folder_to_assembly_matches_oracle_replay(tmp_path, id_start_sample_param):
    from instagraal_amd import io_frags, synth
    from instagraal_amd.simulation import assemble_sampler_args, instagraal_class
    from oracle import oracle_lib as ol
    from oracle.sampler_oracle import OracleSampler

    ol.build()
    data, out = str(tmp_path / "data"), str(tmp_path / "out")
    synth.write_text_dataset(data, n_contigs=10, mean_frags=110, seed=7, contacts_per_frag=40)
    np.random.seed(17)
    p2 = instagraal_class(name="synth", folder_path=data, fasta=os.path.join(data, "genome.fa"), device=0, level=LEVEL,
                          n_iterations_em=30, n_iterations_mcmc=100, is_simu=False, scrambled=False, perform_em=False, use_rippe=True,
                          sample_param=True, thresh_factor=1, output_folder=out)
    s = p2.simulation.sampler
    fitted = {k: float(s.param_simu[k][0]) for k in s.param_simu.dtype.names}
    n_cycles = 2
    p2.full_em(n_cycles=n_cycles, n_neighbours=5, bomb=True, id_start_sample_param=id_start_sample_param)
    folder = p2.simulation.output_folder
    for f in ("info_frags.txt", "genome.fasta", "list_mutations.txt", "list_likelihood.txt", "save_simu_step_1.txt"):
        assert os.path.getsize(os.path.join(folder, f)) > 0, f
    got_state = s.gpu_vect_frags.copy_from_gpu().soa17()

    # ---- the oracle replays the run: same arguments, same fitted parameters, same RNG stream
    args, lev, sub = assemble_sampler_args(p2.simulation.hic_pyr, LEVEL, 30, False, True)
    args.pop("vel"), args.pop("pos")
    o = OracleSampler(**args, vel=None, pos=None, mode=ol.MODE_DET)
    o.set_param_simu(fitted)
    o.bins = s.bins
    o.eval_likelihood_init()
    np.random.seed(17)
    o.bomb_the_genome()
    list_frags = np.arange(0, o.n_new_frags)
    rows, nuis = [], []
    t, n_iter = 0, n_cycles * o.n_new_frags
    for j in range(n_cycles):
        np.random.shuffle(list_frags)
        for id_frag in list_frags:
            r = o.step_sampler(int(id_frag), 5, o.dt)
            rows.append((int(id_frag), int(r[3]), int(r[2]), float(r[0]), float(r[1]), int(r[5])))
            if j > id_start_sample_param:  # IG:241-252: nuisance parameters are sampled after every move
                q = o.step_nuisance_parameters(o.dt, t, n_iter)
                nuis.append((float(q[0]), float(q[4]), float(q[2]), int(q[6])))
            t += 1
    want_mut = "id_fA\tid_fB\tid_mutation\n" + "".join("%s\t%s\t%s\n" % (a, b, m) for a, b, m, _, _, _ in rows)
    assert open(os.path.join(folder, "list_mutations.txt")).read() == want_mut
    assert [float(x) for x in open(os.path.join(folder, "list_likelihood.txt")).read().split()] == [r[3] for r in rows]
    assert [float(x) for x in open(os.path.join(folder, "list_dist_init_genome.txt")).read().split()] == [r[4] for r in rows]
    assert [int(x) for x in open(os.path.join(folder, "list_n_contigs.txt")).read().split()] == [r[5] for r in rows]
    assert np.array_equal(got_state, o.gpu_vect_frags.soa17())
    for name, col in (("fact", 0), ("slope", 1), ("d_max", 2)):
        got = [np.float32(x) for x in open(os.path.join(folder, "list_%s.txt" % name)).read().split()]  # written as float32 repr
        assert got == [np.float32(q[col]) for q in nuis], name
    assert [int(float(x)) for x in open(os.path.join(folder, "list_success.txt")).read().split()] == [q[3] for q in nuis]
    assert (len(nuis) > 0) == (id_start_sample_param == 0)

    class V:
        pass

    v = V()
    g = o.gpu_vect_frags
    for k in ("id_c", "pos", "ori", "activ", "id_d"):
        setattr(v, k, np.asarray(getattr(g, k)))
    spec = p2.simulation.hic_pyr.spec_level[str(LEVEL)]
    fa, info = str(tmp_path / "o.fasta"), str(tmp_path / "o_info.txt")
    io_frags.write_assembly(v, p2.simulation.level.frags_init_contigs, spec["start_pos"], spec["end_pos"],
                            p2.simulation.hic_pyr.dict_sequence_contigs, fa, info)
    assert open(info, "rb").read() == open(os.path.join(folder, "info_frags.txt"), "rb").read()
    assert open(fa, "rb").read() == open(os.path.join(folder, "genome.fasta"), "rb").read()


def test_two_processes_batch_runner():
    """two real processes (torch.distributed.run) sharing the one GPU of the test box, gloo collectives: the slot-split
    BatchRunner of every rank reproduces ig_step_batch bit for bit (tests/_two_rank_worker.py)"""
    import signal
    import socket
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    last = None
    for attempt in range(2):  # a rendezvous that does not come up (port taken in between) is retried once on another port
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(here, "_two_rank_worker.py")]
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)  # the launcher and its two workers (own process group)
            out, err = p.communicate()
            last = "timeout: " + out[-1000:] + err[-2000:]
            continue
        assert p.returncode == 0, out[-2000:] + err[-4000:]
        assert "TWO_RANK_OK" in out
        return
    raise AssertionError(last)


@pytest.mark.parametrize("runner", ["batch", "sharded", "replicas"])
def test_bench_starts_its_own_ranks(runner):
    """`python bench.py --gpus 2` without a launcher: the script starts its two ranks itself (child processes; this rig has
    one GPU, so IG_BENCH_ONE_DEVICE=1 puts both on cuda:0 with gloo collectives) and relays rank 0's JSON line; both ways
    of splitting the chain (batch slots / contact rows) leave the maintained likelihood exact and report the world size."""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, IG_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "small", "--steps", "60", "--warmup", "20",
           "--moves-per-step", "1", "--no-cpu-baseline", "--nuisance-moves", "0", "--runner", runner]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 60 and out["value"] > 0
    assert out["roofline"]["launches"] > 0
    if runner in ("batch", "replicas"):
        assert out["config"]["maintained_likelihood_exact"] is True
    assert out["scaling"] == ("weak" if runner == "replicas" else "strong")


def test_bench_line_contract_one_gpu():
    """`python bench.py --steps K --warmup W` on one GPU: one JSON line, a step is 128 consecutive moves (value in moves/s over
    128 K moves), the roofline and CPU-baseline objects present, the maintained likelihood exact after the run"""
    import json
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "small", "--steps", "4", "--warmup", "1", "--cpu-budget", "2",
                        "--nuisance-moves", "20"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 1 and out["unit"] == "moves/s" and out["higher_is_better"] is True
    assert out["config"]["moves_per_step"] == 128 and out["config"]["moves_timed"] == 512 and out["config"]["moves_warmup"] == 128
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 - 128.0) < 1e-6 * 128.0  # value = moves / s, ms_per_step per step of 128 moves
    # BASELINE.md section 3 read literally: one step_sampler call per move (round 6: a top-level field)
    assert out["value_unchanged_caller"] > 0 and out["value_unchanged_caller"] < out["value"]
    assert out["roofline"]["traffic_replayed"] is None or out["roofline"]["traffic_replayed"]["replayed"] is True
    assert out["vs_baseline"] is None and out["config"]["maintained_likelihood_exact"] is True
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["launches"] >= 4 and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "moves/s" and c["value"] > 0 and c["cores"] >= 1
    assert out["config"]["nuisance_on_moves_per_s"] > 0


def test_bench_refuses_ranks_without_devices():
    """more ranks than GPUs must not silently measure one GPU"""
    import subprocess
    import sys

    import torch

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("IG_BENCH_ONE_DEVICE", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--config", "tiny", "--steps", "5", "--warmup", "0",
                        "--moves-per-step", "1", "--no-cpu-baseline", "--nuisance-moves", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
