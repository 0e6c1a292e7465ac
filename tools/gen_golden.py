#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own sampler Python.

Runs only in the authoring container (needs /root/reference).  The reference's
``instagraal.cuda_lib_gl_single.sampler`` is imported unmodified; ``pycuda`` is
the functional fake in tools/fake_pycuda whose kernels are oracle/ig_oracle_*.c.
What this pins: everything the reference does on the HOST for the hot path --
neighbour draw and RNG consumption (CL:3103-3141), call order and stale buffers
(CL:1401-1465, 1918-1923), slice + sort (CL:1009-1069), argmax (CL:1435-1446),
apply + renumbering (CL:2094-2151, 2715-2881), genome distance (CL:665-716),
nuisance step (CL:2961-3051) -- for both arithmetic modes of the oracle kernels.

Only numeric inputs/outputs are stored; no reference source enters the repo.

usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools", "fake_pycuda"))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference/src")

CASES = {
    # name: (config, seed, n_moves, bomb, n_nuisance_from)
    "tiny_plain": ("tiny", 11, 80, False, None),
    "tiny_bomb": ("tiny", 12, 60, True, None),
    "tiny_nuis": ("tiny", 13, 30, False, 10),
}


def run_case(name, mode, outdir):
    from instagraal_amd import synth
    from oracle import oracle_lib as ol
    import pycuda.driver as cuda
    from pycuda import compiler as fake_compiler
    from instagraal.cuda_lib_gl_single import sampler as ref_sampler

    cfg, seed, n_moves, bomb, nuis_from = CASES[name]
    ol.set_mode(mode)
    prob = synth.make_problem(*synth.CONFIGS[cfg])
    kw = prob.sampler_kwargs()
    np.random.seed(seed)
    s = ref_sampler(*[kw[k] for k in kw])
    # what estimate_parameters_rippe does after the fit (CL:2343-2349), with fixed parameters
    p = prob.params
    par = np.array([(p["kuhn"], p["lm"], p["c1"], p["slope"], p["d"], p["d_max"], p["fact"], p["v_inter"])],
                   dtype=s.param_simu_rippe)
    s.param_simu = par
    s.param_simu_test = s.param_simu
    s.gpu_param_simu = cuda.mem_alloc(s.param_simu.nbytes)
    s.gpu_param_simu_test = cuda.mem_alloc(s.param_simu.nbytes)
    cuda.memcpy_htod(s.gpu_param_simu, s.param_simu)
    cuda.memcpy_htod(s.gpu_param_simu_test, s.param_simu_test)
    s.bins = np.arange(1.0, 60.0, 1.0)  # normally set by estimate_parameters_rippe (CL:2247)
    s.eval_likelihood_init()
    init_nz = float(s.gpu_curr_likelihood_nz.get()[0])

    if bomb:
        s.bomb_the_genome()
    list_frags = np.arange(0, s.n_new_frags)
    np.random.shuffle(list_frags)  # IG:213
    rec = dict(frag=[], cands=[], scores=[], ret=[], valid=[], state_every=[], states=[], nuis=[])
    fake_compiler.CALL_LOG.clear()
    for t, id_frag in enumerate(list_frags[:n_moves]):
        r = s.step_sampler(id_frag, 5, s.dt)
        c = list(s.candidates) + [-1] * (5 - len(s.candidates))
        sc = np.full(5 * 24, np.nan)
        sc[: len(s.all_scores)] = s.all_scores
        rec["frag"].append(int(id_frag))
        rec["cands"].append(c)
        rec["scores"].append(sc)
        rec["ret"].append([float(r[0]), float(r[1]), float(r[2]), float(r[3]), float(r[4]), float(r[5])])
        rec["valid"].append(s.gpu_list_valid_insert.get())
        if nuis_from is not None and t >= nuis_from:
            q = s.step_nuisance_parameters(s.dt, t, n_moves)
            rec["nuis"].append([float(q[0]), float(q[1]), float(q[2]), float(q[3]), float(q[4]),
                                float(np.ravel(q[5])[0]), float(q[6])])
        if t % 10 == 9 or t == n_moves - 1:
            s.gpu_vect_frags.copy_from_gpu()
            g = s.gpu_vect_frags
            rec["state_every"].append(t)
            rec["states"].append(np.stack([getattr(g, k) if k != "next" else g.next for k in ol.FRAG_FIELDS]).astype(np.int32))
    launches = len(fake_compiler.CALL_LOG) / float(n_moves)
    out = os.path.join(outdir, "%s_mode%d.npz" % (name, mode))
    np.savez_compressed(
        out, config=cfg, seed=seed, bomb=bomb, mode=mode, init_nz=init_nz, frag=np.array(rec["frag"], np.int32),
        cands=np.array(rec["cands"], np.int32), scores=np.array(rec["scores"]), ret=np.array(rec["ret"]),
        valid=np.array(rec["valid"], np.int32), state_every=np.array(rec["state_every"], np.int32),
        states=np.array(rec["states"], np.int32), nuis=np.array(rec["nuis"]), launches_per_move=launches,
        nuis_from=-1 if nuis_from is None else nuis_from,
        params=np.array([p[k] for k in ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")], np.float64),
        rng_after=np.array(np.random.get_state()[1][:8], np.uint32))
    print("wrote", out, "launches/move %.0f" % launches, "last ret", rec["ret"][-1])


def estimate_golden(outdir, mode):
    """the reference's own estimate_parameters_rippe (CL:2239-2372) driven as simu_single does
    (SS:157-171) on the synthetic 'small' problem: binned mean contacts, fitted parameters, cut-off, initial likelihood."""
    from instagraal_amd import synth
    from oracle import oracle_lib as ol
    import pycuda.driver as cuda  # noqa: F401
    from instagraal.cuda_lib_gl_single import sampler as ref_sampler

    ol.set_mode(mode)
    prob = synth.make_problem(*synth.CONFIGS["small"])
    kw = prob.sampler_kwargs()
    np.random.seed(5)
    s = ref_sampler(*[kw[k] for k in kw])
    g = s.gpu_vect_frags
    g.copy_from_gpu()
    id_start = np.nonzero(g.start_bp == 0)[0]
    max_dist_kb = g.l_cont_bp[id_start].max() / 1000.0
    mean_size_bin_kb = 1.8  # synthetic sub-fragments are log-normal around 1.8 kb (SS:161 takes the measured mean)
    mvt0 = float(s.mean_value_trans)
    s.estimate_parameters_rippe(max_dist_kb, mean_size_bin_kb / 2.0, False)
    par = s.param_simu
    out = os.path.join(outdir, "small_estimate_mode%d.npz" % mode)
    np.savez_compressed(out, config="small", max_dist_kb=max_dist_kb, size_bin_kb=mean_size_bin_kb / 2.0, mean_value_trans_in=mvt0,
                        mean_value_trans_out=float(s.mean_value_trans), bins_upd=np.asarray(s.bins_upd, np.float64),
                        mean_contacts_upd=np.asarray(s.mean_contacts_upd, np.float64), y_estim=np.asarray(s.y_estim, np.float64),
                        params=np.array([par[k][0] for k in ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")], np.float64),
                        init_nz=float(s.gpu_curr_likelihood_nz.get()[0]), init_full=float(s.likelihood_t))
    print("wrote", out, "params", [float(par[k][0]) for k in ("kuhn", "lm", "slope", "d", "d_max", "fact", "v_inter")],
          "nz", float(s.gpu_curr_likelihood_nz.get()[0]))


def host_helper_goldens(outdir):
    """Reference host helpers that need no kernels ('importable-and-runnable pieces')."""
    from instagraal import optim_rippe_curve_update as opti

    x = np.array([1.0, 2.5, 7.0, 20.0, 55.0, 160.0, 400.0, 900.0])
    p = [50.0, 9.6, -1.5, 3.0e5]
    y = opti.peval(x, p)
    dmax = opti.estimate_max_dist_intra([50.0, 9.6, -1.5, 2.0, 3.0e5], 5e-3)
    dmax_n = opti.estimate_max_dist_intra_nuis([50.0, 9.6, -1.45, 2.0, 3.0e5], 5e-3, dmax)
    fit, y_est = opti.estimate_param_rippe(opti.peval(x, p) * 1.0, x)
    # P(s) itself: the one place the reference states the Rippe curve in runnable form (optim_rippe_curve_update.peval,
    # reference l.21-31).  With A = fact it is the function kernel_sparse_adapt.cu:153-163 evaluates in float
    # (c1 = 0.53 (lm/kuhn)^slope kuhn^-3, CL:2206-2221); the grid pins the oracle's rippe (both arithmetic modes) and the
    # HIP kernels' to it: tests/test_oracle_golden.py::test_rippe_against_reference_peval, tests/test_hip_configs.py.
    from instagraal_amd import synth

    rp = synth.rippe_params(1.8)
    grid = np.exp(np.linspace(np.log(0.05), np.log(0.999 * float(np.float32(rp["d_max"]))), 4000)).astype(np.float32)
    ref = opti.peval(grid.astype(np.float64), [float(np.float32(rp["kuhn"])), float(np.float32(rp["lm"])),
                                               float(np.float32(rp["slope"])), float(np.float32(rp["fact"]))])
    np.savez(os.path.join(outdir, "host_helpers.npz"), x=x, p=np.array(p), peval=y, dmax=dmax, dmax_nuis=dmax_n,
             # the fit: leastsq stops anywhere along a valley (kuhn, lm and the amplitude are not separately identifiable) and
             # WHERE varies from run to run with the last bits of numpy's log: only what every run reproduces is stored --
             # slope (8 decimals), d, the fitted curve in float32
             fit_slope=np.float64(round(float(fit[2]), 8)), fit_d=np.float64(fit[3]), y_est=np.asarray(y_est, np.float32),
             # ... and what the valley leaves fixed, in double: the amplitude product A 0.53 kuhn^-3 (lm / kuhn)^slope (the one
             # identifiable combination of kuhn, lm, A) and the fitted curve itself -- compared at 1e-9 relative
             fit_amp=np.float64(float(fit[4]) * 0.53 * abs(float(fit[0])) ** -3 * (abs(float(fit[1])) / abs(float(fit[0]))) ** float(fit[2])),
             y_est64=np.asarray(y_est, np.float64),
             rippe_grid_s=grid, rippe_grid_peval=np.asarray(ref, np.float64),
             rippe_grid_params=np.array([rp[k] for k in ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")], np.float64))
    print("wrote host_helpers.npz", dmax, dmax_n, fit)


# fields that are NOT bit-reproducible from run to run and are stored for a tolerance test only (their tests say which):
# host_helpers.npz:y_est64 -- the fitted curve in double, last ulp follows numpy's log alignment path (tests use rtol 1e-9)
CHECK_RTOL = {("host_helpers.npz", "y_est64"): 1e-9, ("host_helpers.npz", "fit_amp"): 1e-9}


def check(a):
    """regenerate into a temp dir, diff against a.out: 0 = every array identical (the CHECK_RTOL fields within their tolerance)"""
    import subprocess

    tmp = tempfile.mkdtemp(prefix="ig_golden_check_")
    cmd = [sys.executable, os.path.abspath(__file__), "--out", tmp, "--cases", a.cases, "--extra", a.extra]
    if a.only_host_helpers:
        cmd.append("--only-host-helpers")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    bad = n_arrays = 0
    for f in sorted(os.listdir(tmp)):
        if not f.endswith(".npz"):
            continue
        ref_path = os.path.join(a.out, f)
        if not os.path.exists(ref_path):
            print("MISSING in %s: %s" % (a.out, f))
            bad += 1
            continue
        new, old = np.load(os.path.join(tmp, f), allow_pickle=False), np.load(ref_path, allow_pickle=False)
        if sorted(new.files) != sorted(old.files):
            print("KEYS differ in %s: %s" % (f, sorted(set(new.files) ^ set(old.files))))
            bad += 1
        for k in sorted(set(new.files) & set(old.files)):
            n_arrays += 1
            x, y = new[k], old[k]
            tol = CHECK_RTOL.get((f, k))
            same = x.shape == y.shape and x.dtype == y.dtype and (
                np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")) if x.dtype.kind != "U" else np.array_equal(x, y))
            if not same and tol is not None and x.shape == y.shape:
                same = bool(np.allclose(x, y, rtol=tol, atol=0))
                if same:
                    print("within rtol %g (not bit for bit, as documented): %s:%s" % (tol, f, k))
            if not same:
                print("DIFFERS: %s:%s" % (f, k))
                bad += 1
    print("gen_golden --check: %d arrays compared, %d differences" % (n_arrays, bad))
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--extra", default="estimate")
    ap.add_argument("--only-host-helpers", action="store_true")
    ap.add_argument("--check", action="store_true",
                    help="regenerate everything into a temporary directory and compare it with the committed files (--out), array for array")
    a = ap.parse_args()
    if a.check:
        sys.exit(check(a))
    os.makedirs(a.out, exist_ok=True)
    os.chdir(tempfile.mkdtemp())  # the reference's log.py drops a log file in the CWD
    if a.only_host_helpers:
        host_helper_goldens(a.out)
        return
    for name in [n for n in a.cases.split(",") if n]:
        for mode in (0, 1):
            run_case(name, mode, a.out)
    if "estimate" in a.extra.split(","):
        for mode in (0, 1):
            estimate_golden(a.out, mode)
    host_helper_goldens(a.out)


if __name__ == "__main__":
    main()
