"""P(s) ("Rippe") curve fit used by the sampler's host side.

Same public names and argument meaning as the reference module
``/root/reference/src/instagraal/optim_rippe_curve_update.py``: ``peval`` (l.21-31),
``log_residuals`` (l.34-49), ``estimate_param_rippe`` (l.64-106), ``estimate_max_dist_intra`` (l.120-134),
``estimate_max_dist_intra_nuis`` (l.137-149).  Host-only scipy code in the reference as well -- this is
set-up / nuisance bookkeeping, not part of the device path.
"""
import warnings

import numpy as np
from scipy.optimize import fsolve, leastsq

d = 2  # the exponential cut-off parameter is pinned (reference l.8)


def _rippe_shape(x, kuhn, lm, slope, dd):
    u = lm * x / kuhn
    return 0.53 * (kuhn ** -3.0) * np.power(u, slope) * np.exp((dd - 2) / (np.power(u, 2) + dd))


def peval(x, param):
    """param = (kuhn, lm, slope, A, ...): amplitude is param[3] whatever follows (callers rely on it)."""
    return param[3] * _rippe_shape(x, param[0], param[1], param[2], d)


def log_residuals(p, y, x):
    kuhn, lm, slope, A = p
    with np.errstate(invalid="ignore", divide="ignore"):
        model = (np.log(A) + np.log(0.53) - 3 * np.log(kuhn) + slope * (np.log(lm * x / kuhn))
                 + (d - 2) / (np.power((lm * x / kuhn), 2) + d))
    return y - model


def estimate_param_rippe(y_meas, x_bins):
    kuhn, lm, slope = 50, 9.6, -1.5
    A = np.max(y_meas)
    lower_fact = 7.0
    plsq = leastsq(log_residuals, [kuhn, lm, slope, A], args=(np.log(y_meas / lower_fact), x_bins))
    y_estim = peval(x_bins, plsq[0])
    kuhn_x, lm_x, slope_x, A_x = plsq[0]
    out = [kuhn_x, lm_x, slope_x, d, A_x]
    if np.any(np.isnan(np.array(out))) or slope_x >= 0:
        test = peval(x_bins, [kuhn, lm, slope, A])
        new_A = y_meas[0] * A / test.max()
        out = [kuhn, lm, slope, d, A * new_A]
        y_estim = peval(x_bins, [kuhn, lm, slope, new_A])
    return out, y_estim


def residual_4_max_dist(x, p):
    kuhn, lm, slope, dd, A, y = p
    x[np.isnan(x)] = 0
    x = np.abs(x)
    return np.abs(y - A * _rippe_shape(x, kuhn, lm, slope, dd))


def _solve_fsolve(p, val_inter, s0):
    kuhn, lm, slope, dd, A = p
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        return fsolve(residual_4_max_dist, s0, args=([kuhn, lm, slope, dd, A, val_inter]))[0]


try:  # MINPACK's hybrd as scipy.optimize.fsolve reaches it (scipy/optimize/_minpack_py.py, _root_hybr)
    from scipy.optimize import _minpack as _minpack_ext

    _hybrd = _minpack_ext._hybrd
except Exception:  # pragma: no cover - another scipy: the public function
    _hybrd = None


def _solve(p, val_inter, s0):
    """``fsolve(residual_4_max_dist, s0, args=...)[0]`` without fsolve's Python layers: this root is found twice per nuisance
    step, on the critical path of the host.  Same MINPACK routine, same arguments (xtol, maxfev, band, the forward-difference
    step from the dtype of the residual on the caller's s0, factor), the same residual with its parameter-only factors
    computed once -- same iterates, same bits (tests/test_cpu_abi_and_host.py compares the two on a grid)."""
    if _hybrd is None:
        return _solve_fsolve(p, val_inter, s0)
    kuhn, lm, slope, dd, A = p
    c0 = 0.53 * (kuhn ** -3.0)
    c1 = dd - 2

    def residual(x):
        if x.size != 1 or x[0] != x[0]:
            x[np.isnan(x)] = 0
        u = lm * np.abs(x) / kuhn
        return np.abs(val_inter - A * (c0 * np.power(u, slope) * np.exp(c1 / (np.power(u, 2) + dd))))

    iterate = residual
    if _quiet and c1 == 0:
        # d == 2 (the only value the reference ever uses, l.8): the exponential factor is exp(0 / ...) = 1 exactly, and what is left
        # is IEEE double arithmetic on values that are exact in double (float32 parameters) around ONE library call, numpy's own
        # pow -- the same loop for a Python float as for the one-element array.  Same bits as the expression above
        # (tests/test_cpu_abi_and_host.py compares the two residuals and the roots), an eighth of its time: MINPACK evaluates it
        # seven times per root, and the root is what a nuisance step costs on the host.
        fl, fk, fs, fA, fv, fc0, npow = float(lm), float(kuhn), float(slope), float(A), float(val_inter), float(c0), np.power

        def iterate(x):
            x0 = x[0]
            if x0 != x0:
                x[np.isnan(x)] = 0
                x0 = x[0]
            return abs(fv - fA * (fc0 * float(npow(fl * abs(x0) / fk, fs))))

    x0 = np.asarray(s0).flatten()
    # fsolve's shape check evaluates the residual once on the caller's (possibly float32) s0 and takes the forward-difference step
    # from the dtype of what comes back: a function of the dtypes of the arguments alone -- looked up, evaluated the first time
    key = tuple(type(v) for v in p) + (type(val_inter), x0.dtype)
    if _quiet:  # (the caller holds np.errstate(all="ignore") around a run of calls: nothing here can warn)
        eps = _eps_for.get(key)
        if eps is None:
            res = np.atleast_1d(residual(x0[:1]))
            eps = _eps_for[key] = np.finfo(res.dtype if np.issubdtype(res.dtype, np.inexact) else np.dtype(float)).eps
        return _hybrd(iterate, x0, (), 1, 1.49012e-08, 200 * (x0.size + 1), -10, -10, eps, 100, None)[0][0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        res = np.atleast_1d(residual(x0[:1]))
        dt = res.dtype if np.issubdtype(res.dtype, np.inexact) else np.dtype(float)
        return _hybrd(residual, x0, (), 1, 1.49012e-08, 200 * (x0.size + 1), -10, -10, np.finfo(dt).eps, 100, None)[0][0]


_quiet = False
_eps_for = {}


class quiet_runs:
    """``with quiet_runs():`` around a run of ``estimate_max_dist_intra_nuis`` calls (the nuisance loop: one root per step on the
    host's critical path): floating-point warnings off once for the whole run instead of a ``warnings.catch_warnings`` per call,
    and the dtype probe of the residual looked up instead of evaluated.  Same MINPACK call, same arguments, same bits."""

    def __enter__(self):
        global _quiet
        self._err = np.errstate(all="ignore")
        self._err.__enter__()
        self._was = _quiet
        _quiet = True
        return self

    def __exit__(self, *a):
        global _quiet
        _quiet = self._was
        return self._err.__exit__(*a)


def estimate_max_dist_intra(p, val_inter):
    return np.abs(_solve(p, val_inter, 500))


def estimate_max_dist_intra_nuis(p, val_inter, old_s):
    return _solve(p, val_inter, old_s)
