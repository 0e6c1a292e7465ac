"""P(s) curve helpers of the reference's host side, restated (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/instagraal/optim_rippe_curve_update.py ("OPTI"):
peval l.21-31, log_residuals l.34-49, estimate_param_rippe l.64-106,
residual_4_max_dist l.109-117, estimate_max_dist_intra l.120-134, _nuis l.137-149.
scipy.optimize (MINPACK leastsq / fsolve) is the same third-party code the reference calls.
"""
import warnings

import numpy as np
from scipy.optimize import fsolve, leastsq

D = 2  # OPTI l.8


def peval(x, param):
    return param[3] * (0.53 * (param[0] ** -3.0) * np.power((param[1] * x / param[0]), (param[2]))
                       * np.exp((D - 2) / (np.power((param[1] * x / param[0]), 2) + D)))


def log_residuals(p, y, x):
    kuhn, lm, slope, A = p
    with np.errstate(invalid="ignore", divide="ignore"):
        rippe = (np.log(A) + np.log(0.53) - 3 * np.log(kuhn) + slope * (np.log(lm * x / kuhn))
                 + (D - 2) / (np.power((lm * x / kuhn), 2) + D))
    return y - rippe


def estimate_param_rippe(y_meas, x_bins):
    kuhn, lm, slope = 50, 9.6, -1.5
    A = np.max(y_meas)
    p0 = [kuhn, lm, slope, A]
    plsq = leastsq(log_residuals, p0, args=(np.log(y_meas / 7.0), x_bins))
    y_estim = peval(x_bins, plsq[0])
    kuhn_x, lm_x, slope_x, A_x = plsq[0]
    out = [kuhn_x, lm_x, slope_x, D, A_x]
    if np.any(np.isnan(np.array(out))) or slope_x >= 0:
        test = peval(x_bins, [kuhn, lm, slope, A])
        new_A = y_meas[0] * A / test.max()
        out = [kuhn, lm, slope, D, A * new_A]
        y_estim = peval(x_bins, [kuhn, lm, slope, new_A])
    return out, y_estim


def residual_4_max_dist(x, p):
    kuhn, lm, slope, d, A, y = p
    x[np.isnan(x)] = 0
    x = np.abs(x)
    rippe = A * (0.53 * (kuhn ** -3.0) * np.power((lm * x / kuhn), slope)
                 * np.exp((d - 2) / (np.power((lm * x / kuhn), 2) + d)))
    return np.abs(y - rippe)


def estimate_max_dist_intra(p, val_inter):
    kuhn, lm, slope, d, A = p
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        x = fsolve(residual_4_max_dist, 500, args=([kuhn, lm, slope, d, A, val_inter]))
    return np.abs(x[0])


def estimate_max_dist_intra_nuis(p, val_inter, old_s):
    kuhn, lm, slope, d, A = p
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        x = fsolve(residual_4_max_dist, old_s, args=([kuhn, lm, slope, d, A, val_inter]))
    return x[0]
