"""Host-side mirror of the reference's ``sampler`` (cuda_lib_gl_single.py:91-3186) over the HIP C ABI."""
from __future__ import annotations

import numpy as np

from . import hip_lib

LIST_SIZE = np.array([1, 3, 5, 10, 20, 50, 200, 200], dtype=np.int32)  # CL:417
N_INSERT_BLOCKS = 6  # CL:192
PARAM_NAMES = ("kuhn", "lm", "c1", "slope", "d", "d_max", "fact", "v_inter")  # KA:91-100


def soa17_from_dict(S_o_A_frags, n):
    """17 x N int32 in KA:40-58 order; ``ori`` starts at +1 like create_gpu_struct (CL:537-541)."""
    out = np.zeros((17, n), np.int32)
    for k, name in enumerate(hip_lib.FRAG_FIELDS):
        if name == "ori":
            out[k] = 1
        elif name == "id":
            out[k] = np.arange(n)
        else:
            out[k] = S_o_A_frags[name]
    return out


def problem_to_context(prob, params=None, device_id=0, rank=0, world=1):
    """Upload a synth.SynthProblem (contacts, sub-fragment table, state, fixed parameters)."""
    ctx = hip_lib.Context(device_id)
    ctx.upload_subfrag_table(prob.np_sub_frags_2_frags)
    ctx.upload_contacts(prob.coo_row, prob.coo_col, prob.coo_cnt, prob.n_sub_frags, rank, world)
    max_bounds = LIST_SIZE[:N_INSERT_BLOCKS].max() * np.int32(np.round(prob.S_o_A_frags["sub_len"].mean()) + 1)  # CL:418-420
    ctx.set_insert_config(LIST_SIZE[:N_INSERT_BLOCKS], int(max_bounds))
    ctx.upload_state(soa17_from_dict(prob.S_o_A_frags, prob.n_frags))
    p = prob.params if params is None else params
    mean_kb = np.float32(prob.S_o_A_sub_frags["len_bp"].mean() / 1000.0)  # CL:231, 1120
    ctx.set_params([np.float32(p[k]) for k in PARAM_NAMES], mean_kb, 0)
    ctx.set_params([np.float32(p[k]) for k in PARAM_NAMES], mean_kb, 1)
    return ctx
