"""SourceModule whose functions are the CPU restatements in oracle/ig_oracle_*.c.

Each adapter receives the reference's argument list exactly as
cuda_lib_gl_single.py issues it and forwards it to the igo_* function.  Scalar
arguments are packed by the numpy dtype the caller passed, not by the kernel's
declared type (that is how pycuda behaves): an np.int32 handed to a ``float``
parameter is reinterpreted bit for bit (reference quirk Q8, CL:743).
"""
import ctypes as C
import os
import sys

import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from oracle import oracle_lib as ol  # noqa: E402

from .driver import DeviceAllocation  # noqa: E402
from .gpuarray import GPUArray  # noqa: E402

CALL_LOG = []  # (kernel name) in launch order, for launch-count statistics


def P(a):
    if isinstance(a, (GPUArray, DeviceAllocation)):
        return C.c_void_p(int(a))
    raise TypeError("pointer argument expected, got %r" % type(a))


def I(a):  # noqa: E741
    if isinstance(a, (np.floating,)):
        return C.c_int32(int(np.array([a], np.float32).view(np.int32)[0]))
    return C.c_int32(int(a))


def F(a):
    if isinstance(a, (np.integer, int)):
        return C.c_float(float(np.array([a], np.int32).view(np.float32)[0]))
    return C.c_float(float(a))


def L(a):
    return C.c_int64(int(a))


def _mk(name, conv, pick=None):
    fn = getattr(ol.lib(), "igo_" + name)

    def call(*args, block=None, grid=None, shared=0, texrefs=None, **kw):
        CALL_LOG.append(name)
        sel = args if pick is None else [args[i] for i in pick]
        assert len(sel) == len(conv), (name, len(sel), len(conv))
        fn(*[c(a) for c, a in zip(conv, sel)])

    return call


def _noop(name):
    def call(*args, **kw):
        CALL_LOG.append(name)

    return call


def _dead(name):
    def call(*args, **kw):
        raise RuntimeError("kernel %s is dead on the live path and has no restatement" % name)

    return call


def _table():
    t = {}
    t["fill_vect_dist"] = _mk("fill_vect_dist", [P, P, P, P, P, P, P, I, I], pick=[0, 1, 2, 3, 4, 5, 6, 11, 12])
    t["uni_fill_vect_dist"] = _mk("uni_fill_vect_dist", [P, P, P, P, P, P, P, I], pick=[0, 1, 2, 3, 4, 5, 6, 11])
    t["init_rng"] = _noop("init_rng")
    t["evaluate_likelihood_sparse"] = _mk("evaluate_likelihood_sparse", [P, P, P, P, F, P, P, P, P, P, P, L],
                                          pick=[0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12])
    t["eval_sub_likelihood"] = _mk("eval_sub_likelihood", [P, P, P, P, P, P, F, P, P, P, P, P, P, P, P, I, I])
    t["extract_sub_likelihood"] = _mk("extract_sub_likelihood", [P, P, P, P, P, P, F, P, P, P, P, P, P, I, I])
    t["slice_sp_mat"] = _mk("slice_sp_mat", [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, L])
    t["prepare_sparse_call"] = _mk("prepare_sparse_call", [P, P, P, P, I])
    t["eval_likelihood_on_zero"] = _mk("eval_likelihood_on_zero", [P, P, P, P, P, F, P, P, I])
    t["eval_all_likelihood_on_zero_1st"] = _mk("eval_all_likelihood_on_zero_1st", [P, P, P, P, P, F, P, P, P, P, I])
    t["eval_all_likelihood_on_zero_2nd"] = _mk("eval_all_likelihood_on_zero_2nd", [P, P, P, P, P, P])
    t["extract_uniq_mutations"] = _mk("extract_uniq_mutations", [P, I, I, P, P, P, I])
    t["eval_all_scores"] = _mk("eval_all_scores", [P, P, P, P, P, P, P])
    t["select_uniq_id_c"] = _mk("select_uniq_id_c", [P, P, P, P, I])
    t["make_old_2_new_id_c"] = _mk("make_old_2_new_id_c", [P, P, I])
    t["count_num"] = _mk("count_num", [P, I, P, I])
    t["explode_genome"] = _mk("explode_genome", [P, P, I])
    t["get_bounds"] = _mk("get_bounds", [P, I, I, P, P, P, P, I, I])
    t["extract_block"] = _mk("extract_block", [P, P, P, I, P, I, I, I, I])
    t["insert_block"] = _mk("insert_block", [P, P, P, I, I, P, P, I, I, I, I])
    # gl_update_pos(list_len,pos,vel,pos_gen,vel_gen,frag,old2new,id_contigs,max_id,n,...): only KA:4689-4692 survives
    t["gl_update_pos"] = _mk("renumber_id_c", [P, P, P, F, I], pick=[5, 6, 7, 8, 9])
    t["pop_out_frag"] = _mk("pop_out_frag", [P, P, P, I, I, I])
    t["flip_frag"] = _mk("flip_frag", [P, P, I, I])
    for k in (1, 2, 3):
        t["pop_in_frag_%d" % k] = _mk("pop_in_frag_%d" % k, [P, P, I, I, I, I, I])
    t["split_contig"] = _mk("split_contig", [P, P, P, I, I, I, I])
    t["paste_contigs"] = _mk("paste_contigs", [P, P, I, I, I, I])
    t["simple_copy"] = _mk("simple_copy", [P, P, I])
    t["copy_struct"] = _mk("copy_struct", [P, P, P, I])
    for k in ("update_gpu_vect_frags", "gpu_struct_2_pxl", "update_matrix", "update_gl_buffer",
              "prepare_sparse_call_4_gl", "set_null", "copy_gpu_array", "swap_activity_frag"):
        t[k] = _dead(k)
    return t


class SourceModule:
    def __init__(self, source, no_extern_c=False, options=None, **kw):
        self.source = source
        self._t = _table()

    def get_function(self, name):
        return self._t[name]
