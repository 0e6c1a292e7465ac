"""ctypes binding of the CPU oracle (oracle/_build/libig_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, tools/gen_golden.py,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by instagraal_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libig_oracle.so")

FRAG_FIELDS = ("pos", "sub_pos", "id_c", "start_bp", "len_bp", "sub_len", "circ", "id", "prev", "next", "l_cont",
               "sub_l_cont", "l_cont_bp", "ori", "rep", "activ", "id_d")  # KA:40-58

MODE_LIBM, MODE_DET = 0, 1

PARAM_DTYPE = np.dtype([("kuhn", np.float32), ("lm", np.float32), ("c1", np.float32), ("slope", np.float32),
                        ("d", np.float32), ("d_max", np.float32), ("fact", np.float32), ("v_inter", np.float32)],
                       align=True)  # CL:235-247
INT3 = np.dtype([("x", np.int32), ("y", np.int32), ("z", np.int32)], align=True)


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("ig_oracle_ops.c", "ig_oracle_lik.c", "ig_oracle.h", "../include/ig_detmath.h")
            if os.path.exists(os.path.join(HERE, f))):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.igo_get_mode.restype = C.c_int
    return _lib


_ptr_cache = {}  # id(array) -> (weak reference, address): the sampler's buffers live as long as it does


def ptr(a):
    """address of a numpy array / FragStruct / None"""
    if a is None:
        return C.c_void_p(0)
    e = _ptr_cache.get(id(a))
    if e is not None and e[0]() is a:
        return e[1]
    if isinstance(a, FragStruct):
        return C.c_void_p(a.addr)
    assert a.flags["C_CONTIGUOUS"]
    p = C.c_void_p(a.ctypes.data)
    if a.base is None and a.flags["OWNDATA"]:  # (views are made per call: not worth an entry)
        if len(_ptr_cache) > 4096:
            for k in [k for k, v in _ptr_cache.items() if v[0]() is None]:
                del _ptr_cache[k]
        _ptr_cache[id(a)] = (weakref.ref(a), p)
    return p


class FragStruct:
    """17 int32 arrays + the packed pointer block of KA:40-58 (what GPUStruct.get_ptr() hands to a kernel)."""

    def __init__(self, n, data=None):
        self.n = int(n)
        for k in FRAG_FIELDS:
            if data is not None and k in data:
                arr = np.ascontiguousarray(np.array(data[k], dtype=np.int32, copy=True))
            else:
                arr = np.zeros(self.n, dtype=np.int32)
            assert arr.shape == (self.n,)
            setattr(self, k, arr)
        self._block = (C.c_void_p * 17)(*[getattr(self, k).ctypes.data for k in FRAG_FIELDS])
        self.addr = C.addressof(self._block)

    def copy_from_gpu(self):  # API parity with gpustruct.py:162
        return self

    def as_dict(self):
        return {k: getattr(self, k).copy() for k in FRAG_FIELDS}

    def soa17(self):
        return np.stack([getattr(self, k) for k in FRAG_FIELDS]).astype(np.int32)

    def assign(self, other):
        for k in FRAG_FIELDS:
            getattr(self, k)[:] = getattr(other, k)


def i32(v):
    return C.c_int32(int(v))


def f32(v):
    return C.c_float(float(v))


def set_mode(mode):
    lib().igo_set_mode(C.c_int(mode))


def set_threads(n):
    lib().igo_set_threads(C.c_int(n))


def last_limbs(n=25):
    hi = np.zeros(n, dtype=np.int64)
    lo = np.zeros(n, dtype=np.int64)
    lib().igo_last_limbs(ptr(hi), ptr(lo), C.c_int(n))
    return hi, lo


def eval_terms(s, s_tot, ob, params):
    s = np.ascontiguousarray(s, np.float32)
    s_tot = np.ascontiguousarray(s_tot, np.float32)
    ob = np.ascontiguousarray(ob, np.int32)
    n = s.size
    ex = np.zeros(n, np.float32)
    exc = np.zeros(n, np.float32)
    term = np.zeros(n, np.float64)
    q = np.zeros(n, np.int64)
    p = np.ascontiguousarray(params)
    lib().igo_eval_terms(ptr(s), ptr(s_tot), ptr(ob), C.c_int64(n), ptr(p), ptr(ex), ptr(exc), ptr(term), ptr(q))
    return ex, exc, term, q


def lgf_table():
    out = np.zeros(15, np.float64)
    lib().igo_lgf_table(ptr(out))
    return out
