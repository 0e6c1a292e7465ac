"""ctypes binding of libinstagraal_hip.so (the C ABI of include/instagraal_hip.h).

There is no CPU fallback: if the shared library or a HIP device is missing, every entry
point raises.  The library is built in-tree by ``__graft_entry__.build()`` / ``build_lib()``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "libinstagraal_hip.so")
SRC = os.path.join(HERE, "csrc", "ig_hip.hip")
SRC_HOST = os.path.join(HERE, "csrc", "ig_draw.cpp")  # host-only part: the candidate draw
DEPS = [SRC, SRC_HOST] + [os.path.join(HERE, "csrc", f) for f in ("ig_ops.cuh", "ig_common.cuh", "ig_model.cuh", "ig_kernels_setup.cuh",
                                                          "ig_kernels_score.cuh", "ig_kernels_screen.cuh", "ig_kernels_commit.cuh",
                                                          "ig_kernels_nuis.cuh", "ig_host_core.inc", "ig_host_upload.inc",
                                                          "ig_host_batch.inc", "ig_host_nuis.inc", "ig_host_debug.inc")] + \
       [os.path.join(ROOT, "include", f) for f in ("ig_detmath.h", "ig_detmath_tables.h", "instagraal_hip.h")]

N_TMP_STRUCT = 24
MAX_CANDIDATES = 16

FRAG_FIELDS = ("pos", "sub_pos", "id_c", "start_bp", "len_bp", "sub_len", "circ", "id", "prev", "next", "l_cont",
               "sub_l_cont", "l_cont_bp", "ori", "rep", "activ", "id_d")  # kernel_sparse_adapt.cu:40-58


class MoveResult(C.Structure):
    _fields_ = [("o", C.c_double), ("dist", C.c_double), ("mean_len", C.c_double), ("op_sampled", C.c_int32),
                ("id_f_sampled", C.c_int32), ("n_contigs", C.c_int32), ("n_candidates", C.c_int32),
                ("n_slice", C.c_int64), ("n_evals", C.c_int64), ("bytes_min", C.c_int64), ("error", C.c_int32),
                ("pad", C.c_int32)]


MOVE_RESULT_DTYPE = np.dtype([("o", np.float64), ("dist", np.float64), ("mean_len", np.float64),
                              ("op_sampled", np.int32), ("id_f_sampled", np.int32), ("n_contigs", np.int32),
                              ("n_candidates", np.int32), ("n_slice", np.int64), ("n_evals", np.int64),
                              ("bytes_min", np.int64), ("error", np.int32), ("pad", np.int32)])
assert MOVE_RESULT_DTYPE.itemsize == C.sizeof(MoveResult)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(LIB_PATH) for d in DEPS):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value", "-pthread", "-o", LIB_PATH, SRC, SRC_HOST]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libinstagraal_hip.so is missing (%s): run __graft_entry__.build(); "
                               "the MI355X path has no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME, requested under an
        # unversioned name), so if our library pulled in /opt/rocm's copy first a later `import torch` would bring a
        # second runtime that sees no devices.  Loading torch first makes our NEEDED entry resolve to its copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # the product library, nothing else -- unless the caller says in so many words that this is a tuning session
        # (IG_DEBUG_TUNING=1 IG_HIP_LIB=<a build of the same source with other -D constants>: tools/diff_pass_time.py)
        path = LIB_PATH
        if os.environ.get("IG_HIP_LIB"):
            if os.environ.get("IG_DEBUG_TUNING") == "1":
                path = os.environ["IG_HIP_LIB"]
            else:  # (said out loud: a deployment that used to point IG_HIP_LIB somewhere else would load another build silently)
                import warnings

                warnings.warn("IG_HIP_LIB=%s is ignored without IG_DEBUG_TUNING=1: loading the product library %s"
                              % (os.environ["IG_HIP_LIB"], LIB_PATH), RuntimeWarning, stacklevel=2)
        _lib = C.CDLL(path)
        _lib.ig_last_error.restype = C.c_char_p
        _lib.ig_partials_count.restype = C.c_int64
        _lib.ig_partials_device_ptr.restype = C.c_void_p
        _lib.ig_partials_count.argtypes = [C.c_void_p]
        _lib.ig_partials_device_ptr.argtypes = [C.c_void_p]
    return _lib


class HipError(RuntimeError):
    pass


def _ck(rc):
    if rc != 0:
        raise HipError(lib().ig_last_error().decode())


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


class _NoLock:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_LOCK = _NoLock()


class Neighbours:
    """The jump distributions of setup_distri_frags (CL:3053-3101) in the library's host memory + the draw of
    return_neighbours (CL:3103-3141) on numpy's legacy MT19937 stream (csrc/ig_draw.cpp).  Needs no GPU."""

    def __init__(self, indptr, xk, pk, n_frags, blacklisted=()):
        self._h = C.c_void_p()
        ip = np.ascontiguousarray(indptr, np.int64)
        x = np.ascontiguousarray(xk, np.int32)
        p = np.ascontiguousarray(pk, np.float32)
        b = np.ascontiguousarray(list(blacklisted), np.int32)
        assert ip.size == n_frags + 1 and x.size == p.size == ip[-1]
        _ck(lib().ig_neighbours_create(_p(ip), _p(x), _p(p), C.c_int32(int(n_frags)), _p(b) if b.size else C.c_void_p(0),
                                       C.c_int32(b.size), C.byref(self._h)))
        self.n_frags = int(n_frags)

    def __del__(self):
        try:
            if self._h:
                lib().ig_neighbours_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @staticmethod
    def take_numpy_state():
        """(key[624] uint32 copy, pos, rest) of numpy's GLOBAL legacy generator"""
        st = np.random.get_state()
        assert st[0] == "MT19937"
        return np.array(st[1], dtype=np.uint32, copy=True), int(st[2]), st[3:]

    @staticmethod
    def put_numpy_state(key, pos, rest):
        np.random.set_state(("MT19937", key, int(pos)) + tuple(rest))

    _mt_cache = [None, 0, None]

    @staticmethod
    def numpy_mt_address():
        """address of numpy's GLOBAL legacy generator's mt19937_state {uint32 key[624]; int pos} (numpy/random/src/mt19937/mt19937.h),
        through the bit generator's own ctypes interface -- or 0 where the layout check fails.  np.random.get_state() +
        set_state() cost 50 - 130 us per pair (a tuple and a 2.5 KB copy each way): more than the draw of one move's candidates
        through numpy itself; on the state in place a draw is a few microseconds.  (np.random.seed / set_state work in place.)"""
        bg = np.random.mtrand._rand._bit_generator
        if Neighbours._mt_cache[0] is not bg:
            addr = 0
            try:
                a = int(bg.ctypes.state_address)
                st = np.random.get_state()
                key = np.frombuffer((C.c_uint32 * 624).from_address(a), np.uint32)
                if st[0] == "MT19937" and np.array_equal(key, st[1]) and C.c_int32.from_address(a + 624 * 4).value == int(st[2]):
                    addr = a
            except Exception:
                addr = 0
            Neighbours._mt_cache = [bg, addr, getattr(bg, "lock", None)]
        return Neighbours._mt_cache[1]

    @staticmethod
    def numpy_mt_lock():
        """the bit generator's own lock (numpy's legacy functions take it around every draw): held around the library calls that
        work on the state in place -- they run with the GIL released, and ig_step_batch_draw's drawing thread writes key / pos for
        the whole call, so another Python thread's np.random.* would race with it.  (Where the layout probe of numpy_mt_address
        fails the callers fall back to get_state / set_state copies, which need no lock.)"""
        Neighbours.numpy_mt_address()
        lk = Neighbours._mt_cache[2]
        return lk if lk is not None else _NO_LOCK

    def draw(self, frags, n_neighbours):
        """candidate lists of consecutive moves, consuming numpy's global generator exactly as successive
        return_neighbours calls would -> (n, n_neighbours) int32, sorted, -1 padded"""
        f = np.ascontiguousarray(frags, np.int32)
        out = np.full((f.size, int(n_neighbours)), -1, np.int32)
        addr = self.numpy_mt_address()
        if addr:  # on numpy's state in place
            with self.numpy_mt_lock():
                rc = lib().ig_neighbours_draw(self._h, C.c_void_p(addr), C.c_void_p(addr + 624 * 4), _p(f), C.c_int32(f.size),
                                              C.c_int32(int(n_neighbours)), _p(out))
            _ck(rc)
            return out
        key, pos, rest = self.take_numpy_state()
        cpos = C.c_int32(pos)
        rc = lib().ig_neighbours_draw(self._h, _p(key), C.byref(cpos), _p(f), C.c_int32(f.size), C.c_int32(int(n_neighbours)), _p(out))
        self.put_numpy_state(key, cpos.value, rest)
        _ck(rc)
        return out


def _neighbours_draw_nuisance(self, frags, n_neighbours, skip_normal_3=False):
    """the generator stream of a run of moves with nuisance sampling: per move the candidate list, then choice(4), the
    standard normal behind normal(0, sigma), rand() -- numpy's global generator is advanced exactly as the reference's loop
    (step_sampler; step_nuisance_parameters) would -> (cands, id_modif, normal, uniform)"""
    f = np.ascontiguousarray(frags, np.int32)
    n = f.size
    cands = np.full((n, int(n_neighbours)), -1, np.int32)
    idm = np.zeros(n, np.int32)
    g = np.zeros(n, np.float64)
    u = np.zeros(n, np.float64)
    key, pos, rest = self.take_numpy_state()
    cpos, hg, gz = C.c_int32(pos), C.c_int32(int(rest[0])), C.c_double(float(rest[1]))
    rc = lib().ig_neighbours_draw_nuisance(self._h, _p(key), C.byref(cpos), C.byref(hg), C.byref(gz), _p(f), C.c_int32(n),
                                           C.c_int32(int(n_neighbours)), C.c_int32(int(bool(skip_normal_3))), _p(cands), _p(idm), _p(g), _p(u))
    self.put_numpy_state(key, cpos.value, (hg.value, gz.value))
    _ck(rc)
    return cands, idm, g, u


Neighbours.draw_nuisance = _neighbours_draw_nuisance


def set_batch_width(w):
    """moves scored per launch by ``Context.step_batch`` (1 = one move at a time; results do not depend on it)"""
    _ck(lib().ig_set_batch_width(C.c_int(int(w))))


def set_window(w):
    """slots of the window of scored moves ig_step_batch keeps from launch to launch (0: off -- the batches of rounds 1 - 4)"""
    _ck(lib().ig_set_window(C.c_int(int(w))))


def set_nuis_width(w):
    """moves scored ahead per launch by the nuisance-on loop (``Context.nuis_step_begin``); 0 = follow the run lengths"""
    _ck(lib().ig_set_nuis_width(C.c_int(int(w))))


def set_nuis_screen(on):
    """the Metropolis test of the nuisance steps from the screened pass where it decides (default), or always from the exact one"""
    _ck(lib().ig_set_nuis_screen(C.c_int(int(on))))


def set_nuis_hist(on):
    """the screened pass starts from the histogram of the cis contacts' distances where that pays (1, default: a cost model decides),
    always (2), or never (0)"""
    _ck(lib().ig_set_nuis_hist(C.c_int(int(on))))


def set_nuis_chain(on):
    """runs of (move, nuisance step) pairs decided on the device, as far as they need no host (default), or one pair per library call"""
    global _nuis_chain
    _nuis_chain = bool(on)
    _ck(lib().ig_set_nuis_chain(C.c_int(int(on))))


_nuis_chain = None


def nuis_chain_wanted():
    """what ``set_nuis_chain`` / IG_NUIS_CHAIN say (default: on)"""
    if _nuis_chain is not None:
        return bool(_nuis_chain)
    return os.environ.get("IG_NUIS_CHAIN", "1") != "0"


CHAIN_MAX = 64  # test parameter sets per ig_nuis_chain_begin (csrc/ig_common.cuh)
CHAIN_REASONS = ("sets used up", "test", "conflict", "pending", "overflow", "no scored slots", "unsupported")


def debug_set_zero_inject(every):
    """tests: the decide step treats every n-th move of a two-tier batch as one whose contenders hold a score of exactly 0.0
    (the move is scored again with every column exact); 0 = off"""
    _ck(lib().ig_debug_set_zero_inject(C.c_int(int(every))))


def debug_set_full_hist(on):
    """from-scratch pass over all contacts: tiles of trans pairs only from their count histograms (default) or contact by
    contact -- same exact sums (tests)"""
    _ck(lib().ig_debug_set_full_hist(C.c_int(int(on))))


class Context:
    """One handle per sampler (ig_create .. ig_destroy)."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        _ck(lib().ig_create(C.c_int(device_id), C.byref(self._h)))
        self.N = 0
        self.M = 0

    def close(self):
        if self._h:
            lib().ig_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- upload
    def upload_contacts(self, row, col, cnt, M, rank=0, world=1):
        row = np.ascontiguousarray(row, np.int32)
        col = np.ascontiguousarray(col, np.int32)
        cnt = np.ascontiguousarray(cnt, np.int32)
        _ck(lib().ig_upload_contacts(self._h, _p(row), _p(col), _p(cnt), C.c_int64(row.size), C.c_int32(M), C.c_int32(rank),
                                     C.c_int32(world)))
        self.M = int(M)

    def upload_subfrag_table(self, table):
        """table: (M,) structured float4 or (M, 4) float32"""
        t = np.ascontiguousarray(table)
        if t.dtype.names:
            t = np.stack([t["x"], t["y"], t["z"], t["w"]], axis=1)
        t = np.ascontiguousarray(t, np.float32)
        _ck(lib().ig_upload_subfrag_table(self._h, _p(t), C.c_int32(t.shape[0])))
        self.M = int(t.shape[0])

    def upload_state(self, soa17):
        s = np.ascontiguousarray(soa17, np.int32)
        assert s.ndim == 2 and s.shape[0] == 17
        _ck(lib().ig_upload_state(self._h, _p(s), C.c_int32(s.shape[1])))
        self.N = int(s.shape[1])

    def download_state(self):
        out = np.zeros((17, self.N), np.int32)
        _ck(lib().ig_download_state(self._h, _p(out)))
        return out

    def set_params(self, p8, mean_subfrag_kb, which=0):
        p = np.ascontiguousarray(p8, np.float32)
        assert p.size == 8
        _ck(lib().ig_set_params(self._h, _p(p), C.c_float(float(mean_subfrag_kb)), C.c_int(which)))

    def set_insert_config(self, list_bounds6, max_bounds_insert):
        b = np.ascontiguousarray(list_bounds6, np.int32)
        assert b.size == 6
        _ck(lib().ig_set_insert_config(self._h, _p(b), C.c_int32(int(max_bounds_insert))))

    def set_initial_genome(self, init_prev, init_next, orientable, blacklisted=()):
        ip = np.ascontiguousarray(init_prev, np.int32)
        inn = np.ascontiguousarray(init_next, np.int32)
        o = np.ascontiguousarray(orientable, np.int32)
        b = np.ascontiguousarray(list(blacklisted), np.int32)
        _ck(lib().ig_set_initial_genome(self._h, _p(ip), _p(inn), _p(o), _p(b) if b.size else C.c_void_p(0), C.c_int32(b.size)))

    # ---- likelihood
    def full_likelihood(self, which=0, use_prev_tables=False):
        nz = C.c_double()
        z = C.c_double()
        limbs = np.zeros(5, np.int64)
        _ck(lib().ig_full_likelihood(self._h, C.c_int(which), C.c_int(int(use_prev_tables)), C.byref(nz), C.byref(z), _p(limbs)))
        return nz.value, z.value, limbs

    # ---- moves
    def score_move(self, frag_a, cands):
        c = np.ascontiguousarray(cands, np.int32)
        out = np.zeros(c.size * N_TMP_STRUCT, np.float64)
        _ck(lib().ig_score_move(self._h, C.c_int32(int(frag_a)), _p(c), C.c_int32(c.size), _p(out)))
        return out

    def apply(self, frag_a, frag_b, op):
        _ck(lib().ig_apply(self._h, C.c_int32(int(frag_a)), C.c_int32(int(frag_b)), C.c_int32(int(op))))

    def step(self, frag_a, cands, want_scores=True):
        c = np.ascontiguousarray(cands, np.int32)
        res = MoveResult()
        sc = np.zeros(c.size * N_TMP_STRUCT, np.float64) if want_scores else None
        _ck(lib().ig_step(self._h, C.c_int32(int(frag_a)), _p(c), C.c_int32(c.size), C.byref(res), _p(sc)))
        return res, sc

    def step_draw(self, neighbours, frag_a, n_neighbours, cands=None, want_scores=True):
        """ONE complete step_sampler call in one library call (ig_step_draw): the candidate draw on numpy's generator state in
        place (``cands`` given: the caller's list), lists and results through mapped host memory, the move decided and applied as a
        batch of one -> (MoveResult, scores [C x 24], candidate list).  ``want_scores=False`` -> scores None, and the library is free to
        score the move in two tiers (the exact kernel for the columns that can still win only)"""
        fn = lib().ig_step_draw
        if fn.argtypes is None:
            fn.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32] + [C.c_void_p] * 4
        io = self.__dict__.get("_step_io")
        if io is None:
            sc = np.zeros(MAX_CANDIDATES * N_TMP_STRUCT, np.float64)
            io = self._step_io = (sc, sc.ctypes.data, (C.c_int32 * MAX_CANDIDATES)(), C.c_int32(0))
        sc, sc_addr, cbuf, ncand = io
        res = MoveResult()
        if cands is None:
            addr = Neighbours.numpy_mt_address()
            if not addr:  # (a numpy whose generator state cannot be reached in place: two calls)
                row = neighbours.draw(np.array([frag_a], np.int32), int(n_neighbours))[0]
                lst = [int(x) for x in row if x >= 0]
                r, s2 = self.step(int(frag_a), lst, want_scores=want_scores)
                return r, s2, lst
            with Neighbours.numpy_mt_lock():
                rc = fn(self._h, neighbours._h, addr, addr + 624 * 4, int(frag_a), int(n_neighbours), cbuf, C.byref(ncand), C.byref(res),
                        sc_addr if want_scores else None)
        else:
            n = len(cands)
            if n > MAX_CANDIDATES:
                raise HipError("a move needs 1..%d candidates (got %d)" % (MAX_CANDIDATES, n))
            cbuf[:n] = [int(x) for x in cands]
            ncand.value = n
            rc = fn(self._h, None, None, None, int(frag_a), 0, cbuf, C.byref(ncand), C.byref(res), sc_addr if want_scores else None)
        if rc != 0:
            _ck(rc)
        n = ncand.value
        return res, (sc[:n * N_TMP_STRUCT].copy() if want_scores else None), cbuf[:n]

    def debug_step_stats(self):
        """ig_step_draw: calls that went through mapped memory, and how many of them the one-move tail finished"""
        o = np.zeros(2, np.int64)
        _ck(lib().ig_debug_step_stats(self._h, _p(o)))
        return dict(calls=int(o[0]), tails=int(o[1]))

    def step_batch(self, frags, cands):
        """frags: (n,), cands: (n, max_c) -1 padded -> structured array of MOVE_RESULT_DTYPE"""
        f = np.ascontiguousarray(frags, np.int32)
        c = np.ascontiguousarray(cands, np.int32)
        assert c.ndim == 2 and c.shape[0] == f.size
        res = np.zeros(f.size, MOVE_RESULT_DTYPE)
        _ck(lib().ig_step_batch(self._h, C.c_int32(f.size), _p(f), _p(c), C.c_int32(c.shape[1]), _p(res)))
        return res

    def step_batch_draw(self, neighbours, frags, n_neighbours):
        """len(frags) complete step_sampler calls, candidate draw included (numpy's global generator is advanced exactly as
        the reference's loop would) -> (results, candidate lists)"""
        f = np.ascontiguousarray(frags, np.int32)
        res = np.zeros(f.size, MOVE_RESULT_DTYPE)
        cands = np.full((f.size, int(n_neighbours)), -1, np.int32)
        addr = Neighbours.numpy_mt_address()
        if addr:  # on numpy's generator state in place (this thread waits inside the call while the library's thread draws)
            with Neighbours.numpy_mt_lock():
                rc = lib().ig_step_batch_draw(self._h, neighbours._h, C.c_void_p(addr), C.c_void_p(addr + 624 * 4), C.c_int32(f.size), _p(f),
                                              C.c_int32(int(n_neighbours)), _p(cands), _p(res))
            _ck(rc)
            return res, cands
        key, pos, rest = Neighbours.take_numpy_state()
        cpos = C.c_int32(pos)
        rc = lib().ig_step_batch_draw(self._h, neighbours._h, _p(key), C.byref(cpos), C.c_int32(f.size), _p(f),
                                      C.c_int32(int(n_neighbours)), _p(cands), _p(res))
        Neighbours.put_numpy_state(key, cpos.value, rest)
        _ck(rc)
        return res, cands

    # ---- a move and the nuisance step behind it, in flight together
    def nuis_begin(self, frag_a, cands, p_test8, mean_subfrag_kb):
        c = np.ascontiguousarray(cands, np.int32)
        p = np.ascontiguousarray(p_test8, np.float32)
        _ck(lib().ig_nuis_begin(self._h, C.c_int32(int(frag_a)), _p(c), C.c_int32(c.size), _p(p), C.c_float(float(mean_subfrag_kb))))

    def links_inverse(self):
        return bool(lib().ig_links_inverse(self._h))

    def nuis_run_begin(self, frags, cands):
        """the lists of a run of (move, nuisance step) pairs: the moves are scored ahead in batches (ig_nuis_run_begin)"""
        f = np.ascontiguousarray(frags, np.int32)
        c = np.ascontiguousarray(cands, np.int32)
        assert c.ndim == 2 and c.shape[0] == f.size
        _ck(lib().ig_nuis_run_begin(self._h, C.c_int32(f.size), _p(f), _p(c), C.c_int32(c.shape[1])))

    def nuis_step_begin(self, move, p_test8, mean_subfrag_kb):
        p = np.ascontiguousarray(p_test8, np.float32)
        _ck(lib().ig_nuis_step_begin(self._h, C.c_int32(int(move)), _p(p), C.c_float(float(mean_subfrag_kb))))

    def nuis_step_next(self, temperature, u, p_next_rejected, p_next_accepted, mean_subfrag_kb, has_next):
        """end of the step in flight + acceptance test + promotion + begin of the next step (ig_nuis_step_next)
        -> (move result, nz_test, z_test, accepted: 0 / 1 / 2 = undecided / 3 = accepted ahead of its exact pass: nz_test is the
        screened midpoint, the exact value comes from ``nuis_exact_result``)"""
        res = MoveResult()
        nz, z, acc = C.c_double(), C.c_double(), C.c_int32()
        pr = None if p_next_rejected is None else np.ascontiguousarray(p_next_rejected, np.float32)
        pa = None if p_next_accepted is None else np.ascontiguousarray(p_next_accepted, np.float32)
        _ck(lib().ig_nuis_step_next(self._h, C.c_double(float(temperature)), C.c_double(float(u)), _p(pr), _p(pa),
                                    C.c_float(float(mean_subfrag_kb)), C.c_int32(int(has_next)), C.byref(res), C.byref(nz), C.byref(z),
                                    C.byref(acc)))
        return res, nz.value, z.value, acc.value

    def nuis_chain_begin(self, move, p_tests, u, temperature, mean_subfrag_kb):
        """the pairs move .. move + K - 1 of the run as far as the device can take them alone (ig_nuis_chain_begin; asynchronous):
        p_tests (K, 8) float32, u / temperature (K,) float64"""
        p = np.ascontiguousarray(p_tests, np.float32)
        uu = np.ascontiguousarray(u, np.float64)
        tt = np.ascontiguousarray(temperature, np.float64)
        assert p.ndim == 2 and p.shape[1] == 8 and uu.size == p.shape[0] == tt.size
        _ck(lib().ig_nuis_chain_begin(self._h, C.c_int32(int(move)), C.c_int32(p.shape[0]), _p(p), _p(uu), _p(tt), C.c_float(float(mean_subfrag_kb))))

    def nuis_chain_done(self):
        return bool(lib().ig_nuis_chain_done(self._h))

    def nuis_chain_end(self):
        """-> (pairs completed: each a move decided and a step rejected, index into CHAIN_REASONS of why the chain ended)"""
        n, r = C.c_int32(), C.c_int32()
        _ck(lib().ig_nuis_chain_end(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def debug_nuis_chain_stats(self):
        o = np.zeros(10, np.int64)
        _ck(lib().ig_debug_nuis_chain_stats(self._h, _p(o)))
        return dict(calls=int(o[0]), segments=int(o[1]), pairs=int(o[2]), ends={k: int(v) for k, v in zip(CHAIN_REASONS, o[3:10])})

    def nuis_exact_result(self):
        """exact nz_test of the last step ``nuis_step_next`` reported as accepted = 3 (decided from the screened interval)"""
        nz = C.c_double()
        _ck(lib().ig_nuis_exact_result(self._h, C.byref(nz)))
        return nz.value

    def nuis_end(self):
        res = MoveResult()
        nz, z = C.c_double(), C.c_double()
        _ck(lib().ig_nuis_end(self._h, C.byref(res), C.byref(nz), C.byref(z), C.c_void_p(0)))
        return res, nz.value, z.value

    def nuis_accept(self):
        _ck(lib().ig_nuis_accept(self._h))

    # ---- speculative batches in steps (slots of a batch split over several GPUs)
    def batch_max_width(self, max_c=5):
        return int(lib().ig_batch_max_width(self._h, C.c_int32(int(max_c))))

    def batch_upload(self, frags, cands, max_w):
        f = np.ascontiguousarray(frags, np.int32)
        c = np.ascontiguousarray(cands, np.int32)
        assert c.ndim == 2 and c.shape[0] == f.size
        _ck(lib().ig_batch_upload(self._h, C.c_int32(f.size), _p(f), _p(c), C.c_int32(c.shape[1]), C.c_int32(int(max_w))))

    def batch_score(self, move0, w, slot_begin, slot_end):
        _ck(lib().ig_batch_score(self._h, C.c_int32(int(move0)), C.c_int32(int(w)), C.c_int32(int(slot_begin)), C.c_int32(int(slot_end))))

    def batch_records(self):
        """-> (device pointer, bytes per slot) of the slot-major score records: slot w lives at pointer + w * bytes"""
        p1 = C.c_void_p()
        b1 = C.c_int64()
        _ck(lib().ig_batch_records(self._h, C.byref(p1), C.byref(b1)))
        return p1.value, b1.value

    def batch_commit(self, move0, w):
        n = C.c_int32()
        _ck(lib().ig_batch_commit(self._h, C.c_int32(int(move0)), C.c_int32(int(w)), C.byref(n)))
        return n.value

    def batch_results(self, n_moves):
        res = np.zeros(int(n_moves), MOVE_RESULT_DTYPE)
        _ck(lib().ig_batch_results(self._h, C.c_int32(int(n_moves)), _p(res)))
        return res

    def debug_tile_trace(self):
        """one from-scratch pass -> (n, 4) int64: start, end (100 MHz clock), XCC_ID << 32 | HW_ID, contacts of every workgroup"""
        n = C.c_int64()
        _ck(lib().ig_debug_tile_trace(self._h, C.c_void_p(0), C.c_int64(0), C.byref(n)))
        out = np.zeros((n.value, 4), np.int64)
        _ck(lib().ig_debug_tile_trace(self._h, _p(out), C.c_int64(n.value), C.byref(n)))
        return out

    def debug_diff_trace(self, p_test8, mean_subfrag_kb):
        """one screened nuisance pass under p_test -> ((n, 8) int64 per-workgroup clocks, the pass's 8 output words)"""
        p = np.ascontiguousarray(p_test8, np.float32)
        n = C.c_int64()
        _ck(lib().ig_debug_diff_trace(self._h, _p(p), C.c_float(float(mean_subfrag_kb)), C.c_void_p(0), C.c_int64(0), C.byref(n), C.c_void_p(0)))
        out = np.zeros((n.value, 8), np.int64)
        sums = np.zeros(8, np.int64)
        _ck(lib().ig_debug_diff_trace(self._h, _p(p), C.c_float(float(mean_subfrag_kb)), _p(out), C.c_int64(n.value), C.byref(n), _p(sums)))
        return out, sums

    def debug_nuis_wait(self):
        s = C.c_double()
        _ck(lib().ig_debug_nuis_wait(self._h, C.byref(s)))
        return s.value

    def debug_nuis_screen_stats(self):
        """the screened nuisance pass (csrc/ig_kernels_nuis.cuh): steps screened, rejected from the interval alone, exact passes
        run behind a screened one, void, largest bound, largest |screened - exact| / bound, mean bound, undecided"""
        o = np.zeros(12, np.float64)
        _ck(lib().ig_debug_nuis_screen_stats(self._h, _p(o)))
        n = max(o[0] - o[3], 1.0)
        return dict(screened=int(o[0]), rejected_screened=int(o[1]), exact_passes=int(o[2]), void=int(o[3]), largest_bound=float(o[4]),
                    largest_used_fraction=float(o[5]), mean_bound=float(o[6] / n), undecided=int(o[7]),
                    void_why=dict(parameters=int(o[8]), contact=int(o[9]), sums=int(o[10]), no_record=int(o[11])))

    def debug_nuis_hist_stats(self):
        """the histogram tier of the screened nuisance pass: evaluations, steps rejected / accepted there, void, mean bound, largest
        |screened - exact| / bound, moves walked into the histogram, builds from scratch"""
        o = np.zeros(12, np.float64)
        _ck(lib().ig_debug_nuis_hist_stats(self._h, _p(o)))
        n = max(o[0] - o[3], 1.0)
        return dict(evaluated=int(o[0]), rejected=int(o[1]), accepted=int(o[2]), void=int(o[3]), mean_bound=float(o[4] / n),
                    largest_used_fraction=float(o[5]), walks=int(o[6]), builds=int(o[7]),
                    void_why=dict(parameters=int(o[8]), contact=int(o[9]), sums=int(o[10]), no_record=int(o[11])))

    def debug_zero_fallbacks(self):
        """moves that were scored again with every column exact because a contender's score came out as exactly 0.0"""
        n = C.c_int64()
        _ck(lib().ig_debug_zero_fallbacks(self._h, C.byref(n)))
        return int(n.value)

    def debug_pool_retries(self):
        """moves of the one-move path repeated with a larger slice pool (their lists did not fit)"""
        n = C.c_int64()
        _ck(lib().ig_debug_pool_retries(self._h, C.byref(n)))
        return int(n.value)

    def debug_nuis_hist_check(self):
        """words of the maintained histogram that differ from one built from scratch (-1: no histogram kept)"""
        n = C.c_int64()
        _ck(lib().ig_debug_nuis_hist_check(self._h, C.byref(n)))
        return int(n.value)

    def batch_stats(self):
        o = np.zeros(4, np.int64)
        _ck(lib().ig_batch_stats(self._h, _p(o)))
        return dict(batches=int(o[0]), committed_in_batch=int(o[1]), one_move_tails=int(o[2]), predicted_deltas=int(o[3]))

    def scratch_bytes(self):
        """bytes of the move buffers: (per-window arrays, slice pool, the rest)"""
        o = np.zeros(3, np.int64)
        _ck(lib().ig_scratch_bytes(self._h, _p(o)))
        return int(o[0]), int(o[1]), int(o[2])

    # ---- bookkeeping
    def renumber_contigs(self):
        n = C.c_int32()
        m = C.c_float()
        mx = C.c_int32()
        _ck(lib().ig_renumber_contigs(self._h, C.byref(n), C.byref(m), C.byref(mx)))
        return n.value, np.float32(m.value), mx.value

    def bomb(self, shuffle):
        s = np.ascontiguousarray(shuffle, np.int32)
        _ck(lib().ig_bomb(self._h, _p(s)))

    def genome_distance(self):
        d = C.c_double()
        _ck(lib().ig_genome_distance(self._h, C.byref(d)))
        return d.value

    def valid_insert(self):
        out = np.zeros(12, np.int32)
        _ck(lib().ig_get_valid_insert(self._h, _p(out)))
        return out

    def sync(self):
        _ck(lib().ig_sync(self._h))

    def set_stream(self, hip_stream_ptr):
        _ck(lib().ig_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    # ---- timing
    def reset_timers(self, enable=True):
        _ck(lib().ig_reset_timers(self._h, C.c_int(int(enable))))

    def set_timer_sampling(self, every):
        _ck(lib().ig_set_timer_sampling(self._h, C.c_int(int(every))))

    def kernel_time_ms(self, name):
        avg = C.c_double()
        n = C.c_int64()
        _ck(lib().ig_kernel_time_ms(self._h, name.encode(), C.byref(avg), C.byref(n)))
        return avg.value, n.value

    # ---- multi-GPU two-phase move
    def set_shard(self, rank, world):
        _ck(lib().ig_set_shard(self._h, C.c_int32(rank), C.c_int32(world)))

    def partials(self):
        return lib().ig_partials_device_ptr(self._h), lib().ig_partials_count(self._h)

    def step_begin(self, frag_a, cands):
        c = np.ascontiguousarray(cands, np.int32)
        _ck(lib().ig_step_begin(self._h, C.c_int32(int(frag_a)), _p(c), C.c_int32(c.size)))

    def step_finish(self, n_cands, want_scores=False):
        res = MoveResult()
        sc = np.zeros(n_cands * N_TMP_STRUCT, np.float64) if want_scores else None
        _ck(lib().ig_step_finish(self._h, C.byref(res), _p(sc)))
        return res, sc

    # ---- debug
    def debug_eval_terms(self, s, s_tot, ob):
        s = np.ascontiguousarray(s, np.float32)
        st = np.ascontiguousarray(s_tot, np.float32)
        ob = np.ascontiguousarray(ob, np.int32)
        n = s.size
        ex = np.zeros(n, np.float32)
        exc = np.zeros(n, np.float32)
        term = np.zeros(n, np.float64)
        q = np.zeros(n, np.int64)
        _ck(lib().ig_debug_eval_terms(self._h, _p(s), _p(st), _p(ob), C.c_int64(n), _p(ex), _p(exc), _p(term), _p(q)))
        return ex, exc, term, q

    def debug_candidate_state(self, cand, slot):
        out = np.zeros((17, self.N), np.int32)
        _ck(lib().ig_debug_candidate_state(self._h, C.c_int32(cand), C.c_int32(slot), _p(out)))
        return out

    def debug_last_sums(self, n_cands):
        T = N_TMP_STRUCT
        a = {k: np.zeros(n_cands * T, np.int64) for k in ("nz_hi", "nz_lo", "z_hi", "z_lo", "n_intra")}
        ext_hi = np.zeros(n_cands, np.int64)
        ext_lo = np.zeros(n_cands, np.int64)
        n_slice = np.zeros(n_cands, np.int64)
        n_uniq = np.zeros(n_cands, np.int32)
        uniq = np.zeros(n_cands * T, np.int32)
        _ck(lib().ig_debug_last_sums(self._h, _p(a["nz_hi"]), _p(a["nz_lo"]), _p(a["z_hi"]), _p(a["z_lo"]), _p(a["n_intra"]),
                                     _p(ext_hi), _p(ext_lo), _p(n_slice), _p(n_uniq), _p(uniq)))
        a.update(ext_hi=ext_hi, ext_lo=ext_lo, n_slice=n_slice, n_uniq=n_uniq, uniq=uniq.reshape(n_cands, T))
        return a

    def debug_transcendental_error(self):
        o = np.zeros(2, np.float64)
        _ck(lib().ig_debug_transcendental_error(self._h, _p(o)))
        return float(o[0]), float(o[1])

    def debug_screen_stats(self):
        """(largest used fraction of a bound, largest bound) under IG_SCREEN_VERIFY=1; (columns screened, columns scored exactly,
        terms screened, terms scored exactly)"""
        o = np.zeros(6, np.float64)
        _ck(lib().ig_debug_screen_stats(self._h, _p(o)))
        return float(o[0]), float(o[1]), int(o[2]), int(o[3]), int(o[4]), int(o[5])

    def debug_globals(self):
        sums = np.zeros(5, np.int64)
        ints = np.zeros(6, np.int32)
        _ck(lib().ig_debug_globals(self._h, _p(sums), _p(ints)))
        return sums, ints

    def debug_dbg(self, clear=False):
        """Glob.dbg (what raised a device-side consistency failure; in a tuning build a kernel's tick counters)"""
        o = np.zeros(8, np.int32)
        _ck(lib().ig_debug_dbg(self._h, _p(o), C.c_int32(int(bool(clear)))))
        return o

    def debug_tables(self):
        M = self.M
        d = np.zeros(M, np.float32)
        c = np.zeros(M, np.int32)
        st = np.zeros(M, np.float32)
        p = np.zeros(M, np.int32)
        ln = np.zeros(M, np.int32)
        _ck(lib().ig_debug_tables(self._h, _p(d), _p(c), _p(st), _p(p), _p(ln)))
        return d, c, st, p, ln
