"""A minimal functional stand-in for h5py, used ONLY by tools/gen_golden_pyramid.py to drive the reference's
pyramid builder/loader in a container without h5py.  Files are pickles of nested dicts of numpy arrays.
Covers what pyramid_sparse.py touches: File(path, mode), .attrs, create_group, create_dataset(name, shape, dtype),
dataset[...] read/write, group[name], close()."""
import os
import pickle

import numpy as np


class _Dataset:
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, k):
        return self._a[k]

    def __setitem__(self, k, v):
        self._a[k] = v

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    @property
    def shape(self):
        return self._a.shape


class _Group:
    def __init__(self, store):
        self._s = store

    def create_group(self, name):
        self._s[name] = {}
        return _Group(self._s[name])

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        if data is not None:
            arr = np.array(data)
        else:
            arr = np.zeros(shape, dtype=np.dtype("int32") if dtype == "i" else np.dtype(dtype))
        self._s[name] = arr
        return _Dataset(arr)

    def __getitem__(self, name):
        v = self._s[name]
        return _Group(v) if isinstance(v, dict) else _Dataset(v)

    def __contains__(self, name):
        return name in self._s

    def keys(self):
        return self._s.keys()


_OPEN = {}  # realpath -> blob: like HDF5, every handle of a file in one process sees the same objects


class File(_Group):
    def __init__(self, path, mode="r"):
        self._path, self._mode = path, mode
        key = os.path.realpath(path)
        if key in _OPEN:
            blob = _OPEN[key]
        elif os.path.exists(path):
            with open(path, "rb") as f:
                blob = pickle.load(f)
        else:
            if mode == "r":
                raise OSError("no such file: %s" % path)
            blob = {"attrs": {}, "root": {}}
        _OPEN[key] = blob
        self.attrs = blob["attrs"]
        super().__init__(blob["root"])

    def close(self):
        if self._mode != "r":
            with open(self._path, "wb") as f:
                pickle.dump({"attrs": self.attrs, "root": self._s}, f)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
