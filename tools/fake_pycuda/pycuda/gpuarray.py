import numpy as np


class GPUArray:
    def __init__(self, arr):
        self._np = np.ascontiguousarray(arr)

    @property
    def gpudata(self):
        return self

    def __int__(self):
        return int(self._np.ctypes.data)

    __index__ = __int__

    @property
    def dtype(self):
        return self._np.dtype

    @property
    def shape(self):
        return self._np.shape

    @property
    def nbytes(self):
        return self._np.nbytes

    def get(self, ary=None):
        if ary is None:
            return self._np.copy()
        ary[...] = self._np.reshape(ary.shape)
        return ary

    def fill(self, v):
        self._np.fill(v)
        return self

    def free(self):
        pass


def to_gpu(ary=None, **kw):
    return GPUArray(np.array(ary, copy=True))


def zeros(shape, dtype=np.float32, **kw):
    return GPUArray(np.zeros(shape, dtype=dtype))


def zeros_like(other):
    return GPUArray(np.zeros_like(other._np))


def max(a):  # noqa: A001 - mirrors pycuda.gpuarray.max
    return GPUArray(np.array(a._np.max()))
