"""A numpy stand-in for the handful of TensorFlow 1.x functions the reference's pure-Python helpers use
(dirt/matrices.py, dirt/lighting.py, dirt/projection.py) -- TEST INFRASTRUCTURE.

There is no TensorFlow in this image.  tests/golden/make_helpers_golden.py puts this directory first on sys.path and
imports the reference's three helper modules FROM WHERE THEY LIE (/root/reference/dirt/*.py, by file path, so that the
package's op loader is not touched); their code then runs eagerly over numpy float32 arrays and its outputs become the
committed vectors tests/golden/helpers_ref.npz, against which the torch counterparts in dirt_amd/ are tested
(tests/test_helpers_ref.py).  Only what those three files call exists here, with TF's semantics for it (float32
stays float32, python lists mixed into arithmetic take the tensor's dtype, duplicate indices of scatter_nd /
SparseTensor add up).  Summation order inside reductions is numpy's, not TensorFlow's: comparisons are to 1e-6.
"""
import contextlib

import numpy as np

newaxis = None
__version__ = '2.0.0'   # the reference's scripts take their eager (`.numpy()`) branch
float32, float64, int32, int64 = np.dtype('float32'), np.dtype('float64'), np.dtype('int32'), np.dtype('int64')
uint8 = np.dtype('uint8')


class _Dim(int):
    @property
    def value(self):
        return int(self)


class _Shape(tuple):
    """TensorShape: tuple of ints with .ndims / .dims / .as_list() / slicing."""

    @property
    def ndims(self):
        return len(self)

    @property
    def dims(self):
        return [_Dim(d) for d in self]

    def as_list(self):
        return list(self)

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return _Shape(r) if isinstance(i, slice) else _Dim(r)


class Tensor:
    """An eager tensor: a numpy array behind the few attributes and operators the reference's helpers use."""
    __array_priority__ = 1000

    def __init__(self, a):
        self.a = np.asarray(a)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    dtype = property(lambda self: self.a.dtype)
    shape = property(lambda self: _Shape(self.a.shape))

    def get_shape(self):
        return _Shape(self.a.shape)

    def set_shape(self, _):
        pass

    def __len__(self):
        return len(self.a)

    def __iter__(self):
        return (Tensor(x) for x in self.a)

    def __getitem__(self, i):
        def plain(k):
            return _raw(k) if isinstance(k, Tensor) else k
        return Tensor(self.a[tuple(plain(k) for k in i) if isinstance(i, tuple) else plain(i)])

    def _other(self, o):
        if isinstance(o, Tensor):
            return o.a
        if isinstance(o, (list, tuple)):
            return np.asarray(o, dtype=self.a.dtype)   # a python list takes the tensor's dtype, as in TensorFlow
        if isinstance(o, np.ndarray):
            return o
        return self.a.dtype.type(o) if isinstance(o, (int, float)) and self.a.dtype.kind == 'f' else o

    def __add__(self, o): return Tensor(self.a + self._other(o))
    def __radd__(self, o): return Tensor(self._other(o) + self.a)
    def __sub__(self, o): return Tensor(self.a - self._other(o))
    def __rsub__(self, o): return Tensor(self._other(o) - self.a)
    def __mul__(self, o): return Tensor(self.a * self._other(o))
    def __rmul__(self, o): return Tensor(self._other(o) * self.a)
    def __truediv__(self, o): return Tensor(self.a / self._other(o))
    def __rtruediv__(self, o): return Tensor(self._other(o) / self.a)
    def __floordiv__(self, o): return Tensor(self.a // self._other(o))
    def __mod__(self, o): return Tensor(np.mod(self.a, self._other(o)))   # tf.floormod
    def __neg__(self): return Tensor(-self.a)
    def __int__(self): return int(self.a)
    def __index__(self): return int(self.a)
    def __float__(self): return float(self.a)

    def numpy(self):
        return self.a

    def eval(self, feed_dict=None):   # noqa: A003  (TensorFlow 1.x: everything is already evaluated here)
        return self.a


def _wrap(a):
    return a if isinstance(a, Tensor) else Tensor(a)


def _raw(a):
    return a.a if isinstance(a, Tensor) else a


def convert_to_tensor(value, dtype=None, name=None):
    if isinstance(value, Tensor):
        a = value.a
    elif isinstance(value, np.ndarray):
        a = value
    else:
        def unwrap(v):
            if isinstance(v, Tensor):
                return v.a
            if isinstance(v, (list, tuple)):
                return [unwrap(x) for x in v]
            return v
        a = np.array(unwrap(value))
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        elif a.dtype == np.int64:
            a = a.astype(np.int32)
    if dtype is not None:
        a = a.astype(dtype)
    return _wrap(a)


constant = convert_to_tensor


def _ints(shape):
    return [int(s) for s in np.asarray(_raw(convert_to_tensor(shape))).reshape(-1)]


def shape(t):
    return _wrap(np.array(np.asarray(_raw(t)).shape, dtype=np.int32))


def reshape(t, new_shape):
    return _wrap(np.reshape(np.asarray(_raw(t)), _ints(new_shape)))


def concat(values, axis):
    return _wrap(np.concatenate([np.atleast_1d(_raw(convert_to_tensor(v))) for v in values], axis=axis))


def stack(values, axis=0):
    return _wrap(np.stack([_raw(convert_to_tensor(v)) for v in values], axis=axis))


def matmul(a, b):
    return _wrap(np.matmul(_raw(convert_to_tensor(a)), _raw(convert_to_tensor(b))))


def zeros_like(t):
    return _wrap(np.zeros_like(np.asarray(_raw(t))))


def ones_like(t):
    return _wrap(np.ones_like(np.asarray(_raw(t))))


def zeros(shape_, dtype=float32):
    return _wrap(np.zeros(_ints(shape_) if not isinstance(shape_, int) else [shape_], dtype))


def eye(n, m=None, dtype=float32):
    return _wrap(np.eye(n, m, dtype=dtype))


def transpose(t, perm=None):
    return _wrap(np.transpose(np.asarray(_raw(t)), perm))


def tile(t, multiples):
    return _wrap(np.tile(np.asarray(_raw(t)), _ints(multiples)))


def norm(t, axis=None, keep_dims=False, keepdims=False):
    a = np.asarray(_raw(t))
    return _wrap(np.sqrt(np.sum(a * a, axis=axis, keepdims=keep_dims or keepdims)).astype(a.dtype))


def broadcast_to(t, shape_):
    return _wrap(np.broadcast_to(_raw(convert_to_tensor(t)), _ints(shape_)).copy())


def cast(t, dtype):
    return _wrap(np.asarray(_raw(t)).astype(dtype))


def range(*args, dtype=int32):   # noqa: A001  (the name TensorFlow uses)
    return _wrap(np.arange(*[int(a) for a in args], dtype=dtype))


def maximum(a, b):
    return _wrap(np.maximum(_raw(convert_to_tensor(a)), _raw(convert_to_tensor(b))))


def abs(t):   # noqa: A001
    return _wrap(np.abs(np.asarray(_raw(t))))


def reduce_sum(t, axis=None, keep_dims=False, keepdims=False):
    a = np.asarray(_raw(t))
    return _wrap(np.sum(a, axis=axis, keepdims=keep_dims or keepdims).astype(a.dtype))


def reduce_prod(t, axis=None):
    a = np.asarray(_raw(t))
    return _wrap(np.prod(a, axis=axis).astype(a.dtype))


def map_fn(fn, elems, dtype=None):
    return stack([fn(e) for e in elems])


def gather(params, indices):
    return _wrap(np.asarray(_raw(params))[np.asarray(_raw(indices))])


def squeeze(t, axis=None):
    return _wrap(np.squeeze(np.asarray(_raw(t)), axis=axis))


def expand_dims(t, axis):
    return _wrap(np.expand_dims(np.asarray(_raw(t)), axis))


def sin(t):
    return _wrap(np.sin(np.asarray(_raw(t))))


def cos(t):
    return _wrap(np.cos(np.asarray(_raw(t))))


def pow(a, b):   # noqa: A001
    return _wrap(np.power(_raw(convert_to_tensor(a)), _raw(convert_to_tensor(b))))


def cross(a, b):
    return _wrap(np.cross(np.asarray(_raw(a)), np.asarray(_raw(b))).astype(np.asarray(_raw(a)).dtype))


def scatter_nd(indices, updates, shape):   # noqa: A002
    out = np.zeros(_ints(shape), dtype=np.asarray(_raw(updates)).dtype)
    idx = np.asarray(_raw(indices))
    np.add.at(out, tuple(idx[:, k] for k in np.arange(idx.shape[1])), np.asarray(_raw(updates)))
    return _wrap(out)


def ones(shape_, dtype=float32):
    return _wrap(np.ones(_ints(shape_) if not isinstance(shape_, int) else [shape_], dtype))


def meshgrid(*args):
    return [_wrap(a) for a in np.meshgrid(*[np.asarray(_raw(x)) for x in args])]


def less_equal(a, b):
    return _wrap(np.asarray(_raw(convert_to_tensor(a))) <= np.asarray(_raw(convert_to_tensor(b))).astype(np.asarray(_raw(a)).dtype))


def logical_and(a, b):
    return _wrap(np.logical_and(np.asarray(_raw(a)), np.asarray(_raw(b))))


def floor(t):
    return _wrap(np.floor(np.asarray(_raw(t))))


def clip_by_value(t, lo, hi):
    a = np.asarray(_raw(t))
    return _wrap(np.clip(a, a.dtype.type(lo), a.dtype.type(hi)))


def unstack(t, axis=0):
    return [_wrap(x) for x in np.moveaxis(np.asarray(_raw(t)), axis, 0)]


def gather_nd(params, indices):
    """Index tuples in the last axis of `indices`.  Out-of-range tuples give ZEROS: what TensorFlow's GPU kernel does
    (its CPU kernel raises); samples/textured.py reads row Ht / column Wt for samples inside the last texel."""
    p, idx = np.asarray(_raw(params)), np.asarray(_raw(indices))
    k = idx.shape[-1]
    ok = np.ones(idx.shape[:-1], bool)
    for d in np.arange(k):
        ok &= (idx[..., d] >= 0) & (idx[..., d] < p.shape[d])
    safe = np.where(ok[..., None], idx, 0)
    out = p[tuple(safe[..., d] for d in np.arange(k))]
    return _wrap(np.where(ok.reshape(ok.shape + (1,) * (out.ndim - ok.ndim)), out, np.zeros((), p.dtype)))


class SparseTensor:
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = np.asarray(_raw(indices)), np.asarray(_raw(values)), _ints(dense_shape)


def sparse_reduce_sum(sp, axis=None):
    dense = np.zeros(sp.dense_shape, dtype=sp.values.dtype)
    np.add.at(dense, tuple(sp.indices[:, k] for k in np.arange(sp.indices.shape[1])), sp.values)
    return _wrap(dense.sum(axis=axis).astype(sp.values.dtype))


def matrix_inverse(t):
    a = np.asarray(_raw(t))
    return _wrap(np.linalg.inv(a.astype(np.float64)).astype(a.dtype))


class linalg:   # noqa: N801
    @staticmethod
    def l2_normalize(t, axis=None, epsilon=1e-12):
        a = np.asarray(_raw(convert_to_tensor(t)))
        return _wrap((a / np.sqrt(np.maximum(np.sum(a * a, axis=axis, keepdims=True), a.dtype.type(epsilon)))).astype(a.dtype))

    @staticmethod
    def diag(t):
        a = np.asarray(_raw(t))
        out = np.zeros(a.shape + (a.shape[-1],), dtype=a.dtype)
        i = np.arange(a.shape[-1])
        out[..., i, i] = a
        return _wrap(out)


class version:   # noqa: N801
    VERSION = '1.15.0'


@contextlib.contextmanager
def device(_):
    yield


class Session:
    """tf.Session of the 1.x scripts: nothing to hold, every tensor is already a value."""

    def __init__(self, *args, **kw):
        pass

    @contextlib.contextmanager
    def as_default(self):
        yield self

    def run(self, fetches, feed_dict=None):
        return [np.asarray(_raw(f)) for f in fetches] if isinstance(fetches, (list, tuple)) else np.asarray(_raw(fetches))


# ---- what the reference's sample scripts do around their rendering: files, JPEG, session options -----------------------

written_files = []   # (file name, contents) of every tf.write_file: nothing is written to disk


class _Runnable:
    def run(self, *args, **kw):
        pass


def write_file(filename, contents):
    written_files.append((str(filename), np.asarray(_raw(contents))))
    return _Runnable()


def read_file(filename):
    return str(filename)


class image:   # noqa: N801
    @staticmethod
    def encode_jpeg(t):
        return t          # (kept as the uint8 image: the comparison wants the pixels, not a codec)

    @staticmethod
    def decode_jpeg(path):
        from PIL import Image
        return _wrap(np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8))


def ConfigProto(**kw):   # noqa: N802
    return None


def GPUOptions(**kw):   # noqa: N802
    return None


# ---- what dirt/rasterise_ops.py needs on top of the helpers: the op library and (forward-only) custom_gradient ----------

_op_library = None      # set by oracle/ref.py before the reference's module is imported: an object with .rasterise / .rasterise_grad
_gradients_hook = None  # optional: tf.gradients(ys, xs, grad_ys) for the shader of rasterise_deferred (the caller's autodiff)


def load_op_library(path):
    if _op_library is None:
        raise RuntimeError('no op library bound (oracle/ref.py binds the host-compiled reference kernels)')
    return _op_library


def executing_eagerly():
    return False


def gradients(ys, xs, grad_ys=None):
    if _gradients_hook is None:
        raise NotImplementedError('tf.gradients: no autodiff in the numpy stand-in; bind tensorflow._gradients_hook')
    return _gradients_hook(ys, xs, grad_ys)


class _WithGradient:
    """What tf.custom_gradient makes of `f`: calling it gives the forward value; the gradient closure f returned is kept
    on the value as `.dirt_grad_fn` so that a test can invoke the reference's backward composition directly."""

    def __init__(self, f):
        self.f = f

    def __call__(self, *args, **kw):
        value, grad_fn = self.f(*args, **kw)
        value = _wrap(value)
        value.dirt_grad_fn = grad_fn
        return value


def custom_gradient(f):
    return _WithGradient(f)
