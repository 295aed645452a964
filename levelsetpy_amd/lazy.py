"""HostView: what the fused path hands back to a NumPy caller -- an ndarray-compatible handle on a result that
still lives in HBM.

The reference's drivers are NumPy in / NumPy out and feed every result straight back into the next call
(ValueFuncs/hji_solver.py:542, Notes/rcbrt.ipynb cell 4):

    t, y, schemeData = odeCFL3(termRestrictUpdate, [t, t_plot], y, options, schemeData)

A literal drop-in pays PCIe both ways per call (2.5 ms per step at 201^3 against 0.12 ms of GPU work).  A HostView
keeps the device tensor the kernels wrote and copies it to (page-locked) host memory only when somebody LOOKS at the
values: `np.asarray(y)`, indexing, any NumPy function or ufunc, any ndarray attribute.  Passed back into
odeCFLn / termLaxFriedrichs / termRestrictUpdate it is consumed on the device, so the loop above touches PCIe once
per plot, not twice per step.

What it promises: same shape / dtype / values as the ndarray the reference would have returned (`np.asarray(y)` is
exact and cached); results of arithmetic and NumPy functions are plain ndarrays; inputs are never mutated (kernels
only read their inputs).  Where it differs from an ndarray: `isinstance(y, np.ndarray)` is False; `reshape` /
`flatten` / `ravel` / `squeeze` / `copy` return independent HostViews (not memory-sharing views); the cached host copy
is read-only while the device tensor is attached (write through the view itself, `y[i] = v`, which detaches it from
the device first).  `HJ_LAZY_NUMPY=0` turns the handles off: plain ndarrays, a D2H copy per call, as in rounds 1-3.

`HJ_LAZY_NUMPY=ndarray` (set_lazy("ndarray"), round 5): results are DeviceArray -- a GENUINE np.ndarray subclass (isinstance, np.save,
pickle, the buffer protocol, C extensions all see an ndarray, as with the reference's return values, ode_cfl_3.py:241-272) that
still remembers the device tensor it was copied from, so passing it back in costs no upload.  It pays the download per call: an
ndarray's memory is handed out by C code without any Python hook (np.asarray(y) of a subclass is a base-class view made in C, and
so are memoryview(y) and PyArray_DATA in an extension), so the values have to BE there when the object is returned -- a lazily
filled ndarray would hand uninitialised memory to exactly the callers that make the subclass worth having.  Half of the PCIe
traffic of the eager mode, none of HostView's type caveats; HostView stays the default because it is the one that keeps the
reference's NumPy loop at the GPU's pace.
"""
import os

import numpy as np

def _mode(v):
    if isinstance(v, str):
        v = v.strip().lower()
        return "ndarray" if v == "ndarray" else (False if v in ("0", "", "off", "false") else True)
    return bool(v)


LAZY = _mode(os.environ.get("HJ_LAZY_NUMPY", "1"))        # True: HostView; "ndarray": DeviceArray; False: plain ndarrays


def set_lazy(on):
    """What NumPy callers get back from the fused path: True (default unless HJ_LAZY_NUMPY says otherwise) HostView handles,
    "ndarray" DeviceArray -- a real np.ndarray subclass, downloaded at once, consumed on the device when passed back --,
    False plain ndarrays."""
    global LAZY
    LAZY = _mode(on)


def _host_copy(t):
    """device tensor -> ndarray in page-locked host memory when it is large (the D2H copy then runs at the DMA rate and
    an array handed back later is uploaded at the DMA rate too); context.DeviceGrid.like's policy."""
    from . import context
    import torch
    t = t.detach()
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and nbytes >= (1 << 20) and context._PIN_RESULTS and nbytes <= context._PIN_MAX_BYTES:
        try:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t)
            return host.numpy()
        except RuntimeError:
            pass
    return t.cpu().numpy()


def _unwrap(x):
    if isinstance(x, HostView):
        return x.__array__()
    if isinstance(x, (list, tuple)):
        return type(x)(_unwrap(v) for v in x)
    if isinstance(x, dict):
        return {k: _unwrap(v) for k, v in x.items()}
    return x


class HostView(np.lib.mixins.NDArrayOperatorsMixin):
    """ndarray-compatible, lazily materialised view of a device tensor (module docstring)."""

    __slots__ = ("_t", "_h")
    __array_priority__ = 1000.0

    def __init__(self, tensor, host=None):
        object.__setattr__(self, "_t", tensor)      # device tensor (None once detached)
        object.__setattr__(self, "_h", host)        # cached host ndarray (None until somebody looks)

    # ------------------------------------------------------------------ what the package itself asks for
    def device_tensor(self):
        """The tensor the kernels wrote, or None if this view was written to from the host."""
        return self._t

    # ------------------------------------------------------------------ cheap metadata (no copy)
    @property
    def shape(self):
        return tuple(self._t.shape) if self._t is not None else self._h.shape

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def dtype(self):
        if self._h is not None:
            return self._h.dtype
        return np.dtype(str(self._t.dtype).replace("torch.", ""))

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def nbytes(self):
        return self.size * self.itemsize

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    # ------------------------------------------------------------------ materialisation
    def __array__(self, dtype=None, copy=None):
        h = self._h
        if h is None:
            h = _host_copy(self._t)
            h.flags.writeable = False       # the device copy is still what the next call consumes: no silent divergence
            object.__setattr__(self, "_h", h)
        if dtype is not None and np.dtype(dtype) != h.dtype:
            return h.astype(dtype)
        if copy:
            return h.copy()
        return h

    def _detach(self):
        """Before a write from the host: a private writable host copy, and the device tensor is let go."""
        h = np.array(self.__array__(), copy=True)
        object.__setattr__(self, "_h", h)
        object.__setattr__(self, "_t", None)
        return h

    # ------------------------------------------------------------------ NumPy protocols: results are plain ndarrays
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if "out" in kwargs:
            outs = kwargs["out"]
            kwargs["out"] = tuple(o._detach() if isinstance(o, HostView) else o for o in outs)
        return getattr(ufunc, method)(*_unwrap(inputs), **_unwrap(kwargs))

    def __array_function__(self, func, types, args, kwargs):
        # metadata questions are answered without looking at the values (np.ndim / np.shape / np.size dispatch here too)
        if func in _META and len(args) == 1 and not kwargs and isinstance(args[0], HostView):
            return _META[func](args[0])
        return func(*_unwrap(args), **_unwrap(kwargs))

    # ------------------------------------------------------------------ shape changes stay on the device
    def _lazy(self, fn_t, fn_h):
        if self._t is not None:
            return HostView(fn_t(self._t))
        return fn_h(self._h)

    def reshape(self, *shape, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if kw.get("order", "C") not in ("C", "A") or self._t is None:
            return self.__array__().reshape(*shape, **kw)
        return HostView(self._t.reshape(tuple(int(v) for v in shape)))

    def flatten(self, order="C"):
        if order not in ("C", "A"):
            return self.__array__().flatten(order)
        return self._lazy(lambda t: t.reshape(-1), lambda h: h.flatten())

    def ravel(self, order="C"):
        if order not in ("C", "A"):
            return self.__array__().ravel(order)
        return self._lazy(lambda t: t.reshape(-1), lambda h: h.ravel())

    def squeeze(self, axis=None):
        if axis is not None:
            return self.__array__().squeeze(axis)
        return self._lazy(lambda t: t.squeeze(), lambda h: h.squeeze())

    def copy(self, order="C"):
        return self._lazy(lambda t: t, lambda h: h.copy())     # the device tensor is never written: sharing it IS a copy

    def __copy__(self):
        return HostView(self._t, self._h)

    def __deepcopy__(self, memo):
        return self._lazy(lambda t: t, lambda h: h.copy())

    # ------------------------------------------------------------------ element access
    def __getitem__(self, idx):
        return self.__array__()[idx]

    def __setitem__(self, idx, value):
        self._detach()[idx] = _unwrap(value)

    def __iter__(self):
        return iter(self.__array__())

    def __float__(self):
        return float(self.__array__())

    def __int__(self):
        return int(self.__array__())

    def __bool__(self):
        return bool(self.__array__())

    def __repr__(self):
        where = "device" if self._t is not None else "host"
        return "HostView(%s, shape=%s, dtype=%s)" % (where, self.shape, self.dtype)

    def __getattr__(self, name):
        # every other ndarray attribute / method (sum, min, T, astype, tolist, ...): look, then delegate
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.__array__(), name)

    def __reduce__(self):
        return (np.array, (self.__array__(),))      # pickles as the ndarray it stands for


class DeviceArray(np.ndarray):
    """A real ndarray (page-locked host memory, values present) that remembers the device tensor it is a copy of (module
    docstring).  While the tensor is attached the memory is read-only -- a write through some view would silently diverge from
    what the next call consumes --; writing through the array itself (`y[i] = v`, `out=y`) detaches it first and makes it
    writable.  Views, slices and copies are plain detached arrays of this type."""

    def __new__(cls, host, tensor=None, cell=None):
        obj = np.asarray(host).view(cls)
        obj._hj_t = tensor
        # ALIASES (reshape / ravel / squeeze of an attached array) share the host memory, so they share ONE attachment cell:
        # a write through any of them detaches all of them (`z = y.reshape(..); z[0, 0] = v` used to leave y
        # attached to a tensor that no longer matched the memory both look at)
        obj._hj_cell = (cell if cell is not None else [True]) if tensor is not None else None
        if tensor is not None:
            obj.flags.writeable = False
        return obj

    def __array_finalize__(self, obj):
        self._hj_t = None           # only the object the package returned (and its shape aliases) stands for the tensor
        self._hj_cell = None

    def device_tensor(self):
        """The tensor this array is a copy of, or None once it -- or an alias of it -- was written to (or for any view / copy of it)."""
        t = getattr(self, "_hj_t", None)
        if t is None:
            return None
        cell = getattr(self, "_hj_cell", None)
        if cell is None or not cell[0] or self.flags.writeable:      # an alias wrote, or somebody forced the flag: the host copy may have changed
            if cell is not None:
                cell[0] = False
            self._hj_t = t = None
        return t

    def _detach(self):
        cell = getattr(self, "_hj_cell", None)
        if cell is None:             # a plain view / copy: nothing to detach (a view of attached memory stays read-only)
            return
        cell[0] = False              # every alias of this memory is detached with it
        self._hj_t = None
        self.flags.writeable = True

    def __setitem__(self, idx, value):
        self._detach()
        np.ndarray.__setitem__(self, idx, value)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        for o in kwargs.get("out", ()) or ():
            if isinstance(o, DeviceArray):
                o._detach()
        strip = lambda x: x.view(np.ndarray) if isinstance(x, DeviceArray) else x      # noqa: E731 -- results are plain ndarrays
        if "out" in kwargs:
            kwargs["out"] = tuple(strip(o) for o in kwargs["out"])
        return getattr(ufunc, method)(*[strip(x) for x in inputs], **kwargs)

    def __reduce__(self):
        return (np.array, (self.view(np.ndarray),))       # pickles as the plain ndarray it is

    # C-order shape changes of the returned object keep the tensor (the reference's drivers flatten / reshape between calls:
    # hji_solver.py:542); both sides are views of the same values
    def _reshaped(self, h, fn_t):
        t = self.device_tensor()
        if t is None or not isinstance(h, np.ndarray) or h.base is None:
            return h
        return DeviceArray(h, fn_t(t), self._hj_cell)

    def reshape(self, *shape, **kw):
        h = np.ndarray.reshape(self, *shape, **kw)
        if kw.get("order", "C") not in ("C", "A"):
            return h
        return self._reshaped(h, lambda t: t.reshape(h.shape))

    def ravel(self, order="C"):
        h = np.ndarray.ravel(self, order)
        return h if order not in ("C", "A") else self._reshaped(h, lambda t: t.reshape(-1))

    def squeeze(self, axis=None):
        h = np.ndarray.squeeze(self, axis)
        return self._reshaped(h, lambda t: t.reshape(h.shape))


def device_array(t):
    """device tensor -> DeviceArray (the download happens here)."""
    return DeviceArray(_host_copy(t), t.detach())


_META = {np.ndim: lambda v: v.ndim, np.shape: lambda v: v.shape, np.size: lambda v: v.size}


def is_lazy(x):
    return isinstance(x, HostView)


def attached_tensor(x):
    """The device tensor behind a result of this package (HostView or DeviceArray), else None."""
    if isinstance(x, (HostView, DeviceArray)):
        return x.device_tensor()
    return None
