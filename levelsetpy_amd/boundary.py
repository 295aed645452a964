"""Ghost-cell padding: addGhostExtrapolate / addGhostPeriodic / addGhostAllDims
(reference BoundaryCondition/add_ghost_{extrapolate,periodic,all}.py), on the GPU via hj_ghost.
Signature and error behaviour as the reference: dataOut = f(dataIn, dim, width=None, ghostData=None).
"""
import ctypes as C

import numpy as np

from . import _ffi
from .context import require_gpu, is_tensor
from .utilities import Bundle, error

__all__ = ["addGhostExtrapolate", "addGhostPeriodic", "addGhostAllDims"]


def _pad(dataIn, dim, width, bc, toward_zero):
    torch = require_gpu()
    lib = _ffi.lib()
    if not width:
        width = 1                                   # add_ghost_extrapolate.py:55-56
    shape = tuple(dataIn.shape)
    if dim < 0 or dim >= len(shape):
        error('Illegal dim parameter')
    if width < 0 or width > shape[dim]:
        error('Illegal width parameter')            # :58-59
    nd = len(shape)
    # pad along one axis: view the data as (outer, n, inner), a 3-D grid for the C ABI
    outer = int(np.prod(shape[:dim])) if dim > 0 else 1
    inner = int(np.prod(shape[dim + 1:])) if dim + 1 < nd else 1
    view, vdim = (outer, shape[dim], inner), 1
    # the reference always returns float64 (add_ghost_extrapolate.py:77, add_ghost_periodic.py:74), but
    # evaluates the slopes in the INPUT's precision: float32 data is extrapolated in float32 and then
    # widened, which is what the golden vector for float32 input records
    dev = torch.device("cuda", torch.cuda.current_device())
    f32 = (str(dataIn.dtype) in ("float32", "torch.float32"))
    work = torch.float32 if f32 else torch.float64
    if is_tensor(dataIn):
        src = dataIn.to(device=dev, dtype=work).contiguous()
    else:
        src = torch.from_numpy(np.ascontiguousarray(dataIn, dtype=np.float32 if f32 else np.float64)).to(dev)
    n = (C.c_int64 * len(view))(*view)
    ones = _ffi.darr([1.0] * len(view))
    zeros = _ffi.darr([0.0] * len(view))
    bcs = [_ffi.BC_EXTRAPOLATE] * len(view)
    bcs[vdim] = bc
    tz = [0] * len(view)
    tz[vdim] = 1 if toward_zero else 0
    ctx = C.c_void_p()
    _ffi.check(lib.hj_ctx_create(C.byref(ctx), len(view), n, zeros, ones,
                                 (C.c_int * len(view))(*bcs), (C.c_int * len(view))(*tz),
                                 _ffi.F32 if f32 else _ffi.F64, dev.index))
    try:
        _ffi.check(lib.hj_ctx_set_stream(ctx, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        oshape = list(shape)
        oshape[dim] += 2 * width
        out = torch.empty(oshape, dtype=work, device=dev)
        _ffi.check(lib.hj_ghost(ctx, vdim, int(width), C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr())))
        _ffi.check(lib.hj_sync(ctx))
    finally:
        lib.hj_ctx_destroy(ctx)
    out = out.to(torch.float64)
    return out if is_tensor(dataIn) else out.cpu().numpy()


def addGhostExtrapolate(dataIn, dim, width=None, ghostData=None):
    """add_ghost_extrapolate.py:16.  ghostData.towardZero flips the slope sign (:60-64)."""
    tz = bool(ghostData is not None and isinstance(ghostData, Bundle)
              and getattr(ghostData, "towardZero", False))
    return _pad(dataIn, dim, width, _ffi.BC_EXTRAPOLATE, tz)


def addGhostPeriodic(dataIn, dim, width=None, ghostData=None):
    """add_ghost_periodic.py:12."""
    return _pad(dataIn, dim, width, _ffi.BC_PERIODIC, False)


def addGhostAllDims(grid, dataIn, width):
    """add_ghost_all.py:4: apply grid.bdry[i] with the same width in every dimension."""
    dataOut = dataIn
    for i in range(grid.dim):
        dataOut = grid.bdry[i](dataOut, i, width, grid.bdryData[i])
    return dataOut
