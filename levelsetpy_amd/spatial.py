"""Upwind first-derivative approximations behind the reference's CoStateCalc protocol
    derivL, derivR = upwindFirst*(grid, data, dim, generateAll=False)
(reference SpatialDerivative/upwind_first_{eno2,eno3,eno3a,weno5,weno5a}.py, ENO3aHelper.py),
computed by hj_upwind on the GPU.

WENO5 arithmetic.  As shipped the reference's upwindFirstWENO5 degenerates to the fixed-weight
5th-order stencil (list aliasing at upwind_first_weno5a.py:97).  `upwindFirstWENO5`
reproduces the reference's results ('asshipped', the drop-in default); the intended nonlinear
Osher-Fedkiw scheme is `upwindFirstWENO5Intended` or `set_weno5_mode('weno5')`.
"""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from .context import device_grid, array_dtype_name, is_tensor
from .utilities import Bundle, error

__all__ = ["upwindFirstENO2", "upwindFirstENO3", "upwindFirstENO3a", "upwindFirstWENO5",
           "upwindFirstWENO5a", "upwindFirstWENO5Intended", "upwindFirstENO3aHelper",
           "set_weno5_mode", "get_weno5_mode", "set_eno_mode", "get_eno_mode"]

_WENO5_MODE = "asshipped"
_ENO_MODE = "exact"


def set_eno_mode(mode):
    """'exact' (default): upwindFirstENO2 / ENO3 in fused substeps evaluate the reference's array expressions operation by operation --
    states, t and stepBound equal the reference's bit for bit.  'fast' (opt-in, round 5): the same schemes in a lean arithmetic
    (undivided differences, contracted FMAs, the chosen candidate formed after the selection): 1e-11 from the reference everywhere
    except in cells where two stencil-selector moduli lie within rounding of each other, which may take the other (equally valid)
    stencil (masked comparison).  Only the fused substep path changes; the array-level derivative functions do not."""
    global _ENO_MODE
    if mode not in ("exact", "fast"):
        error("ENO mode must be 'exact' or 'fast'")
    _ENO_MODE = mode


def get_eno_mode():
    return _ENO_MODE


def set_weno5_mode(mode):
    global _WENO5_MODE
    if mode not in ("asshipped", "weno5"):
        error("WENO5 mode must be 'asshipped' or 'weno5'")
    _WENO5_MODE = mode


def get_weno5_mode():
    return _WENO5_MODE


def scheme_id_of(fn):
    """C-ABI scheme id of one of this module's derivative functions (None if foreign)."""
    name = getattr(fn, "_hj_scheme", None)
    if name is None:
        return None
    if name == "WENO5_DEFAULT":
        name = "WENO5" if _WENO5_MODE == "weno5" else "WENO5_ASSHIPPED"
    if _ENO_MODE == "fast" and name in ("ENO2", "ENO3"):
        name += "_FAST"
    return _ffi.SCHEME_IDS[name]


def _upwind(scheme_name, grid, data, dim):
    if dim < 0 or dim >= grid.dim:
        error('Illegal dim parameter')
    dg = device_grid(grid, array_dtype_name(data))
    if tuple(data.shape) != dg.shape:
        error('data parameter does not agree in array size with grid')
    dg.bind_stream()
    phi = dg.to_device(data)
    dL, dR = dg.empty(), dg.empty()
    mm = (C.c_double * 4)()
    _ffi.check(dg.lib.hj_upwind(dg.ctx, _ffi.SCHEME_IDS[scheme_name], int(dim), dg.ptr(phi),
                                dg.ptr(dL), dg.ptr(dR), mm))
    # the four reductions artificialDissipationGLF needs (artificial_diss_glf.py:80-88) came out of
    # the same kernel: they ride on the returned tensor itself (tagged with its partner and both
    # in-place version counters), so they die with it and an edited or foreign array never matches
    if is_tensor(data):
        dL._hj_minmax = (weakref.ref(dR), dL._version, dR._version, min(mm[0], mm[2]), max(mm[1], mm[3]))
    return dg.like(dL, data), dg.like(dR, data)


def upwind_all_dims(fn, grid, data):
    """derivL[d], derivR[d] for every d with ONE native call and one host synchronisation (hj_lf_split_begin)
    when `fn` is one of this module's derivative functions and `data` is a device tensor; else None."""
    sid = scheme_id_of(fn)
    if sid is None or not is_tensor(data):
        return None
    dg = device_grid(grid, array_dtype_name(data))
    if tuple(data.shape) != dg.shape:
        error('data parameter does not agree in array size with grid')
    dg.bind_stream()
    phi = dg.to_device(data)
    dL = [dg.empty() for _ in range(dg.dim)]
    dR = [dg.empty() for _ in range(dg.dim)]
    vp = C.c_void_p * dg.dim
    mm = (C.c_double * (4 * dg.dim))()
    _ffi.check(dg.lib.hj_lf_split_begin(dg.ctx, sid, dg.ptr(phi), vp(*[t.data_ptr() for t in dL]),
                                        vp(*[t.data_ptr() for t in dR]), mm))
    for d in range(dg.dim):
        dL[d]._hj_minmax = (weakref.ref(dR[d]), dL[d]._version, dR[d]._version,
                            min(mm[4 * d], mm[4 * d + 2]), max(mm[4 * d + 1], mm[4 * d + 3]))
    return dL, dR


def cached_minmax(dL, dR):
    """(min, max) over derivL and derivR if this exact, unmodified pair came out of hj_upwind, else None."""
    tag = getattr(dL, "_hj_minmax", None) if is_tensor(dL) else None
    if tag is None or tag[0]() is not dR or tag[1] != dL._version or tag[2] != dR._version:
        return None
    return tag[3], tag[4]


def _candidates(grid, data, dim, order, approx4=False, want_dd=False):
    """generateAll=True: the ENO candidates (upwind_first_eno3a.py:62-80, eno2.py:119-126), from
    the padded array's divided differences (compatibility path: array ops on the device).  With want_dd the
    divided-difference tables come back too, stripped as the reference returns them (ENO3aHelper.py:99-112:
    D1 with N+1 entries along dim, D2 with N+2, D3 with N+3)."""
    dg = device_grid(grid, "float64")
    g = grid.bdry[dim](dg.to_device(data), dim, order, grid.bdryData[dim])
    dx = dg.dx[dim]
    N = data.shape[dim]

    def diff(a):
        return a.narrow(dim, 1, a.shape[dim] - 1) - a.narrow(dim, 0, a.shape[dim] - 1)

    def take(a, lo, hi):
        return a.narrow(dim, lo, hi - lo)

    D1 = (1 / dx) * diff(g)
    D2 = (0.5 / dx) * diff(D1)
    dd = None
    if order == 2:
        D1s = take(D1, 1, N + 2)
        dL = [take(D1s, 0, N) + dx * take(D2, 0, N), take(D1s, 0, N) + dx * take(D2, 1, N + 1)]
        dR = [take(D1s, 1, N + 1) - dx * take(D2, 1, N + 1), take(D1s, 1, N + 1) - dx * take(D2, 2, N + 2)]
    else:
        D3 = (1 / (3 * dx)) * diff(D2)
        D1s, D2s = take(D1, 2, N + 3), take(D2, 1, N + 3)
        l, r = take(D1s, 0, N), take(D1s, 1, N + 1)
        dL = [l + dx * take(D2s, 0, N) + 2 * dx * dx * take(D3, 0, N),
              l + dx * take(D2s, 0, N) + 2 * dx * dx * take(D3, 1, N + 1),
              l + dx * take(D2s, 1, N + 1) - dx * dx * take(D3, 2, N + 2)]
        dR = [r - dx * take(D2s, 1, N + 1) - dx * dx * take(D3, 1, N + 1),
              r - dx * take(D2s, 1, N + 1) - dx * dx * take(D3, 2, N + 2),
              r - dx * take(D2s, 2, N + 2) + 2 * dx * dx * take(D3, 3, N + 3)]
        if approx4:
            # the middle approximation reached right-then-left through the difference tree
            # (ENO3aHelper.py:139-141,148-149,168-170,178-179): equals element [1] up to rounding
            dL.append(l + dx * take(D2s, 1, N + 1) - dx * dx * take(D3, 1, N + 1))
            dR.append(r - dx * take(D2s, 2, N + 2) + 2 * dx * dx * take(D3, 2, N + 2))
        if want_dd:
            dd = Bundle(dict(D1=dg.like(D1s.contiguous(), data), D2=dg.like(D2s.contiguous(), data),
                             D3=dg.like(D3.contiguous(), data)))
    out = [dg.like(a.contiguous(), data) for a in dL], [dg.like(a.contiguous(), data) for a in dR]
    return out + (dd,) if want_dd else out


def upwindFirstENO2(grid, data, dim, generateAll=False):
    """upwind_first_eno2.py:12."""
    if generateAll:
        return _candidates(grid, data, dim, 2)
    return _upwind("ENO2", grid, data, dim)


def upwindFirstENO3(grid, data, dim, generateAll=False):
    """upwind_first_eno3.py:13 -> upwind_first_eno3a.py:14."""
    if generateAll:
        return _candidates(grid, data, dim, 3)
    return _upwind("ENO3", grid, data, dim)


upwindFirstENO3a = upwindFirstENO3


def upwindFirstENO3aHelper(grid, data, dim, approx4=False, stripDD=False):
    """ENO3aHelper.py:11: the three (approx4: four) candidates per side and the divided-difference Bundle
    DD with fields D1, D2, D3.  The reference returns the STRIPPED tables whatever stripDD says (the
    unstripped Bundle built at :93 is overwritten at :112); so does this."""
    dL, dR, DD = _candidates(grid, data, dim, 3, approx4=bool(approx4), want_dd=True)
    return dL, dR, DD


def upwindFirstWENO5(grid, data, dim, generateAll=False):
    """upwind_first_weno5.py:11 -> upwind_first_weno5a.py:13 (arithmetic per set_weno5_mode)."""
    if generateAll:
        return _candidates(grid, data, dim, 3)
    return _upwind("WENO5" if _WENO5_MODE == "weno5" else "WENO5_ASSHIPPED", grid, data, dim)


upwindFirstWENO5a = upwindFirstWENO5


def upwindFirstWENO5Intended(grid, data, dim, generateAll=False):
    """The Osher-Fedkiw WENO5 the reference documents (ENO3bHelper.py:135-160) with the
    'maxOverGrid' epsilon (upwind_first_weno5a.py:69-70,153-156)."""
    if generateAll:
        return _candidates(grid, data, dim, 3)
    return _upwind("WENO5", grid, data, dim)


upwindFirstENO2._hj_scheme = "ENO2"
upwindFirstENO3._hj_scheme = "ENO3"
upwindFirstWENO5._hj_scheme = "WENO5_DEFAULT"
upwindFirstWENO5Intended._hj_scheme = "WENO5"
