"""CPU-only: the C-ABI library loads and exports every symbol include/hj_mi355x.h declares, and
the host-side mirror of the reference interface behaves (no compute calls)."""
import os
import re

import numpy as np
import pytest

import levelsetpy_amd as L
from levelsetpy_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "hj_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hj_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _ffi.SIGNATURES, "ctypes signature missing for " + s
    assert set(_ffi.SIGNATURES) == set(syms)
    assert b"gfx950" in lib.hj_version()


def test_header_enums_match_python():
    txt = open(os.path.join(ROOT, "include", "hj_mi355x.h")).read()
    for name, val in [("HJ_ENO2", _ffi.ENO2), ("HJ_ENO3", _ffi.ENO3), ("HJ_WENO5", _ffi.WENO5),
                      ("HJ_WENO5_ASSHIPPED", _ffi.WENO5_ASSHIPPED), ("HJ_STAGE_RK3_FULL", _ffi.STAGE_RK3_FULL),
                      ("HJ_STAGE_RK2_FULL", _ffi.STAGE_RK2_FULL), ("HJ_OP_MAX_NEG", _ffi.OP_MAX_NEG),
                      ("HJ_HAM_DOUBLE_PENDULUM", _ffi.HAM_DOUBLE_PENDULUM), ("HJ_BC_PERIODIC", _ffi.BC_PERIODIC)]:
        m = re.search(r"\b%s\s*=\s*(\d+)" % name, txt)
        assert m and int(m.group(1)) == val, name


def test_grid_bundle_matches_reference_fields(golden):
    G = golden("deriv.npz")
    n = G["g3_data"].shape
    g = L.createGrid(G["g3_min"].reshape(-1, 1), G["g3_max"].reshape(-1, 1),
                     np.array(n, dtype=np.int64).reshape(-1, 1), 2)
    np.testing.assert_array_equal(np.asarray(g.dx).ravel(), G["g3_dx"])
    assert g.shape == n and g.dim == 3
    assert g.bdry[2] is L.addGhostPeriodic and g.bdry[0] is L.addGhostExtrapolate
    assert g.xs[0].shape == n and g.vs[1].shape == (n[1], 1)
    # pdDims=0 is read as "no periodic axis", as in the reference (create_grid.py:34)
    g0 = L.createGrid(np.zeros((2, 1)), np.ones((2, 1)), 5 * np.ones((2, 1), dtype=np.int64), 0)
    assert g0.bdry[0] is L.addGhostExtrapolate


def test_shapes_match_golden(golden):
    G = golden("ode.npz")
    g = L.createGrid(G["dub_min"].reshape(-1, 1), G["dub_max"].reshape(-1, 1),
                     G["dub_N"].reshape(-1, 1).astype(np.int64), 2)
    np.testing.assert_array_equal(L.shapeCylinder(g, 2, np.zeros((3, 1)), .5), G["dub_data"])
    g2 = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), 32 * np.ones((2, 1), dtype=np.int64), None)
    np.testing.assert_array_equal(L.shapeSphere(g2, np.zeros((2, 1)), .25), G["di_data"])


def test_odecflset_defaults_and_errors():
    o = L.odeCFLset(L.Bundle(dict(factorCFL=.8, singleStep='on')))
    assert o.factorCFL == .8 and o.singleStep == 'on' and o.maxStep == L.realmax and o.stats == 'off'
    assert L.odeCFLset(L.Bundle(dict(realmax=0.1))).maxStep == 0.1       # ode_cfl_set.py:96
    with pytest.raises(ValueError):
        L.odeCFLset(L.Bundle(dict(factorCFL=-1)))
    with pytest.raises(ValueError):
        L.odeCFLset(L.Bundle(dict(postTimeStep=3)))
    with pytest.raises(ValueError):
        L.odeCFLset()


def test_native_detection_is_strict():
    from levelsetpy_amd.term import native_plan
    g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), 8 * np.ones((2, 1), dtype=np.int64), None)
    s = L.DoubleIntegrator(g, 1)
    sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation,
                       dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
    assert native_plan(sd) is not None
    sd2 = L.Bundle(dict(grid=g, hamFunc=lambda *a: 0, partialFunc=s.dissipation,
                        dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
    assert native_plan(sd2) is None
    other = L.DoubleIntegrator(g, 2)
    sd3 = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=other.dissipation,
                        dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
    assert native_plan(sd3) is None


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), 8 * np.ones((2, 1), dtype=np.int64), None)
    with pytest.raises(RuntimeError):
        L.upwindFirstENO3(g, np.zeros((8, 8)), 0)
    with pytest.raises(RuntimeError):
        L.addGhostPeriodic(np.zeros((4, 4)), 0, 1)


def _check_wide_store_hazard():
    import importlib.util, os
    from levelsetpy_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library or llvm-objdump not present")
    spec = importlib.util.spec_from_file_location(
        "check_store_hazard", os.path.join(os.path.dirname(__file__), "..", "tools", "check_store_hazard.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    total, bad = mod.main(_ffi.LIB_PATH)
    assert total > 0 and not bad, bad[:5]


def test_wide_store_hazard_rule_holds_in_built_library():
    """Disassembles the gfx950 code objects of the built library: no VGPR holding the data of a >8-byte store may be
    written within two wait states of the store (hj_fusedv.h).  Static check, no GPU needed; the
    Makefile runs the same check after every link, and so does the GPU suite on the library it actually loads."""
    _check_wide_store_hazard()


@pytest.mark.gpu
def test_wide_store_hazard_rule_holds_in_loaded_library():
    _check_wide_store_hazard()
