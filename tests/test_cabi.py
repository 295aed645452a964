"""CPU-only: the C-ABI library loads and exports every symbol include/hj_mi355x.h declares, and
the host-side mirror of the reference interface behaves (no compute calls)."""
import ctypes as C
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


def test_runtime_hamiltonian_registration_and_compile_check():
    """hj_ham_register / hj_ham_compile_check (round 4): a user's H / alpha expression becomes a Hamiltonian id; hipRTC
    cross-compiles the fused kernel for gfx950 without a GPU; a wrong expression fails with the compiler's message, which
    points into the user's own text."""
    import levelsetpy_amd as L
    src = """
        H = p[0] * par[0] * cos(x[2]) + p[1] * par[0] * sin(x[2]) + par[1] * fabs(p[2]);
        alpha[0] = fabs(par[0] * cos(x[2]));  alpha[1] = fabs(par[0] * sin(x[2]));  alpha[2] = par[1];
    """
    reg = L.register_native_hamiltonian("cabi_dubins_abs", 3, src, nparams=2)
    assert reg.ham_id >= _ffi.HAM_USER_BASE
    assert L.register_native_hamiltonian("cabi_dubins_abs", 3, src, nparams=2).ham_id == reg.ham_id      # same text, same id
    assert L.register_native_hamiltonian("cabi_dubins_abs", 3, src + " ", nparams=2).ham_id != reg.ham_id
    nd, npar, built = C.c_int(), C.c_int(), C.c_int()
    _ffi.check(_ffi.lib().hj_ham_info(reg.ham_id, C.byref(nd), C.byref(npar), C.byref(built)))
    assert (nd.value, npar.value, built.value) == (3, 2, 0)
    reg.check("ENO3")
    reg.check("WENO5")
    bad = L.register_native_hamiltonian("cabi_bad", 2, "H = p[0] * no_such_symbol;\nalpha[0] = 1; alpha[1] = 1;", nparams=0)
    with pytest.raises(ValueError) as e:
        bad.check()
    assert "no_such_symbol" in str(e.value) and "cabi_bad:1" in str(e.value)
    with pytest.raises(ValueError):
        L.register_native_hamiltonian("cabi_5d", 5, "H = 0;", nparams=0)          # 2-D / 3-D / 4-D
    # round 5: 4-D grids, and an alpha that reads the costate range (dmin / dmax: artificial_diss_glf.py:80-99) -- detected from the text,
    # flagged in the registration, compiled with its range pass
    r4 = L.register_native_hamiltonian("cabi_4d", 4, "H = p[0] + x[1] * p[2]; alpha[0] = 1; alpha[1] = 0; alpha[2] = fabs(x[1]); alpha[3] = 0;")
    assert not r4.uses_range
    r4.check("WENO5_ASSHIPPED")
    rr = L.register_native_hamiltonian("cabi_burgers", 2, "H = 0.5 * (p[0] * p[0] + p[1] * p[1]);\n"
                                       "alpha[0] = fmax(fabs(dmin[0]), fabs(dmax[0])); alpha[1] = fmax(fabs(dmin[1]), fabs(dmax[1]));")
    assert rr.uses_range
    fl = C.c_int()
    _ffi.check(_ffi.lib().hj_ham_flags(rr.ham_id, C.byref(fl)))
    assert fl.value == _ffi.HAM_RANGE
    _ffi.check(_ffi.lib().hj_ham_flags(reg.ham_id, C.byref(fl)))
    assert fl.value == 0
    rr.check("ENO2")
    with pytest.raises(ValueError):
        _ffi.check(_ffi.lib().hj_ham_info(9999, None, None, None))
    with pytest.raises(ValueError):
        L.register_native_hamiltonian('quote"d', 2, "H = 0; alpha[0] = 1; alpha[1] = 1;")      # the name goes into #line directives


def test_runtime_hamiltonian_is_selected_by_callable_identity():
    import levelsetpy_amd as L
    from levelsetpy_amd.dynamics import native_of
    from levelsetpy_amd.term import native_plan
    reg = L.register_native_hamiltonian("cabi_di", 2, "H = -(p[0] * x[1]) + par[0] * fabs(p[1]); alpha[0] = fabs(x[1]); alpha[1] = par[0];", nparams=1)
    g = L.createGrid(-np.ones((2, 1)), np.ones((2, 1)), 17 * np.ones((2, 1), dtype=np.int64), None)
    s = reg(g, [1.5])
    sd = L.Bundle(dict(grid=g, hamFunc=s.hamiltonian, partialFunc=s.dissipation, dissFunc=L.artificialDissipationGLF, CoStateCalc=L.upwindFirstENO3))
    nat = native_of(sd.hamFunc, sd.partialFunc)
    assert nat[0] is s and nat[1] == reg.ham_id and nat[2] == [1.5]
    plan = native_plan(sd)
    assert plan is not None and plan[2] == reg.ham_id and plan[3] == [1.5] and native_plan(sd) is plan      # cached
    s.params[0] = 2.0                       # changed in place: the cached plan is re-validated, like system.native()
    assert native_plan(sd)[3] == [2.0]
    with pytest.raises(NotImplementedError):
        s.hamiltonian(0., None, None)       # no Python callback was given: fused only

    class Mine(object):                     # an EXISTING object of the caller's own class
        def __init__(self, grid):
            self.grid, self.u = grid, 0.5

        def hamiltonian(self, t, data, derivs, sd=None):
            return -(derivs[0] * self.grid.xs[1]) + self.u * np.abs(derivs[1])

        def dissipation(self, t, data, dmin, dmax, sd, dim):
            return np.abs(self.grid.xs[1]) if dim == 0 else self.u
    m = reg.attach(Mine(g), params=lambda o: [o.u])
    nat = native_of(m.hamiltonian, m.dissipation)
    assert nat[1] == reg.ham_id and nat[2] == [0.5]
    assert native_of(m.hamiltonian, Mine(g).dissipation) is None          # methods of two different objects
    with pytest.raises(ValueError):
        reg.attach(Mine(g), params=[1.0, 2.0])                            # wrong parameter count


def test_plan_substep_needs_no_device():
    """hj_plan_substep runs the launch code up to the enqueue without a GPU: kernel, tiles and chunks of C4 / C5 / a small grid."""
    from levelsetpy_amd import _ffi
    from levelsetpy_amd.dist import plan_substep
    sid = _ffi.SCHEME_IDS["WENO5_ASSHIPPED"]
    p = plan_substep([65, 513, 513], [0, 0, 1], "float64", sid, _ffi.HAM_DUBINS_REL, _ffi.STAGE_EULER, 0, 65, True, True)
    assert p["kernel"] == "fused_pair_kernel" and p["threads"] == 512
    assert p["tiles"] * p["chunks"] == p["workgroups"] and p["chunks"] * p["chunk_planes"] >= 65
    assert np.prod(p["tile"]) <= 2048 and len(p["tile"]) == 2
    # pad planes of the deep-halo stepper: planes beyond the slab are plannable
    q = plan_substep([65, 513, 513], [0, 0, 1], "float64", sid, _ffi.HAM_DUBINS_REL, _ffi.STAGE_EULER, -6, 71, True, True)
    assert q["chunks"] * q["chunk_planes"] >= 77
    p4 = plan_substep([17, 129, 129, 129], [1, 1, 1, 1], "float32", sid, _ffi.HAM_DOUBLE_PENDULUM, _ffi.STAGE_EULER, 3, 14, True, True)
    assert p4["kernel"] == "fused_flat4_kernel" and p4["tile"] == [3, 5, 129]      # full rows of the contiguous axis (hj_flat4v.h)
    small = plan_substep([51, 51, 51], [0, 0, 1], "float64", sid, _ffi.HAM_DUBINS_REL, _ffi.STAGE_EULER, 0, 51)
    assert small["kernel"] == "fused_substep_kernel" and small["workgroups"] > 0
    with pytest.raises(ValueError):       # a 3-D Hamiltonian on a 2-D grid
        plan_substep([65, 513], [0, 0], "float64", sid, _ffi.HAM_DUBINS_REL, _ffi.STAGE_EULER, 0, 65)


def test_bench_plan_only_prints_every_rank_without_a_gpu():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--plan-only"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["plan_only"] and plan["n_gpus"] == 8 and plan["planes_per_rank"] == [65] + [64] * 7
    assert [e["planes"] for e in plan["ranks"]][:2] == [[0, 65], [65, 129]]
    assert plan["ranks"][0]["lo"] is None and plan["ranks"][7]["hi"] is None and plan["ranks"][3]["lo"] == 2
    mid = plan["ranks"][3]
    assert mid["halo_bytes_sent_per_step"] == 2 * 9 * 513 * 513 * 8
    assert mid["launches_per_substep"]["edges"]["workgroups"] == 2 * mid["launches_per_substep"]["edges"]["tiles"]
    assert plan["predicted"]["ms_per_step_compute_self_ring"] > 0
    assert "rank 7" in r.stderr
