"""levelsetpy_amd -- MI355X (gfx950) implementation of LevelSetPy's Hamilton-Jacobi time-stepping
hot path, behind the reference's own names and callback protocol:

    odeCFL1/2/3, odeCFLset  ->  termLaxFriedrichs / termRestrictUpdate
        ->  upwindFirstENO2 / ENO3 / WENO5  +  artificialDissipationGLF  +  addGhost*

All arithmetic on the path runs in hand-written HIP kernels (csrc/, C ABI in
include/hj_mi355x.h, bound with ctypes).  There is no CPU fallback.
"""
import os as _os

# the slab stepper drives three HIP streams plus RCCL's: with the default 4 hardware queues two of
# them share a queue and serialise.  Only effective before HIP initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .utilities import *            # noqa: F401,F403
from .boundary import addGhostExtrapolate, addGhostPeriodic, addGhostAllDims   # noqa: F401
from .grids import createGrid, processGrid                                      # noqa: F401
from .init_conds import shapeCylinder, shapeSphere                              # noqa: F401
from .spatial import (upwindFirstENO2, upwindFirstENO3, upwindFirstENO3a, upwindFirstWENO5,   # noqa: F401
                      upwindFirstWENO5a, upwindFirstWENO5Intended, upwindFirstENO3aHelper,
                      set_weno5_mode, get_weno5_mode, set_eno_mode, get_eno_mode)
from .dissipation import (artificialDissipationGLF, artificialDissipationLLF,      # noqa: F401
                          artificialDissipationLLLF)
from .dynamics import DubinsVehicleRel, DoubleIntegrator, DoublePendulum4D      # noqa: F401
from .user_ham import register_native_hamiltonian, NativeRegistration, RegisteredSystem, kernel_cache_stats   # noqa: F401
from .trace_ham import trace_callbacks, TraceError                              # noqa: F401
from .term import termLaxFriedrichs, termRestrictUpdate, explain_plan           # noqa: F401
from .integration import (odeCFL1, odeCFL2, odeCFL3, odeCFLset, odeCFLget,      # noqa: F401
                          odeCFLmultipleSteps, odeCFLcallPostTimestep)
from .hji_solver import HJIPDE_solve                                            # noqa: F401
from .gradients import computeGradients                                         # noqa: F401
from .convection import termConvection                                          # noqa: F401
from .normal_reinit import termNormal, termReinit                               # noqa: F401
from .opt_traj import computeOptTraj, find_earliest_BRS_ind                     # noqa: F401

__version__ = "0.1.0"
