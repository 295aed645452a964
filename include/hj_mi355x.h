/*
 * hj_mi355x.h -- C ABI of libhj_mi355x.so: the MI355X (gfx950) implementation of
 * LevelSetPy's Hamilton-Jacobi time-stepping hot path
 *
 *     odeCFLn -> termLaxFriedrichs -> upwindFirst{ENO2,ENO3,WENO5}
 *             -> artificialDissipationGLF -> addGhost{Extrapolate,Periodic}
 *
 * The reference (robotsorcerer/LevelSetPy) is pure Python: the "FFI" a
 * maintainer binds is ctypes (INTEGRATION.md shows the stub).  Every entry
 * point below names the reference callable it replaces (file:line relative to
 * the reference checkout).
 *
 * Conventions
 *  - plain C: pointers, sizes, scalars.  No torch / C++ types.
 *  - every `const void* / void*` array argument is a DEVICE pointer (HIP) to a
 *    C-order (last axis contiguous) array of the ctx's dtype, owned by the
 *    caller.  `double*` outputs documented as "host" are host pointers.
 *  - calls are stream-ordered on the ctx stream (hj_ctx_set_stream) and return
 *    without synchronising unless they hand back a host scalar.
 *  - return 0 on success, a negative HJ_E* code otherwise; the message is in
 *    hj_last_error() (thread-local).  The Python side raises ValueError, as the
 *    reference's error() does (Utilities/matlab_utils.py:134-137).
 *  - one ctx per device per thread; not re-entrant.
 */
#ifndef HJ_MI355X_H
#define HJ_MI355X_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#else        /* hipRTC has no system headers: the kernels of a user Hamiltonian only need the enums below */
typedef signed long long int64_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define HJ_MAX_DIM 4
#define HJ_STENCIL 3 /* ghost width of the widest scheme (ENO3aHelper.py:61) */

enum { HJ_OK = 0, HJ_EINVAL = -1, HJ_EHIP = -2, HJ_EUNSUPPORTED = -3, HJ_ESTATE = -4 };

/* grid.bdry[dim]: addGhostExtrapolate | addGhostPeriodic (Grids/create_grid.py:61-65) */
enum { HJ_BC_EXTRAPOLATE = 0, HJ_BC_PERIODIC = 1 };

/* schemeData.CoStateCalc / derivFunc */
enum {
    HJ_ENO2 = 0,            /* SpatialDerivative/upwind_first_eno2.py:12  */
    HJ_ENO3 = 1,            /* SpatialDerivative/upwind_first_eno3a.py:14 */
    HJ_WENO5 = 2,           /* intended O&F WENO5 (upwind_first_weno5a.py:13, ENO3bHelper.py:135-160) */
    HJ_WENO5_ASSHIPPED = 3, /* what upwind_first_weno5a.py computes as shipped: linear weights */
    /* Round 5, opt-in: ENO2 / ENO3 in the LEAN arithmetic (undivided differences, contracted FMAs, select-then-form) instead of
     * the reference's operation order.  Same scheme, same stencil choices except where two |D2| / |D3| moduli lie within rounding
     * of each other (masked comparison, margin < 1e-12); everything else agrees with the reference to 1e-11.  Accepted
     * by the substep entry points (hj_lf_term, hj_rk_substep, hj_rk_step, hj_rk_integrate, the slab steppers); the array-level
     * derivative / term entry points treat them as HJ_ENO2 / HJ_ENO3. */
    HJ_ENO2_FAST = 4,
    HJ_ENO3_FAST = 5
};

/* schemeData.hamFunc / partialFunc pairs with a native implementation */
enum {
    HJ_HAM_DUBINS_REL = 0,        /* DynamicalSystems/dubins_relative.py:63-111; params {v_e, v_p, w, w_e+w_p}; aux0=cos(vs[2]) aux1=sin(vs[2]) */
    HJ_HAM_DOUBLE_INTEGRATOR = 1, /* DynamicalSystems/double_integrator.py:49-89; params {u_bound} */
    HJ_HAM_DOUBLE_PENDULUM = 2,   /* build-defined 4-D stress case (BASELINE C5); params {u_max}; aux0..3 = sin th1, cos th1, sin th2, cos th2 */
    HJ_HAM_USER_BASE = 100        /* ids >= this: Hamiltonians registered at run time (hj_ham_register) */
};

enum { HJ_F64 = 0, HJ_F32 = 1 };

/* how one fused substep combines its result (ExplicitIntegration/Integration/ode_cfl_{1,2,3}.py) */
enum {
    HJ_STAGE_YDOT = 0,   /* out = ydot                         termLaxFriedrichs (term_lax_friedrich.py:124-128) */
    HJ_STAGE_EULER = 1,  /* out = y + dt*ydot                  ode_cfl_3.py:151,184 */
    HJ_STAGE_RK3_HALF = 2, /* out = 0.25*(3*y0 + (y + dt*ydot))  ode_cfl_3.py:184,193 */
    HJ_STAGE_RK3_FULL = 3, /* out = (1/3)*(y0 + 2*(y + dt*ydot)) ode_cfl_3.py:226,241 */
    HJ_STAGE_RK2_FULL = 4  /* out = 0.5*(y0 + (y + dt*ydot))     ode_cfl_2.py:184,201 */
};

/* hj_minmax_with ops: HJIPDE_solve post-step operators (ValueFuncs/hji_solver.py:566-599) */
enum { HJ_OP_MIN = 0, HJ_OP_MAX = 1, HJ_OP_MAX_NEG = 2 /* max(y, -other): obstacle mask :641-644 */ };

typedef struct hj_ctx hj_ctx;

/* ---- context = the reference's grid Bundle fields the path uses (Grids/process_grid.py:185-293) ----
 * N[ndim] grid.N, xmin[ndim] grid.min, dx[ndim] grid.dx, bc[ndim] HJ_BC_*, toward_zero[ndim]
 * (ghostData.towardZero, add_ghost_extrapolate.py:60-64; may be NULL = all 0). */
int hj_ctx_create(hj_ctx** out, int ndim, const int64_t* N, const double* xmin, const double* dx,
                  const int* bc, const int* toward_zero, int dtype, int device);
void hj_ctx_destroy(hj_ctx* ctx);

/* HIP stream (hipStream_t) all later calls are ordered on; NULL = the default stream. */
int hj_ctx_set_stream(hj_ctx* ctx, void* hip_stream);

/* Exact node coordinates grid.vs[dim] (host, N[dim] doubles; process_grid.py:204).  Optional:
 * default is xmin + i*dx.  Passing np.linspace values keeps coordinates bit-identical. */
int hj_ctx_set_coords(hj_ctx* ctx, int dim, const double* vs_host);

/* Hamiltonian-specific 1-D tables (host, n doubles), e.g. numpy's cos/sin of vs[2] for Dubins so
 * the trig values are bit-identical to the reference's cp.cos(grid.xs[2]) (dubins_relative.py:84-85).
 * Optional: default is libm on the coordinates. */
int hj_ctx_set_aux(hj_ctx* ctx, int slot, const double* table_host, int64_t n);

/* Slab decomposition along axis 0 (no reference counterpart).  When halo_lo / halo_hi
 * is nonzero, the HJ_STENCIL planes below / above every stencil-input array (`y` arguments) are
 * caller-provided ghost planes (received from the neighbour rank) instead of boundary-rule ghosts:
 * the `y` pointer still addresses the first interior plane, and the planes at negative offsets /
 * past N[0] must be readable. */
int hj_ctx_set_slab(hj_ctx* ctx, int halo_lo, int halo_hi);

/* ---- boundary padding: grid.bdry[dim](data, dim, width, bdryData)
 * addGhostExtrapolate (BoundaryCondition/add_ghost_extrapolate.py:16) / addGhostPeriodic
 * (add_ghost_periodic.py:12).  `out` has N[dim]+2*width entries along dim.  Bit-exact. */
int hj_ghost(hj_ctx* ctx, int dim, int width, const void* in, void* out);

/* ---- derivL, derivR = CoStateCalc(grid, data, dim)  (upwindFirst{ENO2,ENO3,WENO5})
 * minmax4 (host, nullable): {min L, max L, min R, max R} -- the four global reductions
 * artificialDissipationGLF takes per dim (artificial_diss_glf.py:80-88); reading them syncs. */
int hj_upwind(hj_ctx* ctx, int scheme, int dim, const void* phi, void* derivL, void* derivR,
              double* minmax4_host);

/* ---- termLaxFriedrichs with FOREIGN hamFunc / partialFunc callbacks (the reference's general case:
 * term_lax_friedrich.py:106-128 around artificial_diss_glf.py:75-109) in two device calls around the callbacks:
 *   hj_lf_split_begin   derivL[d], derivR[d] = CoStateCalc(grid, data, d) for every d (ndim launches back to
 *                       back) and the 4*ndim reductions {min L, max L, min R, max R} per dim with ONE host
 *                       synchronisation (minmax4n_host nullable: no synchronisation).
 *   ... the caller's hamFunc / partialFunc run on the device arrays ...
 *   hj_lf_split_end     diss = sum_d (0.5*(derivR_d - derivL_d))*alpha_d, out = -(ham - diss)  (ham == NULL: out =
 *                       diss, i.e. artificialDissipationGLF's own return value); alpha_d is the device array
 *                       alpha_arr[d] or, where that is NULL, the scalar alpha_scalar[d];
 *                       stepBound = 1/sum_d max(alpha_d)/dx_d with the max taken for the array-valued alphas only
 *                       (artificial_diss_glf.py:101-109).  Bit-identical to the reference's NumPy expressions.
 *   hj_rk_combine       the array expression of one odeCFLn stage for an arbitrary schemeFunc: mode 1
 *                       y + dt*z (ode_cfl_3.py:151), 2 0.25*(3*x0 + (y + dt*z)) (:184-193), 3
 *                       (1/3)*(x0 + 2*(y + dt*z)) (:226-241), 4 0.5*(x0 + (y + dt*z)) (ode_cfl_2.py:184-201). */
int hj_lf_split_begin(hj_ctx* ctx, int scheme, const void* y, void* const* derivL, void* const* derivR,
                      double* minmax4n_host);
int hj_lf_split_end(hj_ctx* ctx, const void* const* derivL, const void* const* derivR, const void* const* alpha_arr,
                    const double* alpha_scalar, const void* ham, void* out, double* step_bound_host,
                    double* alpha_max_host);
int hj_rk_combine(hj_ctx* ctx, int mode, double dt, const void* x0, const void* y, const void* z, void* out, int64_t n);

/* ---- ydot, stepBound = termLaxFriedrichs(t, y, schemeData) with a native hamFunc/partialFunc and
 * artificialDissipationGLF (term_lax_friedrich.py:8, artificial_diss_glf.py:7), one fused kernel.
 * restrict_sign: 0 none; +1 ydot=max(ydot,0); -1 ydot=min(ydot,0) (termRestrictUpdate,
 * term_restrict_update.py:99-102).  step_bound_host nullable (non-null syncs). */
int hj_lf_term(hj_ctx* ctx, int scheme, int ham_id, const double* ham_params, double t,
               int restrict_sign, const void* y, void* ydot, double* step_bound_host);

/* ---- one fused RK substep: out = stage(y0, y + dt*ydot(y)); see HJ_STAGE_*.  y0 may be NULL for
 * YDOT/EULER.  The per-dim max of alpha (CFL reduction) of this substep is left on the device in
 * `bound_slot` (0..HJ_BOUND_SLOTS-1) for hj_read_step_bound.  Planes [plane_begin, plane_end) of
 * axis 0 are updated (use 0, N[0] for all): the slab driver launches edge and interior ranges
 * separately to overlap the halo exchange. */
#define HJ_BOUND_SLOTS 64
int hj_rk_substep(hj_ctx* ctx, int scheme, int ham_id, const double* ham_params, double t,
                  int stage, double dt, int restrict_sign, const void* y, const void* y0, void* out,
                  int bound_slot, int64_t plane_begin, int64_t plane_end);

/* stepBound (artificial_diss_glf.py:107-109) of the substep that used `bound_slot`; synchronises. */
int hj_read_step_bound(hj_ctx* ctx, int bound_slot, double* step_bound_host, double* alpha_max_host);

/* ---- t, y = odeCFL{1,2,3}(termLaxFriedrichs|termRestrictUpdate, [t0, tf], y, options{singleStep:'on'})
 * with a native Hamiltonian whose alpha does not depend on the data (all HJ_HAM_* above): ONE step,
 * `order` fused substeps, no host synchronisation.  dt = min(factor_cfl*stepBound, tf-t0, max_step)
 * (ode_cfl_3.py:142).  work0/work1: caller scratch, same size as y (work1 unused for order<3,
 * may be NULL).  y_out must not alias y_in (stencil).  t_out/dt_out: host.
 * Host synchronisation: none in steady state.  EXCEPTION (grids of >= HJ_AUTOTUNE_MIN_MCELLS = 40 M cells, whole-grid
 * launches of a single domain, HJ_AUTOTUNE=1 = default): the first ncand * HJ_AUTOTUNE_PASSES (at most 9 x 6) launches per
 * (scheme, stage class, kernel configuration) take turns through candidate tile shapes and each ends in a
 * hipEventSynchronize on the ctx stream (hj_rk_substep / hj_rk_step / hj_rk_integrate alike); results do not depend on
 * the shape.  Launches issued while the stream is being CAPTURED never tune (they take the shape chosen so far).
 * HJ_AUTOTUNE=0 turns the rotation off. */
int hj_rk_step(hj_ctx* ctx, int order, int scheme, int ham_id, const double* ham_params, double t0,
               double tf, double factor_cfl, double max_step, int restrict_sign, const void* y_in,
               void* y_out, void* work0, void* work1, double* t_out, double* dt_out);

/* Two RK stages in ONE launch (hj_fused12.h; no reference counterpart -- the reference evaluates every stage
 * as its own chain of array expressions, ode_cfl_3.py:129-193):
 *     y1 = y + dt*L(y);   out = ca*y + cb*(y1 + dt*L(y1))        (ca, cb) = (3/4, 1/4) RK3, (1/2, 1/2) RK2
 * y1 is kept on chip (1R + 1W words per cell instead of 5).  Bitwise equal to hj_rk_substep(EULER) followed by
 * hj_rk_substep(RK3_HALF | RK2_FULL).  2-D / 3-D grids, every scheme but HJ_WENO5 (its epsilon is a global
 * reduction over y1), single domain (no slab halos), arrays below 4 GiB; HJ_EUNSUPPORTED otherwise.
 * OPT-IN: hj_rk_step takes this path only with HJ_FUSE12=1 in the environment of hj_ctx_create (default 0: on this chip the
 * fused pair of stages is 15 % slower than two launches at fp64 -- it trades 2.9x less traffic for recomputed stencils);
 * hj_rk_plan reports the number of kernel launches one hj_rk_step makes and whether stages 1+2 are fused. */
int hj_rk_stage12(hj_ctx* ctx, int scheme, int ham_id, const double* ham_params, double dt, double ca, double cb,
                  const void* y, void* out, int bound_slot);
int hj_rk_plan(hj_ctx* ctx, int order, int scheme, int ham_id, const double* ham_params, int restrict_sign,
               int* launches_host, int* stage_fused_host);

/* The whole odeCFLn loop of a time span in one call (ode_cfl_3.py:125-251 with singleStep off and no
 * postTimeStep / terminalEvent callbacks): steps from t0 until tf - t < 100*eps*|tf| (or max_steps > 0
 * steps; stop_tol >= 0 replaces the stopping test by HJIPDE_solve's `t < tf - stop_tol`,
 * hji_solver.py:536).  y_in is never written; results alternate between buf_a and buf_b, `work` is stage scratch
 * (all of y's size, all distinct).  *result_in = 0 (no step taken: y_in), 1 (buf_a) or 2 (buf_b).
 * No host synchronisation for Hamiltonians with a static stepBound: one call enqueues the span. */
int hj_rk_integrate(hj_ctx* ctx, int order, int scheme, int ham_id, const double* ham_params, double t0,
                    double tf, double factor_cfl, double max_step, int restrict_sign, const void* y_in,
                    void* buf_a, void* buf_b, void* work, int64_t max_steps, double stop_tol,
                    double* t_out, int64_t* steps_out, int* result_in);

/* Post-step operator of HJIPDE_solve fused into the LAST stage of every hj_rk_step / hj_rk_integrate step
 * (hji_solver.py:566-580): HJ_POST_MIN_PREV / HJ_POST_MAX_PREV = min / max of the new state with the
 * state the step started from ('minVOverTime' / 'maxVOverTime'; NaN propagates as in NumPy).  That
 * state is an operand of the last stage anyway, so the operator costs no extra pass over memory. */
enum { HJ_POST_NONE = 0, HJ_POST_MIN_PREV = 1, HJ_POST_MAX_PREV = 2 };
int hj_ctx_set_post_step(hj_ctx* ctx, int op);
/* ... and, after the step, up to two operators against caller arrays of the state's size (device pointers,
 * NULL = none; applied by separate elementwise launches inside hj_rk_step, so that a whole tau interval
 * still is one hj_rk_integrate call): op 1 = min, 2 = max, 3 = max with the NEGATED array.  'minVWithV0'/'maxVWithV0' and
 * 'min/maxVWithL' use the first (hji_solver.py:576-597), the obstacle mask the second (:641-644). */
int hj_ctx_set_post_arrays(hj_ctx* ctx, int op_a, const void* a, int op_b, const void* b);

/* stepBound of a native Hamiltonian on this grid (alpha is data-independent for all HJ_HAM_*);
 * computed once per (ham_id, params) and cached.  Synchronises on the first call.
 * alpha_max_host (nullable): the ndim per-dimension maxima of alpha (a slab-decomposed run
 * all-reduces them and forms stepBound = 1/sum_d max alpha_d / dx_d itself). */
int hj_static_step_bound(hj_ctx* ctx, int ham_id, const double* ham_params, double* step_bound_host,
                         double* alpha_max_host);

/* Which artificial dissipation's CFL bound hj_lf_term / hj_rk_step / hj_static_step_bound report.  For a
 * Hamiltonian whose alpha ignores the data the dissipation TERM of all three Lax-Friedrichs variants is
 * the same array, only the bound differs: global LF 1/sum_d max_x alpha_d/dx_d
 * (artificial_diss_glf.py:101-109), local variants 1/max_x sum_d alpha_d(x)/dx_d
 * (diss_local_laxfried.py:116-121, diss_localsq_laxfried.py:99-104, with the reduction the toolbox intends).
 * A run-time Hamiltonian whose alpha READS the costate range (HJ_HAM_RANGE) distinguishes the two local variants (round 5): HJ_DISS_LLF
 * evaluates alpha_i with the range of dimension i replaced by the node's own [min(p_i^-, p_i^+), max(p_i^-, p_i^+)] and the grid-wide
 * range in the other dimensions (diss_local_laxfried.py:84-111: a range pass + the fused substep per stage), HJ_DISS_LLLF with the node's
 * own range in every dimension (diss_localsq_laxfried.py:87-90: no range pass at all); the bound is reduced inside the substep kernel,
 * and hj_rk_step finds it before the first stage with a pass of its own (deltaT depends on it).  HJ_DISS_LOCAL = HJ_DISS_LLF. */
enum { HJ_DISS_GLF = 0, HJ_DISS_LOCAL = 1, HJ_DISS_LLF = 1, HJ_DISS_LLLF = 2 };
int hj_ctx_set_dissipation(hj_ctx* ctx, int kind);

/* max over the (unstripped) first-divided-difference table of D1^2, per dim: the 'maxOverGrid'
 * epsilon of true WENO5 (upwind_first_weno5a.py:69-70,153-156).  out_dev: ndim values of the ctx
 * dtype on the device.  hj_rk_substep/hj_lf_term with HJ_WENO5 run this themselves unless
 * hj_ctx_set_weno_eps_source was given a caller-reduced vector (slab decomposition: all-reduced). */
int hj_max_d1sq(hj_ctx* ctx, const void* y, void* out_dev);
int hj_ctx_set_weno_eps_source(hj_ctx* ctx, const void* max_d1sq_dev /* NULL = compute per call */);

/* ---- elementwise y = min/max(y, other)  (HJIPDE_solve compMethod, hji_solver.py:566-599,641-644) */
int hj_minmax_with(hj_ctx* ctx, int op, void* y, const void* other, int64_t n);

/* NaN guard of HJIPDE_solve (hji_solver.py:544-545): *has_nan_host = 1 if any NaN.  Synchronises. */
int hj_any_nan(hj_ctx* ctx, const void* y, int64_t n, int* has_nan_host);

/* ---- multi-GPU slab stepping in native code: RCCL halo exchange over xGMI on a second HIP stream,
 * overlapped with the interior stencil compute (no reference counterpart).
 *   hj_comm_unique_id   rank 0 creates the 128-byte ncclUniqueId; the caller broadcasts it.
 *   hj_comm_init        one communicator per ctx.  rccl_path: the librccl.so to dlopen (NULL =
 *                       "librccl.so.1"); lo_rank / hi_rank: neighbour ranks on axis 0 (-1 = physical
 *                       boundary).  Must be called by all ranks.
 *   hj_halo_exchange    fill the HJ_STENCIL pad planes of `buf` (pointer to the first interior plane)
 *                       from the neighbours' edge planes; stream-ordered on the ctx stream.
 *   hj_slab_rk_step     one odeCFL{1,2,3} step with a given dt on padded slab buffers (pointers to
 *                       the first interior plane).  Per substep: the edge plane ranges [0,3) and
 *                       [n-3,n) run on an edge stream, their exchange on a comm stream, the interior
 *                       planes on the ctx stream; the next substep waits for all three.  With
 *                       HJ_WENO5 the per-dimension max(D1^2) is all-reduced (ncclMax) first.
 *                       Result is left in y_out; cur is preserved.  The exchange of the last substep
 *                       is left in flight on return (it overlaps the next step's interior):
 *   hj_slab_join        makes the ctx stream wait for the outstanding edge/comm work; call it
 *                       before anything else reads the slab buffers. */
int hj_comm_unique_id(const char* rccl_path, void* uid128_host);
int hj_comm_init(hj_ctx* ctx, const char* rccl_path, int rank, int nranks, const void* uid128_host,
                 int lo_rank, int hi_rank);
int hj_comm_destroy(hj_ctx* ctx);
/* rank / size as the RCCL communicator itself reports them (ncclCommUserRank / ncclCommCount; the
 * caller-supplied values for hj_comm_init_external) and the neighbour ranks; any pointer may be NULL. */
int hj_comm_info(hj_ctx* ctx, int* rank, int* nranks, int* lo_rank, int* hi_rank);
int hj_halo_exchange(hj_ctx* ctx, void* buf);
int hj_slab_join(hj_ctx* ctx);
int hj_slab_rk_step(hj_ctx* ctx, int order, int scheme, int ham_id, const double* ham_params, double dt,
                    int restrict_sign, const void* cur, void* y_out, void* work0, void* work1);

/* ---- deep-halo slab stepping: ONE exchange per odeCFLn step instead of one per substep.
 * The slab buffers carry D = HJ_STENCIL*order pad planes on every side that has a neighbour.  Stage s
 * also computes HJ_STENCIL*(order-s) planes beyond the slab (redundantly with the neighbour), so
 * the next stage finds its stencil inputs locally; only the final result's D edge planes are exchanged,
 * overlapped with the last interior launch of this step and the first of the next.  Needs a dt known
 * in advance (data-independent alpha: every native Hamiltonian) and n_local >= 2*D.
 *   hj_ctx_set_axis0_pad   axis-0 tables for the pad planes: vs0_ext (and, for Hamiltonians whose
 *                          aux slots 0/1 are indexed by the axis-0 node, aux0_ext/aux1_ext; else NULL)
 *                          hold N[0]+2*pad values, entry `pad` being the slab's first plane.
 *   hj_comm_init_external  like hj_comm_init but WITHOUT a communicator: the caller moves the pad planes
 *                          itself (MPI, torch.distributed, a test harness) after hj_slab_join + sync.
 *   hj_halo_exchange_depth hj_halo_exchange for `depth` planes.
 *   hj_slab_rk_step_deep   one step; same buffers/semantics as hj_slab_rk_step (pads D deep). */
int hj_ctx_set_axis0_pad(hj_ctx* ctx, int pad, const double* vs0_ext, const double* aux0_ext, const double* aux1_ext);
int hj_comm_init_external(hj_ctx* ctx, int rank, int nranks, int lo_rank, int hi_rank);
int hj_halo_exchange_depth(hj_ctx* ctx, void* buf, int depth);
int hj_slab_rk_step_deep(hj_ctx* ctx, int order, int scheme, int ham_id, const double* ham_params, double dt,
                         int restrict_sign, const void* cur, void* y_out, void* work0, void* work1);

/* ---- the other schemeFuncs of the reference that share the upwind derivatives, ONE launch each
 * (hj_terms.h): derivatives of every dimension, Godunov's upwind choice, |grad phi|, ydot and the CFL maxima.
 * The shipped reference functions raise: the formulas are their docstrings' and the oracle's.
 *   hj_term_normal      termNormal      ExplicitIntegration/Term/term_normal.py:7 (:143-181)
 *                       ydot = -a |grad phi|; speed = array of the grid (or null: speed_scalar)
 *   hj_term_reinit      termReinit      ExplicitIntegration/Term/term_reinit.py:7 (:181-312)
 *                       ydot = -S(initial)(|grad phi| - 1), sub-cell fix of order 0 (smeared sign) or 1
 *   hj_term_convection  termConvection  ExplicitIntegration/Term/term_convection.py:7 (:154-180)
 *                       ydot = -V . grad phi; velocity[d] = array of the grid or null (velocity_scalar[d])
 * y, ydot (must differ) and the arrays are device pointers of the ctx dtype; *step_bound as the reference returns it
 * (inf when nothing moves).  One host synchronisation per call (the step bound).  fp64 2-D / 3-D grids of at least
 * HJ_TERM_TILED_FROM cells (environment, default 1 000 000; negative: never) run the tiled substep kernel with the term in
 * the Hamiltonian's place, everything else the one-thread-per-cell kernel: same cell function, same bits. */
int hj_term_normal(hj_ctx* ctx, int scheme, const void* y, const void* speed, double speed_scalar, void* ydot,
                   double* step_bound);
int hj_term_reinit(hj_ctx* ctx, int scheme, const void* y, const void* initial, int subcell_order, void* ydot,
                   double* step_bound);
int hj_term_convection(hj_ctx* ctx, int scheme, const void* y, const void* const* velocity, const double* velocity_scalar,
                       void* ydot, double* step_bound);

/* ---- a user's hamFunc / partialFunc pair as a fused kernel (round 4).  The reference takes ARBITRARY Python callables
 * (ExplicitIntegration/Term/term_lax_friedrich.py:111 hamFunc(t, data, derivC, schemeData);
 * Dissipation/artificial_diss_glf.py:98 partialFunc(t, data, derivMin, derivMax, schemeData, dim)).  `body` is the same
 * pair written once as a device expression: C++ statements that read x[d] (node coordinates), p[d] (costates,
 * = derivC), par[k] (the ham_params of a call, k < nparams <= 8) and assign  H  and  alpha[d]  for d = 0..ndim-1
 * (alpha must not depend on p -- true of every system the reference ships).  Optional `column_body` (ncol <= 8 values):
 * statements assigning col[k] from x[1..] and par, evaluated ONCE per grid column outside the march along axis 0 and
 * readable in `body` as col[k] -- where trigonometric functions of the in-plane coordinates belong (the built-in Dubins
 * kernel reads such values from tables; an expression that calls cos / sin per cell and plane runs 8 % slower).
 * The library wraps it in a Hamiltonian
 * type, compiles the fused substep kernel for it with hipRTC (gfx950; on first use of a scheme, 1-2 s) and returns an id
 * that every entry point taking ham_id accepts (fp64, 2-D / 3-D grids).  include_dir: directory of the library's kernel
 * headers (levelsetpy_amd/csrc); hiprtc_path: the libhiprtc.so to load (NULL: the loader's default).
 * Registering the same (name, ndim, nparams, body) again returns the same id.  A body that does not compile fails at
 * first use (or in hj_ham_compile_check) with the compiler's message in hj_last_error(). */
int hj_ham_register(const char* name, int ndim, int nparams, const char* body, const char* column_body, int ncol,
                    const char* include_dir, const char* hiprtc_path, int* ham_id);
int hj_ham_info(int ham_id, int* ndim, int* nparams, int* kernels_built);
/* Round 5: the same with flags.  HJ_HAM_RANGE: the expression also reads dmin[d] / dmax[d], the costate range the reference's
 * artificialDissipationGLF hands to partialFunc (Dissipation/artificial_diss_glf.py:80-99: derivMin[i] = min over the grid of derivL[i]
 * and derivR[i], derivMax likewise) -- alpha then depends on the data, and max(alpha) (:101-107) is reduced in the launch.  A substep of
 * such a Hamiltonian is two launches (range pass, then the fused substep); hj_rk_step reads the first stage's bound once per step
 * (ode_cfl_3.py:142) and keeps the later stages' bounds for the reference's CFL warning (hj_rk_last_bounds); hj_static_step_bound
 * fails with HJ_EUNSUPPORTED (the bound is a property of the data).  fp64 and fp32 grids, 2-D / 3-D / 4-D. */
#define HJ_HAM_RANGE 1
int hj_ham_register2(const char* name, int ndim, int nparams, const char* body, const char* column_body, int ncol, int flags,
                     const char* include_dir, const char* hiprtc_path, int* ham_id);
int hj_ham_flags(int ham_id, int* flags_host);
/* stepBound of the stages of the last hj_rk_step on this ctx with an HJ_HAM_RANGE Hamiltonian (sb_host[0..*n_host), n <= 3):
 * ode_cfl_3.py:173-175,215-217 warn when deltaT > min(1, 1.2 factorCFL) * stepBound at the later stages. */
int hj_rk_last_bounds(hj_ctx* ctx, double* sb_host /* 3 */, int* n_host);
/* The same without waiting (round 5): hj_rk_step returns while the last stage of a range-dependent step is still running and copies
 * the later stages' bounds to the host asynchronously; hj_rk_last_bounds waits for them, this call does not -- it reports the NEWEST
 * step whose bounds have arrived (usually the previous one: they are decoded at the next step's own synchronisation point) together
 * with that step's deltaT, once (*n_host = 0 when there is nothing new).  The Python integrators raise the reference's "substep
 * violated CFL" warning (ode_cfl_3.py:173-175, 215-217) from it one step late and drain with hj_rk_last_bounds at the end of a
 * multi-step call. */
int hj_rk_prev_bounds(hj_ctx* ctx, double* sb_host /* 3 */, int* n_host, double* dt_host);
/* Decomposed grids: the range of ONE slab is not the grid's.  hj_range_pass reduces derivL / derivR of the ctx's planes (pads read
 * where the slab has neighbours) into 2*HJ_MAX_DIM order-preserving 64-bit keys at keys_dev ([d] max, [HJ_MAX_DIM-independent ndim + d]
 * -min; an element-wise MAX over ranks of the keys is the reduction); hj_ctx_set_range_source makes later launches read the
 * range from such keys instead of running their own pass (NULL: back to per-launch passes).  A context that IS a slab with neighbours refuses
 * to launch such a Hamiltonian before a range source has been set (HJ_ESTATE: its own pass would give a rank-local range), and the native slab
 * steppers refuse it altogether (HJ_EUNSUPPORTED: dist.SlabIntegrator(dynamic=True) is the path).  No reference counterpart. */
int hj_range_pass(hj_ctx* ctx, int scheme, int ham_id, const double* ham_params, const void* y, void* keys_dev);
/* The local Lax-Friedrichs kinds (HJ_DISS_LLF / HJ_DISS_LLLF) of an HJ_HAM_RANGE Hamiltonian: alpha depends on every node's own costates,
 * so stepBound is a maximum over the stencil results.  hj_bound_pass runs the substep kernel over the ctx's planes storing nothing and
 * returns 1 / max_x sum_d alpha_d(x) / dx_d (diss_local_laxfried.py:120-128, diss_localsq_laxfried.py:99-107) -- of ONE slab when the
 * ctx is one: the grid's bound is the MIN over ranks (dist.SlabIntegrator(dynamic=True, diss=...)).  LLF reads the grid-wide range in
 * the other dimensions from hj_ctx_set_range_source when one is set, else from a range pass of its own.  Synchronises the stream. */
int hj_bound_pass(hj_ctx* ctx, int scheme, int ham_id, const double* ham_params, const void* y, double* step_bound_host);
int hj_ctx_set_range_source(hj_ctx* ctx, const void* keys_dev);
/* max over this ctx's nodes of alpha_d(x, range) for the range currently in force (the source set above, else the last pass):
 * amax_host[0..ndim); artificial_diss_glf.py:101-107 -- what the ranks all-reduce for deltaT.  One host synchronisation. */
int hj_range_alpha_max(hj_ctx* ctx, int ham_id, const double* ham_params, double* amax_host);
/* compile the substep kernel of `scheme` and the alpha-bound kernel WITHOUT launching (needs no GPU) */
int hj_ham_compile_check(int ham_id, int scheme);
/* Compiled code objects are kept on disk, keyed by the generated source, the kernel headers' text and the compile options
 * ($HJ_RTC_CACHE, else $XDG_CACHE_HOME/levelsetpy_amd, else $HOME/.cache/levelsetpy_amd; HJ_RTC_CACHE=0 turns it off): a
 * process that registers an expression seen before loads the kernel in milliseconds instead of compiling for 1-2 s.
 * Counters of this process: kernels compiled with hipRTC, kernels loaded from the cache.  No reference counterpart. */
int hj_ham_cache_stats(int* compiled, int* loaded_from_cache);

int hj_sync(hj_ctx* ctx);
const char* hj_last_error(void);
/* Name of the substep kernel the last hj_rk_substep / hj_rk_step / hj_lf_term on this ctx launched:
 * "fused_pair_kernel", "fused_substep_kernel", "fused12_kernel" or "direct_substep_kernel"; after hj_term_*:
 * "fused_substep_kernel" (tiled) or "term_kernel" (bench.py names the kernel its roofline is about; no reference
 * counterpart). */
const char* hj_last_kernel(hj_ctx* ctx);
/* LDS schedule of that launch: plane buffers in the ring (2 = double buffer) and how many planes ahead of its use the halo
 * ring of a plane is parked in LDS (0 = staged in the iteration that consumes it); tests assert the variant that ran. */
int hj_last_launch(hj_ctx* ctx, int* lds_nbuf_host, int* halo_ahead_host);
/* Tile extents of the last tiled launch on the plane axes (extents_host[0] = planes per chunk, [d] = cells on axis d, 0 beyond
 * the grid's dimension; all 0 after a direct launch): tests assert that the launch-time tuner rotates through tile shapes. */
int hj_last_tile(hj_ctx* ctx, int* extents_host /* HJ_MAX_DIM */);
/* Number of writes to the ctx's per-call state so far (hj_ctx_set_stream / _dissipation / _post_step / _post_arrays).  The Python
 * layer skips those calls when nothing changed; it compares this counter with the value it saw after its own writes, so that a
 * second user of the same (cached, shared) ctx cannot leave it with a stale post-step operator or CFL-bound kind.  (No reference
 * counterpart: the reference passes such state in schemeData on every call, term_restrict_update.py:50-80.) */
unsigned long long hj_ctx_state_generation(hj_ctx* ctx);
/* The launch plan of one substep over planes [p0, p1) of axis 0, made WITHOUT a device or a context: what a rank of a slab run will
 * launch (bench.py --gpus N --plan-only prints it for every rank before the node exists).  It runs the launch code of hj_rk_substep up to
 * the point where the kernel would be enqueued -- configuration choice by scheme / grid size, tile search, chunking against num_cus
 * compute units (0: 256, MI355X) -- with the occupancy taken from the kernel's launch bound instead of the runtime's query.
 * halo_lo / halo_hi: the grid is a slab with pad planes on that side (hj_ctx_set_slab).  Built-in Hamiltonians only.
 * out_host[12] = {threads per workgroup, workgroups, tiles per plane, chunks, planes per chunk, tile extents on axes 1..3 (0 beyond the
 * dimension), LDS bytes per workgroup, workgroups per CU, slab flag, 0}; kernel_name_host receives the kernel's name (as hj_last_kernel).
 * (The reference has no launch geometry: its term evaluates whole NumPy arrays, term_lax_friedrich.py:79-133.) */
int hj_plan_substep(int ndim, const int64_t* N_host, const int* bc_host, int dtype, int scheme, int ham, int stage,
                    int64_t p0, int64_t p1, int halo_lo, int halo_hi, int num_cus,
                    int64_t* out_host /* 12 */, char* kernel_name_host, int name_cap);
const char* hj_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HJ_MI355X_H */
