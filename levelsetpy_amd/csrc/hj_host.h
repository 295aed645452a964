// Host-side declarations shared by the translation units of libhj_mi355x.so: the ctx, the tiling of
// the fused kernel, and the per-(dtype, Hamiltonian) launch entry that hj_inst.hip instantiates.
// The library is built from hj_api.hip (C ABI, split-path kernels) plus one object per (dtype, Hamiltonian)
// of hj_inst.hip (the fused / direct substep kernels), so that the expensive kernel instantiations compile
// in parallel (make -j).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>   // types only: the library is dlopen'ed (hj_comm_*), so libhj loads without RCCL

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/hj_mi355x.h"
#include "hj_device.h"
#include "hj_split.h"

namespace hjh {

int fail(int code, const char* fmt, ...);     // records the message for hj_last_error(), returns code

#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return hjh::fail(HJ_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                             \
    } while (0)

constexpr int RING_SLOTS = 2048;  // bound-key ring (each entry: HJ_MAX_DIM keys)
constexpr int RANGE_RING = 1024;  // range-key ring
// one entry: [0, 2*HJ_MAX_DIM) the range keys of a pass; [8, 8+HJ_MAX_DIM] the max(alpha) keys of the bound kernel that follows it in
// hj_rk_step (HJ_MAX_DIM per-dimension maxima + the local-LF sum); [13] that kernel's workgroup counter.  All zeroed in bulk.
constexpr int ALPHA_BLOCKS_MAX = 1024;
constexpr int RANGE_ENTRY = 16, RANGE_ALPHA_AT = 8, RANGE_DONE_AT = 13;

struct Tiling {
    int E[HJ_MAX_DIM], ntile[HJ_MAX_DIM];
    int ntiles, chunk, nchunks, nchunks1, nblocks, bpx, lpitch;
    size_t lds_bytes;
    double score;
    bool ok;
};

struct KernelCfg { int NT, R, KH; };

// Launch-time choice among tile shapes (round 3).  The static score of make_tiling (halo ratio, partial cache lines) cannot see
// how the workgroup count of a shape fills the last round of resident workgroups or how its row length suits the memory system:
// at 513^3 the 15x130 tile beats the best-scored 27x74 by 6 %, at 481^3 and 551^3 the best-scored shape wins by 9 %
// (profiles/r03_tile_shape_sweep.txt).  The results do not depend on the tiling (bitwise: every cell is a pure function of its
// inputs), so the first launches of a launch shape simply take turns through the candidates, each timed with a pair of events,
// and the fastest is kept.
struct TuneState {
    std::vector<Tiling> cand;
    std::vector<float> best_ms;
    int trial = 0, chosen = -1;
};

}  // namespace hjh

struct hj_ctx {
    int ndim, dtype, device;
    int64_t N[HJ_MAX_DIM];
    double xmin[HJ_MAX_DIM], dx[HJ_MAX_DIM];
    int bc[HJ_MAX_DIM], tz[HJ_MAX_DIM];
    int halo_lo, halo_hi;
    hipStream_t stream;
    size_t esz;
    int64_t total;
    void* coord[HJ_MAX_DIM];           // device, dtype
    std::vector<double> coord_host[HJ_MAX_DIM];
    void* aux[4];
    int64_t aux_n[4];
    unsigned long long* ring;          // RING_SLOTS * HJ_MAX_DIM keys
    int ring_pos;
    int slot_ring[HJ_BOUND_SLOTS];     // user slot -> ring index (-1 = none)
    unsigned long long* ring_keep = nullptr;     // where the live slots' entries wait while the ring is zeroed (next_ring)
    unsigned long long* keys;          // scratch keys: [0,32) weno eps (8 groups x 4 dims), [32,36) upwind min/max
    void* weno_vals;                   // HJ_MAX_DIM values of dtype
    const void* weno_src;              // caller-provided eps source or null
    int* flag;                         // nan flag
    double* partials;                  // per-workgroup partial maxima of max_d1sq_kernel
    int partials_cap;
    // static step bound cache
    int sb_ham;
    double sb_par[8], sb_val, sb_local, sb_alpha[HJ_MAX_DIM];
    int diss_local;                    // hj_ctx_set_dissipation: step bound of the local LF variants
    int post_step_op;                  // hj_ctx_set_post_step: fused into the last stage of hj_rk_step
    const void* post_arr[2];           // hj_ctx_set_post_arrays
    int post_arr_op[2];
    bool sb_valid;
    int internal_slot;
    // slab communication (hj_comm_*)
    ncclComm_t comm;
    int comm_rank, comm_size, lo_rank, hi_rank;
    hipStream_t comm_stream, edge_stream;
    hipEvent_t ev_start, ev_edge, ev_edge2, ev_comm;
    hipEvent_t ev_int[3];              // interior of RK stage s done (deep-halo stepper)
    int slab_pending;
    int slab_overlap2 = 0;             // HJ_SLAB_SCHEDULE=overlap2 (experiment)
    long long term_tiled_from = 1000000;   // HJ_TERM_TILED_FROM: grids of at least this many cells run the terms through the tiled kernel
    int slab_gated = 0;                // HJ_SLAB_SCHEDULE=gated (round 4): edges + interior in ONE launch, exchange gated on a counter
    unsigned long long* gate = nullptr;           // device counter the edge workgroups of gated launches add to
    unsigned long long gate_count = 0;            // its value once every gated launch issued so far has published
    int gate_posted = 0;                          // workgroups of the last launch that publish (0: it did not gate)
    hipStream_t compute_stream = nullptr;         // HJ_SLAB_RESERVE_CUS > 0: CU-masked stream the slab launches run on
    int slab_serial = 1;               // HJ_SLAB_SCHEDULE: edges and interior of a substep on ONE stream, edges first (round 3)
    int external_exchange;             // hj_comm_init_external: the caller fills the pad planes itself
    hipEvent_t launch_stop;            // if set, the next tiled launch signals this event on completion
    int ext_events;                    // HJ_EXT_EVENTS (default 1): use that instead of hipEventRecord
    // axis-0 tables extended by pad0 planes either side (deep-halo stepper computes on pad planes)
    int pad0;
    void* coord0_ext;
    void* aux_ext[2];
    // tuning
    hjh::KernelCfg cfg;
    int force_direct, debug, full_rows, num_cus, pd, occ_hint, cfg_from_env, lds_pad;
    int target_blocks, min_chunk, warmup_cost, no_plain;
    int fuse12, f12_r, f12_nt, f12_kh, f12_warm, f12_e2;
    long long direct_below = 0;                     // HJ_DIRECT_BELOW: grids with fewer cells run direct_substep_kernel
    int autotune = 1, autotune_passes = 6;          // HJ_AUTOTUNE, HJ_AUTOTUNE_PASSES
    long long autotune_min_cells = 40000000;        // HJ_AUTOTUNE_MIN_MCELLS
    std::map<long long, hjh::TuneState> tune;       // per launch shape (scheme, stage class, kernel configuration)
    hipEvent_t tune_ev[2] = {nullptr, nullptr};
    int tile_cells = 0;                             // HJ_TILE_CELLS: cap on the cells of a tile (0 = what the configuration holds)
    int pair_dirs = 0;                              // HJ_PAIR_DIRS (resolved in hj_ctx_create): chunks of a tile column march pairwise in opposite directions
    int tile_block[2] = {4, 4};                     // HJ_TB1 / HJ_TB2: 4-D tile order in blocks of this many tiles along axes 1 and 2 (0: plain order)
    int f12_e1 = 0;                                 // tuning / tests: force the tile's row count (pair variant)
    int f12_pair = 1;                               // stage-fused kernel with two cells per lane (hj_fused12v.h): 0 off, 1 if a tiling exists, 2 or fail
    // Hamiltonians whose alpha reads the costate range (hj_rtc.hip, HJ_HAM_RANGE): 2*HJ_MAX_DIM keys the range pass of a substep writes
    unsigned long long* range_keys = nullptr;       // the entry of range_ring the last range pass wrote (what fill_ham hands to the kernels)
    unsigned long long* range_ring = nullptr;       // RANGE_RING entries of RANGE_ENTRY keys, zeroed in bulk (no memset launch per pass)
    // page-locked host words the device writes for the host to poll / read later (hj_rk_step with a range-dependent alpha):
    // [0..8) the bound kernel's keys + sequence number (alpha_bound_kernel), [8..16) the later stages' bound keys of the previous step
    double* dt_dev = nullptr;                       // deltaT of the step in flight, written by alpha_bound_kernel (DtArgs)
    double* alpha_part = nullptr;                   // per-workgroup partial maxima of alpha_bound_kernel (ALPHA_BLOCKS_MAX x 8)
    unsigned long long* host_words = nullptr;
    unsigned long long host_seq = 0;
    int stage_bounds_pending = 0;                   // stages whose bounds are on their way into host_words[8..] (0: none)
    hipEvent_t ev_bounds = nullptr;                 // recorded behind those copies, on the stream they were issued on
    double stage_bounds_dt = 0;                     // deltaT of the step they belong to
    double prev_bounds[3] = {0, 0, 0}, prev_bounds_dt = 0;   // the newest step whose later-stage bounds have been decoded (hj_rk_prev_bounds)
    int prev_bounds_n = 0;
    bool prev_bounds_new = false;
    int range_pos = 0;
    const unsigned long long* range_src = nullptr;  // hj_ctx_set_range_source: keys reduced by the caller (over all ranks); the launches then skip their own range pass
    double last_bounds[3] = {0, 0, 0};              // stepBound of the stages of the last hj_rk_step with such a Hamiltonian (hj_rk_last_bounds)
    int last_bounds_n = 0;
    int tile4_sel = -1;                             // HJ_TILE4_SEL: which tile of HJ_TILE4 (hj_inst.hip) to take (-1: the first that fits)
    int flat4_sel = -1;                             // HJ_FLAT4_SEL: which shape of HJ_FLAT4 (hj_inst.hip) to take (-1: the first that fits)
    int flat4 = 1;                                  // HJ_FLAT4: 4-D fp32 light stencils through the full-row kernel (hj_flat4v.h) where the grid's last axis fits: 1 (default)
                                                    // all-periodic plane axes only (where it is the faster one), 2 every grid it fits, 0 never
    int pair4 = 1;                                  // HJ_PAIR4: 4-D fp32 light stencils through the compile-time-tile kernel (hj_fused4v.h)
    int pair, pair_nt, pair_r, pair_kh, pair_occ;   // two cells per lane (hj_fusedv.h): 0 off, 1 on, 2 at any size; config overrides
    int pair_ring = -1;                             // pair kernel: halo ring parked in LDS 3 planes ahead (HJ_PAIR_RING: 0 never, 1 always, -1 auto)
    int last_nbuf = 2;
    int last_nbase = 2;                             // LDS plane buffers of the last pair launch that are NOT planes parked ahead (2; HJ_TWO_PLANES builds: 4)
    int last_E[HJ_MAX_DIM] = {0, 0, 0, 0};           // chunk length and tile extents of the last tiled launch (hj_last_tile)
    // intended WENO5: max(D1^2) of a stage's output reduced inside the producing launch (hj_fused.h, eps_part) + eps_seam_kernel
    long long eps_fuse_min_cells = 2000000;         // HJ_EPS_FUSE_MIN_CELLS: below, launch floors make the pre-pass as fast (51^3: faster)
    int eps_fuse = 1;                               // HJ_EPS_FUSE=0: always the two-launch pre-pass (max_d1sq_kernel)
    double* eps_prod = nullptr;                     // one row per workgroup of the producing launch
    size_t eps_prod_cap = 0;
    double* eps_rows = nullptr;                     // HJ_EPS_ROWS rows of the seam kernel: what the next launch folds
    bool eps_chain_in = false, eps_chain_out = false;   // hj_rk_integrate: the state stays inside the call between steps
    bool eps_ready = false;                         // eps_rows describe the output of the last launch
    std::string timing_dump_path;                   // HJ_TIMING_DUMP=file, read once when the ctx is created (debug)
    const char* timing_dump = nullptr;              // = timing_dump_path.c_str() or null
    int keep_bounds = 0;                            // HJ_KEEP_BOUNDS: reduce the CFL bound in launches whose bound nobody reads
    int lds_pitch_add = 0;                          // HJ_LDS_PITCH_ADD (tuning): extra cells of LDS row padding
    int pair_ah = 3;                                // planes the halo ring is parked ahead (HJ_PAIR_AH, 1..3)
    // hj_plan_substep: a host-only context (no device, no allocation) whose launches stop after the tile / chunk plan is made
    int diss_kind = 0;                              // HJ_DISS_GLF / HJ_DISS_LLF / HJ_DISS_LLLF as set (diss_local: kind != 0); what a range-reading Hamiltonian is evaluated with
    unsigned long long state_gen = 0;               // bumped by every hj_ctx_set_stream / _dissipation / _post_step / _post_arrays (hj_ctx_state_generation)
    // HJ_XP: 0 never march along axis 1, 1 (default) slab launches whose window is thin against axis 1 (xp_wanted, hj_api.hip), 2 every launch
    // that has a transposed instantiation (tests: bitwise against the axis-0 march on any grid)
    int xp_mode = 1;
    // HJ_COOP (default 0 -- built, bitwise equal, measured slower: a grid barrier costs 4-6 us, a kernel boundary 1.5-1.9): hj_rk_step of order
    // 2 / 3 on grids the direct kernel runs is ONE launch with grid barriers between the stages (1: plain launch, 2: hipLaunchCooperativeKernel)
    int coop = 0;
    void* coop_sync = nullptr;                      // CoopSync (device): barrier counters, only ever growing
    unsigned long long coop_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, coop_all = 0;      // what they read once every launch issued so far has run
    int coop_ok = -1;                               // device attribute hipDeviceAttributeCooperativeLaunch (-1: not asked yet)
    int xp_max_planes = 0;                          // HJ_XP_MAX_PLANES: auto mode takes windows of at most this many planes (0: the built-in rule)
    // Auto mode, per (scheme, plane range): the launch form.  `prior` is what the two launch PLANS say (launch_scheme's cost model; all a dry
    // context has); a live context then TIMES both forms on its first calls -- same bits either way: runs of XP_RUN consecutive calls of one
    // form (two RK3 steps: every stage kind, and the cache state that form leaves for itself) between a pair of events, the forms taking
    // turns, each run read back without waiting at a later call of the key; after HJ_XP_TRIALS runs of each the faster form (minimum against
    // minimum) is kept for the life of the context.
    enum { XP_RUN = 6 };
    struct XpTrial {
        bool prior = false;
        int decided = -1;                           // -1: still sampling; 0: axis-0 march; 1: transposed march
        int n[2] = {0, 0};                          // completed, read runs per form
        float best[2] = {1e30f, 1e30f};
        int form = 0, calls = 0;                    // the run under way (calls == 0: none)
        int started = 0;                            // runs begun (a run during which the axis-0 march was still rotating tile shapes is dropped)
        unsigned seq0 = 0;                          // hj_ctx::tune_seq when the run began
        bool pend[2] = {false, false};              // a finished run of this form has not been read back yet
        hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    };
    std::map<long long, XpTrial> xp_choice;
    unsigned tune_seq = 0;                          // tile-shape trials issued so far (tune_begin)
    int debug_xp = 0;                               // HJ_DEBUG: report the decision once per key
    int xp_trials = 2;                              // HJ_XP_TRIALS (0: the plan model alone decides)
    int dry = 0;
    struct { int ntiles = 0, nchunks = 0, nblocks = 0, threads = 0, wg_per_cu = 0; size_t lds_bytes = 0; } last_plan;
    const char* last_kernel = "";                   // name of the substep kernel of the last launch (hj_last_kernel)
    size_t lds_limit;
    // resident workgroups per CU of (kernel instantiation, dynamic LDS bytes) on THIS ctx's device
    std::map<std::pair<const void*, size_t>, int> occ_cache;
    std::map<int, bool> f12_ok;        // (scheme, Hamiltonian) -> the stage-fused kernel has a tiling for this grid
    // make_tiling's answer per (NT, R, KH, vec, nbuf): it depends on nothing else that changes after hj_ctx_create, and the
    // enumeration it runs (64 row extents in 3-D, 64^2 in 4-D) was paid on EVERY launch (round 4: host cost of small grids)
    mutable std::map<long long, hjh::Tiling> tiling_cache;
};

namespace hjh {

using namespace hj;

int env_int(const char* name, int dflt);
Tiling make_tiling(const hj_ctx* c, const KernelCfg& k, int64_t p0, int64_t p1, int vec = 1, int nbuf = 2, std::vector<Tiling>* all = nullptr);
void choose_chunks(const hj_ctx* c, Tiling& t, int64_t p0, int64_t p1, int blocks_per_cu, int64_t chunk_max = 0, double march_bytes = 0, double fixed_bytes = 0);
Tiling make_tiling_dims(const hj_ctx* c, const KernelCfg& k, int vec, int nbuf, const int64_t* dims);      // transposed launches (kernel-order extents)
int cfg_kh(int nd, int nt, int r);
int eps_rows_to_vals(hj_ctx* c, const double* rows, int nrows, hipStream_t stream);   // rows -> ctx->weno_vals (kernels that do not fold)

int ham_npar(int ham);
template <typename T> void fill_ham(const hj_ctx* c, const double* par, HamTables<T>& H, int ham) {
    for (int d = 0; d < HJ_MAX_DIM; ++d) H.coord[d] = (const T*)c->coord[d];
    for (int s = 0; s < 4; ++s) H.aux[s] = (const T*)c->aux[s];
    if (c->coord0_ext) H.coord[0] = (const T*)c->coord0_ext + c->pad0;
    for (int s = 0; s < 2; ++s)
        if (c->aux_ext[s]) H.aux[s] = (const T*)c->aux_ext[s] + c->pad0;
    // (exactly the entries the caller's array has: hj_mi355x.h, "par: ham_npar values")
    const int np = par ? std::max(0, std::min((int)HJ_PAR_SLOTS, ham_npar(ham))) : 0;
    for (int s = 0; s < (int)HJ_PAR_SLOTS; ++s) H.par[s] = s < np ? (T)par[s] : T(0);
    H.range = c->range_src ? c->range_src : c->range_keys;
    H.local_mode = c->diss_kind;
}

template <typename T, int ND> void fill_grid(const hj_ctx* c, GridArgs<T, ND>& G) {
    long long s = 1;
    for (int d = ND - 1; d >= 0; --d) {
        G.n[d] = (int)c->N[d];
        G.bc[d] = c->bc[d];
        G.km[d] = c->tz[d] ? T(-1) : T(1);
        G.inv_dx[d] = (T)(1.0 / c->dx[d]);
        fill_stencil_constants<T>(c->dx[d], G.K[d]);
        G.stride[d] = s;
        s *= c->N[d];
    }
    G.halo_lo = c->halo_lo;
    G.halo_hi = c->halo_hi;
    G.total = c->total;
}

// costate scale the scheme's stencil leaves out (hj_device.h, scaling note)
template <typename T> T scheme_scale(int scheme, double dx) {
    if (scheme == HJ_WENO5_ASSHIPPED) return (T)((1.0 / dx) * (1.0 / 60.0));
    if (scheme == HJ_WENO5) return (T)((1.0 / dx) * (1.0 / 12.0));
    if (lean_eno(scheme)) return (T)((1.0 / dx) * 0.5);     // lean ENO2 / ENO3: costates on undivided differences, p = q/(2dx)
    return T(1);                      // ENO2 / ENO3 on the reference's divided-difference tables: true costates
}

struct SubstepCall {
    int scheme, ham, stage, restrict_sign;
    const double* par;
    double dt;
    const void *y, *y0;
    void* out;
    unsigned long long* bound;
    int64_t p0, p1;
    int64_t q0 = 0, q1 = 0;   // optional second plane range in the same launch (tiled kernel only)
    // gated slab launch (round 4): up to two EDGE plane ranges ride in this launch, ahead of [p0, p1) in chunk and dispatch
    // order; each of their workgroups adds 1 to ctx->gate when its planes are in memory.  The launch code reports how many
    // will (ctx->gate_posted); 0 = the launch could not gate (direct kernel): the caller orders the exchange with an event
    int64_t e0[2] = {0, 0}, e1[2] = {0, 0};
    bool gated = false;
    const void* term = nullptr;   // TermPar<T> of a TermOp launch (hj_termop.h; the tiled term path of round 4), else null
    int post_op = 0;          // fused post-step min/max with the state the step started from
    bool on_aux = false;      // launch on the ctx's auxiliary (edge) stream instead of the ctx stream
    // intended WENO5 inside hj_rk_step / hj_rk_integrate: this launch's output is the next launch's input (reduce max(D1^2)
    // of it in this launch), and this launch's input was the previous launch's output (its epsilon rows are ready)
    bool want_eps = false, eps_from_prev = false;
    const double* eps_rows = nullptr;   // set by do_substep: fold these rows of max D1^2 (HJ_MAX_DIM doubles each) instead of
    int eps_nrows = 0;                  // reading weno_vals
    // range-dependent alpha (hj_rtc.hip): run ONLY the range pass of [p0, p1) into range_out (2*HJ_MAX_DIM keys, zeroed by the launch code)
    bool range_only = false;
    unsigned long long* range_out = nullptr;
    bool range_ready = false;           // ctx->range_keys already hold the range of this launch's input: no pass of its own
    bool bound_pass = false;            // run ONLY the bound pass of the local Lax-Friedrichs variants (MODE 3 with a bound slot): max_x sum_d alpha_d / dx_d
    const double* dt_dev = nullptr;     // deltaT in device memory (FusedArgs::dt_dev; run-time Hamiltonians with a range-dependent alpha only)
    // TRANSPOSED MARCH (round 6): the caller would like [p0, p1) computed by a launch that marches along grid axis 1 and tiles the slab axis
    // (hj_fusedv.h, XP) -- taken where a transposed instantiation exists for the call (launch_cfg), else the ordinary launch: same bits either way
    bool xp = false;
};
constexpr int HJ_EPS_ROWS = 256;  // workgroups (= rows) of eps_seam_kernel

// termNormal / termReinit / termConvection through the tiled substep kernel (hj_inst.hip compiled with -DHJ_INST_TERM_ND; fp64, 2-D / 3-D)
template <typename T, int ND> int launch_term_tiled(hj_ctx* c, int kind, const SubstepCall& s);

inline hipStream_t call_stream(const hj_ctx* c, const SubstepCall& s) { return s.on_aux ? c->edge_stream : c->stream; }

// (threads per workgroup, cells per thread, halo slots per thread, waves/SIMD hint, prefetch depth) instantiated for
// the tiled kernel.  Only what launch_cfg's default chooser can pick is built (round 3: the round-2 tables also
// carried runners-up that spill and that nobody selected); cfg_built() narrows the table per scheme, so that the
// heavy stencils (ENO3, intended WENO5) are not compiled in the 4-cells-per-thread shapes they would spill in.
// A configuration requested through HJ_NT / HJ_PAIR_NT ... that is not built is an ERROR, not a silent fallback.
#ifndef HJ_CONFIGS
#ifdef HJ_ALL_CONFIGS   // the full sweep table (tools/cfgsweep.sh); ~2.5 min to compile
#define HJ_CONFIGS(X) X(512, 4, 2, 2, 2) X(1024, 2, 1, 4, 2) X(512, 2, 1, 4, 2) X(256, 4, 3, 2, 2) X(256, 2, 2, 4, 2) \
                      X(512, 1, 1, 4, 2) X(256, 1, 2, 6, 2) X(1024, 1, 1, 4, 2) \
                      X(512, 4, 2, 2, 3) X(256, 4, 3, 2, 3) X(512, 1, 1, 4, 3) X(512, 2, 1, 3, 3) X(512, 2, 1, 3, 2) \
                      X(512, 2, 1, 2, 2) X(256, 2, 2, 2, 2) X(512, 1, 1, 2, 2) X(256, 4, 3, 1, 2) X(256, 2, 2, 3, 2)
#else
#define HJ_CONFIGS(X) X(512, 4, 2, 2, 2) X(256, 2, 2, 2, 2) X(512, 1, 1, 4, 2)
#endif
#endif

// 4-D grids tile three plane axes: the halo cross is ~2x the tile, so more halo slots per thread (cfg_built() says which of these
// a (dtype, scheme) really compiles)
#ifndef HJ_CONFIGS_4D
#define HJ_CONFIGS_4D(X) X(512, 2, 4, 2, 2) X(1024, 1, 3, 2, 2) X(512, 1, 4, 2, 2)
#endif

// light stencils: few enough live values that 4 cells (2 pairs) per thread fit in 256 VGPRs without scratch
constexpr bool light_scheme(int scheme) { return scheme == HJ_WENO5_ASSHIPPED || scheme == HJ_ENO2 || scheme == HJ_ENO2_FAST; }
// ... and on 2-D grids the lean ENO3 (HJ_ENO3_FAST) as well: 227-241 VGPRs without scratch in the two-pairs-per-thread shape (3-D: 260 B of scratch)
constexpr bool light_cfg(int scheme, int nd) { return light_scheme(scheme) || (scheme == HJ_ENO3_FAST && nd == 2); }
// is configuration (NT, R) of the one-cell-per-lane / pair kernel compiled for this scheme?
#ifndef HJ_PAIR_EXTRA      // tuning builds: further (threads, pairs per thread) shapes of the pair kernel for the light stencils
#define HJ_PAIR_EXTRA(nt, r) false
#endif
constexpr bool cfg_built(int scheme, int nd, int nt, int r, bool pair, int esz = 8) {
#ifdef HJ_ALL_CONFIGS
    return true;
#else
    // 4-D (three tiled plane axes): fp32 runs 1024 single cells or -- light stencils, round 3 -- 256 threads x 2 pairs
    // (two independent workgroups per CU); fp64 512 x 1 cell (two cells spill 68-308 B and are 20-100 % slower for every stencil
    // but ENO2, tools/experiments/r03_run44.sh).  The other combinations spill and nobody selects them.
    if (nd == 4) return pair ? (esz == 4 && light_scheme(scheme)) : (esz == 4 ? nt == 1024 : (nt == 512 && r == 1));
    if (pair) return light_cfg(scheme, nd) ? (nt == 512 && r == 2) || (nt == 256 && r == 1) || HJ_PAIR_EXTRA(nt, r) : (nt == 256 && r == 1);
    return light_scheme(scheme) ? true : (nt == 256 && r == 2);
#endif
}

// Hamiltonians compiled at run time (hj_rtc.hip)
int next_range_keys(hj_ctx* c);           // advance ctx->range_keys to a zeroed entry of the range ring
bool user_ham_valid(int ham);
bool user_ham_dynamic(int ham);           // alpha depends on the data (the costate range): no static step bound, dt from the first stage's reduction
int user_ham_ndim(int ham);
int user_ham_npar(int ham);
int launch_user(hj_ctx* c, const SubstepCall& s);
// done / host_out / seq: see alpha_bound_kernel (hj_split.h) -- the last workgroup publishes the keys to page-locked host memory
int user_alpha_bound(hj_ctx* c, int ham, const double* par, unsigned long long* keys, unsigned long long* done, bool with_range = false,
                     unsigned long long* host_out = nullptr, unsigned long long seq = 0, const hj::DtArgs* dt = nullptr);
int alpha_partials(hj_ctx* c);        // allocates ctx->alpha_part on first use

// the fused (tiled) or direct substep kernel of one (dtype, Hamiltonian): defined and explicitly
// instantiated in hj_inst.hip
template <typename T, typename HAM> int launch_scheme(hj_ctx* c, const SubstepCall& s);
// the TRANSPOSED launch of a substep (hj_instx.hip: march along grid axis 1, the slab axis tiled): HJ_OK, an error, or HJ_XP_FALLBACK when
// the call has no transposed form (the caller then takes the ordinary launch -- same bits).  Instantiated for the pairs xp_available names.
constexpr int HJ_XP_FALLBACK = -7777;
template <typename T, typename HAM> int launch_xp(hj_ctx* c, const SubstepCall& s);
template <typename T, typename HAM> constexpr bool xp_available() { return HAM::ID == HJ_HAM_DUBINS_REL && !ham_xp<HAM>::value; }
bool xp_wanted(const hj_ctx* c, int64_t p0, int64_t p1);

// a whole odeCFL2 / odeCFL3 step of a small grid in ONE cooperative launch (hj_split.h, coop_rk_kernel): HJ_OK, an error, or HJ_XP_FALLBACK
// (no instantiation / the grid does not fit the resident workgroups): the caller then issues the stages one launch each -- same bits
struct CoopCall {
    int scheme, ham, order, restrict_sign, post_op;
    const double* par;
    double dt;
    const void* y;
    void *s1, *s2, *out;
};
template <typename T, typename HAM> int launch_coop(hj_ctx* c, const CoopCall& s);

// two RK stages in one launch (hj_fused12.h): out = ca*y + cb*(y1 + dt*L(y1)), y1 = y + dt*L(y)
struct Stage12Call {
    int scheme, ham;
    const double* par;
    double dt, ca, cb;
    const void* y;
    void* out;
    unsigned long long* bound;
    bool probe = false;       // only answer whether a tiling exists (HJ_OK) or not (HJ_EUNSUPPORTED)
};
template <typename T, typename HAM> int launch_stage12(hj_ctx* c, const Stage12Call& s);

}  // namespace hjh
