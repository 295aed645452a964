// The fused Lax-Friedrichs RK-substep kernel (gfx950).
//
// One launch = one evaluation of termLaxFriedrichs (term_lax_friedrich.py:94-130) on the whole
// grid -- upwind derivatives in every dimension, analytic Hamiltonian, global-LF dissipation,
// CFL reduction (artificial_diss_glf.py:75-109) -- fused with the RK stage combination of
// odeCFLn (ode_cfl_3.py:151-241), ghost cells synthesised on the fly (add_ghost_*.py).
//
// Decomposition ("2.5-D blocking"):
//   * axis 0 (slowest, C order) is the MARCH axis: a workgroup owns a tile of the remaining
//     D-1 axes (the "plane") and sweeps a chunk of axis-0 planes, keeping the 7-point axis-0
//     stencil of each of its cells in a register queue (every value is loaded once).
//   * the centre plane of the tile, with a 3-cell halo on every in-plane axis, is staged in LDS
//     (double buffered: one s_barrier per plane); in-plane stencils are read from LDS.
//     Halo cells outside the domain are ghost cells computed at load time.
//   * software pipeline: the HBM loads a plane needs (its own cells for the queue, the halo ring,
//     the RK operand y0, the plane's Hamiltonian scalars) are issued PD (2 or 3) planes before they
//     are consumed, into rotating register sets (the loop body is instantiated PD times with the
//     sets permuted, so in-flight data is never moved): a full plane of arithmetic always covers
//     the memory latency and the memory queue does not drain at the barrier.
//   * thread <-> cell: the tile's cells are numbered linearly (last axis fastest) and dealt
//     round-robin to the NT threads, R cells per thread: consecutive lanes touch consecutive
//     addresses in HBM (coalesced, full 64-lane waves even when N is not a multiple of 64) and
//     consecutive 8-byte LDS words (bank-conflict free for ds_read_b64).  Threads beyond the
//     tile's cell count shadow its last cell (they compute, only their LDS/HBM writes are
//     predicated off): no per-lane validity branches around the arithmetic.
//   * blockIdx -> (chunk, tile) is XCD-aware: the 8 XCDs get contiguous ranges of the logical
//     block order, so tiles sharing halo rows share an L2.
//   * per-dim max(alpha) is reduced with wavefront shuffles, then LDS, then one 64-bit
//     atomicMax per workgroup and dimension.
#pragma once
#include "hj_device.h"
#include "hj_termop.h"

// PAIRED CHUNKS (hj_fusedv.h): chunks of a tile column marching pairwise in opposite directions so that the two workgroups of a
// pair request their six common planes at the same moment.  Built and measured in round 4: -4 % fabric traffic at 201^3, but the
// down-marching support costs the plane loop ~90 instructions per two planes (8 more uniform branches, 30 more SGPR-spill moves,
// a second copy of the axis-0 stencil code) in EVERY launch -- and a build without it is faster than pairing everywhere:
// 201^3 36.3-38.5 us per launch against 38.4-38.7 paired, 513^3 +0.5 % (profiles/r04_paired_chunks.txt, last section).
// 0 (default): compiled out, HJ_PAIR_DIRS is ignored; 1: the round-4 experiment.
#ifndef HJ_MAYDOWN
#define HJ_MAYDOWN 0
#endif

// Surplus halo slots shadow slot 0 (same source, same LDS cell, same value).  In 4-D (10 slots per thread, 2-3 % of them
// surplus) their LDS stores go out unpredicated -- a benign duplicate write instead of an exec save + branch per slot and
// plane: C5 +3 % (tools/experiments/r03_run54.sh).  In 2-D / 3-D most slots of the last round are surplus (2-D: 506 of 512)
// and whole waves would store to ONE LDS address, which the LDS serialises (C3 -9 %): there the predicate stays.
// -DHJ_PRED_HALO (tuning builds): the predicate everywhere.
#ifdef HJ_PRED_HALO
#define HJ_SLOT_PRED(k) if (h_real[k])
#else
#define HJ_SLOT_PRED(k) if (ND < 4 ? h_real[k] : true)
#endif
namespace hj {

// cache policies of the streams (tuning macros): y0 and the output are touched once
#ifndef HJ_AUX_Y0
#define HJ_AUX_Y0 0
#endif
#ifndef HJ_AUX_ST
#define HJ_AUX_ST 0
#endif
#ifndef HJ_AUX_OWN
#define HJ_AUX_OWN 0
#endif

// LDS stencil read.  ds_read2_b64 moves 16 B/lane in 8 LDS cycles, two ds_read_b64 in 4
// (MI355X_MICROARCH.md, LDS table), and the LDS pipe is the busiest unit of this kernel (PMC: 65 % at 513^3,
// half of it counted as bank-conflict cycles).  The back end pairs any two LDS reads that share a base
// register and differ by a constant offset, i.e. the six neighbours along the contiguous axis; passing each
// address through an empty asm makes the bases opaque, so they stay single ds_read_b64 (one extra 32-bit
// add per read).  Measured (round 2): the opaque offsets cost 12-18 VGPRs -- one wave per SIMD less at 201^3
// (172 -> 184 VGPRs), scratch at 401^3+ -- and gain nothing; default 0 = let the compiler pair.
#ifndef HJ_LDS_NO_READ2
#define HJ_LDS_NO_READ2 0
#endif
template <typename T> __device__ __forceinline__ T lds_read(const T* buf, int o) {
#if HJ_LDS_NO_READ2
    if constexpr (sizeof(T) == 8) asm("" : "+v"(o));     // the 32-bit element offset, so that `buf` keeps its LDS address space
#endif
    return buf[o];
}

template <typename T, int ND> struct FusedArgs {
    const T* max_d1sq;            // ND values (HJ_WENO5 only)
    unsigned long long* bound;    // ND keys (atomicMax)
    int n[ND];
    int bc[ND];
    int halo_lo, halo_hi;
    T km[ND];                     // slope multiplier (+1, -1 if towardZero)
    T K[ND][HJ_NK];               // per-dim stencil constants (fill_stencil_constants)
    T sc[ND];                     // costate scale of the scheme: 1/(60dx) as-shipped WENO5, else 1
    long long stride0;            // elements per axis-0 plane
    int pstride[ND];              // in-plane element strides (pstride[0] unused)
    int E[ND];                    // tile extents on the plane axes (E[0] unused)
    int lpitch;                   // LDS row pitch (elements) of the last axis: E[ND-1] + halo (+ bank padding)
    int ntile[ND];
    int ntiles;
    int chunk, nchunks;
    int plane_begin, plane_end;
    int plane_begin2, plane_end2, nchunks1;   // optional second plane range (slab edges): chunks >= nchunks1
    int nblocks, blocks_per_xcd;
    // Gated slab launch (round 4; hj_api.hip, slab_substep): the EDGE plane ranges of a slab ride in the same launch as its
    // interior.  Their chunks (echunk planes each) come first in the chunk order -- eplane[0] (nchunks_e1 chunks), then
    // eplane[1], nchunks_e in all -- and first in the dispatch order (the first edge_blocks workgroups, mapped XCD-aware among
    // themselves); every one of their workgroups adds 1 to *gate once its planes are stored and released, and the stream
    // that posts the halo exchange waits on that count (hipStreamWaitValue64).  nchunks_e = 0: an ordinary launch.
    // 4-D (round 4): tiles of a chunk are ordered in blocks of tb[1] x tb[2] tiles of the plane axes 1 and 2 (tile_coords): the
    // ~64 workgroups an XCD holds at a time then form a compact block whose inner tile faces are served by that XCD's L2.
    // tb[1] = 0: plain order, last axis fastest (the 2-D sheets of that order never hold an axis-1 neighbour).  Measured on C5
    // (profiles/r04_c5_tile_order.txt): fabric reads per launch 4.55 -> 4.11 GB (2.47x -> 2.22x the algorithmic bytes), time
    // -1.5 %: the launch is not bound by its traffic (VALU 55-63 % busy, 130 lane-operations per cell)
    int tb[ND];
    // pair kernel, round 4: the first 2*npairs chunks of the main range march in opposite directions, pairwise
    // (hj_fusedv.h, "PAIRED CHUNKS"); 0 = every chunk marches up
    int npairs;
    int eplane[2][2];
    int echunk, nchunks_e1, nchunks_e;
    int edge_blocks, edge_count, edge_bpx;
    unsigned long long* gate;
    // pair kernel only: LDS plane buffers (2 = double buffer) and how many planes ahead of its use the halo ring
    // of a plane is parked in LDS (0 = written in the iteration that consumes it)
    int lds_nbuf, halo_ahead;
    // out = ydot                                   (ydot_only)
    //     = ca*y0 + cb*(y + dt*ydot)               otherwise; y0 is read iff use_y0
    int ydot_only, use_y0;
    int stage;                    // HJ_STAGE_*: which stage expression (the NumPy-order ENO path evaluates the reference's own form)
    T ca, cb, dt;
    int post_op;                  // 0 none, 1/2: out = min/max(out, state at the start of the step)
    int do_clamp;                 // termRestrictUpdate: ydot clamped to [lo, hi]
    T clamp_lo, clamp_hi;
    HamTables<T> ham;
    // intended WENO5 (round 3): epsilon_d = 1e-6 * max(D1_d^2) of the launch's INPUT (upwind_first_weno5a.py:153-156).
    // eps_nrows > 0: folded in the prologue from eps_rows (rows of HJ_MAX_DIM doubles: max D1^2) instead of read from max_d1sq;
    // eps_part != null: this launch reduces its own OUTPUT for the next stage -- one row per workgroup with the largest
    // |forward difference| per dimension over the pairs inside its tile and chunk; tile / chunk seams and periodic wrap pairs
    // are eps_seam_kernel's (hj_split.h), which also turns the maxima into D1^2
    double* eps_part;
    const double* eps_rows;
    int eps_nrows;
    T inv_dx[ND];
    // debug (HJ_TIMING_DUMP): per logical block {start, end} of the constant 100 MHz clock, {xcc id, chunk}
    unsigned long long* timing;
    TermPar<T> term;              // TermOp launches (termNormal / termReinit / termConvection through the tiled kernel; hj_termop.h)
    // MODE 3 (the RANGE PASS of a substep, round 5): nothing is written but 2*ND keys -- [d] max, [ND + d] -min of derivL_d / derivR_d
    // over the launch's cells (artificial_diss_glf.py:80-88: derivMin / derivMax), for Hamiltonians whose alpha reads them
    unsigned long long* range_keys;
    // deltaT from DEVICE memory (round 5; Hamiltonians that define DT_DEV, i.e. the run-time ones whose alpha reads the costate range): the
    // bound kernel that follows the range pass computes deltaT (ode_cfl_3.py:142) and the first stage is already enqueued behind it -- the
    // host learns the same value from page-locked memory meanwhile, the GPU never waits for it.  null: `dt` above.
    const double* dt_dev;
    // TRANSPOSED MARCH (round 6; Hamiltonian types with XPOSED, hj_device.h): the kernel's axes (march, tile axis 1, ...) are the grid's
    // (1, 0, ...).  Every per-axis field above is in KERNEL order; halo_lo / halo_hi describe the MARCH axis (never a slab axis then: 0).
    // The slab axis is tile axis 1: the launch computes the WINDOW [xwin0, xwin1) of it (a plane range of the slab, possibly reaching into
    // the pad planes), cells outside [0, n[1]) are real data where xh_lo / xh_hi say so (pad planes of the slab) and ghost / periodic
    // otherwise; per-lane byte offsets are relative to index xbase (<= 0) of that axis, xspan = bytes from there to the last row the
    // window's halo can touch (march offset excluded).  Unused (0) in ordinary launches.
    int xwin0, xwin1, xbase, xh_lo, xh_hi;
    unsigned xspan;
};

// deltaT of this launch (built-in Hamiltonians: the kernel argument, exactly the code of rounds 1-4)
template <typename H, typename = void> struct ham_dt_dev { static constexpr bool value = false; };
template <typename H> struct ham_dt_dev<H, typename hj_void<decltype(H::DT_DEV)>::type> { static constexpr bool value = H::DT_DEV; };
template <typename HAM, typename T, int ND>
__device__ __forceinline__ T launch_dt(const FusedArgs<T, ND>& A) {
    if constexpr (ham_dt_dev<HAM>::value) return A.dt_dev != nullptr ? (T)__builtin_nontemporal_load(A.dt_dev) : A.dt;
    else return A.dt;
}

// end of a range pass: the per-thread minima / maxima -> 2*ND atomicMax on order-preserving keys (one per workgroup and value)
template <int ND, int NT, typename T>
__device__ __forceinline__ void publish_range(unsigned long long* keys, double (*red)[ND], const T* rmn, const T* rmx, const T* sc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double m = wave_max(half == 0 ? (double)(sc[d] * rmx[d]) : -(double)(sc[d] * rmn[d]));
            if (lane == 0) red[wv][d] = m;
        }
        __syncthreads();
        if ((int)threadIdx.x < ND) {
            double m = red[0][threadIdx.x];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w][threadIdx.x]);
            key_max(keys + half * ND + threadIdx.x, m);
        }
    }
}

// Logical block of this workgroup, -1 for the padding blocks of the launch.  Blocks b, b + 8, b + 16 ... share an XCD
// (round-robin dispatch): every XCD gets a contiguous run of logical blocks.  In a gated slab launch the first edge_blocks
// workgroups are the edge chunks (logical blocks [0, edge_count)), mapped the same way among themselves.
template <typename ARGS> __device__ __forceinline__ int logical_block(const ARGS& A) {
    const int b = blockIdx.x;
    if (b < A.edge_blocks) {
        const int L = (b & 7) * A.edge_bpx + (b >> 3);
        return L < A.edge_count ? L : -1;
    }
    const int b2 = b - A.edge_blocks;
    const int L = A.edge_count + (b2 & 7) * A.blocks_per_xcd + (b2 >> 3);
    return L < A.nblocks ? L : -1;
}

// Tile coordinates tc[1..ND-1] of tile `rem` of a chunk (see FusedArgs::tb).  Blocked order (4-D): block rows of tb[1] tiles
// along axis 1, in each row blocks of tb[2] tiles along axis 2, in each block
// the (up to) tb[1] x tb[2] positions of the block, for each of them ALL tiles of the contiguous axis 3 one after the other
// (a first version that walked axis 3 slowest lost the sharing of the cache lines the 34-cell rows straddle: reads 2.5x ->
// 3.9x); partial blocks at the ends of the axes are enumerated exactly (no padding: every index in [0, ntiles) is one tile).
template <int ND, typename ARGS> __device__ __forceinline__ void tile_coords(const ARGS& A, int rem, int* tc) {
    if constexpr (ND == 4) {
        if (A.tb[1] > 0) {
            const int B1 = A.tb[1], B2 = A.tb[2], n1 = A.ntile[1], n2 = A.ntile[2], n3 = A.ntile[3];
            int b1, r1, b2, r2, t3, r3, w1, w2;
            fdivmod(rem, fdiv_make(B1 * n2 * n3), b1, r1);
            const int h1 = min(B1, n1 - B1 * b1);
            fdivmod(r1, fdiv_make(h1 * B2 * n3), b2, r2);
            const int h2 = min(B2, n2 - B2 * b2);
            fdivmod(r2, fdiv_make(n3), r3, t3);          // the axis-3 tiles of one (axis 1, axis 2) position stay adjacent: they
            fdivmod(r3, fdiv_make(h2), w1, w2);          // share the cache lines their 34-cell rows straddle
            tc[1] = B1 * b1 + w1;
            tc[2] = B2 * b2 + w2;
            tc[3] = t3;
            return;
        }
    }
#pragma unroll
    for (int d = ND - 1; d >= 1; --d) {
        int q;
        fdivmod(rem, fdiv_make(A.ntile[d]), q, tc[d]);
        rem = q;
    }
}

// planes [p_begin, p_end) of chunk `chunk_id`: edge ranges first (gated launches), then the main range, then the optional
// second range (low and high edge planes of a slab in one launch)
template <typename ARGS> __device__ __forceinline__ void chunk_planes(const ARGS& A, int chunk_id, int& p_begin, int& p_end) {
    if (chunk_id < A.nchunks_e) {
        const int w = chunk_id >= A.nchunks_e1 ? 1 : 0;
        p_begin = A.eplane[w][0] + (chunk_id - (w ? A.nchunks_e1 : 0)) * A.echunk;
        p_end = min(p_begin + A.echunk, A.eplane[w][1]);
        return;
    }
    const int cm = chunk_id - A.nchunks_e;
    const bool second = cm >= A.nchunks1;
    p_begin = second ? A.plane_begin2 + (cm - A.nchunks1) * A.chunk : A.plane_begin + cm * A.chunk;
    p_end = min(p_begin + A.chunk, second ? A.plane_end2 : A.plane_end);
}

// end of an edge chunk's workgroup in a gated launch: its planes are in memory (every wave's stores drained, L2 written
// back at system scope: the exchange kernel may run on any XCD, or be a peer's DMA) before the count moves
template <typename ARGS> __device__ __forceinline__ void publish_gate(const ARGS& A, int chunk_id) {
    if (A.gate != nullptr && chunk_id < A.nchunks_e) {          // workgroup-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the compiler may drop the wait behind buffer_wbl2 (MI355X guide)
            __hip_atomic_fetch_add(A.gate, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// epsilon of the intended WENO5 from partial rows: every thread of the workgroup ends up with the ND maxima
template <typename T, int ND, int NT>
__device__ __forceinline__ void fold_eps_rows(const double* __restrict__ rows, int nrows, double (*red)[ND], T* eps) {
    double m[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) m[d] = 0.0;
    for (int i = threadIdx.x; i < nrows; i += NT)
#pragma unroll
        for (int d = 0; d < ND; ++d) m[d] = fmax(m[d], rows[(size_t)i * HJ_MAX_DIM + d]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        double w = red[0][d];
        for (int k = 1; k < NT / 64; ++k) w = fmax(w, red[k][d]);
        eps[d] = T(1e-6) * (T)w + Lim<T>::tiny;
    }
    __syncthreads();          // red is the CFL reduction's scratch as well
}

// one row of partials per workgroup (eps_part): max |forward difference| per dimension, in double.  (fl(fl(K*x)^2) is
// monotone in |x| for K > 0, so the maximum of D1^2 = (inv_dx*(a - b))^2 over pairs is that expression of the maximum
// |a - b|: the seam kernel applies it once, after folding -- two VALU operations per pair here instead of five)
template <int ND, int NT>
__device__ __forceinline__ void store_eps_part(double* __restrict__ row, double (*red)[ND], const double* dmax) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double w = wave_max(dmax[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < ND) {
        double w = red[0][threadIdx.x];
        for (int k = 1; k < NT / 64; ++k) w = fmax(w, red[k][threadIdx.x]);
        row[threadIdx.x] = w;
    }
    __syncthreads();
}

// where the values of axis-0 plane `p` (possibly a ghost plane) come from: wave-uniform
template <typename T> struct PlaneSrc {
    long long off;     // element offset of the source plane (edge plane for extrapolation)
    long long off_in;  // inner plane (extrapolation only)
    T km;              // k * slope multiplier
    bool ghost;
};

template <typename T, int ND>
__device__ __forceinline__ PlaneSrc<T> plane_src(const FusedArgs<T, ND>& A, int p) {
    PlaneSrc<T> s;
    const int n0 = A.n[0];
    s.ghost = false;
    s.km = T(0);
    s.off_in = 0;
    int src = p;
    if (p < 0 && !A.halo_lo) {
        if (A.bc[0] == HJ_BC_PERIODIC) src = p + n0;
        else { src = 0; s.off_in = A.stride0; s.km = T(-p) * A.km[0]; s.ghost = true; }
    } else if (p >= n0 && !A.halo_hi) {
        if (A.bc[0] == HJ_BC_PERIODIC) src = p - n0;
        else {
            src = n0 - 1;
            s.off_in = (long long)(n0 - 2) * A.stride0;
            s.km = T(p - n0 + 1) * A.km[0];
            s.ghost = true;
        }
    }
    s.off = (long long)src * A.stride0;
    return s;
}

// MODE: 0 = every combination (ydot only, termRestrictUpdate clamp, fused post-step operator) decided by
// runtime flags; 1 / 2 = the plain RK stages that make up almost every launch -- Euler (no y0 operand) /
// convex combination with y0 -- with those flags compiled out (the clamp alone is ~12 VALU operations per
// cell and the flags cost ~5 scalar branches per cell).  Same expressions, so the results are bitwise equal.
template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, int PD, int MODE = 0>
__global__ __launch_bounds__(NT, OCC) void fused_substep_kernel(const T* __restrict__ y,
                                                           const T* __restrict__ y0,
                                                           T* __restrict__ out,
                                                           const FusedArgs<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr bool GEN = (MODE == 0);
    constexpr bool NP = np_order(SCHEME);       // ENO2 / ENO3: every operation rounded as NumPy rounds it (hj_device.h)
    // TERM (round 4): HAM is a TermOp -- the launch evaluates termNormal / termReinit / termConvection instead of the
    // Lax-Friedrichs term: the stencils yield derivL / derivR (upwind<SCHEME>), the cell arithmetic is term_cell, the
    // coefficient array 0 (speed / initial / velocity 0) rides in the y0 stream, the output is ydot (MODE 0, A.ydot_only)
    constexpr bool TERM = ham_traits<HAM>::is_term;
    constexpr int TKIND = ham_traits<HAM>::kind;
    static_assert(!TERM || MODE == 0, "terms run the general instantiation");
    constexpr bool RNG = (MODE == 3);           // range pass: derivL / derivR of every cell reduced to their minima / maxima, nothing stored
    const bool use_y0 = GEN ? (A.use_y0 != 0) : (MODE == 2);
    extern __shared__ __align__(16) unsigned char hj_smem[];
    // dynamic LDS only (keeps the carve base 16-byte aligned): [0,512) reduction scratch, then planes
    double (*red)[ND] = reinterpret_cast<double (*)[ND]>(hj_smem);
    T* lds = reinterpret_cast<T*>(hj_smem + 512);
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");

    // ---- XCD-aware block order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch),
    // give each XCD a contiguous run of logical blocks.
    const int b = blockIdx.x;
    const int L = logical_block(A);
    if (L < 0) return;
    const T dt_launch = launch_dt<HAM>(A);
    int chunk_id, rem;
    fdivmod(L, fdiv_make(A.ntiles), chunk_id, rem);     // index divisions through a float reciprocal (hj_device.h)
    if (A.timing && threadIdx.x == 0) {
        A.timing[4 * L + 0] = wall_clock64();
        A.timing[4 * L + 2] = (unsigned long long)(b & 7);
        A.timing[4 * L + 3] = (unsigned long long)chunk_id;
    }
    int org[ND], tc[ND];
    FDiv fE[ND];
    tile_coords<ND>(A, rem, tc);
#pragma unroll
    for (int d = ND - 1; d >= 1; --d) {
        // the last tile on an axis is shifted back so that no tile straddles the domain edge
        // (it recomputes a few cells of its neighbour: identical values, benign duplicate stores)
        org[d] = min(tc[d] * A.E[d], A.n[d] - A.E[d]);
        fE[d] = fdiv_make(A.E[d]);
    }
    // several plane ranges may share a launch (the low and high edge planes of a slab; edges + interior: chunk_planes)
    int p_begin, p_end;
    chunk_planes(A, chunk_id, p_begin, p_end);

    // ---- LDS geometry: halo'd box, last axis contiguous
    // rows of the last axis are A.lpitch apart: E + 6, or E + 32 when LDS allows -- then the jump a
    // wave makes at a row end is a multiple of the bank period and ds_read_b64 / ds_write_b64 of
    // consecutive cells stay conflict free although the tile width is not a multiple of 32
    int ls[ND];
    ls[ND - 1] = 1;
#pragma unroll
    for (int d = ND - 2; d >= 1; --d) ls[d] = (d == ND - 2) ? A.lpitch : ls[d + 1] * (A.E[d + 1] + 2 * HJ_STENCIL);
    const int lds_plane = (ND >= 3) ? ls[1] * (A.E[1] + 2 * HJ_STENCIL) : A.lpitch;
    int tile_cells = 1;
#pragma unroll
    for (int d = 1; d < ND; ++d) tile_cells *= A.E[d];

    const int tid = threadIdx.x;

    // ---- own cells (loop-invariant along the march); surplus threads shadow the last cell
    int own_lds[R];
    unsigned own_g[R];
    typename HAM::Cell hcell[R];
    typename HAM::Raw hraw[R];
    // only the last round of the deal can run past the tile: shadows compute but do not write
    const bool last_real = (tid + (R - 1) * NT) < tile_cells;
    unsigned nbv[R];              // bit d: the forward neighbour on plane axis d lies inside the tile (eps_part pairs)
    unsigned e_lo[R], e_hi[R];    // TERM: bit d: the cell has a lower / upper neighbour INSIDE the grid on plane axis d
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int c = min(tid + r * NT, tile_cells - 1);
        int lo = 0, g = 0;
        int idx[ND];
        idx[0] = 0;
        nbv[r] = 0u;
        e_lo[r] = 0u; e_hi[r] = 0u;
#pragma unroll
        for (int d = ND - 1; d >= 1; --d) {
            int q, j;
            fdivmod(c, fE[d], q, j);
            c = q;
            const int gi = org[d] + j;
            idx[d] = gi;
            lo += (j + HJ_STENCIL) * ls[d];
            g += gi * A.pstride[d];
            if (j + 1 < A.E[d]) nbv[r] |= 1u << d;
            if (TERM && gi > 0) e_lo[r] |= 1u << d;
            if (TERM && gi + 1 < A.n[d]) e_hi[r] |= 1u << d;
        }
        own_lds[r] = lo;
        own_g[r] = (unsigned)g * (unsigned)sizeof(T);   // byte offset within a plane
        // table loads of the per-column Hamiltonian constants go out FIRST (loads return in order: whatever waits
        // for them later waits for nothing else); the arithmetic on them follows the halo setup
        hraw[r] = HAM::cell_raw(A.ham, idx);
    }

    // ---- The loads every workgroup needs before its first plane (7 planes of its own cells, the first
    // prefetch sets, the RK operand) are issued HERE, ahead of the halo-slot index arithmetic and the table
    // loads below: all workgroups of a launch start together, so this burst is not hidden by anybody else's
    // arithmetic -- the setup that follows is what it overlaps with.
    // ---- loaders.  p is wave-uniform and clamped by the callers to planes that exist.  One buffer
    // descriptor per array for the whole chunk (base = 3 planes below the chunk; SGPRs), the plane
    // goes into the scalar offset and the cell into the per-lane 32-bit byte offset; the hardware
    // range check covers the chunk's planes.  Ghost / wrapped planes of axis 0 (first and last chunk
    // only) build their own descriptor.
    const unsigned plane_bytes = (unsigned)(A.stride0 * (long long)sizeof(T));
    const int p_lo = p_begin - HJ_STENCIL;                       // lowest plane the chunk touches
    const unsigned span = (unsigned)(p_end + HJ_STENCIL - p_lo) * plane_bytes;   // host keeps this < 4 GiB
    const __amdgpu_buffer_rsrc_t ry = make_srd(y + (long long)p_lo * A.stride0, span);
    const __amdgpu_buffer_rsrc_t ry0 = make_srd(y0 + (long long)p_lo * A.stride0, use_y0 ? span : 0u);
    const __amdgpu_buffer_rsrc_t rout = make_srd(out + (long long)p_lo * A.stride0, span);
    // TERM: velocity components 1 .. ND-1 of termConvection (component 0 is the y0 stream); a null array = its scalar
    __amdgpu_buffer_rsrc_t rcf[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
        rcf[d] = make_srd((TERM && d >= 1 && A.term.arr[d]) ? A.term.arr[d] + (long long)p_lo * A.stride0 : y, (TERM && d >= 1 && A.term.arr[d]) ? span : 0u);
    auto load_own = [&](int p, T* dst) {
        const bool direct = (p >= 0 || A.halo_lo) && (p < A.n[0] || A.halo_hi);
        if (direct) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load<HJ_AUX_OWN>(ry, own_g[r], so, T());
        } else {
            const PlaneSrc<T> s = plane_src<T, ND>(A, p);
            const __amdgpu_buffer_rsrc_t rb = make_srd(y + s.off, plane_bytes);
            if (!s.ghost) {
#pragma unroll
                for (int r = 0; r < R; ++r) dst[r] = buf_load(rb, own_g[r], 0u, T());
            } else {
                const __amdgpu_buffer_rsrc_t ri = make_srd(y + s.off_in, plane_bytes);
#pragma unroll
                for (int r = 0; r < R; ++r)
                    dst[r] = ghost_value(buf_load(rb, own_g[r], 0u, T()), buf_load(ri, own_g[r], 0u, T()), s.km);
            }
        }
    };
    auto load_y0 = [&](int p, T* dst) {
        if (use_y0) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load<HJ_AUX_Y0>(ry0, own_g[r], so, T());
        }
    };
    const int p_last = p_end - 1;

    // ---- prologue: axis-0 queue q[r][j] <-> plane p-3+j, j = 0..6
    T q[R][7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        T tmp[R];
        load_own(p_begin - 3 + j, tmp);
#pragma unroll
        for (int r = 0; r < R; ++r) q[r][j] = tmp[r];
    }
    // register sets of the pipeline.  PD < 10: one depth for everything; else PD = 100*y0 + 10*halo + own
    // (e.g. 241: own cells 1 plane ahead, halo ring 4, y0 2 -- the halo ring of plane P is then
    // requested in the same iteration in which the neighbouring tiles request P as their own cells,
    // so the second requester finds the lines in L2).  Set (p - p_begin) % depth serves plane p.
    constexpr int PDO = PD < 10 ? PD : PD % 10;
    constexpr int PDH = PD < 10 ? PD : (PD / 10) % 10;
    constexpr int PDY = PD < 10 ? PD : PD / 100;
    T own[PDO][R], y0s[PDY][R];
    typename HAM::Plane pls[PDY];
#pragma unroll
    for (int s = 0; s < PDO; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) own[s][r] = T(0);
        // own[s] holds plane p+4 for the iteration of plane p = p_begin+s; the last set is filled
        // by the first iteration
        if (s < PDO - 1) load_own(min(p_begin + 4 + s, p_end + 2), own[s]);
    }
#pragma unroll
    for (int s = 0; s < PDY; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) y0s[s][r] = T(0);
        const int ps = min(p_begin + s, p_last);
        load_y0(ps, y0s[s]);
        pls[s] = HAM::plane(A.ham, ps, A.sc);
    }

    // ---- halo slots: for each plane axis d, 3 cells below and 3 above the tile, over the
    // tile's extent on the other axes (a "cross": no corners).  Surplus slots shadow slot 0.
    int h_lds[KH], h_dlt[KH];
    unsigned h_src[KH];
    T h_km[KH];
    bool h_real[KH];
    {
        int area[ND], base[ND + 1];
        base[1] = 0;
        FDiv fA[ND];
#pragma unroll
        for (int d = 1; d < ND; ++d) {
            area[d] = 1;
#pragma unroll
            for (int e = 1; e < ND; ++e) if (e != d) area[d] *= A.E[e];
            base[d + 1] = base[d] + 2 * HJ_STENCIL * area[d];
            fA[d] = fdiv_make(area[d]);
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            int h = tid + k * NT;
            h_real[k] = h < base[ND];
            if (!h_real[k]) h = 0;                    // shadow of slot 0: loads it; writes LDS only in 4-D (HJ_SLOT_PRED)
            h_lds[k] = 0; h_src[k] = 0; h_dlt[k] = 0; h_km[k] = T(0);
            // static loop over the axis the slot belongs to (runtime-indexed local arrays
            // would be demoted to scratch)
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                if (h < base[d] || h >= base[d + 1]) continue;
                const int hh = h - base[d];
                int lay, c;                            // lay 0..5: which halo layer; c: index over the other axes
                if (d == ND - 1) {
                    // contiguous axis: the LAYER runs fastest, so that consecutive lanes fetch the 3 + 3 cells either side
                    // of one tile row (two cache lines) instead of one cell from each of 64 rows (64 lines per wave
                    // instruction: round 3 found the halo columns issuing more line requests than the whole tile)
                    c = hh / (2 * HJ_STENCIL);
                    lay = hh - c * (2 * HJ_STENCIL);
                } else {
                    fdivmod(hh, fA[d], lay, c);
                }
                const int jd = (lay < HJ_STENCIL) ? (lay - HJ_STENCIL) : (A.E[d] + lay - HJ_STENCIL);
                int lo = 0, g = 0;
#pragma unroll
                for (int e = ND - 1; e >= 1; --e) {
                    if (e == d) continue;
                    int q, j;
                    fdivmod(c, fE[e], q, j);
                    c = q;
                    lo += (j + HJ_STENCIL) * ls[e];
                    g += (org[e] + j) * A.pstride[e];
                }
                lo += (jd + HJ_STENCIL) * ls[d];
                int gi = org[d] + jd;                  // in [-3, n+2]: tiles never straddle the edge
                const int nd = A.n[d];
                int dlt = 0;
                T km = T(0);
                if (gi < 0) {
                    if (A.bc[d] == HJ_BC_PERIODIC) gi += nd;
                    else { km = T(-gi) * A.km[d]; dlt = A.pstride[d]; gi = 0; }
                } else if (gi >= nd) {
                    if (A.bc[d] == HJ_BC_PERIODIC) gi -= nd;
                    else { km = T(gi - nd + 1) * A.km[d]; dlt = -A.pstride[d]; gi = nd - 1; }
                }
                h_lds[k] = lo;
                h_src[k] = (unsigned)(g + gi * A.pstride[d]) * (unsigned)sizeof(T);
                h_dlt[k] = dlt * (int)sizeof(T);
                h_km[k] = km;
            }
        }
    }

    // does any halo slot of this tile lie outside the domain on an extrapolated axis?  (block-uniform:
    // interior tiles skip the ghost arithmetic and its second load altogether)
    // (wave-uniform, from the tile's position: the halo layers reach HJ_STENCIL cells past both ends of the tile.
    // Until round 2 this was a __syncthreads_or over the slots -- a barrier that also waited for every load above)
    bool tile_ghost = false;
#pragma unroll
    for (int d = 1; d < ND; ++d)
        tile_ghost = tile_ghost || (A.bc[d] != HJ_BC_PERIODIC && (org[d] < HJ_STENCIL || org[d] + A.E[d] + HJ_STENCIL > A.n[d]));

    T eps[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) eps[d] = T(0);
    WenoK<T> wk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { wk[d].c13 = T(0); wk[d].c4 = T(0); }
    if constexpr (SCHEME == HJ_WENO5) {
        if (A.eps_nrows > 0) fold_eps_rows<T, ND, NT>(A.eps_rows, A.eps_nrows, red, eps);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (A.eps_nrows <= 0) eps[d] = T(1e-6) * A.max_d1sq[d] + Lim<T>::tiny;
            if constexpr (TERM) eps[d] = weno_eps_uncontracted<T>(A.max_d1sq[d]);     // as term_kernel forms it (product, then sum)
            wk[d] = weno_consts<T>(eps[d], A.K[d]);
        }
    }
    // max(D1^2) of the output (eps_part): the outputs of a plane are parked in one of two LDS planes behind the two
    // input planes and meet their in-plane forward neighbours one iteration later, behind that iteration's barrier
    const bool eps_prod = SCHEME == HJ_WENO5 && A.eps_part != nullptr;
    T* const obuf = lds + 2 * lds_plane;
    double dmax[ND];
    T oprev[R];
#pragma unroll
    for (int d = 0; d < ND; ++d) dmax[d] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) oprev[r] = T(0);

    // the halo ring of a tile that touches an extrapolated edge needs two loads per slot (edge and
    // inner cell).  Both are only ISSUED here; the ghost arithmetic is done when the slot is
    // consumed, PD planes later -- forming the ghost value at load time would make the wave wait
    // out the full memory latency every plane (measured: 21 % of the 401^3 launch, since the
    // edge tiles then set the duration of the single round of workgroups).
    auto load_halo = [&](int p, T* dst, T* dst_in) {
        const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
        for (int k = 0; k < KH; ++k) dst[k] = buf_load(ry, h_src[k], so, T());
        if (tile_ghost) {
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                // only the slots that really are ghost cells fetch their inner neighbour (exec-masked
                // load): the other lanes would just re-request the line of their first load
                dst_in[k] = T(0);
                if (h_dlt[k] != 0) dst_in[k] = buf_load(ry, h_src[k] + (unsigned)h_dlt[k], so, T());
            }
        }
    };

    T hal[PDH][KH], hin[PDH][KH];
#pragma unroll
    for (int s = 0; s < PDH; ++s) {
#pragma unroll
        for (int k = 0; k < KH; ++k) { hal[s][k] = T(0); hin[s][k] = T(0); }
        load_halo(min(p_begin + s, p_last), hal[s], hin[s]);
    }

    // the arithmetic on the Hamiltonian tables, now that every load of the setup is in flight
#pragma unroll
    for (int r = 0; r < R; ++r) hcell[r] = HAM::cell_fin(A.ham, hraw[r], A.sc);
    double amax[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) amax[d] = -1.0e300;
    // LOCAL Lax-Friedrichs with a range-reading Hamiltonian, and the bound pass (see hj_fusedv.h)
    constexpr bool RR = ham_reads_range<HAM>::value;
    bool local_lf = false;
    if constexpr (RR) local_lf = A.ham.local_mode != 0;
    const bool BPASS = RNG && local_lf && A.bound != nullptr;
    T lw[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        lw[d] = local_lf ? (A.inv_dx[d] / A.inv_dx[0]) * (A.sc[0] / A.sc[d]) : T(0);
        if (local_lf && d > 0) amax[d] = 0.0;
    }
    auto acc_alpha = [&](const T* alpha) {
        if (RR && local_lf) {
            T ssum = T(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) ssum += alpha[d] * lw[d];
            amax[0] = max_acc(amax[0], (double)ssum);
        } else {
#pragma unroll
            for (int d = 0; d < ND; ++d)
                if ((HAM::PLANE_DEP >> d) & 1u) amax[d] = max_acc(amax[d], (double)alpha[d]);
        }
    };
    T rmn[ND], rmx[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { rmn[d] = -Lim<T>::lowest; rmx[d] = Lim<T>::lowest; }

    // one plane.  own_c holds plane p+4 (joins the queue at the end); own_n is refilled with plane
    // p+3+PD; hal_c / y0_c / pl_c hold plane p's halo ring, RK operand and Hamiltonian scalars and
    // are refilled for plane p+PD once consumed.
#ifdef HJ_STAMP
    // diagnostic build: shader-clock time of the phases of a plane iteration, summed per wave
    unsigned long long st_acc[4] = {0, 0, 0, 0};
#define HJ_ST(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define HJ_ST(var)
#endif
    auto body = [&](int p, T* own_c, T* own_n, T* hal_c, T* hin_c, T* y0_c, typename HAM::Plane& pl_c) {
        HJ_ST(st0);
        T* buf = lds + ((p - p_begin) & 1) * lds_plane;
        load_own(min(p + 3 + PDO, p_end + 2), own_n);
        // TERM: what term_cell needs besides the stencils, requested here and consumed behind the barrier and the stencil
        // arithmetic: the further velocity components (termConvection), the six neighbours of the INITIAL array (sub-cell fix)
        T tcf[R][ND], tlo[R][ND], thi[R][ND];
        if constexpr (TERM) {
            const unsigned so_t = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int d = 0; d < ND; ++d) { tcf[r][d] = A.term.scal[d]; tlo[r][d] = T(0); thi[r][d] = T(0); }
                if constexpr (TKIND == HJ_TERM_CONVECTION) {
#pragma unroll
                    for (int d = 1; d < ND; ++d)
                        if (A.term.arr[d]) tcf[r][d] = buf_load(rcf[d], own_g[r], so_t, T());
                }
                if constexpr (TKIND == HJ_TERM_REINIT) {
                    if (A.term.subcell_order == 1) {
                        // a missing neighbour re-reads the cell itself (term_cell does not look at it)
                        tlo[r][0] = buf_load(ry0, own_g[r], p > 0 ? so_t - plane_bytes : so_t, T());
                        thi[r][0] = buf_load(ry0, own_g[r], p + 1 < A.n[0] ? so_t + plane_bytes : so_t, T());
#pragma unroll
                        for (int d = 1; d < ND; ++d) {
                            const unsigned sb = (unsigned)A.pstride[d] * (unsigned)sizeof(T);
                            tlo[r][d] = buf_load(ry0, ((e_lo[r] >> d) & 1u) ? own_g[r] - sb : own_g[r], so_t, T());
                            thi[r][d] = buf_load(ry0, ((e_hi[r] >> d) & 1u) ? own_g[r] + sb : own_g[r], so_t, T());
                        }
                    }
                }
            }
        }
        // stage the centre plane
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < R - 1 || last_real) buf[own_lds[r]] = q[r][3];
        if (tile_ghost) {
            // h_km = 0 for in-domain slots: edge + 0*slope = edge
#pragma unroll
            for (int k = 0; k < KH; ++k)
                HJ_SLOT_PRED(k) buf[h_lds[k]] = ghost_value(hal_c[k], hin_c[k], h_km[k]);
        } else {
#pragma unroll
            for (int k = 0; k < KH; ++k)
                HJ_SLOT_PRED(k) buf[h_lds[k]] = hal_c[k];
        }
        HJ_ST(st1);
        __syncthreads();
        HJ_ST(st2);
        const int p2 = min(p + PDY, p_last);
        load_halo(min(p + PDH, p_last), hal_c, hin_c);
        const unsigned so_out = (unsigned)(p - p_lo) * plane_bytes;
        const typename HAM::Plane pl_use = pl_c;
        pl_c = HAM::plane(A.ham, p2, A.sc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T pc[ND], hd[ND];
#if defined(HJ_ABLATE) && (HJ_ABLATE & 1)
            // timing experiment: keep every operand alive, skip the arithmetic
            pc[0] = q[r][0]; hd[0] = q[r][6];
#pragma unroll
            for (int j = 1; j < 6; ++j) asm volatile("" ::"v"(q[r][j]));
#else
            if constexpr (TERM) upwind<SCHEME, T>(q[r], A.K[0], eps[0], pc[0], hd[0]);          // pc = derivL, hd = derivR
            else upwind_cd<SCHEME, T>(q[r], A.K[0], eps[0], wk[0], pc[0], hd[0]);
#endif
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                T v[7];
                const T* c = buf + own_lds[r];
#if defined(HJ_ABLATE) && (HJ_ABLATE & 2)
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = q[r][j];
#else
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int o = own_lds[r] + (j - 3) * ls[d];
                    // only the contiguous axis yields constant offsets from one base (what the back end pairs)
                    v[j] = (j == 3) ? q[r][3] : (d == ND - 1 ? lds_read(buf, o) : buf[o]);
                }
#endif
#if defined(HJ_ABLATE) && (HJ_ABLATE & 1)
                pc[d] = v[0]; hd[d] = v[6];
#pragma unroll
                for (int j = 1; j < 6; ++j) asm volatile("" ::"v"(v[j]));
#else
                if constexpr (TERM) upwind<SCHEME, T>(v, A.K[d], eps[d], pc[d], hd[d]);
                else upwind_cd<SCHEME, T>(v, A.K[d], eps[d], wk[d], pc[d], hd[d]);
#endif
            }
            if constexpr (RNG) {
                if (r < R - 1 || last_real) {
                    if (!BPASS) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) range_acc(rmn[d], rmx[d], pc[d], hd[d]);      // derivL = sc (pc - hd), derivR = sc (pc + hd); scaled in publish_range
                    }
                    if constexpr (RR && !TERM) {
                        if (BPASS) {
                            T Hb, ab[ND];
                            lf_eval_local<NP, HAM>(A.ham, hcell[r], pl_use, A.sc, pc, hd, Hb, ab);
                            acc_alpha(ab);
                        }
                    }
                }
                continue;
            }
            T ydot;
            if constexpr (TERM) {
                tcf[r][0] = use_y0 ? y0_c[r] : A.term.scal[0];
                double mm[ND + 1];
#pragma unroll
                for (int d = 0; d <= ND; ++d) mm[d] = -1e300;
                const unsigned hl = e_lo[r] | (p > 0 ? 1u : 0u), hh = e_hi[r] | (p + 1 < A.n[0] ? 1u : 0u);
                ydot = term_cell<TKIND, T, ND>(A.term, pc, hd, tcf[r], q[r][3], tlo[r], thi[r], hl, hh, mm);
                // termNormal's single maximum travels in slot 0 (the host reads key 0 for it)
                if constexpr (TKIND == HJ_TERM_NORMAL) amax[0] = fmax(amax[0], mm[ND]);
                else {
#pragma unroll
                    for (int d = 0; d < ND; ++d) amax[d] = fmax(amax[d], mm[d]);
                }
            } else {
                T alpha[ND];
                ydot = lf_ydot<NP, HAM>(A.ham, hcell[r], pl_use, A.sc, pc, hd, alpha);
                acc_alpha(alpha);
            }
            // termRestrictUpdate clamp; written so that a NaN stays a NaN
            if (GEN && A.do_clamp) {
                ydot = (ydot < A.clamp_lo) ? A.clamp_lo : ydot;
                ydot = (ydot > A.clamp_hi) ? A.clamp_hi : ydot;
            }
            T o;
            if (GEN && A.ydot_only) o = ydot;
            else {
                o = rk_stage_out<NP>(A.stage, A.ca, A.cb, dt_launch, y0_c[r], q[r][3], ydot);
                // the step started from y0 (stages that read it) or from y itself (Euler step)
                if (GEN && A.post_op) o = post_step(A.post_op, o, use_y0 ? y0_c[r] : q[r][3]);
            }
#if defined(HJ_ABLATE) && (HJ_ABLATE & 4)
            asm volatile("" ::"v"(o));
#else
            if (r < R - 1 || last_real) buf_store<HJ_AUX_ST>(o, rout, own_g[r], so_out);
#endif
            if constexpr (SCHEME == HJ_WENO5) {
                if (eps_prod) {
                    T* ob = obuf + ((p - p_begin) & 1) * lds_plane;
                    if (p > p_begin) {
                        dmax[0] = fmax(dmax[0], (double)t_abs(o - oprev[r]));
                        const T* op = obuf + ((p - p_begin - 1) & 1) * lds_plane + own_lds[r];     // plane p - 1
#pragma unroll
                        for (int d = 1; d < ND; ++d)
                            if ((nbv[r] >> d) & 1u) dmax[d] = fmax(dmax[d], (double)t_abs(op[ls[d]] - oprev[r]));
                    }
                    ob[own_lds[r]] = o;
                    oprev[r] = o;
                }
            }
        }
        HJ_ST(st3);
        load_y0(p2, y0_c);
        // rotate the queue: own_c was loaded two iterations ago
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 6; ++j) q[r][j] = q[r][j + 1];
            q[r][6] = own_c[r];
        }
#ifdef HJ_STAMP
        {
            const unsigned long long st4 = __builtin_readcyclecounter();
            st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3;
        }
#endif
    };

    constexpr int L1 = PDO * PDH / (PDO % PDH == 0 ? PDH : (PDH % PDO == 0 ? PDO : 1));   // lcm for 1..4
    constexpr int UNR = L1 * PDY / (L1 % PDY == 0 ? PDY : (PDY % L1 == 0 ? L1 : 1));
    static_assert(UNR <= 6 && UNR % PDO == 0 && UNR % PDH == 0 && UNR % PDY == 0, "unsupported depth mix");
#define HJ_BODY(u)                                                                                   \
    if constexpr (UNR > (u)) {                                                                       \
        if (p + (u) < p_end)                                                                         \
            body(p + (u), own[(u) % PDO], own[((u) + PDO - 1) % PDO], hal[(u) % PDH], hin[(u) % PDH],  \
                 y0s[(u) % PDY], pls[(u) % PDY]);                                                                    \
    }
#ifdef HJ_STAMP
    const unsigned long long st_loop0 = wall_clock64(), st_cyc0 = __builtin_readcyclecounter();
#endif
    for (int p = p_begin; p < p_end; p += UNR) {
        HJ_BODY(0) HJ_BODY(1) HJ_BODY(2) HJ_BODY(3) HJ_BODY(4) HJ_BODY(5)
    }
#undef HJ_BODY
#ifdef HJ_STAMP
    const unsigned long long st_loop1 = wall_clock64(), st_cyc1 = __builtin_readcyclecounter();
#endif

    if constexpr (RNG) {
        if (!BPASS) {
            publish_range<ND, NT>(A.range_keys, red, rmn, rmx, A.sc);
            return;
        }
    }
    if constexpr (SCHEME == HJ_WENO5) {
        if (eps_prod) {
            __syncthreads();
            if (p_end > p_begin) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const T* op = obuf + ((p_end - 1 - p_begin) & 1) * lds_plane + own_lds[r];
#pragma unroll
                    for (int d = 1; d < ND; ++d)
                        if ((nbv[r] >> d) & 1u) dmax[d] = fmax(dmax[d], (double)t_abs(op[ls[d]] - oprev[r]));
                }
            }
            store_eps_part<ND, NT>(A.eps_part + (size_t)L * HJ_MAX_DIM, red, dmax);
        }
    }
    if (A.bound) {   // a launch whose bound nobody reads (hj_rk_step: dt comes from the static bound) skips the reduction
        if constexpr (!TERM) {   // alphas that are constant along the march: one max per column, taken once (any plane does)
            T pz[ND], Hz, az[ND];
    #pragma unroll
            for (int d = 0; d < ND; ++d) pz[d] = T(0);
    #pragma unroll
            for (int r = 0; r < R; ++r) {
                HAM::eval(A.ham, hcell[r], pls[0], A.sc, pz, Hz, az);
    #pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (!((HAM::PLANE_DEP >> d) & 1u)) amax[d] = fmax(amax[d], (double)az[d]);
            }
        }

        // ---- CFL reduction: wavefront shuffles -> LDS -> one atomicMax per block and dim
        const int lane = tid & 63, wv = tid >> 6;
    #pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double m = wave_max(amax[d]) / (double)A.sc[d];   // alpha_s = sc*alpha
            if (lane == 0) red[wv][d] = m;
        }
        __syncthreads();
        if (tid < ND) {
            double m = red[0][tid];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w][tid]);
            if (m > -1.0e299) key_max(A.bound + tid, m);
        }
    }
    publish_gate(A, chunk_id);
    if (A.timing && tid == 0) A.timing[4 * L + 1] = wall_clock64();
#ifdef HJ_STAMP
    // phases of wave 0 and of the last wave, after the 4*nblocks words of the start/end records
    if (A.timing && (tid == 0 || tid == NT - 64)) {
        unsigned long long* dst = A.timing + 4 * (size_t)A.nblocks + 8 * (size_t)L + (tid == 0 ? 0 : 4);
        for (int k = 0; k < 4; ++k) dst[k] = st_acc[k];
        // the last wave's record carries the loop's wall-clock window and its shader-cycle length instead of
        // phases C, D: (loop start, loop end) on the 100 MHz clock, cycles
        if (tid != 0) { dst[1] = st_loop0; dst[2] = st_loop1; dst[3] = st_cyc1 - st_cyc0; }
    }
#endif
}

}  // namespace hj
