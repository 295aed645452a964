// fused_substep_kernel with TWO ADJACENT CELLS PER LANE along the contiguous (last) grid axis (gfx950).
//
// Same algorithm, same per-cell arithmetic (so the results are bitwise those of fused_substep_kernel), other
// thread <-> cell map: a "slot" is a pair of cells (j, j+1) of one tile row with j even in tile coordinates, and
// a thread owns R slots dealt round-robin like the single cells of the scalar kernel.  What it buys (round-2
// counters, DESIGN.md 4.2: both kernels are bound by memory / LDS instruction issue per cell, not by bytes):
//   * HBM: one 16-byte buffer_load / buffer_store per pair instead of two 8-byte ones (own cells, y0, output);
//   * LDS staging: one ds_write_b128 per pair;
//   * LDS stencil reads of a pair: the row-direction neighbours j-3 .. j+4 are one ds_read_b64 + two ds_read_b128
//     + one ds_read_b64 (8 distinct values; the scalar kernel reads 12, as 6 ds_read2_b64 at half rate), the
//     other plane axis is 6 ds_read_b128 (aligned: the LDS row starts on an even cell, left pad 4 instead of 3, and
//     the row pitch is even): 36 LDS cycles per pair instead of 72, 10 LDS instructions instead of 18-24;
//   * half the address arithmetic.
// Halo slots stay single cells (the halo columns are 3 cells wide).  The tile extent along the last axis must
// be even; the last tile of an axis is shifted back inside the domain as before, so a global pair may start on
// an odd cell (8-byte aligned 16-byte buffer access: legal, one more cache line per wave instruction).
#pragma once
#include "hj_fused.h"

namespace hj {

// HJ_ST_UNSAFE (no wait states after the 16-byte store), HJ_ABLATE / HJ_ABLATE_NOSYNC (parts of the kernel removed) exist
// to price those parts; the results are WRONG with them, so only a tuning build may define them
#if (defined(HJ_ST_UNSAFE) || defined(HJ_ABLATE_NOSYNC) || defined(HJ_ABLATE)) && !defined(HJ_TUNE_BUILD)
#error "HJ_ST_UNSAFE / HJ_ABLATE* give wrong results: tuning builds (-DHJ_TUNE_BUILD) only"
#endif

template <typename T> struct Pair;
template <> struct Pair<double> { typedef double V __attribute__((ext_vector_type(2))); };
template <> struct Pair<float> { typedef float V __attribute__((ext_vector_type(2))); };

// cache-policy experiments (gfx940+ aux encoding: 1 = sc0, 2 = nt, 16 = sc1): -DHJ_PAIR_AUX_Y0=2 / -DHJ_PAIR_AUX_OUT=2
#ifndef HJ_PAIR_AUX_Y0
#define HJ_PAIR_AUX_Y0 0
#endif
#ifndef HJ_PAIR_AUX_OUT
#define HJ_PAIR_AUX_OUT 0
#endif
template <int AUX = 0>
__device__ __forceinline__ Pair<double>::V buf_load2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff, double) {
    return __builtin_bit_cast(Pair<double>::V, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ Pair<float>::V buf_load2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff, float) {
    return __builtin_bit_cast(Pair<float>::V, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, AUX));
}
// HAZARD (found in round 2, gfx950 / ROCm 7.2): a buffer store with MORE than 8 bytes of data reads its data
// VGPRs late; a VALU instruction that overwrites them within two wait states corrupts the store (one wait state
// was seen to corrupt as well).  hipcc's hazard recogniser inserts wait states only when the store has NO SGPR in
// its soffset field (it assumes the SGPR form is safe); here the plane offset IS in soffset, and the next slot's
// arithmetic did overwrite the data registers in the very next instruction: racy, wrong values (an intermediate
// of the next cell's stencil) in a few lanes of a plane.  A bare `asm volatile("s_nop")` after the builtin is not
// enough - the scheduler moves VALU work between the two.  Default form: the compiler's store (so it stays in the
// vmcnt accounting), followed by a nop statement that takes the data as an INPUT and clobbers memory: the clobber
// keeps it after the store, the input keeps the data registers live (unwritable) up to the nop.
// tools/check_store_hazard.py disassembles the built library and verifies the rule for every wide store
// (tests/test_cabi.py runs it).  -DHJ_ST_ASM: store + wait states as ONE asm statement (2-3 % slower: the
// compiler no longer counts the store in vmcnt; it also needs the leading s_nop 4, because the hazard recogniser
// does not look inside asm and the SRD / soffset may just have been restored from a spill lane by v_readlane).
// 1: the intended WENO5 shares the MIDDLE axis' smoothness values between lanes through LDS (WX below).  Built and measured in round 5 and
// left OFF: the values are the same bits and 18 fp64 operations per cell go, but the instantiation was at 252 VGPRs -- what has to live
// across the extra barrier spills (36-68 B of scratch) and the launch is 19 % SLOWER (201^3: 6.23 against 7.70e10; profiles/r05_weno5_eno_fast.txt)
#ifndef HJ_WENO_LDS_SHARE
#define HJ_WENO_LDS_SHARE 0
#endif
#ifndef HJ_ST_PRE
#define HJ_ST_PRE 4
#endif
#ifndef HJ_ST_POST
#define HJ_ST_POST 2
#endif
#define HJ_STR2(x) #x
#define HJ_STR(x) HJ_STR2(x)
__device__ __forceinline__ void buf_store2(Pair<double>::V v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
#if defined(HJ_ST_ASM)
    asm volatile("s_nop " HJ_STR(HJ_ST_PRE) "\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop " HJ_STR(HJ_ST_POST)
                 :: "v"(v), "v"(off), "s"(r), "s"(soff) : "memory");
#else
    using W = decltype(__builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(W, v), r, off, soff, HJ_PAIR_AUX_OUT);
#if !defined(HJ_ST_UNSAFE)      // HJ_ST_UNSAFE: tuning only (what the wait states cost); results can be wrong
    asm volatile("s_nop " HJ_STR(HJ_ST_POST) :: "v"(v) : "memory");
#endif
#endif
}
__device__ __forceinline__ void buf_store2(Pair<float>::V v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    using W = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(W, v), r, off, soff, 0);
}

// round 4: every load of the setup is issued before the first wait (the rings of the first AH planes used to be requested and
// waited for one after the other: two serialised memory round trips in front of the plane loop), and HJ_EARLY_ARGS batches
// the kernel-argument loads (profiles/r04_headline_skeleton.txt)
#ifndef HJ_EARLY_ARGS
#define HJ_EARLY_ARGS 1
#endif
// TWO PLANES PER BARRIER (round 6 experiment): the two plane iterations of a loop pass stage their centre planes and halo rings first, meet at ONE
// barrier, then compute both -- half the synchronisations of the per-plane chain barrier -> LDS reads -> arithmetic -> store (profiles/r06_stage1_bound.txt).
// Needs 4 + halo_ahead LDS plane buffers instead of 2 + halo_ahead (the host side sizes them: hj_inst.hip).  Light stencils only (the intended
// WENO5's epsilon producer reads the previous plane's outputs behind the plane's own barrier).
#ifndef HJ_TWO_PLANES
#define HJ_TWO_PLANES 0
#endif
// VERTICAL PAIRS (round 6): on 3-D grids the TWO pair slots of a thread (R = 2) are the pairs of the same columns in two ADJACENT tile rows, not two
// slots dealt half a tile apart.  The middle-axis stencils of the two then overlap in all but two rows, and each one's missing neighbour row is the
// other's own pair (in registers): 6 ds_read_b128 per thread and plane for that axis instead of 12 -- the LDS stencil reads are what the stage-1
// launch of the headline cannot hide (13.6 % at 513^3, 16 % at 201^3: profiles/r06_stage1_bound.txt).  Same cells, same arithmetic: same bits.
// The host gives such launches an EVEN tile extent on axis 1 (make_tiling, hj_api.hip).
// MEASURED SLOWER (-3 %, same-run A/B at 201^3 and 513^3) and compiled out: HJ_VPAIR defaults to 0 (hj_device.h).

template <int K> struct IntTag { static constexpr int value = K; };
constexpr bool defined_ablate2() {
#if defined(HJ_ABLATE) && (HJ_ABLATE & 2)
    return true;
#else
    return false;
#endif
}
constexpr bool light_scheme_dev(int s) { return s == HJ_WENO5_ASSHIPPED || s == HJ_ENO2 || s == HJ_ENO2_FAST; }
constexpr int HJ_VPAD = 4;      // left pad of an LDS row (cells): even, so that tile cell 0 of a row is 16-byte aligned

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, int MODE = 0>
__global__ __launch_bounds__(NT, OCC) void fused_pair_kernel(const T* __restrict__ y, const T* __restrict__ y0,
                                                             T* __restrict__ out, const FusedArgs<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr int LA = ND - 1;                  // the contiguous axis
    constexpr int PD = 2;
    constexpr bool GEN = (MODE == 0);
    constexpr bool RNG = (MODE == 3);           // range pass (FusedArgs::range_keys): derivL / derivR reduced to minima / maxima, nothing stored
    constexpr bool NP = np_order(SCHEME);       // ENO2 / ENO3: every operation rounded as NumPy rounds it (hj_device.h)
    // TRANSPOSED MARCH (round 6; hj_device.h, ham_xp): the march axis is the grid's axis 1 and tile axis 1 its axis 0 -- the slab axis -- of which
    // the launch computes the window [A.xwin0, A.xwin1); FusedArgs is filled in kernel order, byte offsets along tile axis 1 are relative to A.xbase
    constexpr bool XP = ham_xp<HAM>::value;
    static_assert(!XP || (ND == 3 && MODE != 3), "transposed march: 3-D substeps");
    using V = typename Pair<T>::V;
    const bool use_y0 = GEN ? (A.use_y0 != 0) : (MODE == 2);
    extern __shared__ __align__(16) unsigned char hj_smem[];
    double (*red)[ND] = reinterpret_cast<double (*)[ND]>(hj_smem);
    T* lds = reinterpret_cast<T*>(hj_smem + 512);
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");
    const int xb = XP ? A.xbase : 0;            // index of tile axis 1 that byte offset 0 stands for

#if HJ_EARLY_ARGS
    // Round 4: the setup below reads ~45 fields of the 700-byte kernel-argument block.  The compiler places each scalar load next to
    // its first use, so the prologue was a chain of 6-8 dependent s_load -> s_waitcnt round trips before the first
    // buffer_load went out (1.8 us after the workgroup started: profiles/r04_prologue.txt).  Naming the fields as inputs of
    // one empty asm statement makes them all live HERE: the loads are issued back to back and waited for once.
    asm volatile("" ::"s"(A.nblocks), "s"(A.ntiles), "s"(A.blocks_per_xcd), "s"(A.ntile[1]), "s"(A.ntile[ND - 1]), "s"(A.E[1]),
                 "s"(A.E[ND - 1]), "s"(A.n[0]), "s"(A.n[1]), "s"(A.n[ND - 1]), "s"(A.nchunks1), "s"(A.plane_begin), "s"(A.plane_end),
                 "s"(A.plane_begin2), "s"(A.plane_end2), "s"(A.chunk), "s"(A.lpitch), "s"(A.pstride[1]), "s"(A.pstride[ND - 1]),
                 "s"(A.stride0), "s"(A.halo_lo), "s"(A.halo_hi), "s"(A.lds_nbuf), "s"(A.halo_ahead), "s"(A.timing), "s"(A.edge_blocks), "s"(A.edge_count),
                 "s"(A.nchunks_e));
    asm volatile("" ::"s"(y), "s"(y0), "s"(out), "s"(A.ham.coord[0]), "s"(A.ham.coord[1]), "s"(A.ham.coord[ND - 1]), "s"(A.ham.aux[0]),
                 "s"(A.ham.aux[1]), "s"(A.bc[1]), "s"(A.bc[ND - 1]), "s"(A.use_y0));
#endif
    const int L = logical_block(A);
    if (L < 0) return;
    const T dt_launch = launch_dt<HAM>(A);
    // PAIRED CHUNKS (round 4, A.npairs > 0): the main range's chunks 2k and 2k+1 share their boundary plane B; chunk 2k marches
    // DOWN from B - 1, chunk 2k+1 UP from B, and the two workgroups of a pair are neighbours in the logical order (same XCD):
    // both then request the six planes B-3 .. B+2 that fill their register queues in the same microsecond, and the second
    // request of a line is served by L2 / merges with the miss in flight.  Every chunk start costs 6 planes of loads beyond
    // the chunk (1.35x the stencil source at 201^3); an ablation in which 3 of them hit the cache runs the 201^3 launch
    // 4 % faster (profiles/r04_shared_warmup_ablation.txt).  Same values: a cell's result does not depend on the direction.
    int chunk_id, rem;
    bool down = false;
    {
        const int paired = HJ_MAYDOWN ? 2 * A.npairs * A.ntiles : 0;      // (0: `down` is provably false and the loop is the lean one)
        if (L < paired) {
            int pr, r2;
            fdivmod(L, fdiv_make(2 * A.ntiles), pr, r2);
            rem = r2 >> 1;
            chunk_id = 2 * pr + (r2 & 1);
            down = (r2 & 1) == 0;
        } else {
            fdivmod(L, fdiv_make(A.ntiles), chunk_id, rem);
        }
    }
    if (A.timing && threadIdx.x == 0) {         // HJ_TIMING_DUMP: start clock, XCC the hardware put us on, chunk, HW_ID
        A.timing[4 * L + 0] = wall_clock64();
        A.timing[4 * L + 2] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);
        A.timing[4 * L + 3] = (unsigned long long)chunk_id;
        A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    int org[ND], tc[ND];
    FDiv fE[ND];
    tile_coords<ND>(A, rem, tc);
#pragma unroll
    for (int d = ND - 1; d >= 1; --d) {
        org[d] = min(tc[d] * A.E[d], A.n[d] - A.E[d]);
        if (XP && d == 1) org[d] = A.xwin0 + min(tc[d] * A.E[d], (A.xwin1 - A.xwin0) - A.E[d]);      // tiles of the window, the last one shifted back inside it
        fE[d] = fdiv_make(A.E[d]);
    }
    int p_begin, p_end;
    chunk_planes(A, chunk_id, p_begin, p_end);
    // march position m = 0, 1, ... <-> plane P(m) = p_begin + m (up) or p_end - 1 - m (down)
    const int p_first = down ? p_end - 1 : p_begin, dirn = down ? -1 : 1;
    auto plane_at = [&](int m) { return p_first + dirn * m; };
    auto clamp_q = [&](int p) { return min(max(p, p_begin - HJ_STENCIL), p_end + HJ_STENCIL - 1); };   // planes a queue may hold
    auto clamp_c = [&](int p) { return min(max(p, p_begin), p_end - 1); };                            // planes that are computed

    // ---- LDS geometry: rows of the last axis are A.lpitch (even) apart, cell j of a row sits at j + HJ_VPAD;
    // the other plane axes keep the 3-cell pad
    int ls[ND];
    ls[LA] = 1;
#pragma unroll
    for (int d = ND - 2; d >= 1; --d) ls[d] = (d == ND - 2) ? A.lpitch : ls[d + 1] * (A.E[d + 1] + 2 * HJ_STENCIL);
    const int lds_plane = (ND >= 3) ? ls[1] * (A.E[1] + 2 * HJ_STENCIL) : A.lpitch;
    auto pad_of = [](int d) { return d == LA ? HJ_VPAD : HJ_STENCIL; };
    int tile_cells = 1;
#pragma unroll
    for (int d = 1; d < ND; ++d) tile_cells *= A.E[d];
    const int tile_slots = tile_cells >> 1;

    const int tid = threadIdx.x;

    // ---- own slots (pairs); surplus threads shadow the last slot
    int own_lds[R];
    unsigned own_g[R];
    typename HAM::Cell hcell[R][2];
    typename HAM::Raw hraw[R][2];
    constexpr bool VPAIR = HJ_VPAIR && ND == 3 && R == 2 && !HJ_WENO_LDS_SHARE && !defined_ablate2();
    constexpr bool vp_on = VPAIR;                                         // (the host hands such launches an EVEN row count: make_tiling, hj_api.hip)
    const int vp_hp = A.E[LA] >> 1, vp_half = (A.E[1] >> 1) * vp_hp;      // pairs of a tile row; threads that hold two real slots
    const bool last_real = vp_on ? (tid < vp_half) : ((tid + (R - 1) * NT) < tile_slots);
    unsigned nbv[R];              // bit d: the pair's forward neighbour on plane axis d lies inside the tile (eps_part pairs)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int c = 2 * min(tid + r * NT, tile_slots - 1);
        if constexpr (vp_on) {    // slot r of thread t: row 2 * (t / hp) + r, pair t % hp  (surplus threads shadow the last thread that holds real slots)
            const int tt = min(tid, vp_half - 1);
            int rp, col;
            fdivmod(tt, fdiv_make(vp_hp), rp, col);
            c = 2 * ((2 * rp + r) * vp_hp + col);
        }
        int lo = 0, g = 0;
        int idx[ND];
        idx[0] = 0;
        nbv[r] = 0u;
#pragma unroll
        for (int d = ND - 1; d >= 1; --d) {
            int qd, j;
            fdivmod(c, fE[d], qd, j);
            c = qd;
            const int gi = org[d] + j;
            idx[d] = gi;
            lo += (j + pad_of(d)) * ls[d];
            g += (d == 1 ? gi - xb : gi) * A.pstride[d];
            if (j + (d == LA ? 2 : 1) < A.E[d]) nbv[r] |= 1u << d;
        }
        own_lds[r] = lo;
        own_g[r] = (unsigned)g * (unsigned)sizeof(T);
        // table loads of the per-column Hamiltonian constants go out FIRST (loads return in order: whatever waits
        // for them later waits for nothing else); the arithmetic on them follows the halo setup
        hraw[r][0] = HAM::cell_raw(A.ham, idx);
        idx[LA] += 1;
        hraw[r][1] = HAM::cell_raw_next(A.ham, idx, hraw[r][0]);
    }

    // ---- loaders of the own pairs, issued ahead of the rest of the setup (as in the scalar kernel)
    const unsigned plane_bytes = (unsigned)(A.stride0 * (long long)sizeof(T));
    const int p_lo = p_begin - HJ_STENCIL;
    // (transposed: a "plane" of the march is one ROW per index of tile axis 1, pstride[1] elements apart: the descriptors start at index xb of
    //  that axis and reach over the rows the window and its halo can touch)
    const long long xoff = XP ? (long long)xb * A.pstride[1] : 0ll;
    const unsigned xspan = XP ? A.xspan : 0u;
    const unsigned span = xspan + (unsigned)(p_end + HJ_STENCIL - p_lo) * plane_bytes;
    const __amdgpu_buffer_rsrc_t ry = make_srd(y + xoff + (long long)p_lo * A.stride0, span);
    const __amdgpu_buffer_rsrc_t ry0 = make_srd(y0 + xoff + (long long)p_lo * A.stride0, use_y0 ? span : 0u);
    const __amdgpu_buffer_rsrc_t rout = make_srd(out + xoff + (long long)p_lo * A.stride0, span);
    auto load_own = [&](int p, V* dst) {
        const bool direct = (p >= 0 || A.halo_lo) && (p < A.n[0] || A.halo_hi);
        if (direct) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2(ry, own_g[r], so, T());
        } else {
            const PlaneSrc<T> s = plane_src<T, ND>(A, p);
            const __amdgpu_buffer_rsrc_t rb = make_srd(y + xoff + s.off, xspan + plane_bytes);
            if (!s.ghost) {
#pragma unroll
                for (int r = 0; r < R; ++r) dst[r] = buf_load2(rb, own_g[r], 0u, T());
            } else {
                const __amdgpu_buffer_rsrc_t ri = make_srd(y + xoff + s.off_in, xspan + plane_bytes);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const V e = buf_load2(rb, own_g[r], 0u, T()), in = buf_load2(ri, own_g[r], 0u, T());
                    V gv;
                    gv.x = ghost_value<T>(e.x, in.x, s.km);
                    gv.y = ghost_value<T>(e.y, in.y, s.km);
                    dst[r] = gv;
                }
            }
        }
    };
    auto load_y0 = [&](int p, V* dst) {
        if (use_y0) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2<HJ_PAIR_AUX_Y0>(ry0, own_g[r], so, T());
        }
    };

    // axis-0 queue q[r][c][j] <-> plane P(m - 3 + j) of cell c of slot r; filled in ASCENDING plane order whatever the
    // direction (the two workgroups of a pair then ask for their common planes at the same moment)
    // (8 deep: the two plane iterations of a loop pass read windows [0, 7) and [1, 8) of it and the queue is shifted by TWO places
    //  once per pass -- 6 register moves per cell and pass instead of 12; round 4, late: the 201^3 launch is bound by instruction issue)
    T q[R][2][8];
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
        V tmp[R];
        load_own(p_begin - 3 + jj + (down ? (p_end - 1 - p_begin) : 0), tmp);      // up: P(jj - 3); down: P(3 - jj)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (down) { q[r][0][6 - jj] = tmp[r].x; q[r][1][6 - jj] = tmp[r].y; }
            else { q[r][0][jj] = tmp[r].x; q[r][1][jj] = tmp[r].y; }
        }
    }
    if (A.timing && threadIdx.x == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 0] = wall_clock64();
    if (A.timing && threadIdx.x == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 1] = wall_clock64();
    // ---- halo slots (single cells): the cross around the tile, as in the scalar kernel.
    // 4-D (HP, round 3): the halo layers of the plane axes OTHER than the contiguous one are rows of the tile's own row structure,
    // so they are dealt as PAIRS (8 / 16-byte loads and LDS stores, KP = KH - 1 slots per thread); only the 3 + 3 cells either
    // side of a row stay single cells (KS = 1 slot per thread).  Halves the halo's load / store instructions and slot registers:
    // the halo loads were 22 % of the C5 launch (tools/experiments/r03_run60.sh).
    constexpr bool HP = (ND == 4);
    constexpr int KS = HP ? 1 : KH;               // single-cell slots per thread
    constexpr int KP = HP ? KH - 1 : 1;           // pair slots per thread (unused unless HP)
    int h_lds[KS], h_dlt[KS];
    unsigned h_src[KS];
    T h_km[KS];
    bool h_real[KS];
    int hp_lds[KP], hp_dlt[KP];
    unsigned hp_src[KP];
    T hp_km[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { hp_lds[k] = 0; hp_dlt[k] = 0; hp_src[k] = 0; hp_km[k] = T(0); }
    if constexpr (HP) {
        // pair slots: axes d = 1 .. LA-1, 6 layers each, over the tile's extent on the other axes with the contiguous axis in pairs
        int areap[ND], basep[ND + 1];
        basep[1] = 0;
        FDiv fAp[ND];
        const int half = A.E[LA] >> 1;
        const FDiv fHalf = fdiv_make(half);
#pragma unroll
        for (int d = 1; d < LA; ++d) {
            areap[d] = half;
#pragma unroll
            for (int e = 1; e < LA; ++e) if (e != d) areap[d] *= A.E[e];
            basep[d + 1] = basep[d] + 2 * HJ_STENCIL * areap[d];
            fAp[d] = fdiv_make(areap[d]);
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            int h = tid + k * NT;
            if (h >= basep[LA]) h = 0;                // surplus slots shadow slot 0: same source, same LDS cells, same values
#pragma unroll
            for (int d = 1; d < LA; ++d) {
                if (h < basep[d] || h >= basep[d + 1]) continue;
                int lay, c;
                fdivmod(h - basep[d], fAp[d], lay, c);
                const int jd = (lay < HJ_STENCIL) ? (lay - HJ_STENCIL) : (A.E[d] + lay - HJ_STENCIL);
                int lo = 0, g = 0;
                {
                    int qe, j;
                    fdivmod(c, fHalf, qe, j);
                    c = qe;
                    lo += (2 * j + pad_of(LA)) * ls[LA];
                    g += (org[LA] + 2 * j) * A.pstride[LA];
                }
#pragma unroll
                for (int e = LA - 1; e >= 1; --e) {
                    if (e == d) continue;
                    int qe, j;
                    fdivmod(c, fE[e], qe, j);
                    c = qe;
                    lo += (j + pad_of(e)) * ls[e];
                    g += (org[e] + j) * A.pstride[e];
                }
                lo += (jd + pad_of(d)) * ls[d];
                int gi = org[d] + jd;
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
                hp_lds[k] = lo;
                hp_src[k] = (unsigned)(g + gi * A.pstride[d]) * (unsigned)sizeof(T);
                hp_dlt[k] = dlt * (int)sizeof(T);
                hp_km[k] = km;
            }
        }
    }
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
        // HP: the single slots cover the contiguous axis only (the other axes' layers are the pair slots above)
        const int first = HP ? base[LA] : 0;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            int h = first + tid + k * NT;
            h_real[k] = h < base[ND];
            if (!h_real[k]) h = first;                // shadow of the first slot (HJ_SLOT_PRED, hj_fused.h)
            h_lds[k] = 0; h_src[k] = 0; h_dlt[k] = 0; h_km[k] = T(0);
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
                    int qe, j;
                    fdivmod(c, fE[e], qe, j);
                    c = qe;
                    lo += (j + pad_of(e)) * ls[e];
                    g += (org[e] + j - (e == 1 ? xb : 0)) * A.pstride[e];
                }
                lo += (jd + pad_of(d)) * ls[d];
                int gi = org[d] + jd;
                const int nd = A.n[d];
                int dlt = 0;
                T km = T(0);
                // (transposed, tile axis 1 = the slab axis: beyond the slab's own planes lie its PAD planes where xh_lo / xh_hi say so -- real data)
                if (gi < 0 && !(XP && d == 1 && A.xh_lo)) {
                    if (A.bc[d] == HJ_BC_PERIODIC) gi += nd;
                    else { km = T(-gi) * A.km[d]; dlt = A.pstride[d]; gi = 0; }
                } else if (gi >= nd && !(XP && d == 1 && A.xh_hi)) {
                    if (A.bc[d] == HJ_BC_PERIODIC) gi -= nd;
                    else { km = T(gi - nd + 1) * A.km[d]; dlt = -A.pstride[d]; gi = nd - 1; }
                }
                h_lds[k] = lo;
                h_src[k] = (unsigned)(g + (gi - (d == 1 ? xb : 0)) * A.pstride[d]) * (unsigned)sizeof(T);
                h_dlt[k] = dlt * (int)sizeof(T);
                h_km[k] = km;
            }
        }
    }
    // does any halo slot of this tile lie outside a non-periodic plane axis?  (wave-uniform; the halo layers reach
    // HJ_STENCIL cells past both ends of the tile)
    bool tile_ghost = false;
#pragma unroll
    for (int d = 1; d < ND; ++d) {
        const bool below = org[d] < HJ_STENCIL && !(XP && d == 1 && A.xh_lo), above = org[d] + A.E[d] + HJ_STENCIL > A.n[d] && !(XP && d == 1 && A.xh_hi);
        tile_ghost = tile_ghost || (A.bc[d] != HJ_BC_PERIODIC && (below || above));
    }
    if (A.timing && threadIdx.x == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 2] = wall_clock64();

    T eps[ND];
    WenoK<T> wk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { eps[d] = T(0); wk[d].c13 = T(0); wk[d].c4 = T(0); }
    if constexpr (SCHEME == HJ_WENO5) {
        if (A.eps_nrows > 0) fold_eps_rows<T, ND, NT>(A.eps_rows, A.eps_nrows, red, eps);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            if (A.eps_nrows <= 0) eps[d] = T(1e-6) * A.max_d1sq[ham_gaxis<HAM>(d)] + Lim<T>::tiny;      // (the values are stored in grid order)
            wk[d] = weno_consts<T>(eps[d], A.K[d]);
        }
    }
    // intended WENO5 (round 5): the left-biased smoothness values of every own cell along axis 0 travel with the march -- a cell's
    // right-biased ones are the next plane's left-biased ones (hj_device.h, weno5_cd_carry): three per cell and plane instead of six
    constexpr bool WCARRY = SCHEME == HJ_WENO5 && (MODE == 1 || MODE == 2);      // (the general instantiation would spill: it forms all six per cell)
    T wl[R][2][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            wl[r][c][0] = T(0); wl[r][c][1] = T(0); wl[r][c][2] = T(0);
            if constexpr (WCARRY) {
                if (!down) weno5_left_q(weno5_line(q[r][c]), wk[0], wl[r][c]);      // the queue holds planes P(-3) .. P(3): the cell of plane P(0)
            }
        }
    // max(D1^2) of the output (eps_part, hj_fused.h): two more LDS planes behind the ring park the outputs of a plane
    // until the next iteration's barrier
    const bool eps_prod = SCHEME == HJ_WENO5 && A.eps_part != nullptr;
    T* const obuf = lds + A.lds_nbuf * lds_plane;
    // ... and (HJ_WENO_LDS_SHARE=1 builds only: measured slower, see the macro) on 3-D grids the MIDDLE axis shares them too, between lanes: every cell writes its three left-biased values to
    // an LDS plane each (behind the two planes of the epsilon producer), and after one more barrier takes the three of the cell ONE ROW UP as
    // its own right-biased ones; the cells of a tile's last row form theirs as before.  18 fp64 operations per cell replaced by 3 + 3
    // 16-byte LDS accesses per pair; the values are the same bits (weno5_cd_carry's identity, across lanes instead of across planes).
    constexpr bool WX = WCARRY && ND == 3 && HJ_WENO_LDS_SHARE;
    T* const qbuf = lds + (A.lds_nbuf + 2) * lds_plane;
    double dmax[ND];
    V oprev[R];
#pragma unroll
    for (int d = 0; d < ND; ++d) dmax[d] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) { oprev[r].x = T(0); oprev[r].y = T(0); }

    auto load_halo = [&](int p, T* dst, T* dst_in, V* dstp, V* dstp_in) {
        const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#if defined(HJ_ABLATE) && (HJ_ABLATE & 8)          // timing experiment (tuning builds; HJ_ABLATE bits: 1 no Hamiltonian arithmetic,
        (void)so;                                  // 2 no stencil LDS reads, 8 no halo loads, 16 no halo LDS stores)
        return;
#endif
        if constexpr (HP) {
#pragma unroll
            for (int k = 0; k < KP; ++k) dstp[k] = buf_load2(ry, hp_src[k], so, T());
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) dst[k] = buf_load(ry, h_src[k], so, T());
        if (tile_ghost) {
            if constexpr (HP) {
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    dstp_in[k].x = T(0); dstp_in[k].y = T(0);
                    if (hp_dlt[k] != 0) dstp_in[k] = buf_load2(ry, hp_src[k] + (unsigned)hp_dlt[k], so, T());
                }
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                dst_in[k] = T(0);
                if (h_dlt[k] != 0) dst_in[k] = buf_load(ry, h_src[k] + (unsigned)h_dlt[k], so, T());
            }
        }
    };
    // the ring of one plane -> its LDS buffer
    auto park_halo = [&](T* bufh, const T* hv, const T* hi, const V* pv, const V* pi) {
#if !(defined(HJ_ABLATE) && (HJ_ABLATE & 16))
        if (tile_ghost) {
            if constexpr (HP) {
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    V gv;
                    gv.x = ghost_value(pv[k].x, pi[k].x, hp_km[k]);
                    gv.y = ghost_value(pv[k].y, pi[k].y, hp_km[k]);
                    *reinterpret_cast<V*>(bufh + hp_lds[k]) = gv;
                }
            }
#pragma unroll
            for (int k = 0; k < KS; ++k)
                HJ_SLOT_PRED(k) bufh[h_lds[k]] = ghost_value(hv[k], hi[k], h_km[k]);
        } else {
            if constexpr (HP) {
#pragma unroll
                for (int k = 0; k < KP; ++k) *reinterpret_cast<V*>(bufh + hp_lds[k]) = pv[k];
            }
#pragma unroll
            for (int k = 0; k < KS; ++k)
                HJ_SLOT_PRED(k) bufh[h_lds[k]] = hv[k];
        }
#else
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" ::"v"(hv[k]));
        if constexpr (HP) {
#pragma unroll
            for (int k = 0; k < KP; ++k) asm volatile("" ::"v"(pv[k]));
        }
#endif
    };
    // Halo ring schedule.  AH = 0: the ring of plane P is requested PD planes ahead and written to LDS in iteration P.
    // AH = 3 (large grids, HJ_PAIR_RING): it is requested PD + 3 = 5 planes ahead -- in the very iteration in which
    // the neighbouring workgroup requests the same cells as ITS OWN cells of plane P, so that the second of the two
    // requests finds the lines in L2 (three iterations later they are gone: an XCD's 32 workgroups stream ~1.7 MB per
    // iteration through a 4 MB L2) -- and parked in the LDS buffer of plane P, a ring of NB = 5 plane buffers, from
    // iteration P - 3 on.  Same values either way.
    const int NB = A.lds_nbuf, AH = A.halo_ahead;
    T hal[PD][KS], hin[PD][KS];
    V halp[PD][KP], hinp[PD][KP];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int k = 0; k < KS; ++k) { hal[s][k] = T(0); hin[s][k] = T(0); }
#pragma unroll
        for (int k = 0; k < KP; ++k) { halp[s][k].x = T(0); halp[s][k].y = T(0); hinp[s][k].x = T(0); hinp[s][k].y = T(0); }
    }
    // every load of the setup goes out before anything waits: the rings of the first AH planes (parked in LDS below, after
    // the Hamiltonian constants: by then they have landed), the two rings the loop consumes from registers, then the
    // prefetched own plane and the y0 planes -- in the order the loop needs them (loads return in order)
    constexpr int AHM = HP ? 1 : 3;                 // 4-D runs without the parked ring (AH = 0)
    T th[AHM][KS], ti[AHM][KS];
    V tp[AHM][KP], tq[AHM][KP];
#pragma unroll
    for (int a = 0; a < AHM; ++a) {
#pragma unroll
        for (int k = 0; k < KS; ++k) { th[a][k] = T(0); ti[a][k] = T(0); }
#pragma unroll
        for (int k = 0; k < KP; ++k) { tp[a][k].x = T(0); tp[a][k].y = T(0); tq[a][k].x = T(0); tq[a][k].y = T(0); }
        if (a < AH) load_halo(clamp_c(plane_at(a)), th[a], ti[a], tp[a], tq[a]);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s) load_halo(clamp_c(plane_at(AH + s)), hal[s], hin[s], halp[s], hinp[s]);
    V own[PD][R], y0s[PD][R];
    typename HAM::Plane pls[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) { own[s][r].x = T(0); own[s][r].y = T(0); }
        if (s < PD - 1) load_own(clamp_q(plane_at(4 + s)), own[s]);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) { y0s[s][r].x = T(0); y0s[s][r].y = T(0); }
        const int ps = clamp_c(plane_at(s));
        load_y0(ps, y0s[s]);
        pls[s] = HAM::plane(A.ham, ps, A.sc);
    }


    if (A.timing && threadIdx.x == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 3] = wall_clock64();
    // the arithmetic on the Hamiltonian tables, now that every load of the setup is in flight
#pragma unroll
    for (int r = 0; r < R; ++r) {
        hcell[r][0] = HAM::cell_fin(A.ham, hraw[r][0], A.sc);
        hcell[r][1] = HAM::cell_fin(A.ham, hraw[r][1], A.sc);
    }
    // the rings of the first AH planes -> their LDS buffers (the first barrier of the loop orders them)
#pragma unroll
    for (int a = 0; a < AHM; ++a)
        if (a < AH) park_halo(lds + a * lds_plane, th[a], ti[a], tp[a], tq[a]);
    double amax[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) amax[d] = -1.0e300;
    // LOCAL Lax-Friedrichs with a Hamiltonian that reads the costate range (HamTables::local_mode, round 5): the bound is 1 / max_x sum_d
    // alpha_d(x) / dx_d -- the sum travels in slot 0 scaled so that the reader's generic  sum_d key_d / dx_d  returns it, the other slots hold 0.
    // BPASS: a MODE 3 launch with a bound slot = the pass that finds this maximum BEFORE the first stage (deltaT depends on it); nothing stored.
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
    // one stencil: centred costate + half jump of the substep, or (range pass) derivL / derivR themselves, reduced
    auto sten = [&](auto dim_tag, const T* v, T& pcv, T& hdv, bool real) {
        constexpr int d = decltype(dim_tag)::value;
        if constexpr (RNG) {
            // derivL = sc (pc - hd), derivR = sc (pc + hd) with the centred costate and half jump the substep forms anyway (sc > 0:
            // the minima / maxima are taken unscaled and scaled once, in publish_range)
            upwind_cd<SCHEME, T>(v, A.K[d], eps[d], wk[d], pcv, hdv);
            if (real && !BPASS) range_acc(rmn[d], rmx[d], pcv, hdv);
        } else {
            upwind_cd<SCHEME, T>(v, A.K[d], eps[d], wk[d], pcv, hdv);
        }
    };

#ifdef HJ_STAMP
    unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
    int ring_c = 0;                                            // LDS buffer of the plane the next iteration computes
    constexpr bool TWOB = HJ_TWO_PLANES && light_scheme_dev(SCHEME) && !HJ_MAYDOWN && ND <= 3 && MODE != 3 && !ham_xp<HAM>::value;
    int ring_of[2] = {0, 0};                                   // TWOB: the buffer each plane of the pass was staged in
    // PH: 0 the whole plane iteration (staging, barrier, compute); 1 the staging only, 2 the compute only (TWOB: one barrier per PASS between them)
    auto body = [&](auto off_tag, auto ph_tag, int m, V* own_c, V* own_n, T* hal_c, T* hin_c, V* halp_c, V* hinp_c, V* y0_c, typename HAM::Plane& pl_c) {
        constexpr int OFF = decltype(off_tag)::value;          // window [OFF, OFF + 7) of the queue is planes P(m - 3) .. P(m + 3)
        constexpr int PH = decltype(ph_tag)::value;
        const int p = plane_at(m);                              // the plane this iteration computes
#ifdef HJ_STAMP
        const unsigned long long st0 = __builtin_readcyclecounter();
#endif
        if constexpr (PH != 2) ring_of[OFF] = ring_c;
        T* buf = lds + ring_of[OFF] * lds_plane;                // plane p
      if constexpr (PH != 2) {
        int ring_h = ring_c + AH;
        if (ring_h >= NB) ring_h -= NB;
        T* bufh = lds + ring_h * lds_plane;                     // plane p + AH: where hal_c goes
        ring_c = (ring_c + 1 == NB) ? 0 : ring_c + 1;
        load_own(clamp_q(plane_at(m + 3 + PD)), own_n);
        // stage the centre plane: one 16-byte LDS store per pair
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((!vp_on && r < R - 1) || last_real) {
                V c2;
                c2.x = q[r][0][3 + OFF];
                c2.y = q[r][1][3 + OFF];
                *reinterpret_cast<V*>(buf + own_lds[r]) = c2;
            }
        park_halo(bufh, hal_c, hin_c, halp_c, hinp_c);
      }
      if constexpr (PH == 1) return;
#ifdef HJ_STAMP
        const unsigned long long st1 = __builtin_readcyclecounter();
#endif
#ifndef HJ_ABLATE_NOSYNC        // timing experiment only (results are wrong without the barrier)
        if constexpr (PH == 0) __syncthreads();
#endif
#ifdef HJ_STAMP
        const unsigned long long st2 = __builtin_readcyclecounter();
#endif
        const int p2 = clamp_c(plane_at(m + PD));
        load_halo(clamp_c(plane_at(m + PD + AH)), hal_c, hin_c, halp_c, hinp_c);
        const unsigned so_out = (unsigned)(p - p_lo) * plane_bytes;
        const typename HAM::Plane pl_use = pl_c;
        pl_c = HAM::plane(A.ham, p2, A.sc);
        // VPAIR: the middle-axis stencils of BOTH slots from six shared rows (the slots' rows are j1 and j1 + 1: rows j1 - 3 .. j1 - 1 and
        // j1 + 2 .. j1 + 4 from LDS, each slot's missing neighbour row is the other's own pair) -- formed first, so that the rows do not stay live
        T vpc[R][2], vhd[R][2];
        if constexpr (vp_on) {
            const T* base0 = buf + own_lds[0];
            V vrow[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) vrow[k] = *reinterpret_cast<const V*>(base0 + (k < 3 ? k - 3 : k - 1) * ls[1]);
            V own0, own1;
            own0.x = q[0][0][3 + OFF]; own0.y = q[0][1][3 + OFF];
            own1.x = q[1][0][3 + OFF]; own1.y = q[1][1][3 + OFF];
            const V l0[7] = {vrow[0], vrow[1], vrow[2], own0, own1, vrow[3], vrow[4]};
            const V l1[7] = {vrow[1], vrow[2], own0, own1, vrow[3], vrow[4], vrow[5]};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                T va[7], vb[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) { va[j] = r == 0 ? l0[j].x : l1[j].x; vb[j] = r == 0 ? l0[j].y : l1[j].y; }
                sten(IntTag<1>(), va, vpc[r][0], vhd[r][0], last_real);
                sten(IntTag<1>(), vb, vpc[r][1], vhd[r][1], last_real);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T pc[2][ND], hd[2][ND];
            const bool slot_real = (!vp_on && r < R - 1) || last_real;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (down) {
                    // the queue holds the planes in MARCH order: hand the stencil the same seven values in grid order
                    T qg[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) qg[j] = q[r][c][6 - j + OFF];
                    sten(IntTag<0>(), qg, pc[c][0], hd[c][0], slot_real);
                } else if constexpr (WCARRY) {
                    weno5_cd_carry(q[r][c] + OFF, wk[0], wl[r][c], pc[c][0], hd[c][0]);
                } else {
                    sten(IntTag<0>(), q[r][c] + OFF, pc[c][0], hd[c][0], slot_real);
                }
            }
            const T* base = buf + own_lds[r];
            // plane axes other than the contiguous one: the pair's neighbours are pairs (16-byte LDS reads)
#pragma unroll
            for (int d = 1; d < LA; ++d) {
                if constexpr (vp_on) {
                    pc[0][1] = vpc[r][0]; hd[0][1] = vhd[r][0]; pc[1][1] = vpc[r][1]; hd[1][1] = vhd[r][1];
                    continue;
                }
                T va[7], vb[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (j == 3) { va[j] = q[r][0][3 + OFF]; vb[j] = q[r][1][3 + OFF]; continue; }
#if defined(HJ_ABLATE) && (HJ_ABLATE & 2)
                    va[j] = q[r][0][j + OFF]; vb[j] = q[r][1][j + OFF];
#else
                    const V n2 = *reinterpret_cast<const V*>(base + (j - 3) * ls[d]);
                    va[j] = n2.x;
                    vb[j] = n2.y;
#endif
                }
                if constexpr (WX) {
                    T lqa[3], lqb[3], rqa[3], rqb[3];
                    WenoCand<T> ca, cb;
                    {
                        const WenoLine<T> la = weno5_line(va), lb = weno5_line(vb);
                        weno5_left_q(la, wk[1], lqa);
                        weno5_left_q(lb, wk[1], lqb);
                        ca = weno5_candidates(la);
                        cb = weno5_candidates(lb);
                        if (!(nbv[r] & 2u)) {      // a tile's last row has no row above it in LDS: its own right-biased values, as ever
                            weno5_right_q(la, wk[1], rqa);
                            weno5_right_q(lb, wk[1], rqb);
                        }
                    }
                    T* const qb = qbuf + own_lds[r];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        V w2;
                        w2.x = lqa[k]; w2.y = lqb[k];
                        *reinterpret_cast<V*>(qb + k * lds_plane) = w2;
                    }
                    __syncthreads();
                    if (nbv[r] & 2u) {           // the row above is a row of this tile: its left-biased values, reversed, are ours on the right
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const V n2 = *reinterpret_cast<const V*>(qb + ls[1] + (2 - k) * lds_plane);
                            rqa[k] = n2.x; rqb[k] = n2.y;
                        }
                    }
                    weno5_combine_cand(ca, lqa, rqa, pc[0][1], hd[0][1]);
                    weno5_combine_cand(cb, lqb, rqb, pc[1][1], hd[1][1]);
                } else if (d == 1) { sten(IntTag<1>(), va, pc[0][1], hd[0][1], slot_real); sten(IntTag<1>(), vb, pc[1][1], hd[1][1], slot_real); }
                else { sten(IntTag<(ND > 3 ? 2 : 1)>(), va, pc[0][d], hd[0][d], slot_real); sten(IntTag<(ND > 3 ? 2 : 1)>(), vb, pc[1][d], hd[1][d], slot_real); }
            }
            {   // the contiguous axis: cells j-3 .. j+4 = [b64][b128][own pair][b128][b64]
                T w[8];
#if defined(HJ_ABLATE) && (HJ_ABLATE & 2)
                w[0] = q[r][0][0 + OFF]; w[1] = q[r][0][1 + OFF]; w[2] = q[r][0][2 + OFF]; w[5] = q[r][1][4 + OFF]; w[6] = q[r][1][5 + OFF]; w[7] = q[r][1][6 + OFF];
                w[3] = q[r][0][3 + OFF]; w[4] = q[r][1][3 + OFF];
#else
                w[0] = base[-3];
                const V l2 = *reinterpret_cast<const V*>(base - 2);
                w[1] = l2.x; w[2] = l2.y;
                w[3] = q[r][0][3 + OFF]; w[4] = q[r][1][3 + OFF];
                const V r2 = *reinterpret_cast<const V*>(base + 2);
                w[5] = r2.x; w[6] = r2.y;
                w[7] = base[4];
#endif
                if constexpr (WCARRY) {
                    weno5_cd_pair(w, wk[LA], pc[0][LA], hd[0][LA], pc[1][LA], hd[1][LA]);      // the pair shares three of its twelve smoothness values
                } else {
                    sten(IntTag<LA>(), w, pc[0][LA], hd[0][LA], slot_real);
                    sten(IntTag<LA>(), w + 1, pc[1][LA], hd[1][LA], slot_real);
                }
            }
            if constexpr (RNG) {
                if constexpr (RR) {
                    if (BPASS && slot_real) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            T Hb, ab[ND];
                            lf_eval_local<NP, HAM>(A.ham, hcell[r][c], pl_use, A.sc, pc[c], hd[c], Hb, ab);
                            acc_alpha(ab);
                        }
                    }
                }
                continue;
            }
            V o2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                T alpha[ND];
#if defined(HJ_ABLATE) && (HJ_ABLATE & 1)
                T ydot = pc[c][0] + hd[c][ND - 1];
#pragma unroll
                for (int d = 0; d < ND; ++d) { alpha[d] = pc[c][d]; ydot += hd[c][d]; }
#else
                T ydot = lf_ydot<NP, HAM>(A.ham, hcell[r][c], pl_use, A.sc, pc[c], hd[c], alpha);
#endif
                acc_alpha(alpha);
                if (GEN && A.do_clamp) {
                    ydot = (ydot < A.clamp_lo) ? A.clamp_lo : ydot;
                    ydot = (ydot > A.clamp_hi) ? A.clamp_hi : ydot;
                }
                const T y0v = c == 0 ? y0_c[r].x : y0_c[r].y;
                T o;
                if (GEN && A.ydot_only) o = ydot;
                else {
                    o = rk_stage_out<NP>(A.stage, A.ca, A.cb, dt_launch, y0v, q[r][c][3 + OFF], ydot);
                    if (GEN && A.post_op) o = post_step(A.post_op, o, use_y0 ? y0v : q[r][c][3 + OFF]);
                }
                if (c == 0) o2.x = o; else o2.y = o;
            }
            if ((!vp_on && r < R - 1) || last_real) buf_store2(o2, rout, own_g[r], so_out);
            if constexpr (SCHEME == HJ_WENO5) {
                if (eps_prod) {
                    T* ob = obuf + ((p - p_begin) & 1) * lds_plane;
                    dmax[LA] = fmax(dmax[LA], (double)t_abs(o2.y - o2.x));      // the pair itself
                    if (p > p_begin) {
                        dmax[0] = fmax(dmax[0], fmax((double)t_abs(o2.x - oprev[r].x), (double)t_abs(o2.y - oprev[r].y)));
                        const T* op = obuf + ((p - p_begin - 1) & 1) * lds_plane + own_lds[r];     // plane p - 1
#pragma unroll
                        for (int d = 1; d < LA; ++d)
                            if ((nbv[r] >> d) & 1u) {
                                const V n2 = *reinterpret_cast<const V*>(op + ls[d]);
                                dmax[d] = fmax(dmax[d], fmax((double)t_abs(n2.x - oprev[r].x), (double)t_abs(n2.y - oprev[r].y)));
                            }
                        if ((nbv[r] >> LA) & 1u) dmax[LA] = fmax(dmax[LA], (double)t_abs(op[2] - oprev[r].y));
                    }
                    *reinterpret_cast<V*>(ob + own_lds[r]) = o2;
                    oprev[r] = o2;
                }
            }
        }
#ifdef HJ_STAMP
        const unsigned long long st3 = __builtin_readcyclecounter();
#endif
        load_y0(p2, y0_c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (OFF == 0) {            // the pass's first plane: the new plane joins behind the window, nothing moves
                if constexpr (PH == 0) {         // (TWOB: appended when the pass begins, before its second staging reuses the register set)
                    q[r][0][7] = own_c[r].x;
                    q[r][1][7] = own_c[r].y;
                }
            } else {                              // its second plane: two places down, ready for the next pass
#pragma unroll
                for (int j = 0; j < 6; ++j) { q[r][0][j] = q[r][0][j + 2]; q[r][1][j] = q[r][1][j + 2]; }
                q[r][0][6] = own_c[r].x;
                q[r][1][6] = own_c[r].y;
            }
        }
#ifdef HJ_STAMP
        {   // diagnostic build: shader-clock time of the phases of a plane iteration, summed per wave
            const unsigned long long st4 = __builtin_readcyclecounter();
            st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3;
        }
#endif
    };

    if (A.timing && tid == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 5] = wall_clock64();   // loop start
    const int nplanes = p_end - p_begin;
    for (int m = 0; m < nplanes; m += PD) {
        if constexpr (TWOB) {
            const bool two = m + 1 < nplanes;
#pragma unroll
            for (int r = 0; r < R; ++r) { q[r][0][7] = own[0][r].x; q[r][1][7] = own[0][r].y; }      // plane P(m + 4), requested a pass ago
            body(IntTag<0>(), IntTag<1>(), m, own[0], own[1], hal[0], hin[0], halp[0], hinp[0], y0s[0], pls[0]);
            if (two) body(IntTag<1>(), IntTag<1>(), m + 1, own[1], own[0], hal[1], hin[1], halp[1], hinp[1], y0s[1], pls[1]);
            __syncthreads();
            body(IntTag<0>(), IntTag<2>(), m, own[0], own[1], hal[0], hin[0], halp[0], hinp[0], y0s[0], pls[0]);
            if (two) body(IntTag<1>(), IntTag<2>(), m + 1, own[1], own[0], hal[1], hin[1], halp[1], hinp[1], y0s[1], pls[1]);
        } else {
            body(IntTag<0>(), IntTag<0>(), m, own[0], own[1], hal[0], hin[0], halp[0], hinp[0], y0s[0], pls[0]);
            if (m + 1 < nplanes) body(IntTag<1>(), IntTag<0>(), m + 1, own[1], own[0], hal[1], hin[1], halp[1], hinp[1], y0s[1], pls[1]);
        }
        if (A.timing && tid == 0 && m == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 7] = wall_clock64();
    }
    if (A.timing && tid == 0) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + 6] = wall_clock64();   // loop end

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
                    for (int d = 1; d < LA; ++d)
                        if ((nbv[r] >> d) & 1u) {
                            const V n2 = *reinterpret_cast<const V*>(op + ls[d]);
                            dmax[d] = fmax(dmax[d], fmax((double)t_abs(n2.x - oprev[r].x), (double)t_abs(n2.y - oprev[r].y)));
                        }
                    if ((nbv[r] >> LA) & 1u) dmax[LA] = fmax(dmax[LA], (double)t_abs(op[2] - oprev[r].y));
                }
            }
            store_eps_part<ND, NT>(A.eps_part + (size_t)L * HJ_MAX_DIM, red, dmax);
        }
    }
    if (A.bound) {   // a launch whose bound nobody reads (hj_rk_step: dt comes from the static bound) skips the reduction
        {   // alpha of the dimensions that do not vary along the march: column constants, taken once (any plane does)
            T pz[ND], Hz, az[ND];
    #pragma unroll
            for (int d = 0; d < ND; ++d) pz[d] = T(0);
    #pragma unroll
            for (int r = 0; r < R; ++r)
    #pragma unroll
                for (int c = 0; c < 2; ++c) {
                    HAM::eval(A.ham, hcell[r][c], pls[0], A.sc, pz, Hz, az);
    #pragma unroll
                    for (int d = 0; d < ND; ++d)
                        if (!((HAM::PLANE_DEP >> d) & 1u)) amax[d] = fmax(amax[d], (double)az[d]);
                }
        }

        const int lane = tid & 63, wv = tid >> 6;
    #pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double m = wave_max(amax[d]) / (double)A.sc[d];
            if (lane == 0) red[wv][d] = m;
        }
        __syncthreads();
        if (tid < ND) {
            double m = red[0][tid];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w][tid]);
            if (m > -1.0e299) key_max(A.bound + (XP ? (tid == 0 ? 1 : (tid == 1 ? 0 : tid)) : tid), m);      // the keys are in grid order
        }
    }
    publish_gate(A, chunk_id);
    if (A.timing && tid == 0) A.timing[4 * L + 1] = wall_clock64();
#ifdef HJ_STAMP
    if (A.timing && tid == 0)
        for (int k = 0; k < 4; ++k) A.timing[4 * (size_t)A.nblocks + 8 * (size_t)L + k] = st_acc[k];
#endif
}

}  // namespace hj
