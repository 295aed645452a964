// The 4-D substep kernel (round 5; gfx950): fused_pair_kernel's algorithm -- a 3-D tile of the plane axes 1..3 marching along
// axis 0, two adjacent cells of the contiguous axis per lane slot -- rebuilt around what the round-5 measurements say binds
// BASELINE config C5 (129^4 fp32): the launch is bound by the number of instructions a wave issues (profiles/r05_valu_rate.txt:
// one wave issues at most one instruction per ~5 cycles, two waves per SIMD ~2.5 cycles apart, and every packed-fp32 / 64-bit /
// SGPR-operand vector instruction costs ~4.4 cycles of the SIMD), so the kernel drops instructions, not bytes:
//
//   * COMPILE-TIME TILE (E1 x E2 x E3 cells, template arguments).  The LDS strides, the plane size and the slot counts are
//     constants: every LDS stencil read and every staging store is `base VGPR + immediate offset` -- the generic kernel
//     spends 8 vector instructions per cell and plane on LDS addresses (v_lshl_add / v_add of run-time strides) and two
//     dozen scalar ones on the ring of plane buffers.  The two planes of an unrolled loop pass use the two LDS buffers at
//     constant offsets.  A grid needs n[d] >= E[d] (the last tile of an axis is shifted back inside as before); smaller
//     grids take the generic kernel.
//   * ROW COEFFICIENTS IN LDS.  A Hamiltonian whose coefficients depend on (axis-0 plane, one plane axis) only -- the double
//     pendulum: HamDoublePendulum::Row, hj_device.h -- has them evaluated ONCE per (plane of the chunk, tile row) in the
//     prologue into an LDS table (chunk x E_row x 8 values) and read back with one ds_read_b128 + ds_read_b64 per slot and
//     plane: no trigonometric recombination, no division, no ten-term polynomial per cell.
//   * the CFL maxima accumulate in the kernel's own precision (one v_max per cell and dimension instead of a conversion to
//     double + a 64-bit max); ghosts of the plane axes only in the instantiation that can meet them (PG).
//   * no debug stamps, no intended-WENO5 epsilon plumbing (light stencils only), no down-marching.
// Same per-cell stencil arithmetic (upwind_cd) and stage expressions as every other substep kernel; the pendulum's drift is
// the factored form in ALL kernels, so tiled = direct stays bitwise.
#pragma once
#include "hj_fusedv.h"

// HJ_ABLATE4 (tuning builds only; the results are WRONG): bits 1 no stencil / Hamiltonian arithmetic, 2 no LDS stencil reads, 8 no halo
// loads, 16 no halo LDS stores, 32 no barrier, 64 no y0 loads -- what each part of the plane loop costs (profiles/r05_c5_ablation.txt)
#if defined(HJ_ABLATE4) && !defined(HJ_TUNE_BUILD)
#error "HJ_ABLATE4 gives wrong results: tuning builds (-DHJ_TUNE_BUILD) only"
#endif
#ifndef HJ_ABLATE4
#define HJ_ABLATE4 0
#endif
// planes the own cells / the RK operand y0 are requested ahead of their use (register sets; the loop is unrolled by their common multiple)
// cache policy of the streams (gfx940+ aux encoding: 1 = sc0, 2 = nt, 16 = sc1): own cells, y0, output
#ifndef HJ_P4_AUX_OWN
#define HJ_P4_AUX_OWN 0
#endif
#ifndef HJ_P4_AUX_Y0
#define HJ_P4_AUX_Y0 0
#endif
#ifndef HJ_P4_AUX_OUT
#define HJ_P4_AUX_OUT 0
#endif
#ifndef HJ_P4_AUX_HALO
#define HJ_P4_AUX_HALO 0
#endif
#ifndef HJ_P4_DO
#define HJ_P4_DO 2
#endif
#ifndef HJ_P4_DY
#define HJ_P4_DY 2
#endif

namespace hj {

template <int AUX>
__device__ __forceinline__ void buf_store2_aux(Pair<float>::V v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    using W2 = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(W2, v), r, off, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store2_aux(Pair<double>::V v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) { buf_store2(v, r, off, soff); }

// geometry of an (E1, E2, E3) tile, in cells of T
template <int E1, int E2, int E3> struct Tile4 {
    static constexpr int W = HJ_STENCIL, VP = HJ_VPAD;
    static constexpr int PITCH = E3 + 2 * VP;                 // row pitch (even)
    static constexpr int LS2 = PITCH, LS1 = PITCH * (E2 + 2 * W);
    static constexpr int PLANE = LS1 * (E1 + 2 * W);          // cells of one LDS plane buffer
    static constexpr int HALF = E3 / 2;
    static constexpr int SLOTS = E1 * E2 * HALF;              // own pairs
    static constexpr int NP1 = 2 * W * E2 * HALF, NP2 = 2 * W * E1 * HALF;      // halo pairs of axes 1 and 2
    static constexpr int NPC = 4 * E1 * E2;                   // halo pairs of axis 3: cells -4 .. -1 and E3 .. E3+3 of every own row
    static constexpr int NPS = NP1 + NP2 + NPC;
    static constexpr int ROWF = 8;                            // values per row-table entry (6 used; 32-byte entries)
    static_assert(E3 % 2 == 0, "even extent on the contiguous axis");
    // Slot s of a set of `nrows` tile rows (HALF pairs each) -> (row, pair of the row).  A 16-lane group of a wave is what
    // the LDS serves in one pass for the 8-byte accesses (ds_write_b64, ds_read2_b64: 32 banks of 4 bytes), so the slots are
    // numbered such that 16 consecutive lanes hold 16 CONSECUTIVE pairs of ONE row (32 consecutive words: every bank once);
    // the pairs a row has beyond a multiple of 16 come last, one row after the other (for one such pair per row the 16 lanes
    // of a group are PITCH words apart: conflict free as well for the pitches used).  With the plain numbering (pair fastest
    // through the rows) nearly every group of a 17-pair row straddles a row end, where the bank sequence jumps by the row
    // padding: half of the LDS cycles of the round-4 C5 kernel were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.49).
    static constexpr int A16 = (HALF / 16) * 16, B16 = HALF % 16;
    __device__ static __forceinline__ void row_pair(int s, int nrows, int& row, int& pidx) {
        if constexpr (A16 == 0) { row = s / HALF; pidx = s % HALF; return; }
        const int main = nrows * A16;
        if (B16 == 0 || s < main) { row = s / A16; pidx = s % A16; }
        else { const int e = s - main; row = e / (B16 > 0 ? B16 : 1); pidx = A16 + e % (B16 > 0 ? B16 : 1); }
    }
};

// per-cell constants of a Hamiltonian in this kernel: its TCell when it factors its coefficients into rows, else its Cell
template <typename HAM, bool ROWS> struct RowCell {
    typedef typename HAM::Cell type;
    template <typename P, typename RAW, typename T> __device__ static __forceinline__ type make(const P& p, const RAW& r, const T* sc) {
        return HAM::cell_fin(p, r, sc);
    }
    template <typename P, typename PL, typename T>
    __device__ static __forceinline__ void eval_any(const P& p, const type& c, const PL& pl, const T* sc, const T* q, T& H, T* alpha) {
        HAM::eval(p, c, pl, sc, q, H, alpha);
    }
};
template <typename HAM> struct RowCell<HAM, true> {
    typedef typename HAM::TCell type;
    template <typename P, typename RAW, typename T> __device__ static __forceinline__ type make(const P& p, const RAW& r, const T* sc) {
        return HAM::tcell(HAM::cell_fin(p, r, sc), sc);
    }
    // alphas that do not depend on the row (the caller only reads those): a zero row
    template <typename P, typename PL, typename T>
    __device__ static __forceinline__ void eval_any(const P& p, const type& c, const PL&, const T* sc, const T* q, T& H, T* alpha) {
        typename HAM::Row z = {};
        HAM::eval_row(p, c, z, sc, q, H, alpha);
    }
};

// the plane axis the rows of a row-factored Hamiltonian vary along (0: none)
template <typename HAM, bool ROWS> struct RowAxis { static constexpr int value = 0; };
template <typename HAM> struct RowAxis<HAM, true> { static constexpr int value = HAM::ROW_AXIS; };

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int E3, int OCC, bool PG, int MODE>
__global__ __launch_bounds__(NT, OCC) void fused_pair4_kernel(const T* __restrict__ y, const T* __restrict__ y0,
                                                              T* __restrict__ out, const FusedArgs<T, 4> A) {
    constexpr int ND = 4, LA = 3, PD = 2, W = HJ_STENCIL, VP = HJ_VPAD;
    static_assert(HAM::ND == 4, "4-D Hamiltonians");
    static_assert(SCHEME == HJ_WENO5_ASSHIPPED || SCHEME == HJ_ENO2 || SCHEME == HJ_ENO2_FAST, "light stencils");
    using G = Tile4<E1, E2, E3>;
    constexpr int LS1 = G::LS1, LS2 = G::LS2, PLANE = G::PLANE, HALF = G::HALF;
    constexpr int KP = (G::NPS + NT - 1) / NT;
    static_assert(G::SLOTS <= NT * R, "tile does not fit the slots");
    constexpr bool GEN = (MODE == 0);
    constexpr bool NP = np_order(SCHEME);
    constexpr bool ROWS = ham_has_rows<HAM>::value;
    constexpr int RAX = RowAxis<HAM, ROWS>::value;
    constexpr int ER = RAX == 1 ? E1 : E2;             // rows of a tile
    static_assert(!ROWS || RAX == 1 || RAX == 2, "rows along plane axis 1 or 2");
    using V = typename Pair<T>::V;
    const bool use_y0 = GEN ? (A.use_y0 != 0) : (MODE == 2);
    extern __shared__ __align__(16) unsigned char hj_smem[];
    T* const lds = reinterpret_cast<T*>(hj_smem + 512);
    T* const rowtab = lds + 2 * PLANE;                 // ROWS: (planes of the chunk) x E_row x ROWF
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");

    asm volatile("" ::"s"(A.nblocks), "s"(A.ntiles), "s"(A.blocks_per_xcd), "s"(A.ntile[1]), "s"(A.ntile[2]), "s"(A.ntile[3]), "s"(A.n[0]),
                 "s"(A.n[1]), "s"(A.n[2]), "s"(A.n[3]), "s"(A.nchunks1), "s"(A.plane_begin), "s"(A.plane_end), "s"(A.plane_begin2),
                 "s"(A.plane_end2), "s"(A.chunk), "s"(A.pstride[1]), "s"(A.pstride[2]), "s"(A.stride0), "s"(A.halo_lo), "s"(A.halo_hi),
                 "s"(A.edge_blocks), "s"(A.edge_count), "s"(A.nchunks_e));
    asm volatile("" ::"s"(y), "s"(y0), "s"(out), "s"(A.ham.coord[1]), "s"(A.ham.coord[3]), "s"(A.ham.aux[0]), "s"(A.ham.aux[1]),
                 "s"(A.ham.aux[2]), "s"(A.ham.aux[3]), "s"(A.tb[1]), "s"(A.tb[2]));
    const int L = logical_block(A);
    if (L < 0) return;
    int chunk_id, rem;
    fdivmod(L, fdiv_make(A.ntiles), chunk_id, rem);
    int org[ND], tc[ND];
    tile_coords<ND>(A, rem, tc);
    org[1] = min(tc[1] * E1, A.n[1] - E1);
    org[2] = min(tc[2] * E2, A.n[2] - E2);
    org[3] = min(tc[3] * E3, A.n[3] - E3);
    int p_begin, p_end;
    chunk_planes(A, chunk_id, p_begin, p_end);
    auto clamp_q = [&](int p) { return min(max(p, p_begin - W), p_end + W - 1); };
    auto clamp_c = [&](int p) { return min(max(p, p_begin), p_end - 1); };
    const int tid = threadIdx.x;

    // ---- own pair slots
    int own_lds[R];                 // cell index of the pair in a plane buffer, MINUS W*LS1 (every stencil offset is then >= 0)
    unsigned own_g[R];
    int rowoff[R];                  // ROWS: offset of the slot's row in one plane's entries of the row table (values of T)
    typename RowCell<HAM, ROWS>::type hcell[R][2];      // per-cell constants of the Hamiltonian (ROWS: its TCell)
    const bool last_real = (tid + (R - 1) * NT) < G::SLOTS;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = min(tid + r * NT, G::SLOTS - 1);
        int t, pidx;
        G::row_pair(s, E1 * E2, t, pidx);
        const int j3 = 2 * pidx, j2 = t % E2, j1 = t / E2;
        int idx[ND] = {0, org[1] + j1, org[2] + j2, org[3] + j3};
        own_lds[r] = j1 * LS1 + (j2 + W) * LS2 + (j3 + VP);
        own_g[r] = (unsigned)(idx[1] * A.pstride[1] + idx[2] * A.pstride[2] + idx[3]) * (unsigned)sizeof(T);
        rowoff[r] = (RAX == 1 ? j1 : j2) * G::ROWF;
        const typename HAM::Raw r0 = HAM::cell_raw(A.ham, idx);
        idx[LA] += 1;
        const typename HAM::Raw r1 = HAM::cell_raw_next(A.ham, idx, r0);
        hcell[r][0] = RowCell<HAM, ROWS>::make(A.ham, r0, A.sc);
        hcell[r][1] = RowCell<HAM, ROWS>::make(A.ham, r1, A.sc);
    }

    const unsigned plane_bytes = (unsigned)(A.stride0 * (long long)sizeof(T));
    const int p_lo = p_begin - W;
    const unsigned span = (unsigned)(p_end + W - p_lo) * plane_bytes;
    const __amdgpu_buffer_rsrc_t ry = make_srd(y + (long long)p_lo * A.stride0, span);
    const __amdgpu_buffer_rsrc_t ry0 = make_srd(y0 + (long long)p_lo * A.stride0, use_y0 ? span : 0u);
    const __amdgpu_buffer_rsrc_t rout = make_srd(out + (long long)p_lo * A.stride0, span);
    auto load_own = [&](int p, V* dst) {
        const bool direct = (p >= 0 || A.halo_lo) && (p < A.n[0] || A.halo_hi);
        if (direct) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2<HJ_P4_AUX_OWN>(ry, own_g[r], so, T());
        } else {
            const PlaneSrc<T> s = plane_src<T, ND>(A, p);
            const __amdgpu_buffer_rsrc_t rb = make_srd(y + s.off, plane_bytes);
            if (!s.ghost) {
#pragma unroll
                for (int r = 0; r < R; ++r) dst[r] = buf_load2(rb, own_g[r], 0u, T());
            } else {
                const __amdgpu_buffer_rsrc_t ri = make_srd(y + s.off_in, plane_bytes);
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
        if (use_y0 && !(HJ_ABLATE4 & 64)) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2<HJ_P4_AUX_Y0>(ry0, own_g[r], so, T());
        }
    };

    // axis-0 queue, 8 deep, shifted by two places once per loop pass (hj_fusedv.h)
    T q[R][2][8];
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) {
        V tmp[R];
        load_own(p_begin - 3 + jj, tmp);
#pragma unroll
        for (int r = 0; r < R; ++r) { q[r][0][jj] = tmp[r].x; q[r][1][jj] = tmp[r].y; }
    }

    // ---- halo slots, all PAIRS (8-byte loads, ds_write_b64): the 3 + 3 layers of axes 1 and 2 over the tile's extent on the other
    // axes (rows of HALF pairs, numbered by row_pair), then the columns either side of every own row as the two pairs at cells
    // -4 .. -1 and the two at E3 .. E3+3 (the stencil reads -3 .. -1 and E3 .. E3+2).  Until this form the columns were 6 single
    // cells per row, fetched by 4-byte loads that touch a cache line per 3 lanes: a wave-instruction of those costs the CU's
    // texture-address path 60-90 cycles (profiles/r05_l1_rate.txt), 15-20 % of the launch for 3 % of its bytes.  A pair never
    // straddles an end of the axis: the host takes this kernel only for grids where no tile begins or ends 1 or 3 cells from an
    // end (tile4_fits, hj_inst.hip).  Surplus slots shadow slot 0 (same source, same LDS cells, same values).
    // PG (a plane axis may be extrapolated): a ghost PAIR of axis 1 / 2 is formed cell by cell from the edge pair and the inner pair
    // (hp_dlt bytes further in); a ghost pair of axis 3 lies k and k -+ 1 cells beyond ONE edge cell: both cells come from the pair
    // (edge, inner) that is loaded anyway -- hp_mode 1: left end, the pair holds (edge, inner); 2: right end, (inner, edge).
    int hp_lds[KP];
    unsigned hp_src[KP];
    int hp_dlt[KP], hp_mode[KP];
    T hp_km[KP], hp_km2[KP];              // k*slope multiplier of the pair's first / second cell
    auto wrap = [&](int d, int& gi, int& dlt, T& km) {       // global index along plane axis d of a halo cell -> source cell
        const int nd = A.n[d];
        dlt = 0;
        km = T(0);
        if (gi < 0) {
            if (!PG || A.bc[d] == HJ_BC_PERIODIC) gi += nd;
            else { km = T(-gi) * A.km[d]; dlt = A.pstride[d]; gi = 0; }
        } else if (gi >= nd) {
            if (!PG || A.bc[d] == HJ_BC_PERIODIC) gi -= nd;
            else { km = T(gi - nd + 1) * A.km[d]; dlt = -A.pstride[d]; gi = nd - 1; }
        }
    };
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        int h = tid + k * NT;
        if (h >= G::NPS) h = 0;
        int lo, g, dlt = 0, mode = 0;
        T km = T(0), km2 = T(0);
        if (h < G::NP1 + G::NP2) {
            int hrow, hpidx;            // rows of the halo: 2W*E2 of axis 1 (layer, j2), then 2W*E1 of axis 2 (layer, j1)
            G::row_pair(h, 2 * W * (E1 + E2), hrow, hpidx);
            if (hrow < 2 * W * E2) {    // axis 1
                const int lay = hrow / E2, j2 = hrow % E2, j3 = 2 * hpidx;
                const int jd = lay < W ? lay - W : E1 + lay - W;
                int gi = org[1] + jd;
                wrap(1, gi, dlt, km);
                lo = (jd + W) * LS1 + (j2 + W) * LS2 + (j3 + VP);
                g = gi * A.pstride[1] + (org[2] + j2) * A.pstride[2] + org[3] + j3;
            } else {                    // axis 2
                const int hh = hrow - 2 * W * E2;
                const int lay = hh / E1, j1 = hh % E1, j3 = 2 * hpidx;
                const int jd = lay < W ? lay - W : E2 + lay - W;
                int gi = org[2] + jd;
                wrap(2, gi, dlt, km);
                lo = (j1 + W) * LS1 + (jd + W) * LS2 + (j3 + VP);
                g = (org[1] + j1) * A.pstride[1] + gi * A.pstride[2] + org[3] + j3;
            }
            km2 = km;
        } else {                        // the columns of axis 3
            const int hh = h - (G::NP1 + G::NP2);
            const int row = hh / 4, kk = hh % 4, j2 = row % E2, j1 = row / E2;
            const int jd = kk < 2 ? 2 * kk - 4 : E3 + 2 * (kk - 2);
            int gi = org[3] + jd;
            const int n3 = A.n[3];
            if (gi < 0) {
                if (!PG || A.bc[3] == HJ_BC_PERIODIC) gi += n3;
                else { km = T(-gi) * A.km[3]; km2 = T(-gi - 1) * A.km[3]; mode = 1; gi = 0; }
            } else if (gi >= n3) {
                if (!PG || A.bc[3] == HJ_BC_PERIODIC) gi -= n3;
                else { km = T(gi - n3 + 1) * A.km[3]; km2 = T(gi - n3 + 2) * A.km[3]; mode = 2; gi = n3 - 2; }
            }
            lo = (j1 + W) * LS1 + (j2 + W) * LS2 + (jd + VP);
            g = (org[1] + j1) * A.pstride[1] + (org[2] + j2) * A.pstride[2] + gi;
        }
        hp_lds[k] = lo;
        hp_src[k] = (unsigned)g * (unsigned)sizeof(T);
        hp_dlt[k] = dlt * (int)sizeof(T);
        hp_km[k] = km;
        hp_km2[k] = km2;
        hp_mode[k] = mode;
    }
    bool tile_ghost = false;
    if constexpr (PG) {
        const int Ed[ND] = {0, E1, E2, E3};
#pragma unroll
        for (int d = 1; d < ND; ++d)
            tile_ghost = tile_ghost || (A.bc[d] != HJ_BC_PERIODIC && (org[d] < VP || org[d] + Ed[d] + VP > A.n[d]));
    }

    struct Halo { V p[KP]; V pi[KP]; };
    auto load_halo = [&](int p, Halo& h) {
        const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
        if (HJ_ABLATE4 & 8) return;
#pragma unroll
        for (int k = 0; k < KP; ++k) h.p[k] = buf_load2<HJ_P4_AUX_HALO>(ry, hp_src[k], so, T());
        if constexpr (PG) {
            if (tile_ghost) {
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    h.pi[k].x = T(0); h.pi[k].y = T(0);
                    if (hp_dlt[k] != 0) h.pi[k] = buf_load2(ry, hp_src[k] + (unsigned)hp_dlt[k], so, T());
                }
            }
        }
    };
    auto park_halo = [&](T* buf, const Halo& h) {
        if (HJ_ABLATE4 & 16) {
#pragma unroll
            for (int k = 0; k < KP; ++k) asm volatile("" ::"v"(h.p[k]));
            return;
        }
        if (PG && tile_ghost) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const V pv = h.p[k], qv = h.pi[k];
                const int m = hp_mode[k];
                const T ex = m == 2 ? pv.y : pv.x, ey = m == 1 ? pv.x : pv.y;
                const T ix = m == 0 ? qv.x : (m == 1 ? pv.y : pv.x), iy = m == 0 ? qv.y : (m == 1 ? pv.y : pv.x);
                V gv;
                gv.x = ghost_value(ex, ix, hp_km[k]);
                gv.y = ghost_value(ey, iy, hp_km2[k]);
                *reinterpret_cast<V*>(buf + hp_lds[k]) = gv;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KP; ++k) *reinterpret_cast<V*>(buf + hp_lds[k]) = h.p[k];
        }
    };

    Halo hal[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int k = 0; k < KP; ++k) { hal[s].pi[k].x = T(0); hal[s].pi[k].y = T(0); }
        load_halo(clamp_c(p_begin + s), hal[s]);
    }
    constexpr int DO = HJ_P4_DO, DY = HJ_P4_DY;
    constexpr int UNR = (DO == 4 || DY == 4) ? 4 : ((DO == 3 || DY == 3) ? 6 : 2);
    static_assert(UNR % DO == 0 && UNR % DY == 0 && UNR % PD == 0, "prefetch depths 2, 3 or 4");
    V own[DO][R], y0s[DY][R];
    typename HAM::Plane pls[PD];
#pragma unroll
    for (int s = 0; s < DO; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) { own[s][r].x = T(0); own[s][r].y = T(0); }
        if (s < DO - 1) load_own(clamp_q(p_begin + 4 + s), own[s]);
    }
#pragma unroll
    for (int s = 0; s < DY; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) { y0s[s][r].x = T(0); y0s[s][r].y = T(0); }
        load_y0(clamp_c(p_begin + s), y0s[s]);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if constexpr (!ROWS) pls[s] = HAM::plane(A.ham, clamp_c(p_begin + s), A.sc);

    // ---- ROWS: the Hamiltonian's rows of every plane of the chunk -> LDS (the loop's first barrier orders them)
    if constexpr (ROWS) {
        const int nent = (p_end - p_begin) * ER;
        for (int e = tid; e < nent; e += NT) {
            const int pi = e / ER, j = e - pi * ER;
            const typename HAM::Row rw = HAM::row_at(A.ham, p_begin + pi, org[RAX] + j, A.sc);
            T* dst = rowtab + e * G::ROWF;
            typedef T V4 __attribute__((ext_vector_type(4)));
            V4 lo4 = {rw.a1, rw.b1, rw.c1, rw.a3};
            V hi2;
            hi2.x = rw.b3; hi2.y = rw.c3;
            *reinterpret_cast<V4*>(dst) = lo4;
            *reinterpret_cast<V*>(dst + 4) = hi2;
        }
    }
    T amax[ND][2];
#pragma unroll
    for (int d = 0; d < ND; ++d) { amax[d][0] = Lim<T>::lowest; amax[d][1] = Lim<T>::lowest; }

    auto stencil = [&](const T* v, const T* K, T& pcv, T& hdv) {
        if (HJ_ABLATE4 & 1) {
            pcv = v[0]; hdv = v[6];
#pragma unroll
            for (int j = 1; j < 6; ++j) asm volatile("" ::"v"(v[j]));
        } else {
            upwind_cd<SCHEME, T>(v, K, T(0), WenoK<T>{T(0), T(0)}, pcv, hdv);
        }
    };
    auto body = [&](auto off_tag, int p, V* own_c, V* own_n, Halo& hal_c, V* y0_c, typename HAM::Plane& pl_c) {
        constexpr int OFF = decltype(off_tag)::value;          // window [OFF, OFF + 7) of the queue; LDS buffer OFF
        T* const buf = lds + OFF * PLANE;
        load_own(clamp_q(p + 3 + DO), own_n);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < R - 1 || last_real) {
                V c2;
                c2.x = q[r][0][3 + OFF];
                c2.y = q[r][1][3 + OFF];
                *reinterpret_cast<V*>(buf + W * LS1 + own_lds[r]) = c2;
            }
        park_halo(buf, hal_c);
        if (!(HJ_ABLATE4 & 32)) __syncthreads();
        const int p2 = clamp_c(p + PD);
        load_halo(p2, hal_c);
        const unsigned so_out = (unsigned)(p - p_lo) * plane_bytes;
        typename HAM::Plane pl_use = pl_c;
        if constexpr (!ROWS) pl_c = HAM::plane(A.ham, p2, A.sc);
        const int row_plane = (p - p_begin) * (ER * G::ROWF);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T pc[2][ND], hd[2][ND];
#pragma unroll
            for (int c = 0; c < 2; ++c) stencil(q[r][c] + OFF, A.K[0], pc[c][0], hd[c][0]);
            const T* base = buf + own_lds[r];           // = the pair's cell - W*LS1
#pragma unroll
            for (int d = 1; d < LA; ++d) {
                constexpr int LSD[3] = {0, LS1, LS2};
                T va[7], vb[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (j == 3) { va[j] = q[r][0][3 + OFF]; vb[j] = q[r][1][3 + OFF]; continue; }
                    if (HJ_ABLATE4 & 2) { va[j] = q[r][0][j + OFF]; vb[j] = q[r][1][j + OFF]; continue; }
                    const V n2 = *reinterpret_cast<const V*>(base + W * LS1 + (j - 3) * LSD[d]);
                    va[j] = n2.x;
                    vb[j] = n2.y;
                }
                stencil(va, A.K[d], pc[0][d], hd[0][d]);
                stencil(vb, A.K[d], pc[1][d], hd[1][d]);
            }
            {
                T w[8];
                const T* ctr = base + W * LS1;
                if (HJ_ABLATE4 & 2) {
                    w[0] = q[r][0][0 + OFF]; w[1] = q[r][0][1 + OFF]; w[2] = q[r][0][2 + OFF]; w[5] = q[r][1][4 + OFF]; w[6] = q[r][1][5 + OFF]; w[7] = q[r][1][6 + OFF];
                    w[3] = q[r][0][3 + OFF]; w[4] = q[r][1][3 + OFF];
                } else {
                w[0] = ctr[-3];
                const V l2 = *reinterpret_cast<const V*>(ctr - 2);
                w[1] = l2.x; w[2] = l2.y;
                w[3] = q[r][0][3 + OFF]; w[4] = q[r][1][3 + OFF];
                const V r2 = *reinterpret_cast<const V*>(ctr + 2);
                w[5] = r2.x; w[6] = r2.y;
                w[7] = ctr[4];
                }
                stencil(w, A.K[LA], pc[0][LA], hd[0][LA]);
                stencil(w + 1, A.K[LA], pc[1][LA], hd[1][LA]);
            }
            V o2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                T alpha[ND], ydot;
                if (HJ_ABLATE4 & 1) {
                    ydot = pc[c][0] + hd[c][ND - 1];
#pragma unroll
                    for (int d = 0; d < ND; ++d) { alpha[d] = pc[c][d]; ydot += hd[c][d]; }
                } else
                if constexpr (ROWS) {
                    typedef T V4 __attribute__((ext_vector_type(4)));
                    const T* rp = rowtab + row_plane + rowoff[r];
                    const V4 lo4 = *reinterpret_cast<const V4*>(rp);
                    const V hi2 = *reinterpret_cast<const V*>(rp + 4);
                    typename HAM::Row rw;
                    rw.a1 = lo4.x; rw.b1 = lo4.y; rw.c1 = lo4.z; rw.a3 = lo4.w; rw.b3 = hi2.x; rw.c3 = hi2.y;
                    ydot = lf_ydot_row<NP, HAM>(A.ham, hcell[r][c], rw, A.sc, pc[c], hd[c], alpha);
                } else {
                    ydot = lf_ydot<NP, HAM>(A.ham, hcell[r][c], pl_use, A.sc, pc[c], hd[c], alpha);
                }
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if ((HAM::PLANE_DEP >> d) & 1u) amax[d][c] = max_acc(amax[d][c], alpha[d]);
                if (GEN && A.do_clamp) {
                    ydot = (ydot < A.clamp_lo) ? A.clamp_lo : ydot;
                    ydot = (ydot > A.clamp_hi) ? A.clamp_hi : ydot;
                }
                const T y0v = c == 0 ? y0_c[r].x : y0_c[r].y;
                T o;
                if (GEN && A.ydot_only) o = ydot;
                else {
                    o = rk_stage_out<NP>(A.stage, A.ca, A.cb, A.dt, y0v, q[r][c][3 + OFF], ydot);
                    if (GEN && A.post_op) o = post_step(A.post_op, o, use_y0 ? y0v : q[r][c][3 + OFF]);
                }
                if (c == 0) o2.x = o; else o2.y = o;
            }
            if (r < R - 1 || last_real) buf_store2_aux<HJ_P4_AUX_OUT>(o2, rout, own_g[r], so_out);
        }
        load_y0(clamp_c(p + DY), y0_c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (OFF == 0) {
                q[r][0][7] = own_c[r].x;
                q[r][1][7] = own_c[r].y;
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) { q[r][0][j] = q[r][0][j + 2]; q[r][1][j] = q[r][1][j + 2]; }
                q[r][0][6] = own_c[r].x;
                q[r][1][6] = own_c[r].y;
            }
        }
    };

#define HJ_BODY4(u)                                                                                                     \
    if constexpr (UNR > (u)) {                                                                                          \
        if (p + (u) < p_end)                                                                                            \
            body(IntTag<(u) & 1>(), p + (u), own[(u) % DO], own[((u) + DO - 1) % DO], hal[(u) % PD], y0s[(u) % DY], pls[(u) % PD]); \
    }
    for (int p = p_begin; p < p_end; p += UNR) {
        HJ_BODY4(0) HJ_BODY4(1) HJ_BODY4(2) HJ_BODY4(3) HJ_BODY4(4) HJ_BODY4(5)
    }
#undef HJ_BODY4

    if (A.bound) {
        {   // alpha of the dimensions that do not vary along the march: column constants, taken once
            T pz[ND], Hz, az[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) pz[d] = T(0);
            const typename HAM::Plane pl0 = HAM::plane(A.ham, clamp_c(p_begin), A.sc);
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    RowCell<HAM, ROWS>::eval_any(A.ham, hcell[r][c], pl0, A.sc, pz, Hz, az);
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        if (!((HAM::PLANE_DEP >> d) & 1u)) amax[d][c] = t_max(amax[d][c], az[d]);
                }
        }
        const int lane = tid & 63, wv = tid >> 6;
        __syncthreads();            // (red aliases nothing the loop uses, but the last plane's LDS reads must be over before a reuse)
        double (*redd)[ND] = reinterpret_cast<double (*)[ND]>(hj_smem);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const T ml = t_max(amax[d][0], amax[d][1]);
            const double m = wave_max(ml > Lim<T>::lowest ? (double)ml : -1.0e300) / (double)A.sc[d];
            if (lane == 0) redd[wv][d] = m;
        }
        __syncthreads();
        if (tid < ND) {
            double m = redd[0][tid];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, redd[w][tid]);
            if (m > -1.0e299) key_max(A.bound + tid, m);
        }
    }
    publish_gate(A, chunk_id);
}

}  // namespace hj
