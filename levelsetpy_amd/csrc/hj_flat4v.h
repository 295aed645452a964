// The 4-D substep kernel with FULL-ROW tiles and 16-byte row loads (round 6; gfx950) -- fused_pair4_kernel's algorithm and per-cell
// arithmetic (bitwise the same results) with another data path for the stencil source.
//
// Why (profiles/r05_c5_tiles.txt, r05_l1_rate.txt): on BASELINE config C5 (129^4 fp32) the pair kernel is bound by the CU's texture-address
// path -- every access is an 8-byte one on 264-byte row pieces at a 516-byte stride (23 cycles per wave instruction against 15.5 on whole
// lines), the 3 + 3 halo cells either side of a row piece are lane gathers at 60-90 cycles, and rows of 129 floats start on no 16-byte
// boundary but every fourth.  A 16-byte load over CONTIGUOUS memory moves 40 B/clk/CU against 22.
//
// What changes:
//   * a tile spans the WHOLE contiguous axis 3 (E1 x E2 rows of n3 cells; n3 is a run-time value below the LDS row pitch P3): the halo
//     columns of axis 3 do not exist -- periodic wrap / ghost cells of axis 3 are cells of the SAME row, written into the row's pad by
//     the lanes that own the cells at its ends;
//   * every row that is loaded -- the 6 x E2 + 6 x E1 halo rows of the centre plane, the E1 x E2 own rows of the plane that enters the
//     axis-0 register queue -- is a contiguous run of n3 values: `buffer_load_dwordx4` in chunks of 4 cells, consecutive lanes on
//     consecutive chunks, `ds_write_b128` into a row of the LDS box (the rows of the box are 16-byte aligned; a chunk is loaded from
//     wherever the row starts: a misaligned 16-byte access on a contiguous run costs the address path 5 % -- `line shifted` in
//     r05_l1_rate.txt -- not the 3x of a short row piece);
//   * the compute slots stay PAIRS of adjacent cells (all of fused_pair4_kernel's stencil and Hamiltonian code, row table included);
//     their own cells of the entering plane are read back from the staged rows (one ds_read_b64 per pair and plane).  An odd n3 leaves a
//     last slot of ONE real cell per row: its second cell holds the cell that FOLLOWS the row end (periodic: the row's cell 0; else the
//     inner neighbour, from which the ghost is formed) -- it is computed like any other, never stored, never reduced.
// y0 and the output keep the pair accesses (8 bytes at the row's alignment).
#pragma once
#include "hj_fused4v.h"

namespace hj {

typedef float F4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>
__device__ __forceinline__ F4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    return __builtin_bit_cast(F4, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, AUX));
}

// geometry of an E1 x E2 x (whole axis 3) tile with LDS row pitch P3, in cells
template <int E1, int E2, int P3> struct Flat4 {
    static constexpr int W = HJ_STENCIL, VP = HJ_VPAD;
    static constexpr int LS2 = P3, LS1 = P3 * (E2 + 2 * W);
    static constexpr int PLANE = LS1 * (E1 + 2 * W);          // cells of one LDS box
    static constexpr int NR = E1 * E2;                        // own rows
    static constexpr int NH = 2 * W * (E1 + E2);              // halo rows: 2W x E2 of axis 1, then 2W x E1 of axis 2
    static constexpr int STAGE = NR * P3;                     // cells of one buffer of staged own rows
    static constexpr int ROWF = 8;
    // write-only scratch: the W x W corner rows of a box (both halo layers at once) hold no cell anybody reads -- lanes without a pad copy
    // to make write theirs there instead of being predicated off (two cells per thread)
    static constexpr int DUMP_ROWS = W * W, DUMP_PER_ROW = 128;
    static_assert(P3 >= DUMP_PER_ROW, "corner rows hold the scratch cells");
    static_assert(P3 % 4 == 0 && VP % 4 == 0, "16-byte aligned LDS rows");
    // largest n3 a row of the box holds: cells -3 .. n3 + 3 are read by the (dummy) second cell of an odd row's last slot
    static constexpr int N3MAX = P3 - VP - 4;
    // spare cells of a staged row (behind everything a chunk can write): the pair (cell n3 - 1, cell 0) of an odd periodic row
    static constexpr int TAILP = P3 - 4;
    static_assert(TAILP >= VP + 4 * ((N3MAX + 3) / 4) && TAILP % 2 == 0, "the spare pair lies behind the last chunk of the longest row");
};

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int P3, int OCC, bool PG, int MODE>
__global__ __launch_bounds__(NT, OCC) void fused_flat4_kernel(const T* __restrict__ y, const T* __restrict__ y0,
                                                              T* __restrict__ out, const FusedArgs<T, 4> A) {
    constexpr int ND = 4, LA = 3, PD = 2, W = HJ_STENCIL, VP = HJ_VPAD;
    static_assert(HAM::ND == 4 && sizeof(T) == 4, "4-D fp32");
    static_assert(SCHEME == HJ_WENO5_ASSHIPPED || SCHEME == HJ_ENO2 || SCHEME == HJ_ENO2_FAST, "light stencils");
    using G = Flat4<E1, E2, P3>;
    constexpr int LS1 = G::LS1, LS2 = G::LS2, PLANE = G::PLANE, NR = G::NR, NH = G::NH, STAGE = G::STAGE;
    constexpr bool GEN = (MODE == 0);
    constexpr bool NP = np_order(SCHEME);
    constexpr bool ROWS = ham_has_rows<HAM>::value;
    constexpr int RAX = RowAxis<HAM, ROWS>::value;
    constexpr int ER = RAX == 1 ? E1 : E2;
    static_assert(!ROWS || RAX == 1 || RAX == 2, "rows along plane axis 1 or 2");
    using V = typename Pair<T>::V;
    const bool use_y0 = GEN ? (A.use_y0 != 0) : (MODE == 2);
    extern __shared__ __align__(16) unsigned char hj_smem[];
    T* const lds = reinterpret_cast<T*>(hj_smem + 512);       // two boxes
    T* const stage = lds + 2 * PLANE;                         // two buffers of staged own rows
    T* const rowtab = stage + 2 * STAGE;                      // ROWS: (planes of the chunk) x E_row x ROWF
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");
    static_assert(2 * NT <= G::DUMP_ROWS * G::DUMP_PER_ROW, "two scratch cells per thread");

    asm volatile("" ::"s"(A.nblocks), "s"(A.ntiles), "s"(A.blocks_per_xcd), "s"(A.ntile[1]), "s"(A.ntile[2]), "s"(A.n[0]),
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
    org[3] = 0;
    int p_begin, p_end;
    chunk_planes(A, chunk_id, p_begin, p_end);
    auto clamp_q = [&](int p) { return min(max(p, p_begin - W), p_end + W - 1); };
    auto clamp_c = [&](int p) { return min(max(p, p_begin), p_end - 1); };
    const int tid = threadIdx.x;
    const int n3 = A.n[3];
    const int HALFP = (n3 + 1) >> 1;                  // pair slots of a row (an odd row ends in a slot of one real cell)
    const int CH = (n3 + 3) >> 2;                     // 16-byte chunks of a row
    const int nslots = NR * HALFP;
    const bool odd3 = (n3 & 1) != 0;
    const bool per3 = !PG || A.bc[3] == HJ_BC_PERIODIC;

    // ---- own pair slots.  Numbering as in Tile4::row_pair: 16 consecutive lanes hold 16 consecutive pairs of ONE row; what a row has
    // beyond a multiple of 16 pairs comes last, row after row (the odd rows' single-cell slots among them: whole waves without one skip
    // their special cases)
    int own_lds[R];                 // cell index of the pair in a box, MINUS W*LS1
    int st_x[R], st_y[R];           // where the slot's two cells sit in a buffer of staged own rows
    unsigned own_g[R];
    int rowoff[R];
    int wx[R], wy[R];               // periodic axis 3: where the copy of cell .x / .y of the centre pair goes (an index from `lds`, box 0) -- for the
                                    // pairs at the row ends the row's pad, for everybody else a cell of `dump` (one unpredicated write each)
    int gmode[R];                   // extrapolated axis 3: 1 the pair is (edge, inner) of the row's left end, 2 (inner, edge) of its right end,
                                    // 3 (edge, inner-as-second-cell) of an odd row's right end; 0 none
    bool tail[R];                   // an odd row's last slot: the second cell is not a cell of the row
    typename RowCell<HAM, ROWS>::type hcell[R][2];
    const bool last_real = (tid + (R - 1) * NT) < nslots;
    {
        const int a16 = (HALFP / 16) * 16, b16 = HALFP - a16;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = min(tid + r * NT, nslots - 1);
            int t, pidx;
            const int main = NR * a16;
            if (a16 == 0) { t = s / HALFP; pidx = s - t * HALFP; }
            else if (b16 == 0 || s < main) { t = s / a16; pidx = s - t * a16; }
            else { const int e = s - main; t = e / b16; pidx = a16 + (e - t * b16); }
            const int j3 = 2 * pidx, j2 = t % E2, j1 = t / E2;
            tail[r] = odd3 && j3 == n3 - 1;
            int idx[ND] = {0, org[1] + j1, org[2] + j2, j3};
            own_lds[r] = j1 * LS1 + (j2 + W) * LS2 + (j3 + VP);
            own_g[r] = (unsigned)(idx[1] * A.pstride[1] + idx[2] * A.pstride[2] + j3) * (unsigned)sizeof(T);
            rowoff[r] = (RAX == 1 ? j1 : j2) * G::ROWF;
            st_x[r] = t * P3 + VP + j3;
            // an odd row's last slot: its second cell is the cell after the row end -- periodic: the row's cell 0, which the staging lanes put
            // NEXT to cell n3 - 1 in the row's spare cells (TAILP: one ds_read_b64 like every other slot); extrapolated (PG builds only): the
            // inner neighbour n3 - 2, from which the ghost is formed, fetched by a second read
            st_y[r] = tail[r] ? (per3 ? t * P3 + VP : st_x[r] - 1) : st_x[r] + 1;
            if (tail[r] && per3) st_x[r] = t * P3 + G::TAILP;
            int cx = 0, cy = 0;
            gmode[r] = 0;
            if (per3) {
                if (j3 == 0) { cx = n3; cy = n3; }
                else if (j3 == 2) { cx = n3; }
                if (odd3) {
                    if (j3 == n3 - 1) cx = -n3;
                    else if (j3 == n3 - 3) { cx = -n3; cy = -n3; }
                } else {
                    if (j3 == n3 - 2) { cx = -n3; cy = -n3; }
                    else if (j3 == n3 - 4) cy = -n3;
                }
            }
            // this thread's two scratch cells, in the corner rows (j1 < W, j2 < W): cell tid and cell NT + tid of them (consecutive lanes on consecutive banks)
            auto corner = [&](int d) { const int drow = d / G::DUMP_PER_ROW; return (drow / W) * LS1 + (drow % W) * LS2 + (d - drow * G::DUMP_PER_ROW); };
            const int dump0 = corner(tid), dump1 = corner(NT + tid);
            wx[r] = cx != 0 ? W * LS1 + own_lds[r] + cx : dump0;
            wy[r] = cy != 0 ? W * LS1 + own_lds[r] + 1 + cy : dump1;
            if (!per3) {
                if (j3 == 0) gmode[r] = 1;
                else if (odd3 && j3 == n3 - 1) gmode[r] = 3;
                else if (!odd3 && j3 == n3 - 2) gmode[r] = 2;
            }
            const typename HAM::Raw r0 = HAM::cell_raw(A.ham, idx);
            idx[LA] = tail[r] ? (per3 ? 0 : n3 - 2) : j3 + 1;
            const typename HAM::Raw r1 = HAM::cell_raw_next(A.ham, idx, r0);
            hcell[r][0] = RowCell<HAM, ROWS>::make(A.ham, r0, A.sc);
            hcell[r][1] = RowCell<HAM, ROWS>::make(A.ham, r1, A.sc);
        }
    }

    const unsigned plane_bytes = (unsigned)(A.stride0 * (long long)sizeof(T));
    const int p_lo = p_begin - W;
    const unsigned span = (unsigned)(p_end + W - p_lo) * plane_bytes;
    const __amdgpu_buffer_rsrc_t ry = make_srd(y + (long long)p_lo * A.stride0, span);
    const __amdgpu_buffer_rsrc_t ry0 = make_srd(y0 + (long long)p_lo * A.stride0, use_y0 ? span : 0u);
    const __amdgpu_buffer_rsrc_t rout = make_srd(out + (long long)p_lo * A.stride0, span);

    // ---- the staged own rows: chunk `tid` of the NR x CH chunks of a plane's own rows
    // (a thread beyond the NR x CH chunks shadows chunk tid mod NR*CH: the same load, the same LDS cells, the same values -- no predicate)
    unsigned os_g;
    int os_lds, os_tail;            // os_tail: where this lane's share of the row's (cell n3 - 1, cell 0) pair goes, or -1
    int os_e = 0;
    {
        const int h = tid % (NR * CH);
        const int t = h / CH, ck = h - t * CH;
        os_tail = -1;
        if (odd3 && per3) {
            if (ck == CH - 1) { os_tail = t * P3 + G::TAILP; os_e = (n3 - 1) & 3; }
            else if (ck == 0) { os_tail = t * P3 + G::TAILP + 1; os_e = 0; }
        }
        const int j2 = t % E2, j1 = t / E2;
        os_g = (unsigned)((org[1] + j1) * A.pstride[1] + (org[2] + j2) * A.pstride[2] + 4 * ck) * (unsigned)sizeof(T);
        os_lds = t * P3 + VP + 4 * ck;
    }
    // axis-0 plane p (possibly a ghost / wrapped plane) of the own rows, as one chunk per lane
    auto load_stage = [&](int p) -> F4 {
        const bool direct = (p >= 0 || A.halo_lo) && (p < A.n[0] || A.halo_hi);
        if (direct) return buf_load4<HJ_P4_AUX_OWN>(ry, os_g, (unsigned)(p - p_lo) * plane_bytes);
        const PlaneSrc<T> s = plane_src<T, ND>(A, p);
        const __amdgpu_buffer_rsrc_t rb = make_srd(y + s.off, plane_bytes);
        const F4 e = buf_load4(rb, os_g, 0u);
        if (!s.ghost) return e;
        const __amdgpu_buffer_rsrc_t ri = make_srd(y + s.off_in, plane_bytes);
        const F4 in = buf_load4(ri, os_g, 0u);
        F4 gv;
        gv.x = ghost_value<T>(e.x, in.x, s.km); gv.y = ghost_value<T>(e.y, in.y, s.km);
        gv.z = ghost_value<T>(e.z, in.z, s.km); gv.w = ghost_value<T>(e.w, in.w, s.km);
        return gv;
    };
    auto write_stage = [&](T* sb, const F4& f) {
        *reinterpret_cast<F4*>(sb + os_lds) = f;
        if (os_tail >= 0) sb[os_tail] = os_e == 0 ? f.x : (os_e == 1 ? f.y : (os_e == 2 ? f.z : f.w));
    };
    auto read_own = [&](const T* sb, V* dst) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            V v = *reinterpret_cast<const V*>(sb + st_x[r]);
            if constexpr (PG) {
                if (tail[r] && !per3) v.y = sb[st_y[r]];
            }
            dst[r] = v;
        }
    };
    auto load_y0 = [&](int p, V* dst) {
        if (use_y0) {
            const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2<HJ_P4_AUX_Y0>(ry0, own_g[r], so, T());
        }
    };

    // ---- the axis-0 queue: the own rows of the seven planes p_begin - 3 .. p_begin + 3, staged through the (still unused) boxes
    T q[R][2][8];
    {
        static_assert(7 * STAGE <= 2 * PLANE, "the boxes hold the seven staged planes of the prologue");
        F4 f[7];
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) f[jj] = load_stage(p_begin - 3 + jj);
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) write_stage(lds + jj * STAGE, f[jj]);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) {
            V tmp[R];
            read_own(lds + jj * STAGE, tmp);
#pragma unroll
            for (int r = 0; r < R; ++r) { q[r][0][jj] = tmp[r].x; q[r][1][jj] = tmp[r].y; }
        }
        __syncthreads();          // the boxes are rewritten from the first iteration on
    }

    // ---- halo rows of the centre plane: NH rows x CH chunks, KH chunk slots per thread (consecutive lanes: consecutive chunks of a row)
    constexpr int KHMAX = 4;
    static_assert((long long)NH * ((G::N3MAX + 3) / 4) <= (long long)KHMAX * NT, "halo chunk slots");
    int hc_lds[KHMAX];
    unsigned hc_src[KHMAX];
    int hc_dlt[KHMAX];
    T hc_km[KHMAX];
    bool tile_ghost = false;
    {
        const int nhc = NH * CH;
#pragma unroll
        for (int k = 0; k < KHMAX; ++k) {
            int h = tid + k * NT;
            if (h >= nhc) h -= nhc * (h / nhc);      // a slot beyond the NH x CH chunks shadows another chunk: same load, same LDS cells, same values
            const int hrow = h / CH, ck = h - hrow * CH;
            int lo, g, dlt = 0;
            T km = T(0);
            auto wrapd = [&](int d, int& gi) {
                const int nd = A.n[d];
                if (gi < 0) {
                    if (!PG || A.bc[d] == HJ_BC_PERIODIC) gi += nd;
                    else { km = T(-gi) * A.km[d]; dlt = A.pstride[d]; gi = 0; }
                } else if (gi >= nd) {
                    if (!PG || A.bc[d] == HJ_BC_PERIODIC) gi -= nd;
                    else { km = T(gi - nd + 1) * A.km[d]; dlt = -A.pstride[d]; gi = nd - 1; }
                }
            };
            if (hrow < 2 * W * E2) {    // axis 1: (layer, j2)
                const int lay = hrow / E2, j2 = hrow - lay * E2;
                const int jd = lay < W ? lay - W : E1 + lay - W;
                int gi = org[1] + jd;
                wrapd(1, gi);
                lo = (jd + W) * LS1 + (j2 + W) * LS2 + VP + 4 * ck;
                g = gi * A.pstride[1] + (org[2] + j2) * A.pstride[2] + 4 * ck;
            } else {                    // axis 2: (layer, j1)
                const int hh = hrow - 2 * W * E2;
                const int lay = hh / E1, j1 = hh - lay * E1;
                const int jd = lay < W ? lay - W : E2 + lay - W;
                int gi = org[2] + jd;
                wrapd(2, gi);
                lo = (j1 + W) * LS1 + (jd + W) * LS2 + VP + 4 * ck;
                g = (org[1] + j1) * A.pstride[1] + gi * A.pstride[2] + 4 * ck;
            }
            hc_lds[k] = lo;
            hc_src[k] = (unsigned)g * (unsigned)sizeof(T);
            hc_dlt[k] = dlt * (int)sizeof(T);
            hc_km[k] = km;
        }
        if constexpr (PG) {
            tile_ghost = (A.bc[1] != HJ_BC_PERIODIC && (org[1] < W || org[1] + E1 + W > A.n[1])) ||
                         (A.bc[2] != HJ_BC_PERIODIC && (org[2] < W || org[2] + E2 + W > A.n[2]));
        }
    }
    struct Halo { F4 p[KHMAX]; F4 pi[KHMAX]; };
    auto load_halo = [&](int p, Halo& h) {
        const unsigned so = (unsigned)(p - p_lo) * plane_bytes;
#pragma unroll
        for (int k = 0; k < KHMAX; ++k) h.p[k] = buf_load4<HJ_P4_AUX_HALO>(ry, hc_src[k], so);      // (unconditional: a predicated load makes the compiler drain vmcnt at the join)
        if constexpr (PG) {
            if (tile_ghost) {
#pragma unroll
                for (int k = 0; k < KHMAX; ++k) {
                    h.pi[k] = F4{T(0), T(0), T(0), T(0)};
                    if (hc_dlt[k] != 0) h.pi[k] = buf_load4(ry, hc_src[k] + (unsigned)hc_dlt[k], so);
                }
            }
        }
    };
    auto park_halo = [&](T* buf, const Halo& h) {
        if (PG && tile_ghost) {
#pragma unroll
            for (int k = 0; k < KHMAX; ++k) {
                const F4 e = h.p[k], in = h.pi[k];
                F4 gv;
                gv.x = ghost_value(e.x, in.x, hc_km[k]); gv.y = ghost_value(e.y, in.y, hc_km[k]);
                gv.z = ghost_value(e.z, in.z, hc_km[k]); gv.w = ghost_value(e.w, in.w, hc_km[k]);
                *reinterpret_cast<F4*>(buf + hc_lds[k]) = gv;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KHMAX; ++k)
                *reinterpret_cast<F4*>(buf + hc_lds[k]) = h.p[k];
        }
    };

    Halo hal[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int k = 0; k < KHMAX; ++k) { hal[s].pi[k] = F4{T(0), T(0), T(0), T(0)}; hal[s].p[k] = F4{T(0), T(0), T(0), T(0)}; }
        load_halo(clamp_c(p_begin + s), hal[s]);
    }
    // own rows in flight: osf[u & 1] holds plane p + 4 when iteration p begins (staged, read back and appended to the queue in that iteration)
    F4 osf[2];
    osf[0] = load_stage(clamp_q(p_begin + 4));
    osf[1] = F4{T(0), T(0), T(0), T(0)};
    V y0s[2][R];
    typename HAM::Plane pls[PD];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int r = 0; r < R; ++r) { y0s[s][r].x = T(0); y0s[s][r].y = T(0); }
        load_y0(clamp_c(p_begin + s), y0s[s]);
    }
#pragma unroll
    for (int s = 0; s < PD; ++s)
        if constexpr (!ROWS) pls[s] = HAM::plane(A.ham, clamp_c(p_begin + s), A.sc);

    if constexpr (ROWS) {
        const int nent = (p_end - p_begin) * ER;
        for (int e = tid; e < nent; e += NT) {
            const int pi = e / ER, j = e - pi * ER;
            const typename HAM::Row rw = HAM::row_at(A.ham, p_begin + pi, org[RAX] + j, A.sc);
            T* dst = rowtab + e * G::ROWF;
            F4 lo4 = {rw.a1, rw.b1, rw.c1, rw.a3};
            V hi2;
            hi2.x = rw.b3; hi2.y = rw.c3;
            *reinterpret_cast<F4*>(dst) = lo4;
            *reinterpret_cast<V*>(dst + 4) = hi2;
        }
    }
    T amax[ND][2];
#pragma unroll
    for (int d = 0; d < ND; ++d) { amax[d][0] = Lim<T>::lowest; amax[d][1] = Lim<T>::lowest; }

    auto stencil = [&](const T* v, const T* K, T& pcv, T& hdv) { upwind_cd<SCHEME, T>(v, K, T(0), WenoK<T>{T(0), T(0)}, pcv, hdv); };
    // the as-shipped WENO5 of BOTH cells of a pair, written on 2-vectors: upwind_cd's expressions element by element (same contractions, same
    // bits) -- handed to the compiler as packed operations instead of hoping that it pairs two scalar copies (it de-paired a third of them)
    constexpr bool VST = SCHEME == HJ_WENO5_ASSHIPPED;
    auto stencil2 = [&](const V* v, V& pcv, V& hdv) {
        const V D1 = v[4] - v[2], D2 = v[5] - v[1], D3 = v[6] - v[0];
        const V S1 = v[4] + v[2], S2 = v[5] + v[1], S3 = v[6] + v[0];
        pcv = T(45) * D1 + (T(-9) * D2 + D3);
        hdv = T(15) * S1 + (T(-6) * S2 + (S3 + T(-20) * v[3]));
    };
    auto body = [&](auto off_tag, int p, F4& os_c, F4& os_n, Halo& hal_c, V* y0_c, typename HAM::Plane& pl_c) {
        constexpr int OFF = decltype(off_tag)::value;          // window [OFF, OFF + 7) of the queue; box OFF; stage buffer OFF
        T* const buf = lds + OFF * PLANE;
        T* const sb = stage + OFF * STAGE;
        os_n = load_stage(clamp_q(p + 5));
        // the centre plane: every pair from the queue into the box, the cells at the row ends into the row's pad as well
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < R - 1 || last_real) {
                V c2;
                c2.x = q[r][0][3 + OFF];
                c2.y = q[r][1][3 + OFF];
                T* const ctr = buf + W * LS1 + own_lds[r];
                // (an odd row's last slot writes its second cell too: periodic -> the row's cell 0, the very value the pad cell gets from the
                //  owner of cell 0; extrapolated -> overwritten by this lane's own ghost write below)
                *reinterpret_cast<V*>(ctr) = c2;
                if (!PG || per3) {
                    buf[wx[r]] = c2.x;
                    buf[wy[r]] = c2.y;
                }
                if constexpr (PG) {
                    const int gm = gmode[r];
                    if (gm != 0) {
                        const T e = gm == 2 ? c2.y : c2.x, in = gm == 2 ? c2.x : c2.y;
                        const int at = gm == 1 ? 0 : (gm == 2 ? 1 : 0);          // where the edge cell sits in the pair
                        const int dir = gm == 1 ? -1 : 1;
#pragma unroll
                        for (int k = 1; k <= W; ++k) ctr[at + dir * k] = ghost_value(e, in, T(k) * A.km[LA]);
                    }
                }
            }
        park_halo(buf, hal_c);
        write_stage(sb, os_c);                      // the own rows of plane p + 4
        __syncthreads();
        const int p2 = clamp_c(p + PD);
        load_halo(p2, hal_c);
        const unsigned so_out = (unsigned)(p - p_lo) * plane_bytes;
        typename HAM::Plane pl_use = pl_c;
        if constexpr (!ROWS) pl_c = HAM::plane(A.ham, p2, A.sc);
        const int row_plane = (p - p_begin) * (ER * G::ROWF);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            T pc[2][ND], hd[2][ND];
            auto put = [&](int d, const V& pv, const V& hv) { pc[0][d] = pv.x; pc[1][d] = pv.y; hd[0][d] = hv.x; hd[1][d] = hv.y; };
            if constexpr (VST) {
                V v2[7], pv, hv;
#pragma unroll
                for (int j = 0; j < 7; ++j) { v2[j].x = q[r][0][j + OFF]; v2[j].y = q[r][1][j + OFF]; }
                stencil2(v2, pv, hv);
                put(0, pv, hv);
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c) stencil(q[r][c] + OFF, A.K[0], pc[c][0], hd[c][0]);
            }
            const T* base = buf + own_lds[r];           // = the pair's cell - W*LS1
#pragma unroll
            for (int d = 1; d < LA; ++d) {
                constexpr int LSD[3] = {0, LS1, LS2};
                V v2[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    if (j == 3) { v2[j].x = q[r][0][3 + OFF]; v2[j].y = q[r][1][3 + OFF]; continue; }
                    v2[j] = *reinterpret_cast<const V*>(base + W * LS1 + (j - 3) * LSD[d]);
                }
                if constexpr (VST) {
                    V pv, hv;
                    stencil2(v2, pv, hv);
                    put(d, pv, hv);
                } else {
                    T va[7], vb[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) { va[j] = v2[j].x; vb[j] = v2[j].y; }
                    stencil(va, A.K[d], pc[0][d], hd[0][d]);
                    stencil(vb, A.K[d], pc[1][d], hd[1][d]);
                }
            }
            {
                T w[8];
                const T* ctr = base + W * LS1;
                w[0] = ctr[-3];
                const V l2 = *reinterpret_cast<const V*>(ctr - 2);
                w[1] = l2.x; w[2] = l2.y;
                w[3] = q[r][0][3 + OFF];
                w[4] = q[r][1][3 + OFF];
                if constexpr (PG) {      // an odd row's last cell at an extrapolated end: its right neighbour is a ghost, not the slot's second cell
                    if (gmode[r] == 3) w[4] = ghost_value(w[3], w[4], A.km[LA]);
                }
                const V r2 = *reinterpret_cast<const V*>(ctr + 2);
                w[5] = r2.x; w[6] = r2.y;
                w[7] = ctr[4];
                if constexpr (VST) {
                    V v2[7], pv, hv;
#pragma unroll
                    for (int j = 0; j < 7; ++j) { v2[j].x = w[j]; v2[j].y = w[j + 1]; }
                    stencil2(v2, pv, hv);
                    put(LA, pv, hv);
                } else {
                    stencil(w, A.K[LA], pc[0][LA], hd[0][LA]);
                    stencil(w + 1, A.K[LA], pc[1][LA], hd[1][LA]);
                }
            }
            V o2;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                T alpha[ND], ydot;
                if constexpr (ROWS) {
                    const T* rp = rowtab + row_plane + rowoff[r];
                    const F4 lo4 = *reinterpret_cast<const F4*>(rp);
                    const V hi2 = *reinterpret_cast<const V*>(rp + 4);
                    typename HAM::Row rw;
                    rw.a1 = lo4.x; rw.b1 = lo4.y; rw.c1 = lo4.z; rw.a3 = lo4.w; rw.b3 = hi2.x; rw.c3 = hi2.y;
                    ydot = lf_ydot_row<NP, HAM>(A.ham, hcell[r][c], rw, A.sc, pc[c], hd[c], alpha);
                } else {
                    ydot = lf_ydot<NP, HAM>(A.ham, hcell[r][c], pl_use, A.sc, pc[c], hd[c], alpha);
                }
                // (the second cell of an odd row's last slot carries the constants of a REAL cell -- the row's cell 0 or n3 - 2 -- and alpha does not
                //  depend on the costate: its alpha is that cell's, the maxima are the same with or without it)
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
            if (r < R - 1 || last_real) {
                if (tail[r]) buf_store<HJ_P4_AUX_OUT>(o2.x, rout, own_g[r], so_out);
                else buf_store2_aux<HJ_P4_AUX_OUT>(o2, rout, own_g[r], so_out);
            }
        }
        load_y0(clamp_c(p + 2), y0_c);
        V own_c[R];
        read_own(sb, own_c);            // plane p + 4 of the own cells, staged before this iteration's barrier
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

    for (int p = p_begin; p < p_end; p += 2) {
        body(IntTag<0>(), p, osf[0], osf[1], hal[0], y0s[0], pls[0]);
        if (p + 1 < p_end) body(IntTag<1>(), p + 1, osf[1], osf[0], hal[1], y0s[1], pls[1]);
    }

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
        __syncthreads();
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
