// Two RK stages in ONE launch with TWO ADJACENT CELLS PER LANE (gfx950; round 3).
//
// Same scheme as fused12_kernel (hj_fused12.h: stage 1 on plane q for the tile plus a 3-cell ring, stage 2 on
// plane q-3 for the tile, y1 never leaves the CU), same per-cell arithmetic -- the results are bitwise those of
// the two unfused launches -- but the access pattern of fused_pair_kernel (hj_fusedv.h).  Round 2 found the
// one-cell-per-lane version bound by LDS instruction issue (SQ_ACTIVE_INST_LDS + bank conflicts ~ 90 % of the CU
// cycles: 35 ds_read_b64 per output cell, issued as ds_read2_b64 at half rate).  Here
//   * an "A slot" is a PAIR of cells (j, j+1), j even in tile coordinates, of T1 = tile + ring: 16-byte global
//     loads / stores, one ds_write_b128 per pair into the y and y1 boxes;
//   * the stencils of a pair read their neighbours as pairs: the other plane axis 6 ds_read_b128, the contiguous
//     axis [b64][b128][own][b128][b64], the axis-0 stencil of stage 2 6 ds_read_b128 from the y1 ring:
//     96 LDS-array cycles per output pair (both stages) against ~280 for two single cells;
//   * the ring along the contiguous axis is 4 cells (2 pairs) wide instead of 3: the outer column only carries y
//     (its y1 is never read), so the LDS rows are padded by 8 (y box) and 4 (y1 box) cells and every pair
//     starts 16-byte aligned.
// "H slots" (the deep halo, the corner blocks and the stand-ins for ghost positions of an extrapolated boundary)
// stay single cells, exactly as in fused12_kernel.  The host only offers tilings in which no pair straddles a
// domain edge (make_tiling12v), so a pair is in the domain or outside it as a whole.
#pragma once
#include "hj_fused12.h"
#include "hj_fusedv.h"

namespace hj {

#ifdef HJ_F12_STAMP
#define HJ_F12_ST(k) { const unsigned long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_last; st_last = now_; }
#else
#define HJ_F12_ST(k)
#endif

#ifndef HJ_F12_STAGGER
#define HJ_F12_STAGGER 0
#endif
// HJ_F12_ABLATE (tuning builds only, WRONG results): 1 no barrier, 2 no LDS stencil reads, 4 no global loads in the loop
#ifndef HJ_F12_ABLATE
#define HJ_F12_ABLATE 0
#endif
#if HJ_F12_ABLATE && !defined(HJ_TUNE_BUILD)
#error "HJ_F12_ABLATE gives wrong results: tuning builds only"
#endif

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC>
__global__ __launch_bounds__(NT, OCC) void fused12_pair_kernel(const T* __restrict__ y, T* __restrict__ out,
                                                               const Fused12Args<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr int LA = ND - 1;
    constexpr int W = HJ_STENCIL;
    constexpr int PYL = 8, PWL = 4;             // left pads (cells) of a row of the y / y1 box: even
    constexpr bool NP = np_order(SCHEME);
    using V = typename Pair<T>::V;
    static_assert(ND == 2 || ND == 3, "fused12: 2-D and 3-D grids");
    static_assert(SCHEME != HJ_WENO5, "fused12: the intended WENO5 needs a global reduction between the stages");
    extern __shared__ __align__(16) unsigned char hj_smem[];
    double (*red)[ND] = reinterpret_cast<double (*)[ND]>(hj_smem);
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");
    T* ldsY = reinterpret_cast<T*>(hj_smem + 512);

    // ---- XCD-aware block order (as fused_substep_kernel)
    const int b = blockIdx.x;
    const int L = (b & 7) * A.blocks_per_xcd + (b >> 3);
    if (L >= A.nblocks) return;
    const int chunk_id = L / A.ntiles;
    int rem = L - chunk_id * A.ntiles;
    int org[ND];
    org[0] = 0;
#pragma unroll
    for (int d = ND - 1; d >= 1; --d) {
        const int qd = rem / A.ntile[d];
        org[d] = min((rem - qd * A.ntile[d]) * A.E[d], A.n[d] - A.E[d]);
        rem = qd;
    }
    const int p_begin = A.plane_begin + chunk_id * A.chunk;
    const int p_end = min(p_begin + A.chunk, A.plane_end);
    const int n0 = A.n[0];

    // ---- LDS boxes.  y: [-2W, E+2W) on axis 1, [-8, E+8) on the contiguous axis; y1: [-W, E+W) and [-4, E+4)
    auto padY = [](int d) { return d == LA ? PYL : 2 * W; };
    auto padW = [](int d) { return d == LA ? PWL : W; };
    int lsY[ND], lsW[ND];
    lsY[LA] = lsW[LA] = 1;
#pragma unroll
    for (int d = ND - 2; d >= 1; --d) {
        lsY[d] = A.E[LA] + 2 * PYL;
        lsW[d] = A.E[LA] + 2 * PWL;
    }
    const int ybox = (ND == 3) ? lsY[1] * (A.E[1] + 4 * W) : A.E[LA] + 2 * PYL;
    const int wbox = (ND == 3) ? lsW[1] * (A.E[1] + 2 * W) : A.E[LA] + 2 * PWL;
    T* ldsW = ldsY + 2 * ybox;
    const int ls1Y = (ND == 3) ? lsY[1] : 0, ls1W = (ND == 3) ? lsW[1] : 0;     // row strides of the two boxes
    const int E1 = (ND == 3) ? A.E[1] : 1;      // rows of the tile
    const int E2 = A.E[LA], E2p = E2 >> 1;      // cells / pairs of a tile row
    const int n_int = E1 * E2;                  // interior cells
    const int n_intp = E1 * E2p;                // interior pairs
    int area[ND];
#pragma unroll
    for (int d = 1; d < ND; ++d) area[d] = n_int / A.E[d];

    const int tid = threadIdx.x;

    // ---- A slots (pairs of T1): interior pairs first, then (3-D) the 2W ring rows of axis 1 over the tile's
    // columns, then the 4 ring columns (2 pairs) either side of the contiguous axis over the tile's rows
    int a_oy[R], a_ow[R];
    unsigned a_g[R];
    bool a_in[R], a_w0[R], a_w1[R], a_int[R];
    typename HAM::Cell hcell[R][2];
    {
        const int nring1 = (ND == 3) ? 2 * W * E2p : 0;
        const int nA = n_intp + nring1 + 4 * E1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int a = tid + r * NT;
            bool in = a < nA;
            if (!in) a = 0;
            a_int[r] = in && a < n_intp;
            int j1 = 0, j2 = 0;
            if (a < n_intp) {
                j1 = a / E2p;
                j2 = 2 * (a - j1 * E2p);
            } else if (a < n_intp + nring1) {
                const int hh = a - n_intp;
                const int lay = hh / E2p;
                j2 = 2 * (hh - lay * E2p);
                j1 = (lay < W) ? (lay - W) : (E1 + lay - W);
            } else {
                // the pair column runs fastest: consecutive lanes fetch the two ring pairs either side of ONE row
                const int hh = a - n_intp - nring1;
                j1 = hh >> 2;
                const int lay = hh & 3;                   // 0..3: pair columns -4, -2, E2, E2+2
                j2 = (lay < 2) ? (2 * lay - 4) : (E2 + 2 * (lay - 2));
            }
            int oy = 0, ow = 0, g = 0;
            int idx[ND];
            idx[0] = 0;
            if (ND == 3) {
                int gi = org[1] + j1;
                const int nd = A.n[1];
                if (gi < 0 || gi >= nd) {
                    if (A.bc[1] == HJ_BC_PERIODIC) { gi %= nd; if (gi < 0) gi += nd; }
                    else { in = false; gi = gi < 0 ? 0 : nd - 1; }        // ghost row: H slots stand in
                }
                idx[1] = gi;
                oy += (j1 + 2 * W) * lsY[1];
                ow += (j1 + W) * lsW[1];
                g += gi * A.pstride[1];
            }
            {
                int gi = org[LA] + j2;                                     // first cell of the pair
                const int nd = A.n[LA];
                // no pair straddles an edge (the host rejects such tilings): gi in [0, nd) implies gi + 1 < nd
                if (gi < 0 || gi >= nd) {
                    if (A.bc[LA] == HJ_BC_PERIODIC) { gi %= nd; if (gi < 0) gi += nd; }
                    else { in = false; gi = gi < 0 ? 0 : nd - 2; }            // ghost columns: H slots stand in
                }
                idx[LA] = gi;
                oy += j2 + PYL;
                ow += j2 + PWL;
                g += gi * A.pstride[LA];
            }
            a_in[r] = in;
            a_int[r] = a_int[r] && in;
            // the pad columns -4 and E2+3 only carry y
            a_w0[r] = in && j2 >= -W;
            a_w1[r] = in && j2 + 1 < E2 + W;
            a_oy[r] = oy;
            a_ow[r] = ow;
            a_g[r] = (unsigned)g * (unsigned)sizeof(T);
            hcell[r][0] = HAM::cell(A.ham, idx, A.sc);
            idx[LA] = min(idx[LA] + 1, A.n[LA] - 1);
            hcell[r][1] = HAM::cell(A.ham, idx, A.sc);
        }
    }

    // ---- H slots (single cells).  Index space: per plane axis 4W layers (-2W..-1, E..E+2W-1) over the tile's
    // extent on the other axes, then (ND = 3) the four W x W corner blocks.  A ring-layer position that lies in
    // the domain belongs to an A slot; on the contiguous axis the A slots reach one column further (-4, E2+3).
    int h_oy[KH], h_dlt[KH], h_ow[KH], h_we[KH], h_wd[KH];
    unsigned h_src[KH];
    T h_km[KH];
    bool h_act[KH], h_fix[KH];
    {
        int hbase[ND + 1];
        hbase[1] = 0;
#pragma unroll
        for (int d = 1; d < ND; ++d) hbase[d + 1] = hbase[d] + 4 * W * area[d];
        const int ncorner = (ND == 3) ? 4 * W * W : 0;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            int h = tid + k * NT;
            bool act = h < hbase[ND] + ncorner;
            if (!act) h = 0;
            int j[ND];
            bool ringlay[ND], alay[ND], outside[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) { j[d] = 0; ringlay[d] = false; alay[d] = false; outside[d] = false; }
            bool corner = false;
            if (h < hbase[ND]) {
#pragma unroll
                for (int d = 1; d < ND; ++d) {
                    if (h < hbase[d] || h >= hbase[d + 1]) continue;
                    const int hh = h - hbase[d];
                    int lay, c;
                    if (d == LA) {          // contiguous axis: the layer runs fastest (6 + 6 cells of one row: two cache lines)
                        c = hh / (4 * W);
                        lay = hh - c * (4 * W);
                    } else {
                        lay = hh / area[d];
                        c = hh - lay * area[d];
                    }
                    j[d] = (lay < 2 * W) ? (lay - 2 * W) : (A.E[d] + lay - 2 * W);
                    outside[d] = true;
                    ringlay[d] = (j[d] >= -W && j[d] < A.E[d] + W);
                    // layers the A slots cover when the position is in the domain
                    alay[d] = (d == LA) ? (j[d] >= -W - 1 && j[d] < A.E[d] + W + 1) : ringlay[d];
#pragma unroll
                    for (int e = ND - 1; e >= 1; --e) {
                        if (e == d) continue;
                        const int qe = c / A.E[e];
                        j[e] = c - qe * A.E[e];
                        c = qe;
                    }
                }
            } else if (ND == 3) {
                corner = true;
                int c = h - hbase[ND];
                const int blk = c / (W * W);
                c -= blk * (W * W);
                const int c1 = c / W, c2 = c - c1 * W;
                j[1] = (blk & 1) ? A.E[1] + c1 : c1 - W;
                j[LA] = (blk & 2) ? A.E[LA] + c2 : c2 - W;
                outside[1] = outside[LA] = true;
                ringlay[1] = ringlay[LA] = true;
                alay[1] = alay[LA] = true;
            }
            int oy = 0, ow = 0, g = 0, dlt = 0, we = 0, wd = 0, nghost = 0;
            T km = T(0);
            bool fix = false;
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                int gi = org[d] + j[d];
                const int nd = A.n[d];
                oy += (j[d] + padY(d)) * lsY[d];
                int jw = j[d];                                // y1-box coordinate of the (edge) cell
                if (gi < 0 || gi >= nd) {
                    if (A.bc[d] == HJ_BC_PERIODIC) {
                        gi %= nd; if (gi < 0) gi += nd;
                        if (outside[d] && alay[d] && !corner) act = false;     // an A slot loads the wrapped pair
                    } else {
                        const int kk = gi < 0 ? -gi : gi - nd + 1;
                        if (kk > W) act = false;              // deeper than any stencil reaches
                        ++nghost;
                        km = T(kk) * A.km[d];
                        dlt = (gi < 0 ? A.pstride[d] : -A.pstride[d]) * (int)sizeof(T);
                        wd = gi < 0 ? lsW[d] : -lsW[d];
                        jw = gi < 0 ? -org[d] : nd - 1 - org[d];
                        gi = gi < 0 ? 0 : nd - 1;
                        fix = ringlay[d] && !corner;
                    }
                } else if (outside[d] && alay[d] && !corner) {
                    act = false;                              // an in-domain ring cell: an A slot owns it
                }
                ow += (j[d] + padW(d)) * lsW[d];
                we += (jw + padW(d)) * lsW[d];
                g += gi * A.pstride[d];
            }
            if (nghost > 1) act = false;                      // ghost on two axes: no stencil reads it
            if (nghost == 0) { km = T(0); dlt = 0; fix = false; }
            h_act[k] = act;
            h_fix[k] = act && fix;
            h_oy[k] = oy;
            h_ow[k] = ow;
            h_we[k] = we;
            h_wd[k] = wd;
            h_src[k] = (unsigned)g * (unsigned)sizeof(T);
            h_dlt[k] = dlt;
            h_km[k] = km;
        }
    }
    bool any_ghost = false, any_fix = false;
#pragma unroll
    for (int k = 0; k < KH; ++k) { any_ghost = any_ghost || (h_act[k] && h_dlt[k] != 0); any_fix = any_fix || h_fix[k]; }
    const bool tile_ghost = __syncthreads_or(any_ghost ? 1 : 0) != 0;
    const bool tile_fix = __syncthreads_or(any_fix ? 1 : 0) != 0;

    T eps[ND];
    WenoK<T> wk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { eps[d] = T(0); wk[d].c13 = T(0); wk[d].c4 = T(0); }

    // ---- loaders: one descriptor for the whole array, the plane in the scalar offset
    const unsigned plane_bytes = (unsigned)(A.stride0 * (long long)sizeof(T));
    const __amdgpu_buffer_rsrc_t ry = make_srd(y, A.total_bytes);
    const __amdgpu_buffer_rsrc_t rout = make_srd(out, A.total_bytes);
    const bool per0 = A.bc[0] == HJ_BC_PERIODIC;
    auto wrap0 = [&](int p) { int m = p % n0; return m < 0 ? m + n0 : m; };     // prologue only (integer division)
    auto load_own = [&](int p, int pw, V* dst) {
        if ((p >= 0 && p < n0) || per0) {
            const unsigned so = (unsigned)(per0 ? pw : p) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load2(ry, a_g[r], so, T());
        } else {
            const unsigned se = (p < 0 ? 0u : (unsigned)(n0 - 1)) * plane_bytes;
            const unsigned si = (p < 0 ? 1u : (unsigned)(n0 - 2)) * plane_bytes;
            const T km0 = T(p < 0 ? -p : p - n0 + 1) * A.km[0];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const V e = buf_load2(ry, a_g[r], se, T()), in = buf_load2(ry, a_g[r], si, T());
                V gv;
                gv.x = ghost_value<T>(e.x, in.x, km0);
                gv.y = ghost_value<T>(e.y, in.y, km0);
                dst[r] = gv;
            }
        }
    };
    auto load_halo = [&](int pw, T* dst, T* dst_in) {
        const unsigned so = (unsigned)pw * plane_bytes;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            dst[k] = T(0);
            if (h_act[k]) dst[k] = buf_load(ry, h_src[k], so, T());
        }
        if (tile_ghost) {
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                dst_in[k] = T(0);
                if (h_act[k] && h_dlt[k] != 0) dst_in[k] = buf_load(ry, h_src[k] + (unsigned)h_dlt[k], so, T());
            }
        }
    };
    auto s1_plane = [&](int q) { return per0 || (q >= 0 && q < n0); };
    auto inc0 = [&](int pw) { return pw + 1 >= n0 ? pw + 1 - n0 : pw + 1; };     // next plane, wrapped

    // ---- prologue: yq[r][c][j] <-> plane q-3+j of cell c of slot r
    const int q0 = p_begin - W, q1 = p_end + W;
    T yq[R][2][7];
    int pw_load = wrap0(q0 - W);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        V tmp[R];
        load_own(q0 - W + j, pw_load, tmp);
        pw_load = inc0(pw_load);
#pragma unroll
        for (int r = 0; r < R; ++r) { yq[r][0][j] = tmp[r].x; yq[r][1][j] = tmp[r].y; }
    }
    // Prefetch: ONE register set each (256 VGPRs hold the 7-deep queue of two pairs, their Hamiltonian constants and
    // the arithmetic of a pair; a second set spilled).  The own cells of plane q+4 and the H values of plane q+1 are
    // requested right after the barrier of iteration q; the former join the queue at the end of the iteration, the
    // latter are staged at the start of iteration q+1.  An iteration is two stage evaluations of the whole tile
    // (several microseconds): it covers the memory latency.
    V own[R];
    T hal[KH], hin[KH];
#pragma unroll
    for (int r = 0; r < R; ++r) { own[r].x = T(0); own[r].y = T(0); }
    int pw_h = wrap0(q0);
#pragma unroll
    for (int k = 0; k < KH; ++k) { hal[k] = T(0); hin[k] = T(0); }
    if (s1_plane(q0)) load_halo(pw_h, hal, hin);
    pw_h = inc0(pw_h);
    int qw = wrap0(q0);
    // y1 ring: sW[j] = LDS element offset of the slot that holds plane q-6+j (j = 0..6; slot = plane mod 7)
    int sW[7];
    {
        int m = (q0 - 6) % 7;
        if (m < 0) m += 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) { sW[j] = m * wbox; m = (m == 6) ? 0 : m + 1; }
    }
    int off_e = 0, off_i = 0;
    bool w_act[R], w_int[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { w_act[r] = __any(a_in[r] ? 1 : 0) != 0; w_int[r] = __any(a_int[r] ? 1 : 0) != 0; }

    double amax[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) amax[d] = -1.0e300;
    {   // alphas that are constant along the march: one max per column (interior cells)
        T pz[ND], Hz, az[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) pz[d] = T(0);
        const typename HAM::Plane plz = HAM::plane(A.ham, min(max(p_begin, 0), n0 - 1), A.sc);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                HAM::eval(A.ham, hcell[r][c], plz, A.sc, pz, Hz, az);
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (!((HAM::PLANE_DEP >> d) & 1u) && a_int[r]) amax[d] = fmax(amax[d], (double)az[d]);
            }
    }

    // The Lax-Friedrichs right-hand side of a PAIR: v0[c] = the 7 axis-0 values of cell c, in-plane values from
    // `buf` at box offset `o` (first cell of the pair, 16-byte aligned) with row stride ls1.  Same per-cell
    // expressions as fused_substep_kernel / fused_pair_kernel.
    auto lf_rhs2 = [&](const T (*v0)[7], const T* buf, int o, int ls1, const typename HAM::Cell* hc,
                       const typename HAM::Plane& pl, T* ydot, T (*alpha)[ND]) {
        T pc[2][ND], hd[2][ND];
#pragma unroll
        for (int c = 0; c < 2; ++c) upwind_cd<SCHEME, T>(v0[c], A.K[0], eps[0], wk[0], pc[c][0], hd[c][0]);
        const T* base = buf + o;
        if constexpr (ND == 3) {
            T va[7], vb[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if (j == 3) { va[j] = v0[0][3]; vb[j] = v0[1][3]; continue; }
#if HJ_F12_ABLATE & 2      // timing experiment: no LDS stencil reads (results are wrong)
                va[j] = v0[0][j]; vb[j] = v0[1][j]; continue;
#endif
                const V n2 = *reinterpret_cast<const V*>(base + (j - 3) * ls1);
                va[j] = n2.x;
                vb[j] = n2.y;
            }
            upwind_cd<SCHEME, T>(va, A.K[1], eps[1], wk[1], pc[0][1], hd[0][1]);
            upwind_cd<SCHEME, T>(vb, A.K[1], eps[1], wk[1], pc[1][1], hd[1][1]);
        }
        {   // the contiguous axis: cells j-3 .. j+4 = [b64][b128][own pair][b128][b64]
            T w[8];
#if HJ_F12_ABLATE & 2
            w[0] = v0[0][0]; w[1] = v0[0][1]; w[2] = v0[0][2]; w[3] = v0[0][3]; w[4] = v0[1][3]; w[5] = v0[1][4]; w[6] = v0[1][5]; w[7] = v0[1][6];
#else
            w[0] = base[-3];
            const V l2 = *reinterpret_cast<const V*>(base - 2);
            w[1] = l2.x; w[2] = l2.y;
            w[3] = v0[0][3]; w[4] = v0[1][3];
            const V r2 = *reinterpret_cast<const V*>(base + 2);
            w[5] = r2.x; w[6] = r2.y;
            w[7] = base[4];
#endif
            upwind_cd<SCHEME, T>(w, A.K[LA], eps[LA], wk[LA], pc[0][LA], hd[0][LA]);
            upwind_cd<SCHEME, T>(w + 1, A.K[LA], eps[LA], wk[LA], pc[1][LA], hd[1][LA]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) ydot[c] = lf_ydot<NP, HAM>(A.ham, hc[c], pl, A.sc, pc[c], hd[c], alpha[c]);
    };

    V o2p[R];                   // stage-2 results of the previous iteration, not yet stored
#pragma unroll
    for (int r = 0; r < R; ++r) { o2p[r].x = T(0); o2p[r].y = T(0); }
    bool st_pending = false;
    unsigned st_so = 0u;
    // one iteration: stage 1 on plane q, stage 2 on plane q - W
#ifdef HJ_F12_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
    auto body = [&](int q) {
#ifdef HJ_F12_STAMP
        st_last = __builtin_readcyclecounter();
#endif
        V* own_c = own;
        T* hal_c = hal;
        T* hin_c = hin;
        const bool s1 = s1_plane(q);
        T* bufY = ldsY + (q & 1) * ybox;
        T* bufWq = ldsW + sW[6];
        if (s1) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (a_in[r]) {
                    V c2;
                    c2.x = yq[r][0][3];
                    c2.y = yq[r][1][3];
                    *reinterpret_cast<V*>(bufY + a_oy[r]) = c2;
                }
            if (tile_ghost) {
#pragma unroll
                for (int k = 0; k < KH; ++k)
                    if (h_act[k]) bufY[h_oy[k]] = ghost_value(hal_c[k], hin_c[k], h_km[k]);
            } else {
#pragma unroll
                for (int k = 0; k < KH; ++k)
                    if (h_act[k]) bufY[h_oy[k]] = hal_c[k];
            }
        }
        HJ_F12_ST(0)               // phase 0: wait for the H values, stage the y plane
#if !(HJ_F12_ABLATE & 1)    // timing experiment: no barrier (results are wrong)
        __syncthreads();
#endif
        HJ_F12_ST(1)               // phase 1: barrier
#if HJ_F12_STAGGER > 0
        // the waves of a SIMD leave the barrier together and would run their LDS-read bursts and their arithmetic in
        // lock-step; the second half of the workgroup starts late, so that one wave computes while its partner reads
        if (tid >= NT / 2) __builtin_amdgcn_s_sleep(HJ_F12_STAGGER);
#endif
        // ---- every vector-memory operation of the iteration is issued HERE, right after the barrier, and none is
        // waited for before the end of the iteration (the queue rotation) or the top of the next one (the H values):
        // the results of the PREVIOUS iteration's stage 2 go out first, then the loads.  The compiler cannot count
        // the stores behind the wave-uniform stage-2 branches, so wherever it waits for a load it waits for
        // vmcnt(0); with the stores issued at the end of stage 2 (first version) that wait sat right behind them
        // and every iteration paid the store round trip (ablation, DESIGN.md 4.3: 0.36 of 1.39 ms at 513^3).
        if (st_pending) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (w_int[r] && a_int[r]) buf_store2(o2p[r], rout, a_g[r], st_so);
            st_pending = false;
        }
#if !(HJ_F12_ABLATE & 4)
        if (q + W + 1 < q1 + W) load_own(q + W + 1, pw_load, own_c);
        if (s1_plane(q + 1) && q + 1 < q1) load_halo(pw_h, hal_c, hin_c);
#endif
        pw_load = inc0(pw_load);
        pw_h = inc0(pw_h);
        // ---- y1 ghost cells of plane q-1 on extrapolated in-plane boundaries (boundary rule applied to y1)
        if (tile_fix && q - 1 >= q0 && s1_plane(q - 1)) {
            T* bw = ldsW + sW[5];
#pragma unroll
            for (int k = 0; k < KH; ++k)
                if (h_fix[k]) bw[h_ow[k]] = ghost_value(bw[h_we[k]], bw[h_we[k] + h_wd[k]], h_km[k]);
        }
        HJ_F12_ST(2)               // phase 2: stores of the previous results, loads, y1 ghost fix
        // ---- stage 1 on plane q
        V y1n[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { y1n[r].x = T(0); y1n[r].y = T(0); }
        if (s1) {
            const typename HAM::Plane pl1 = HAM::plane(A.ham, qw, A.sc);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!w_act[r]) continue;
                T ydot[2], alpha[2][ND];
                lf_rhs2(yq[r], bufY, a_oy[r], ls1Y, hcell[r], pl1, ydot, alpha);
                // the Euler stage as fused_substep_kernel forms it (ca = 0, cb = 1, no y0 operand)
                y1n[r].x = rk_stage_out<NP>(HJ_STAGE_EULER, T(0), T(1), A.dt, T(0), yq[r][0][3], ydot[0]);
                y1n[r].y = rk_stage_out<NP>(HJ_STAGE_EULER, T(0), T(1), A.dt, T(0), yq[r][1][3], ydot[1]);
                if (a_w0[r] && a_w1[r]) *reinterpret_cast<V*>(bufWq + a_ow[r]) = y1n[r];
                else if (a_w0[r]) bufWq[a_ow[r]] = y1n[r].x;
                else if (a_w1[r]) bufWq[a_ow[r] + 1] = y1n[r].y;
            }
        } else if (q >= n0) {
            // a ghost plane of y1 beyond an extrapolated axis-0 boundary (high side): boundary rule on the own column
            const T kmq = T(q - n0 + 1) * A.km[0];
            const T* be = ldsW + off_e;
            const T* bi = ldsW + off_i;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (a_int[r]) {
                    const V e = *reinterpret_cast<const V*>(be + a_ow[r]), in = *reinterpret_cast<const V*>(bi + a_ow[r]);
                    y1n[r].x = ghost_value<T>(e.x, in.x, kmq);
                    y1n[r].y = ghost_value<T>(e.y, in.y, kmq);
                    *reinterpret_cast<V*>(bufWq + a_ow[r]) = y1n[r];
                }
        }
        HJ_F12_ST(3)               // phase 3: stage 1 (all A slots)
        if (q == n0 - 1) off_e = sW[6];
        if (q == n0 - 2) off_i = sW[6];
        if (!per0 && q == 1) {
            // planes -1, -2, -3 of y1 from planes 0 (slot sW[5]) and 1 (just computed)
            const T* be = ldsW + sW[5];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!a_int[r]) continue;
                const V e0 = *reinterpret_cast<const V*>(be + a_ow[r]);
#pragma unroll
                for (int k = 1; k <= W; ++k) {
                    V gv;
                    gv.x = ghost_value<T>(e0.x, y1n[r].x, T(k) * A.km[0]);
                    gv.y = ghost_value<T>(e0.y, y1n[r].y, T(k) * A.km[0]);
                    *reinterpret_cast<V*>(ldsW + sW[5 - k] + a_ow[r]) = gv;
                }
            }
        }
        // ---- stage 2 on plane p = q - W
        const int p = q - W;
        if (p >= p_begin) {
            const typename HAM::Plane pl2 = HAM::plane(A.ham, p, A.sc);
            const unsigned so_out = (unsigned)p * plane_bytes;
            const T* bufWp = ldsW + sW[3];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!w_int[r]) continue;
                T v0[2][7];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
#if HJ_F12_ABLATE & 2
                    v0[0][j] = yq[r][0][j]; v0[1][j] = yq[r][1][j]; continue;
#endif
                    const V n2 = *reinterpret_cast<const V*>(ldsW + sW[j] + a_ow[r]);
                    v0[0][j] = n2.x;
                    v0[1][j] = n2.y;
                }
                v0[0][6] = y1n[r].x;
                v0[1][6] = y1n[r].y;
                T ydot[2], alpha[2][ND];
                lf_rhs2(v0, bufWp, a_ow[r], ls1W, hcell[r], pl2, ydot, alpha);
                V o2;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
#pragma unroll
                    for (int d = 0; d < ND; ++d)
                        if (((HAM::PLANE_DEP >> d) & 1u) && a_int[r]) amax[d] = fmax(amax[d], (double)alpha[c][d]);
                    // yq[r][c][0] is y on plane p: the y0 operand of the second stage
                    const T o = rk_stage_out<NP>(A.stage2, A.ca, A.cb, A.dt, yq[r][c][0], v0[c][3], ydot[c]);
                    if (c == 0) o2.x = o; else o2.y = o;
                }
                o2p[r] = o2;                              // stored after the next barrier (or after the loop)
            }
            st_pending = true;
            st_so = so_out;
        }
        HJ_F12_ST(4)               // phase 4: stage 2 (interior slots)
        // ---- rotate the queue and the ring
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 6; ++j) { yq[r][0][j] = yq[r][0][j + 1]; yq[r][1][j] = yq[r][1][j + 1]; }
            yq[r][0][6] = own_c[r].x;
            yq[r][1][6] = own_c[r].y;
        }
        {
            const int s0 = sW[0];
#pragma unroll
            for (int j = 0; j < 6; ++j) sW[j] = sW[j + 1];
            sW[6] = s0;
        }
        qw = inc0(qw);
        HJ_F12_ST(5)               // phase 5: wait for the own cells of plane q+4, rotate
    };

#ifdef HJ_F12_STAMP
    const unsigned long long lp_c0 = __builtin_readcyclecounter(), lp_w0 = wall_clock64();
#endif
    for (int q = q0; q < q1; ++q) body(q);
#ifdef HJ_F12_STAMP
    if (A.timing && (tid & 63) == 0) {
        unsigned long long* dst = A.timing + ((size_t)L * (NT / 64) + (tid >> 6)) * 10;
        for (int k = 0; k < 6; ++k) dst[k] = st_acc[k];
        dst[6] = __builtin_readcyclecounter() - lp_c0;
        dst[7] = wall_clock64() - lp_w0;
        dst[8] = (unsigned long long)(q1 - q0);
        dst[9] = (unsigned long long)((w_int[0] ? 1 : 0) + (R > 1 && w_int[R - 1] ? 1 : 0));
    }
#endif
    if (st_pending) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (w_int[r] && a_int[r]) buf_store2(o2p[r], rout, a_g[r], st_so);
    }

    if (A.bound) {
        // ---- CFL reduction: wavefront shuffles -> LDS -> one atomicMax per block and dim
        const int lane = tid & 63, wv = tid >> 6;
        __syncthreads();
    #pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double m = wave_max(amax[d]) / (double)A.sc[d];
            if (lane == 0) red[wv][d] = m;
        }
        __syncthreads();
        if (tid < ND) {
            double m = red[0][tid];
            for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w][tid]);
            if (m > -1.0e299) key_max(A.bound + tid, m);
        }
    }
}

}  // namespace hj
