// Two RK stages in ONE launch (gfx950): the first two substeps of odeCFL3 / both substeps of odeCFL2
//     y1  = y + dt*L(y)                              (ode_cfl_3.py:151,  ode_cfl_2.py:151)
//     out = ca*y + cb*(y1 + dt*L(y1))                (ode_cfl_3.py:184-193: ca=3/4, cb=1/4;  ode_cfl_2.py:184-201: 1/2, 1/2)
// with L = -(H - dissipation) of termLaxFriedrichs (term_lax_friedrich.py:94-130, artificial_diss_glf.py:75-109).
// y1 never goes to HBM: 2 words per cell (1R + 1W) instead of 5 (1R+1W, 2R+1W) for the two unfused launches;
// the price is stage-1 work on a 3-cell ring around every tile and 6 more warm-up planes per chunk.
//
// Same 2.5-D blocking as fused_substep_kernel (axis 0 is the march axis), with two in-flight planes:
//   iteration q:   stage 1 on plane q       (inputs: y planes q-3..q+3 from the register queue, y plane q in LDS)
//                  stage 2 on plane p = q-3 (inputs: y1 planes p-3..p+3 and the y1 tile of plane p, all in LDS)
//   * "A slots" (R per thread): the cells of T1 = tile + 3-cell cross ring, interior cells first.  Every A
//     slot carries the 7-deep axis-0 queue of y and evaluates stage 1; interior slots also evaluate stage 2.
//   * "H slots" (KH per thread): the rest of the stage-1 footprint T0 = T1 + 3-cell cross (the deep halo
//     and the corner blocks), staged into LDS only; they also stand in for ring positions that are ghost
//     cells of an extrapolated boundary (their y comes from the boundary rule, and so does their y1:
//     addGhostExtrapolate applied to y1, add_ghost_extrapolate.py:88-110, formed from the LDS copy of y1).
//   * LDS: y planes double buffered on the (E+12)^(ND-1) box; y1 on the (E+6)^(ND-1) box in a ring of 7
//     planes (plane mod 7), which serves both the in-plane stencils of stage 2 and its axis-0 stencil --
//     a second register queue would not fit the 256-VGPR budget.
// Per-cell arithmetic is the same source expression as in fused_substep_kernel, and the library is built
// with -ffp-contract=on, so out equals the two unfused launches BITWISE (tests/test_gpu_configs.py).
// Not for HJ_WENO5 (its epsilon is a global reduction over y1), not for slab halos, 2-D and 3-D grids only.
#pragma once
#include "hj_device.h"

namespace hj {

template <typename T, int ND> struct Fused12Args {
    unsigned long long* bound;    // ND keys (atomicMax)
    int n[ND];
    int bc[ND];
    T km[ND];                     // slope multiplier (+1, -1 if towardZero)
    T K[ND][HJ_NK];
    T sc[ND];
    long long stride0;            // elements per axis-0 plane
    unsigned total_bytes;         // whole array (< 4 GiB: one buffer descriptor)
    int pstride[ND];
    int E[ND];
    int ntile[ND];
    int ntiles;
    int chunk, nchunks;
    int plane_begin, plane_end;
    int nblocks, blocks_per_xcd;
    int nA, nH;                   // A / H index-space sizes (host-computed, = the formulas in the kernel)
    int stage2;                   // HJ_STAGE_RK3_HALF or HJ_STAGE_RK2_FULL: the second stage's expression
    T ca, cb, dt;
    HamTables<T> ham;
    // diagnostic build (-DHJ_F12_STAMP, HJ_TIMING_DUMP=file): per workgroup and wave 10 words -- shader cycles of the phases of
    // the plane loop summed over the iterations, then the loop's shader cycles and its 100 MHz wall-clock length
    unsigned long long* timing;
};

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC>
__global__ __launch_bounds__(NT, OCC) void fused12_kernel(const T* __restrict__ y, T* __restrict__ out,
                                                          const Fused12Args<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr int W = HJ_STENCIL;
    constexpr bool NP = np_order(SCHEME);
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

    // ---- LDS boxes: y on [-2W, E+2W), y1 on [-W, E+W) per plane axis, last axis contiguous
    int lsY[ND], lsW[ND];
    lsY[ND - 1] = lsW[ND - 1] = 1;
#pragma unroll
    for (int d = ND - 2; d >= 1; --d) {
        lsY[d] = lsY[d + 1] * (A.E[d + 1] + 4 * W);
        lsW[d] = lsW[d + 1] * (A.E[d + 1] + 2 * W);
    }
    const int ybox = lsY[1] * (A.E[1] + 4 * W), wbox = lsW[1] * (A.E[1] + 2 * W);
    T* ldsW = ldsY + 2 * ybox;
    int n_int = 1;
#pragma unroll
    for (int d = 1; d < ND; ++d) n_int *= A.E[d];
    int area[ND];
#pragma unroll
    for (int d = 1; d < ND; ++d) area[d] = n_int / A.E[d];

    const int tid = threadIdx.x;

    // ---- A slots: cells of T1, interior first, then for each plane axis its 2W ring layers
    int a_oy[R], a_ow[R];
    unsigned a_g[R];
    bool a_act[R], a_int[R];
    typename HAM::Cell hcell[R];
    {
        int abase[ND + 1];
        abase[1] = n_int;
#pragma unroll
        for (int d = 1; d < ND; ++d) abase[d + 1] = abase[d] + 2 * W * area[d];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int a = tid + r * NT;
            bool act = a < abase[ND];
            if (!act) a = 0;
            a_int[r] = act && a < n_int;
            int j[ND];
            j[0] = 0;
            if (a < n_int) {
                int c = a;
#pragma unroll
                for (int d = ND - 1; d >= 1; --d) {
                    const int qd = c / A.E[d];
                    j[d] = c - qd * A.E[d];
                    c = qd;
                }
            } else {
#pragma unroll
                for (int d = 1; d < ND; ++d) j[d] = 0;
#pragma unroll
                for (int d = 1; d < ND; ++d) {
                    if (a < abase[d] || a >= abase[d + 1]) continue;
                    const int hh = a - abase[d];
                    const int lay = hh / area[d];
                    int c = hh - lay * area[d];
                    j[d] = (lay < W) ? (lay - W) : (A.E[d] + lay - W);
#pragma unroll
                    for (int e = ND - 1; e >= 1; --e) {
                        if (e == d) continue;
                        const int qe = c / A.E[e];
                        j[e] = c - qe * A.E[e];
                        c = qe;
                    }
                }
            }
            int oy = 0, ow = 0, g = 0;
            int idx[ND];
            idx[0] = 0;
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                int gi = org[d] + j[d];
                const int nd = A.n[d];
                if (gi < 0 || gi >= nd) {
                    if (A.bc[d] == HJ_BC_PERIODIC) { gi %= nd; if (gi < 0) gi += nd; }
                    else { act = false; gi = gi < 0 ? 0 : nd - 1; }      // ghost position: an H slot stands in
                }
                idx[d] = gi;
                oy += (j[d] + 2 * W) * lsY[d];
                ow += (j[d] + W) * lsW[d];
                g += gi * A.pstride[d];
            }
            a_act[r] = act;
            a_int[r] = a_int[r] && act;
            a_oy[r] = oy;
            a_ow[r] = ow;
            a_g[r] = (unsigned)g * (unsigned)sizeof(T);
            hcell[r] = HAM::cell(A.ham, idx, A.sc);
        }
    }

    // ---- H slots.  Index space: per plane axis 4W layers (-2W..-1, E..E+2W-1) over the tile's extent on
    // the other axes, then (ND = 3) the four W x W corner blocks.
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
            bool ringlay[ND], outside[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) { j[d] = 0; ringlay[d] = false; outside[d] = false; }
            bool corner = false;
            if (h < hbase[ND]) {
#pragma unroll
                for (int d = 1; d < ND; ++d) {
                    if (h < hbase[d] || h >= hbase[d + 1]) continue;
                    const int hh = h - hbase[d];
                    const int lay = hh / area[d];
                    int c = hh - lay * area[d];
                    j[d] = (lay < 2 * W) ? (lay - 2 * W) : (A.E[d] + lay - 2 * W);
                    outside[d] = true;
                    ringlay[d] = (j[d] >= -W && j[d] < A.E[d] + W);
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
                j[ND - 1] = (blk & 2) ? A.E[ND - 1] + c2 : c2 - W;
                outside[1] = outside[ND - 1] = true;
                ringlay[1] = ringlay[ND - 1] = true;
            }
            int oy = 0, ow = 0, g = 0, dlt = 0, we = 0, wd = 0, nghost = 0;
            T km = T(0);
            bool fix = false;
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                int gi = org[d] + j[d];
                const int nd = A.n[d];
                oy += (j[d] + 2 * W) * lsY[d];
                int jw = j[d];                                // W-box coordinate of the (edge) cell
                if (gi < 0 || gi >= nd) {
                    if (A.bc[d] == HJ_BC_PERIODIC) { gi %= nd; if (gi < 0) gi += nd; }
                    else {
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
                } else if (outside[d] && ringlay[d] && !corner) {
                    act = false;                              // an in-domain ring cell: an A slot owns it
                }
                if (d != 0) ow += (j[d] + W) * lsW[d];
                we += (jw + W) * lsW[d];
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
    // plane p of y for the A slots (any p: wrapped on a periodic axis 0, boundary-rule ghost plane otherwise);
    // pw = p wrapped into [0, n0) (only meaningful on a periodic axis 0)
    auto load_own = [&](int p, int pw, T* dst) {
        if ((p >= 0 && p < n0) || per0) {
            const unsigned so = (unsigned)(per0 ? pw : p) * plane_bytes;
#pragma unroll
            for (int r = 0; r < R; ++r) dst[r] = buf_load(ry, a_g[r], so, T());
        } else {
            const unsigned se = (p < 0 ? 0u : (unsigned)(n0 - 1)) * plane_bytes;
            const unsigned si = (p < 0 ? 1u : (unsigned)(n0 - 2)) * plane_bytes;
            const T km0 = T(p < 0 ? -p : p - n0 + 1) * A.km[0];
#pragma unroll
            for (int r = 0; r < R; ++r)
                dst[r] = ghost_value(buf_load(ry, a_g[r], se, T()), buf_load(ry, a_g[r], si, T()), km0);
        }
    };
    // plane pw (already wrapped; a plane stage 1 runs on) for the H slots; both loads of a ghost slot are only
    // issued here, the ghost value is formed when the slot is written to LDS
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
    // does stage 1 run on plane q?  (a plane of the domain, or any plane when axis 0 is periodic)
    auto s1_plane = [&](int q) { return per0 || (q >= 0 && q < n0); };
    auto inc0 = [&](int pw) { return pw + 1 >= n0 ? pw + 1 - n0 : pw + 1; };     // next plane, wrapped

    // ---- prologue: yq[r][j] <-> plane q-3+j
    const int q0 = p_begin - W, q1 = p_end + W;
    T yq[R][7];
    int pw_load = wrap0(q0 - W);                 // wrapped index of the next plane load_own fetches
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        T tmp[R];
        load_own(q0 - W + j, pw_load, tmp);
        pw_load = inc0(pw_load);
#pragma unroll
        for (int r = 0; r < R; ++r) yq[r][j] = tmp[r];
    }
    constexpr int PD = 2;
    T own[PD][R], hal[PD][KH], hin[PD][KH];
#pragma unroll
    for (int s = 0; s < PD; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r) own[s][r] = T(0);
    load_own(q0 + W + 1, pw_load, own[0]);       // plane q0+4; the second set is filled by the first iteration
    pw_load = inc0(pw_load);
    int pw_h = wrap0(q0);                        // wrapped index of the next plane load_halo fetches
#pragma unroll
    for (int s = 0; s < PD; ++s) {
#pragma unroll
        for (int k = 0; k < KH; ++k) { hal[s][k] = T(0); hin[s][k] = T(0); }
        if (s1_plane(q0 + s)) load_halo(pw_h, hal[s], hin[s]);
        pw_h = inc0(pw_h);
    }
    int qw = wrap0(q0);                          // plane q wrapped
    // y1 ring: sW[j] = LDS element offset of the slot that holds plane q-6+j (j = 0..6; slot = plane mod 7)
    int sW[7];
    {
        int m = (q0 - 6) % 7;
        if (m < 0) m += 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) { sW[j] = m * wbox; m = (m == 6) ? 0 : m + 1; }
    }
    int off_e = 0, off_i = 0;
    // which waves hold any active / interior A slot (wave-uniform: whole waves skip the slots they do not own)
    bool w_act[R], w_int[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { w_act[r] = __any(a_act[r] ? 1 : 0) != 0; w_int[r] = __any(a_int[r] ? 1 : 0) != 0; }

    double amax[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) amax[d] = -1.0e300;
    {   // alphas that are constant along the march: one max per column (interior cells)
        T pz[ND], Hz, az[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) pz[d] = T(0);
        const typename HAM::Plane plz = HAM::plane(A.ham, min(max(p_begin, 0), n0 - 1), A.sc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            HAM::eval(A.ham, hcell[r], plz, A.sc, pz, Hz, az);
#pragma unroll
            for (int d = 0; d < ND; ++d)
                if (!((HAM::PLANE_DEP >> d) & 1u) && a_int[r]) amax[d] = fmax(amax[d], (double)az[d]);
        }
    }

    // the Lax-Friedrichs right-hand side of one cell: v0 = its 7 axis-0 values, in-plane values from `buf` at
    // box offset `o` with strides ls[]; returns ydot (and the alphas)
    auto lf_rhs = [&](const T* v0, const T* buf, int o, const int* ls, const typename HAM::Cell& hc,
                      const typename HAM::Plane& pl, T* alpha) {
        T pc[ND], hd[ND];
        upwind_cd<SCHEME, T>(v0, A.K[0], eps[0], wk[0], pc[0], hd[0]);
#pragma unroll
        for (int d = 1; d < ND; ++d) {
            T v[7];
            const T* c = buf + o;
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = (j == 3) ? v0[3] : c[(j - 3) * ls[d]];
            upwind_cd<SCHEME, T>(v, A.K[d], eps[d], wk[d], pc[d], hd[d]);
        }
        return lf_ydot<NP, HAM>(A.ham, hc, pl, A.sc, pc, hd, alpha);
    };

    // one iteration: stage 1 on plane q, stage 2 on plane q - W.  own_c holds plane q+4 (joins the queue at the
    // end), own_n is refilled with plane q+5; hal_c / hin_c hold plane q's H values and are refilled for q+2.
    // Inactive slots compute on (valid) stand-in data; only their LDS / HBM writes are predicated off.
    auto body = [&](int q, T* own_c, T* own_n, T* hal_c, T* hin_c) {
        const bool s1 = s1_plane(q);
        T* bufY = ldsY + (q & 1) * ybox;
        T* bufWq = ldsW + sW[6];
        if (q + W + PD < q1 + W) load_own(q + W + PD, pw_load, own_n);
        pw_load = inc0(pw_load);
        if (s1) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (a_act[r]) bufY[a_oy[r]] = yq[r][3];
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
        __syncthreads();
        if (s1_plane(q + PD) && q + PD < q1) load_halo(pw_h, hal_c, hin_c);
        pw_h = inc0(pw_h);
        // ---- y1 ghost cells of plane q-1 on extrapolated in-plane boundaries (boundary rule applied to y1)
        if (tile_fix && q - 1 >= q0 && s1_plane(q - 1)) {
            T* bw = ldsW + sW[5];
#pragma unroll
            for (int k = 0; k < KH; ++k)
                if (h_fix[k]) bw[h_ow[k]] = ghost_value(bw[h_we[k]], bw[h_we[k] + h_wd[k]], h_km[k]);
        }
        // ---- stage 1 on plane q
        T y1n[R];
#pragma unroll
        for (int r = 0; r < R; ++r) y1n[r] = T(0);
        if (s1) {
            const typename HAM::Plane pl1 = HAM::plane(A.ham, qw, A.sc);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!w_act[r]) continue;
                T alpha[ND];
                const T ydot = lf_rhs(yq[r], bufY, a_oy[r], lsY, hcell[r], pl1, alpha);
                // the Euler stage as fused_substep_kernel forms it (ca = 0, cb = 1, no y0 operand)
                y1n[r] = rk_stage_out<NP>(HJ_STAGE_EULER, T(0), T(1), A.dt, T(0), yq[r][3], ydot);
                if (a_act[r]) bufWq[a_ow[r]] = y1n[r];
            }
        } else if (q >= n0) {
            // a ghost plane of y1 beyond an extrapolated axis-0 boundary (high side; the low side is filled
            // when plane 1 is done, below): boundary rule on the thread's own column
            const T kmq = T(q - n0 + 1) * A.km[0];
            const T* be = ldsW + off_e;                   // planes n0-1 and n0-2 (slots noted when they were written)
            const T* bi = ldsW + off_i;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (a_int[r]) { y1n[r] = ghost_value(be[a_ow[r]], bi[a_ow[r]], kmq); bufWq[a_ow[r]] = y1n[r]; }
        }
        if (q == n0 - 1) off_e = sW[6];
        if (q == n0 - 2) off_i = sW[6];
        if (!per0 && q == 1) {
            // planes -1, -2, -3 of y1 (slots 4, 3, 2 back from plane 1... i.e. sW[4], sW[3], sW[2]) from planes 0 and 1
            const T* be = ldsW + sW[5];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!a_int[r]) continue;
                const T e0 = be[a_ow[r]];
#pragma unroll
                for (int k = 1; k <= W; ++k) (ldsW + sW[5 - k])[a_ow[r]] = ghost_value(e0, y1n[r], T(k) * A.km[0]);
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
                T v0[7];
#pragma unroll
                for (int j = 0; j < 6; ++j) v0[j] = (ldsW + sW[j])[a_ow[r]];
                v0[6] = y1n[r];
                T alpha[ND];
                const T ydot = lf_rhs(v0, bufWp, a_ow[r], lsW, hcell[r], pl2, alpha);
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    if (((HAM::PLANE_DEP >> d) & 1u) && a_int[r]) amax[d] = fmax(amax[d], (double)alpha[d]);
                // yq[r][0] is y on plane p: the y0 operand of the second stage
                const T o = rk_stage_out<NP>(A.stage2, A.ca, A.cb, A.dt, yq[r][0], v0[3], ydot);
                if (a_int[r]) buf_store(o, rout, a_g[r], so_out);
            }
        }
        // ---- rotate the queue and the ring
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 6; ++j) yq[r][j] = yq[r][j + 1];
            yq[r][6] = own_c[r];
        }
        {
            const int s0 = sW[0];
#pragma unroll
            for (int j = 0; j < 6; ++j) sW[j] = sW[j + 1];
            sW[6] = s0;
        }
        qw = inc0(qw);
    };

    for (int q = q0; q < q1; q += PD) {
        body(q, own[0], own[1], hal[0], hin[0]);
        if (q + 1 < q1) body(q + 1, own[1], own[0], hal[1], hin[1]);
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
