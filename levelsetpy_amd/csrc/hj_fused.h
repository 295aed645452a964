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
//   * thread <-> cell: the tile's cells are numbered linearly (last axis fastest) and dealt
//     round-robin to the NT threads, R cells per thread: consecutive lanes touch consecutive
//     addresses in HBM (coalesced, full 64-lane waves even when N is not a multiple of 64) and
//     consecutive 8-byte LDS words (bank-conflict free for ds_read_b64).
//   * blockIdx -> (chunk, tile) is XCD-aware: the 8 XCDs get contiguous ranges of the logical
//     block order, so tiles sharing halo rows share an L2.
//   * per-dim max(alpha) is reduced with wavefront shuffles, then LDS, then one 64-bit
//     atomicMax per workgroup and dimension.
#pragma once
#include "hj_device.h"

namespace hj {

template <typename T, int ND> struct FusedArgs {
    const T* y;    // stencil input; first interior plane
    const T* y0;   // stage operand (may be null)
    T* out;
    const T* max_d1sq;            // ND values (HJ_WENO5 only)
    unsigned long long* bound;    // ND keys (atomicMax)
    int n[ND];
    int bc[ND];
    int halo_lo, halo_hi;
    T km[ND];                     // slope multiplier (+1, -1 if towardZero)
    T dx[ND], inv_dx[ND];
    long long stride0;            // elements per axis-0 plane
    int pstride[ND];              // in-plane element strides (pstride[0] unused)
    int E[ND];                    // tile extents on the plane axes (E[0] unused)
    int ntile[ND];
    int ntiles;
    int chunk, nchunks;
    int plane_begin, plane_end;
    int nblocks, blocks_per_xcd;
    int stage, restrict_sign;
    T dt;
    HamTables<T> ham;
};

template <typename T, int ND>
__device__ __forceinline__ T load_axis0(const FusedArgs<T, ND>& A, int p, int g) {
    const int n0 = A.n[0];
    if (p >= 0 && p < n0) return A.y[(long long)p * A.stride0 + g];
    if (p < 0) {
        if (A.halo_lo) return A.y[(long long)p * A.stride0 + g];
        if (A.bc[0] == HJ_BC_PERIODIC) return A.y[(long long)(p + n0) * A.stride0 + g];
        const T e = A.y[g], i = A.y[A.stride0 + g];
        return ghost_value(e, i, T(-p) * A.km[0]);
    }
    if (A.halo_hi) return A.y[(long long)p * A.stride0 + g];
    if (A.bc[0] == HJ_BC_PERIODIC) return A.y[(long long)(p - n0) * A.stride0 + g];
    const T e = A.y[(long long)(n0 - 1) * A.stride0 + g], i = A.y[(long long)(n0 - 2) * A.stride0 + g];
    return ghost_value(e, i, T(p - n0 + 1) * A.km[0]);
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH>
__global__ __launch_bounds__(NT) void fused_substep_kernel(const FusedArgs<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr int PD = ND - 1;  // plane dims
    extern __shared__ __align__(16) unsigned char hj_smem[];
    // dynamic LDS only (keeps the carve base 16-byte aligned): [0,512) reduction scratch, then planes
    double (*red)[ND] = reinterpret_cast<double (*)[ND]>(hj_smem);
    T* lds = reinterpret_cast<T*>(hj_smem + 512);
    static_assert((NT / 64) * ND * 8 <= 512, "reduction scratch");

    // ---- XCD-aware block order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch),
    // give each XCD a contiguous run of logical blocks.
    const int b = blockIdx.x;
    const int L = (b & 7) * A.blocks_per_xcd + (b >> 3);
    if (L >= A.nblocks) return;
    const int chunk_id = L / A.ntiles;
    int rem = L - chunk_id * A.ntiles;
    int org[ND];
#pragma unroll
    for (int d = ND - 1; d >= 1; --d) {
        const int q = rem / A.ntile[d];
        // the last tile on an axis is shifted back so that no tile straddles the domain edge
        // (it recomputes a few cells of its neighbour: identical values, benign duplicate stores)
        org[d] = min((rem - q * A.ntile[d]) * A.E[d], A.n[d] - A.E[d]);
        rem = q;
    }
    const int p_begin = A.plane_begin + chunk_id * A.chunk;
    const int p_end = min(p_begin + A.chunk, A.plane_end);

    // ---- LDS geometry: halo'd box, last axis contiguous
    int ls[ND];
    ls[ND - 1] = 1;
#pragma unroll
    for (int d = ND - 2; d >= 1; --d) ls[d] = ls[d + 1] * (A.E[d + 1] + 2 * HJ_STENCIL);
    const int lds_plane = ls[1] * (A.E[1] + 2 * HJ_STENCIL);
    int tile_cells = 1;
#pragma unroll
    for (int d = 1; d < ND; ++d) tile_cells *= A.E[d];

    const int tid = threadIdx.x;

    // ---- own cells (loop-invariant along the march)
    int own_lds[R], own_g[R], own_i[R][PD];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int c = tid + r * NT;
        bool ok = c < tile_cells;
        int lo = 0, g = 0;
#pragma unroll
        for (int d = ND - 1; d >= 1; --d) {
            const int q = c / A.E[d];
            const int j = c - q * A.E[d];
            c = q;
            const int gi = org[d] + j;
            ok = ok && (gi < A.n[d]);
            own_i[r][d - 1] = gi;
            lo += (j + HJ_STENCIL) * ls[d];
            g += gi * A.pstride[d];
        }
        own_lds[r] = lo;
        own_g[r] = ok ? g : -1;
    }

    // ---- halo slots: for each plane axis d, 3 cells below and 3 above the tile, over the
    // tile's extent on the other axes (a "cross": no corners).
    int h_lds[KH], h_src[KH], h_dlt[KH];
    T h_km[KH];
    {
        int area[ND], base[ND + 1];
        base[1] = 0;
#pragma unroll
        for (int d = 1; d < ND; ++d) {
            area[d] = tile_cells / A.E[d];
            base[d + 1] = base[d] + 2 * HJ_STENCIL * area[d];
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) {
            const int h = tid + k * NT;
            h_lds[k] = -1; h_src[k] = 0; h_dlt[k] = 0; h_km[k] = T(0);
            // static loop over the axis the slot belongs to (runtime-indexed local arrays
            // would be demoted to scratch)
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                if (h < base[d] || h >= base[d + 1]) continue;
                const int hh = h - base[d];
                const int lay = hh / area[d];          // 0..5: which halo layer
                int c = hh - lay * area[d];            // index over the other axes (last fastest)
                const int jd = (lay < HJ_STENCIL) ? (lay - HJ_STENCIL) : (A.E[d] + lay - HJ_STENCIL);
                bool ok = true;
                int lo = 0, g = 0;
#pragma unroll
                for (int e = ND - 1; e >= 1; --e) {
                    if (e == d) continue;
                    const int q = c / A.E[e];
                    const int j = c - q * A.E[e];
                    c = q;
                    const int gi = org[e] + j;
                    ok = ok && (gi < A.n[e]);
                    lo += (j + HJ_STENCIL) * ls[e];
                    g += gi * A.pstride[e];
                }
                lo += (jd + HJ_STENCIL) * ls[d];
                int gi = org[d] + jd;
                const int nd = A.n[d];
                if (gi >= nd + HJ_STENCIL) ok = false;   // never read
                if (!ok) continue;
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
                h_src[k] = g + gi * A.pstride[d];
                h_dlt[k] = dlt;
                h_km[k] = km;
            }
        }
    }

    auto fetch_halo = [&](int p, int k) -> T {
        const T* base = A.y + (long long)p * A.stride0;
        const T e = base[h_src[k]];
        if (h_dlt[k] != 0) {
            const T i = base[h_src[k] + h_dlt[k]];
            return ghost_value(e, i, h_km[k]);
        }
        return e;
    };

    T eps[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) eps[d] = T(0);
    if constexpr (SCHEME == HJ_WENO5) {
#pragma unroll
        for (int d = 0; d < ND; ++d) eps[d] = T(1e-6) * A.max_d1sq[d] + Lim<T>::tiny;
    }

    // ---- prologue: axis-0 queue q[r][j] <-> plane p-3+j, j = 0..6; qn = plane p+4
    T q[R][7];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 7; ++j) q[r][j] = T(0);
        if (own_g[r] >= 0) {
#pragma unroll
            for (int j = 0; j < 7; ++j) q[r][j] = load_axis0<T, ND>(A, p_begin - 3 + j, own_g[r]);
        }
    }
    T hcur[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) hcur[k] = (h_lds[k] >= 0) ? fetch_halo(p_begin, k) : T(0);

    double amax[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) amax[d] = -1.0e300;

    for (int p = p_begin; p < p_end; ++p) {
        T* buf = lds + ((p - p_begin) & 1) * lds_plane;
        // loads for the next iteration first: their latency hides behind this plane's math
        T qn[R], hn[KH], y0v[R];
        const bool more = (p + 1 < p_end);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            qn[r] = T(0);
            if (more && own_g[r] >= 0) qn[r] = load_axis0<T, ND>(A, p + 4, own_g[r]);
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) hn[k] = (more && h_lds[k] >= 0) ? fetch_halo(p + 1, k) : T(0);
        if (A.stage >= HJ_STAGE_RK3_HALF) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                y0v[r] = (own_g[r] >= 0) ? A.y0[(long long)p * A.stride0 + own_g[r]] : T(0);
        }
        // stage the centre plane
#pragma unroll
        for (int r = 0; r < R; ++r) if (own_g[r] >= 0) buf[own_lds[r]] = q[r][3];
#pragma unroll
        for (int k = 0; k < KH; ++k) if (h_lds[k] >= 0) buf[h_lds[k]] = hcur[k];
        __syncthreads();

#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (own_g[r] < 0) continue;
            T dL[ND], dR[ND], pc[ND];
            upwind<SCHEME, T>(q[r], A.dx[0], A.inv_dx[0], eps[0], dL[0], dR[0]);
#pragma unroll
            for (int d = 1; d < ND; ++d) {
                T v[7];
                const T* c = buf + own_lds[r];
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = (j == 3) ? q[r][3] : c[(j - 3) * ls[d]];
                upwind<SCHEME, T>(v, A.dx[d], A.inv_dx[d], eps[d], dL[d], dR[d]);
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) pc[d] = T(0.5) * (dL[d] + dR[d]);
            int idx[ND];
            idx[0] = p;
#pragma unroll
            for (int d = 1; d < ND; ++d) idx[d] = own_i[r][d - 1];
            T H, alpha[ND];
            HAM::eval(A.ham, idx, pc, H, alpha);
            T diss = T(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                diss += (T(0.5) * (dR[d] - dL[d])) * alpha[d];
                amax[d] = fmax(amax[d], (double)alpha[d]);
            }
            T ydot = -(H - diss);
            if (A.restrict_sign > 0) ydot = t_max(ydot, T(0));
            else if (A.restrict_sign < 0) ydot = t_min(ydot, T(0));
            T o;
            if (A.stage == HJ_STAGE_YDOT) o = ydot;
            else {
                const T ye = q[r][3] + A.dt * ydot;
                if (A.stage == HJ_STAGE_EULER) o = ye;
                else if (A.stage == HJ_STAGE_RK3_HALF) o = T(0.25) * (T(3) * y0v[r] + ye);
                else if (A.stage == HJ_STAGE_RK3_FULL) o = (T(1) / T(3)) * (y0v[r] + T(2) * ye);
                else o = T(0.5) * (y0v[r] + ye);
            }
            A.out[(long long)p * A.stride0 + own_g[r]] = o;
        }
        // rotate the queue
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < 6; ++j) q[r][j] = q[r][j + 1];
            q[r][6] = qn[r];
        }
#pragma unroll
        for (int k = 0; k < KH; ++k) hcur[k] = hn[k];
    }

    // ---- CFL reduction: wavefront shuffles -> LDS -> one atomicMax per block and dim
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double m = wave_max(amax[d]);
        if (lane == 0) red[wv][d] = m;
    }
    __syncthreads();
    if (tid < ND) {
        double m = red[0][tid];
        for (int w = 1; w < NT / 64; ++w) m = fmax(m, red[w][tid]);
        if (m > -1.0e299) atomicMax(A.bound + tid, max_key(m));
    }
}

}  // namespace hj
