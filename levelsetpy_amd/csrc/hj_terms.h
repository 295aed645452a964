// termNormal, termReinit, termConvection as ONE kernel launch each (gfx950; round 3).
//
// Until round 2 these schemeFuncs took their one-sided derivatives from the HIP upwind kernels and then ran
// ~30 elementwise torch / NumPy launches per call (Godunov selection, |grad phi|, the sub-cell fix, the CFL maxima).
// Here every cell does the whole chain in registers: the 7-point stencils of all dimensions (ghost cells on the
// fly, upwind<SCHEME> of hj_device.h: derivL, derivR), the upwind choice, the magnitude, ydot, and the CFL
// reduction through the DPP wave maximum -> LDS -> one atomicMax per workgroup.
//
//   termNormal      ydot = -a |grad phi|                    term_normal.py:143-181
//   termReinit      ydot = -S(phi_0)(|grad phi| - 1)        term_reinit.py:181-312   (sub-cell fix of order 0 / 1)
//   termConvection  ydot = -V . grad phi                    term_convection.py:154-180
//
// The shipped reference functions raise, so the formulas are those of their docstrings and
// of the toolbox they port, as restated in oracle/hj_oracle.py (term_normal / term_reinit / term_convection):
// parity UNPINNED, checked against the oracle.  Contraction is off: the expressions are evaluated operation by
// operation in the order of the host implementation they replace (levelsetpy_amd/normal_reinit.py, convection.py),
// so the two agree to the last bit wherever the derivatives do.
// One thread per cell, neighbours straight from global memory, every stencil load issued unconditionally (gather_stencils
// of hj_split.h, as in direct_substep_kernel: L1/L2 absorb the stencil reuse).
#pragma once
#include "hj_split.h"
#include "hj_termop.h"

namespace hj {

template <typename T, int ND> struct TermArgs {
    const T* y;
    T* out;
    GridArgs<T, ND> G;
    const T* max_d1sq;            // HJ_WENO5 only
    TermPar<T> P;                 // coefficient arrays / scalars, spacings, termReinit's constants (hj_termop.h)
    unsigned long long* keys;     // ND + 1 atomicMax keys: per-dimension maxima, then the scalar one
};

template <typename T, int ND, int SCHEME, int KIND>
__global__ __launch_bounds__(256) void term_kernel(const TermArgs<T, ND> A) {
#pragma clang fp contract(off)
    T eps[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        eps[d] = T(0);
        if constexpr (SCHEME == HJ_WENO5) eps[d] = T(1e-6) * A.max_d1sq[d] + Lim<T>::tiny;
    }
    double m[ND + 1];
#pragma unroll
    for (int d = 0; d <= ND; ++d) m[d] = -1e300;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < A.G.total;
         t += (long long)gridDim.x * blockDim.x) {
        int idx[ND];
        decode<T, ND>(A.G, t, idx);
        T dL[ND], dR[ND];
        T pc0v;
        {
            // every stencil load issued back to back (hj_split.h, gather_stencils), then the derivatives
            const T* pc0 = A.y + t;
            T v[ND][7];
            pc0v = pc0[0];
            gather_stencils<T, ND>(A.G, pc0, idx, pc0v, v);
#pragma unroll
            for (int d = 0; d < ND; ++d) upwind<SCHEME, T>(v[d], A.G.K[d], eps[d], dL[d], dR[d]);
        }
        // the cell's arithmetic: term_cell (hj_termop.h), shared with the tiled kernel
        T coef[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) coef[d] = A.P.arr[d] ? A.P.arr[d][t] : A.P.scal[d];
        T nb_lo[ND], nb_hi[ND];
        unsigned has_lo = 0u, has_hi = 0u;
#pragma unroll
        for (int d = 0; d < ND; ++d) { nb_lo[d] = T(0); nb_hi[d] = T(0); }
        if constexpr (KIND == HJ_TERM_REINIT) {
            if (A.P.subcell_order == 1) {
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const T* p = A.P.arr[0] + t;
                    const long long st = A.G.stride[d];
                    if (idx[d] > 0) { has_lo |= 1u << d; nb_lo[d] = p[-st]; }
                    if (idx[d] + 1 < A.G.n[d]) { has_hi |= 1u << d; nb_hi[d] = p[st]; }
                }
            }
        }
        const T o = term_cell<KIND, T, ND>(A.P, dL, dR, coef, pc0v, nb_lo, nb_hi, has_lo, has_hi, m);
        A.out[t] = o;
    }
    __shared__ double red[4][ND + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d <= ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if (threadIdx.x <= ND) {
        const int d = threadIdx.x;
        const double w = fmax(fmax(red[0][d], red[1][d]), fmax(red[2][d], red[3][d]));
        if (w > -1e299) key_max(A.keys + d, w);
    }
}

}  // namespace hj
