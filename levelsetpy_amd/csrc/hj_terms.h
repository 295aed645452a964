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

namespace hj {

enum { HJ_TERM_NORMAL = 0, HJ_TERM_REINIT = 1, HJ_TERM_CONVECTION = 2 };

template <typename T, int ND> struct TermArgs {
    const T* y;
    T* out;
    GridArgs<T, ND> G;
    const T* max_d1sq;            // HJ_WENO5 only
    // termNormal: speed array or scalar;  termConvection: velocity arrays or scalars;  termReinit: initial
    const T* arr[HJ_MAX_DIM];
    T scal[HJ_MAX_DIM];
    T dx_inv[ND], dx[ND], max_dx;
    int subcell_order;            // termReinit
    T small2, tiny;               // (1e6 eps)^2 and eps of termReinit
    unsigned long long* keys;     // ND + 1 atomicMax keys: per-dimension maxima, then the scalar one
};

template <typename T> __device__ __forceinline__ T t_sign(T a) { return a > T(0) ? T(1) : (a < T(0) ? T(-1) : T(0)); }

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
        {
            // every stencil load issued back to back (hj_split.h, gather_stencils), then the derivatives
            const T* pc0 = A.y + t;
            T v[ND][7];
            gather_stencils<T, ND>(A.G, pc0, idx, pc0[0], v);
#pragma unroll
            for (int d = 0; d < ND; ++d) upwind<SCHEME, T>(v[d], A.G.K[d], eps[d], dL[d], dR[d]);
        }
        T o;
        if constexpr (KIND == HJ_TERM_NORMAL) {
            const T speed = A.arr[0] ? A.arr[0][t] : A.scal[0];
            T magnitude = T(0), sbi = T(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const T prodL = speed * dL[d], prodR = speed * dR[d];
                const T magL = t_abs(prodL), magR = t_abs(prodR);
                const bool conv = (prodL >= T(0)) && (prodR <= T(0));
                const bool flowL = ((prodL >= T(0)) && (prodR >= T(0))) || (conv && (magL >= magR));
                const bool flowR = ((prodL <= T(0)) && (prodR <= T(0))) || (conv && (magL < magR));
                const T fl = flowL ? T(1) : T(0), fr = flowR ? T(1) : T(0);
                magnitude = magnitude + ((dL[d] * dL[d]) * fl + (dR[d] * dR[d]) * fr);
                const T vel = magL * fl + magR * fr;
                sbi = sbi + vel / A.dx[d];
            }
            magnitude = sqrt(magnitude);
            o = -(speed * magnitude);
            if (magnitude > T(0)) m[ND] = fmax(m[ND], (double)(sbi / magnitude));
        } else if constexpr (KIND == HJ_TERM_CONVECTION) {
            T delta = T(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const T v = A.arr[d] ? A.arr[d][t] : A.scal[d];
                const T deriv = dL[d] * (v > T(0) ? T(1) : T(0)) + dR[d] * (v < T(0) ? T(1) : T(0));
                delta = delta + deriv * v;
                m[d] = fmax(m[d], (double)t_abs(v));
            }
            o = -delta;
        } else {
            const T init = A.arr[0][t];
            const T data = A.y[t];
            T S;
            if (A.subcell_order) S = t_sign(init);
            else S = init / sqrt(init * init + A.max_dx * A.max_dx);
            T deriv[ND];
            T mag = T(0);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const T sL = S * dL[d], sR = S * dR[d];
                bool flowL = (sR <= T(0)) && (sL <= T(0));
                bool flowR = (sR >= T(0)) && (sL >= T(0));
                const bool flows = (sR < T(0)) && (sL > T(0));
                T den = dR[d] - dL[d];
                den = den + (den == T(0) ? T(1) : T(0));
                const T s = S * (t_abs(dR[d]) - t_abs(dL[d])) / den;
                flowL = flowL || (flows && (s < T(0)));
                flowR = flowR || (flows && (s >= T(0)));
                deriv[d] = dL[d] * (flowR ? T(1) : T(0)) + dR[d] * (flowL ? T(1) : T(0));
                mag = mag + deriv[d] * deriv[d];
            }
            mag = sqrt(mag);
            mag = mag > A.tiny ? mag : A.tiny;
            T delta = -S;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const T v = S * deriv[d] / mag;
                delta = delta + v * deriv[d];
                m[d] = fmax(m[d], (double)t_abs(v));
            }
            if (A.subcell_order == 1) {
                // Russo & Smereka's sub-cell fix with the robust distance estimate (long differences, short ones
                // where they are larger), applied at the nodes next to the interface
                T denom = T(0);
                bool near = (t_sign(init) == T(0));
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const int i = idx[d], n = A.G.n[d];
                    const long long s = A.G.stride[d];
                    const T* p = A.arr[0] + t;
                    const T di = A.dx_inv[d];
                    const T lo = i > 0 ? p[-s] : init, hi = i + 1 < n ? p[s] : init;
                    T diff2;
                    if (i > 0 && i + 1 < n) { const T c = (T(0.5) * di) * (hi - lo); diff2 = c * c; }
                    else if (i == 0) { const T c = di * (hi - init); diff2 = c * c; }
                    else { const T c = di * (init - lo); diff2 = c * c; }
                    if (i + 1 < n) { const T c = di * (hi - init); const T s2 = c * c; diff2 = diff2 > s2 ? diff2 : s2; }
                    if (i > 0) { const T c = di * (init - lo); const T s2 = c * c; diff2 = diff2 > s2 ? diff2 : s2; }
                    diff2 = diff2 > A.small2 ? diff2 : A.small2;
                    denom = denom + diff2;
                    const T sg = t_sign(init);
                    if (i > 0) near = near || (t_sign(lo) != sg);
                    if (i + 1 < n) near = near || (t_sign(hi) != sg);
                }
                const T D = init / sqrt(denom);
                const T nr = near ? T(1) : T(0), fr = near ? T(0) : T(1);
                delta = delta * fr + (S * t_abs(data) - D) / A.max_dx * nr;
            }
            o = -delta;
        }
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
        if (w > -1e299) atomicMax(A.keys + d, max_key(w));
    }
}

}  // namespace hj
