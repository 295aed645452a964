// termNormal / termReinit / termConvection: the arithmetic of ONE cell, shared by the direct kernel (hj_terms.h: one thread
// per cell, stencils straight from global memory) and the tiled kernel (hj_fused.h with a TermOp "Hamiltonian": LDS-staged
// stencils, register queue along axis 0; round 4) -- one function, so the two kernels agree to the last bit.
//
//   termNormal      ydot = -a |grad phi|                    term_normal.py:143-181
//   termReinit      ydot = -S(phi_0)(|grad phi| - 1)        term_reinit.py:181-312   (sub-cell fix of order 0 / 1)
//   termConvection  ydot = -V . grad phi                    term_convection.py:154-180
//
// The shipped reference functions raise, so the formulas are those of their docstrings and of the
// toolbox they port, as restated in oracle/hj_oracle.py (term_normal / term_reinit / term_convection): parity UNPINNED,
// checked against the oracle.  Contraction is off: the expressions are evaluated operation by operation in the order of
// the host implementation they replace (levelsetpy_amd/normal_reinit.py, convection.py), so that the kernels and the
// array path agree to the last bit wherever the derivatives do.
#pragma once
#include "hj_device.h"

namespace hj {

enum { HJ_TERM_NORMAL = 0, HJ_TERM_REINIT = 1, HJ_TERM_CONVECTION = 2 };

// what a term needs besides the grid: coefficient arrays (speed / initial / velocity components; null = the scalar), the
// spacings and termReinit's constants
template <typename T> struct TermPar {
    const T* arr[HJ_MAX_DIM];
    T scal[HJ_MAX_DIM];
    T dx[HJ_MAX_DIM], dx_inv[HJ_MAX_DIM], max_dx;
    T small2, tiny;               // (1e6 eps)^2 and eps of termReinit (term_reinit.py:128,206,274)
    int subcell_order;            // termReinit: 0 smeared sign, 1 Russo-Smereka sub-cell fix
};

template <typename T> __device__ __forceinline__ T t_sign(T a) { return a > T(0) ? T(1) : (a < T(0) ? T(-1) : T(0)); }

// dL / dR: one-sided derivatives of this cell per dimension (upwind<SCHEME>);  coef: termNormal speed = coef[0], termReinit
// initial = coef[0], termConvection velocity = coef[0..ND-1];  data: phi at the cell;  nb_lo / nb_hi, has_lo / has_hi (bit d):
// the neighbours of the INITIAL array one cell down / up along dimension d where they exist (sub-cell fix only);
// m[0..ND]: running maxima -- per dimension (convection, reinit) and m[ND] (normal) -- that become the step bound.
template <int KIND, typename T, int ND>
__device__ __forceinline__ T term_cell(const TermPar<T>& P, const T* dL, const T* dR, const T* coef, T data, const T* nb_lo, const T* nb_hi,
                                       unsigned has_lo, unsigned has_hi, double* m) {
#pragma clang fp contract(off)
    T o;
    if constexpr (KIND == HJ_TERM_NORMAL) {
        const T speed = coef[0];
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
            sbi = sbi + vel / P.dx[d];
        }
        magnitude = sqrt(magnitude);
        o = -(speed * magnitude);
        if (magnitude > T(0)) m[ND] = fmax(m[ND], (double)(sbi / magnitude));
    } else if constexpr (KIND == HJ_TERM_CONVECTION) {
        T delta = T(0);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const T v = coef[d];
            const T deriv = dL[d] * (v > T(0) ? T(1) : T(0)) + dR[d] * (v < T(0) ? T(1) : T(0));
            delta = delta + deriv * v;
            m[d] = fmax(m[d], (double)t_abs(v));
        }
        o = -delta;
    } else {
        const T init = coef[0];
        T S;
        if (P.subcell_order) S = t_sign(init);
        else S = init / sqrt(init * init + P.max_dx * P.max_dx);
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
        mag = mag > P.tiny ? mag : P.tiny;
        T delta = -S;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const T v = S * deriv[d] / mag;
            delta = delta + v * deriv[d];
            m[d] = fmax(m[d], (double)t_abs(v));
        }
        if (P.subcell_order == 1) {
            // Russo & Smereka's sub-cell fix with the robust distance estimate (long differences, short ones where they
            // are larger), applied at the nodes next to the interface
            T denom = T(0);
            bool near = (t_sign(init) == T(0));
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const bool lo_ok = (has_lo >> d) & 1u, hi_ok = (has_hi >> d) & 1u;
                const T di = P.dx_inv[d];
                const T lo = lo_ok ? nb_lo[d] : init, hi = hi_ok ? nb_hi[d] : init;
                T diff2;
                if (lo_ok && hi_ok) { const T c = (T(0.5) * di) * (hi - lo); diff2 = c * c; }
                else if (!lo_ok) { const T c = di * (hi - init); diff2 = c * c; }
                else { const T c = di * (init - lo); diff2 = c * c; }
                if (hi_ok) { const T c = di * (hi - init); const T s2 = c * c; diff2 = diff2 > s2 ? diff2 : s2; }
                if (lo_ok) { const T c = di * (init - lo); const T s2 = c * c; diff2 = diff2 > s2 ? diff2 : s2; }
                diff2 = diff2 > P.small2 ? diff2 : P.small2;
                denom = denom + diff2;
                const T sg = t_sign(init);
                if (lo_ok) near = near || (t_sign(lo) != sg);
                if (hi_ok) near = near || (t_sign(hi) != sg);
            }
            const T D = init / sqrt(denom);
            const T nr = near ? T(1) : T(0), fr = near ? T(0) : T(1);
            delta = delta * fr + (S * t_abs(data) - D) / P.max_dx * nr;
        }
        o = -delta;
    }
    return o;
}

// The "Hamiltonian" type under which the tiled substep kernels run a term: no per-column / per-plane constants, the cell
// arithmetic is term_cell above (the kernels branch on ham_traits<HAM>::is_term at compile time).
template <typename T, int ND_, int KIND_> struct TermOp {
    static constexpr int ND = ND_;
    static constexpr int ID = -1;
    static constexpr int KIND = KIND_;
    static constexpr unsigned PLANE_DEP = 0xFu;
    struct Cell {};
    struct Plane {};
    using Raw = Cell;
    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>&, const int*) { return Raw(); }
    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>&, const int*, const Raw& r) { return r; }
    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>&, const Raw& r, const T*) { return r; }
    __device__ static __forceinline__ Cell cell(const HamTables<T>&, const int*, const T*) { return Cell(); }
    __device__ static __forceinline__ Plane plane(const HamTables<T>&, int, const T*) { return Plane(); }
};
template <typename HAM> struct ham_traits { static constexpr bool is_term = false; static constexpr int kind = -1; };
template <typename T, int ND, int KIND> struct ham_traits<TermOp<T, ND, KIND>> { static constexpr bool is_term = true; static constexpr int kind = KIND; };

}  // namespace hj
