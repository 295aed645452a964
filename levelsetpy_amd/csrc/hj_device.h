// Device-side building blocks shared by the fused and split kernels (gfx950 only).
//
//  * ghost_value     -- addGhostExtrapolate's rule          (add_ghost_extrapolate.py:88-110)
//  * upwind<SCHEME>  -- left/right first derivative from the 7 values phi[i-3..i+3]
//                       ENO2 (upwind_first_eno2.py:78-148), ENO3 (ENO3aHelper.py:76-189 +
//                       upwind_first_eno3a.py:105-141), WENO5 as shipped / intended
//                       (upwind_first_weno5a.py:106-196)
//  * Ham*            -- native hamFunc/partialFunc pairs (dubins_relative.py:83-111,
//                       double_integrator.py:71-89, build-defined double pendulum)
//
// Arithmetic follows the reference's divided-difference tables (D1, D2, D3) in the same
// operation order so that results agree to rounding (not bit-for-bit: the compiler may
// contract a*b+c into FMAs here; ghost_value is the exception and is bit-exact).
#pragma once
#ifndef __HIPCC_RTC__      // hipRTC (user Hamiltonians, hj_rtc.hip) brings the HIP device declarations itself
#include <hip/hip_runtime.h>
#endif
#include "../../include/hj_mi355x.h"

namespace hj {

// VERTICAL PAIRS (hj_fusedv.h): 1 = the two pair slots of a thread are vertically adjacent on 3-D grids (shared middle-axis rows); the tilings
// (hj_api.hip, make_tiling) then give such launches an even row count.  Built and measured in round 6 (same bits, 12 -> 6 ds_read_b128 per
// thread and plane on the middle axis) and left OFF: 3 % SLOWER at 201^3 and 513^3 in a same-run A/B (profiles/r06_stage1_bound.txt, appendix 3)
#ifndef HJ_VPAIR
#define HJ_VPAIR 0
#endif
template <typename T> struct Lim;
template <> struct Lim<double> { static constexpr double tiny = 1e-99; static constexpr double lowest = -1.0e300; };
template <> struct Lim<float>  { static constexpr float  tiny = 1e-30f; static constexpr float lowest = -3.0e38f; };

// epsilon of the intended WENO5 formed as a product, then a sum (no FMA): what the term kernels and the host-side array path use
template <typename T> __device__ __forceinline__ T weno_eps_uncontracted(T max_d1sq) {
#pragma clang fp contract(off)
    const T a = T(1e-6) * max_d1sq;
    return a + Lim<T>::tiny;
}

template <typename T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }
template <> __device__ __forceinline__ double t_abs<double>(double x) { return __builtin_fabs(x); }
template <> __device__ __forceinline__ float  t_abs<float>(float x)   { return __builtin_fabsf(x); }

// Index division by a runtime divisor through one float reciprocal: ~9 instructions instead of the ~35 of the
// exact 32-bit sequence.  Valid for 0 <= a < 2^22, 0 < d (every index of a launch: hj_inst.hip checks nblocks):
// the float quotient is within 1 of the true one, and one correction in each direction settles it.  The kernels'
// setup code is instruction-bound (two waves per SIMD), its dozen index divisions were a quarter of it.
// max for the per-thread alpha accumulation of the plane loop: ONE v_max_f64.  fmax() costs three there (the
// compiler canonicalises both operands first -- the running maximum is a loop-carried value it cannot prove quiet);
// same result for every input that is not a signalling NaN (a NaN operand is ignored, as fmax does).
__device__ __forceinline__ double max_acc(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_acc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lo = a - |b|, hi = a + |b| folded into running min / max: one instruction each (v_min / v_max with source modifiers) -- the range
// pass of a substep (derivL = pc - hd, derivR = pc + hd: min(derivL, derivR) = pc - |hd|, max = pc + |hd|)
__device__ __forceinline__ void range_acc(double& mn, double& mx, double a, double b) {
    double lo, hi;
    asm("v_add_f64 %0, %1, -|%2|" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_add_f64 %0, %1, |%2|" : "=v"(hi) : "v"(a), "v"(b));
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(lo));
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(hi));
}
__device__ __forceinline__ void range_acc(float& mn, float& mx, float a, float b) {
    float lo, hi;
    asm("v_sub_f32 %0, %1, |%2|" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_add_f32 %0, %1, |%2|" : "=v"(hi) : "v"(a), "v"(b));
    asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(lo));
    asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(hi));
}
struct FDiv { int d; float r; };
__device__ __forceinline__ FDiv fdiv_make(int d) { return FDiv{d, __builtin_amdgcn_rcpf((float)d)}; }
__device__ __forceinline__ void fdivmod(int a, const FDiv& f, int& q, int& rem) {
    q = (int)((float)a * f.r);
    rem = a - q * f.d;
    if (rem < 0) { --q; rem += f.d; }
    if (rem >= f.d) { ++q; rem -= f.d; }
}
template <typename T> __device__ __forceinline__ T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> __device__ __forceinline__ T t_min(T a, T b) { return a < b ? a : b; }

// ghost cell k cells outside the edge: edge + k*slope, slope = mult*|edge-inner|*sign(edge),
// sign(0)=0.  `km` = k*mult (mult=+-1, exact).  Contraction is disabled so the value is
// bit-identical to NumPy's (edge + k*(mult*abs(d)*sign(edge))).
template <typename T>
__device__ __forceinline__ T ghost_value(T edge, T inner, T km) {
#pragma clang fp contract(off)
    T d = edge - inner;
    T sgn = (edge > T(0)) ? T(1) : ((edge < T(0)) ? T(-1) : T(0));
    T slope = t_abs(d) * sgn;
    T ks = km * slope;
    return edge + ks;
}

// post-step operator of HJIPDE_solve fused into the last RK stage (hji_solver.py:566-580):
// 1: min(new, previous state), 2: max(new, previous state); NaN propagates like NumPy's minimum/maximum
template <typename T>
__device__ __forceinline__ T post_step(int op, T a, T b) {
    if (op == 0) return a;
    T r;
    if (a != a) r = a;
    else if (b != b) r = b;
    else if (op == 1) r = a < b ? a : b;
    else r = a > b ? a : b;
    return r;
}

// ------------------------------------------------------------------------------------------
// v[0..6] = phi at i-3 .. i+3 along the differentiated axis.
// local tables: D1[j] = (v[j+1]-v[j])/dx (j=0..5), D2[j] = (D1[j+1]-D1[j])/(2dx) (j=0..4),
//               D3[j] = (D2[j+1]-D2[j])/(3dx) (j=0..3)
template <typename T> struct DD {
    T D1[6], D2[5], D3[4];
};

// Per-dimension stencil constants, computed on the HOST and passed in the kernel arguments so they
// live in SGPRs (there is no scalar fp64 multiply: derived in-kernel they would be wave-uniform
// values pinned in VGPR pairs for the whole march).
//   K[0] = 1/dx        K[1] = 1/(2dx)     K[2] = 1/(3dx)     K[3] = dx    K[4] = dx^2   K[5] = 2dx^2
//   as-shipped WENO5 (c = 1/(60dx)):  K[6] = 45c  K[7] = -9c  K[8] = c  K[9] = 15c  K[10] = -6c  K[11] = -20c
constexpr int HJ_NK = 12;
// 1 (default): ENO2 / ENO3 select their stencils on the reference's divided-difference tables, bit for bit (upwind_cd);
// 0: the lean arithmetic of rounds 1-2 on undivided differences (same result except at exact |D2| / |D3| ties)
#ifndef HJ_ENO_EXACT
#define HJ_ENO_EXACT 1
#endif
// NumPy operation order ("NP") for a scheme: with HJ_ENO_EXACT the ENO2 / ENO3 substep is evaluated operation by
// operation as the reference's array expressions evaluate it, contraction off from the derivatives to the RK update:
// the one-sided derivatives (upwind_cd), the Hamiltonian (HAM::eval<true>), the dissipation sum and -(H - diss)
// (lf_ydot), and the stage expression (rk_stage_out).  Why the WHOLE substep: the selectors alone are not enough --
// after one substep the state differs from the reference's in the last bit wherever an FMA was formed, and a near-tie
// |D2_a| ~ |D2_b| then flips on that noise (measured in round 3: with bit-faithful selectors only, 1.9 % of the cells of
// the symmetric 5-step golden still differed).  With every operation rounded as NumPy rounds it the states are the
// reference's bit for bit and so is every later choice.  The WENO5 arithmetics keep their contracted forms (linear /
// smooth in the data: no discrete choices to flip).
constexpr bool np_order(int scheme) { return HJ_ENO_EXACT && (scheme == HJ_ENO2 || scheme == HJ_ENO3); }
// the scheme an opt-in fast variant stands for (HJ_ENO2_FAST / HJ_ENO3_FAST: the lean arithmetic of upwind_cd, include/hj_mi355x.h)
constexpr int base_scheme(int scheme) { return scheme == HJ_ENO2_FAST ? HJ_ENO2 : (scheme == HJ_ENO3_FAST ? HJ_ENO3 : scheme); }
constexpr bool lean_eno(int scheme) { return scheme == HJ_ENO2_FAST || scheme == HJ_ENO3_FAST || (!HJ_ENO_EXACT && (scheme == HJ_ENO2 || scheme == HJ_ENO3)); }
template <typename T> inline void fill_stencil_constants(double dx, T* K) {
    const double inv = 1.0 / dx;
    K[0] = (T)inv;
    K[1] = (T)(0.5 * inv);
    K[2] = (T)((1.0 / 3.0) * inv);
    K[3] = (T)dx;
    K[4] = (T)(dx * dx);
    K[5] = (T)(2 * (dx * dx));
    const double c = inv * (1.0 / 60.0);
    K[6] = (T)(45 * c); K[7] = (T)(-9 * c); K[8] = (T)c;
    K[9] = (T)(15 * c); K[10] = (T)(-6 * c); K[11] = (T)(-20 * c);
}

template <typename T>
__device__ __forceinline__ void dd_tables(const T* v, const T* K, DD<T>& t) {
    const T inv_dx = K[0], h2 = K[1], h3 = K[2];
#pragma unroll
    for (int j = 0; j < 6; ++j) t.D1[j] = inv_dx * (v[j + 1] - v[j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) t.D2[j] = h2 * (t.D1[j + 1] - t.D1[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) t.D3[j] = h3 * (t.D2[j + 1] - t.D2[j]);
}

// the three third-order candidates per side (ENO3aHelper.py:116-189)
template <typename T>
__device__ __forceinline__ void eno3_candidates(const DD<T>& t, const T* K, T* dL, T* dR) {
    const T dx = K[3], dx2 = K[4], tdx2 = K[5];
    const T l = t.D1[2] + dx * t.D2[1];
    dL[0] = l + tdx2 * t.D3[0];
    dL[1] = l + tdx2 * t.D3[1];
    dL[2] = (t.D1[2] + dx * t.D2[2]) - dx2 * t.D3[2];
    const T r = t.D1[3] - dx * t.D2[2];
    dR[0] = r - dx2 * t.D3[1];
    dR[1] = r - dx2 * t.D3[2];
    dR[2] = (t.D1[3] - dx * t.D2[3]) + tdx2 * t.D3[3];
}

template <typename T>
__device__ __forceinline__ T weno_combine(T d0, T d1, T d2, T s0, T s1, T s2, T w0, T w1, T w2,
                                          T eps) {
    // weightWENO (upwind_first_weno5a.py:177-196): alpha_k = w_k/(s_k+eps)^2.
    // Same quotient written with ONE division: multiply through by prod (s_k+eps)^2, each
    // factor first scaled by 1/eps so the products stay O(1..1e30) (eps >= 1e-99 > 0).
    const T ie = T(1) / eps;
    T q0 = (s0 + eps) * ie, q1 = (s1 + eps) * ie, q2 = (s2 + eps) * ie;
    q0 *= q0; q1 *= q1; q2 *= q2;
    const T a0 = w0 * (q1 * q2), a1 = w1 * (q0 * q2), a2 = w2 * (q0 * q1);
    return (a0 * d0 + a1 * d1 + a2 * d2) / (a0 + a1 + a2);
}

template <typename T>
__device__ __forceinline__ void weno_smooth(T v1, T v2, T v3, T v4, T v5, T& s1, T& s2, T& s3) {
    const T c = T(13) / T(12), q = T(0.25);
    T a = v1 - T(2) * v2 + v3, b = v1 - T(4) * v2 + T(3) * v3;
    s1 = c * a * a + q * b * b;
    a = v2 - T(2) * v3 + v4; b = v2 - v4;
    s2 = c * a * a + q * b * b;
    a = v3 - T(2) * v4 + v5; b = T(3) * v3 - T(4) * v4 + v5;
    s3 = c * a * a + q * b * b;
}

// eps: only used by HJ_WENO5 (= 1e-6*max(D1^2)+tiny for this dim)
template <int SCHEME_IN, typename T>
__device__ __forceinline__ void upwind(const T* v, const T* K, T eps, T& L, T& R) {
    constexpr int SCHEME = base_scheme(SCHEME_IN);
    DD<T> t;
    dd_tables(v, K, t);
    const T dx = K[3];
    if constexpr (SCHEME == HJ_ENO2) {
        // upwind_first_eno2.py:97-148 in local indices
        const bool sl = t_abs(t.D2[1]) < t_abs(t.D2[2]);
        const bool sr = t_abs(t.D2[2]) < t_abs(t.D2[3]);
        L = t.D1[2] + dx * (sl ? t.D2[1] : t.D2[2]);
        R = t.D1[3] - dx * (sr ? t.D2[2] : t.D2[3]);
    } else {
        T dL[3], dR[3];
        eno3_candidates(t, K, dL, dR);
        if constexpr (SCHEME == HJ_ENO3) {
            // upwind_first_eno3a.py:105-141: strict '<', ties go right
            const bool sL0 = t_abs(t.D2[1]) < t_abs(t.D2[2]);
            const bool sL1 = t_abs(t.D2[2]) < t_abs(t.D2[3]);
            const bool sT0 = t_abs(t.D3[0]) < t_abs(t.D3[1]);
            const bool sT1 = t_abs(t.D3[1]) < t_abs(t.D3[2]);
            const bool sT2 = t_abs(t.D3[2]) < t_abs(t.D3[3]);
            // left: LL = sT0&sL0, M = (sT1&!sL0)|(!sT0&sL0), RR = !sT1&!sL0
            L = sL0 ? (sT0 ? dL[0] : dL[1]) : (sT1 ? dL[1] : dL[2]);
            R = sL1 ? (sT1 ? dR[0] : dR[1]) : (sT2 ? dR[1] : dR[2]);
        } else if constexpr (SCHEME == HJ_WENO5_ASSHIPPED) {
            // all smoothness estimates are (rounding-level) zero as shipped, so
            // weightWENO returns the fixed-weight combination.
            L = T(0.1) * dL[0] + T(0.6) * dL[1] + T(0.3) * dL[2];
            R = T(0.3) * dR[0] + T(0.6) * dR[1] + T(0.1) * dR[2];
        } else {
            T s1, s2, s3;
            weno_smooth(t.D1[0], t.D1[1], t.D1[2], t.D1[3], t.D1[4], s1, s2, s3);
            L = weno_combine(dL[0], dL[1], dL[2], s1, s2, s3, T(0.1), T(0.6), T(0.3), eps);
            weno_smooth(t.D1[5], t.D1[4], t.D1[3], t.D1[2], t.D1[1], s1, s2, s3);
            // dR[0]=psi3, dR[1]=psi2, dR[2]=psi1 (upwind_first_weno5a.py:136-147)
            R = weno_combine(dR[0], dR[1], dR[2], s3, s2, s1, T(0.3), T(0.6), T(0.1), eps);
        }
    }
}

// Centred costate pc = (L+R)/2 and half jump hd = (R-L)/2: all the Lax-Friedrichs term needs
// (term_lax_friedrich.py:108, artificial_diss_glf.py:91,100).  The as-shipped WENO5 is linear in
// phi, so both are short fixed stencils on the undivided differences u_j = v[j+1]-v[j]:
//   L = (2u0 - 13u1 + 47u2 + 27u3 - 3u4)/(60dx),  R = (-3u1 + 27u2 + 47u3 - 13u4 + 2u5)/(60dx)
// (SURVEY Appendix B), which collapse to the stencils below.  Nonlinear schemes go through upwind<>.
// per-dimension constants of the intended WENO5 that depend on the (device-side) epsilon
template <typename T> struct WenoK {
    T c13, c4;
};
template <typename T>
__device__ __forceinline__ WenoK<T> weno_consts(T eps, const T* K) {
    WenoK<T> w;
    const T e2 = eps * K[4];            // epsilon in undivided units: eps*dx^2
    const T ie = T(1) / e2;
    w.c13 = (T(13) / T(12)) * ie;
    w.c4 = T(0.25) * ie;
    return w;
}

// ---- the intended WENO5 in pieces (round 5).  u_j = v[j+1] - v[j] (undivided), candidates x6, smoothness x dx^2 (so is epsilon:
// WenoK), weights x10, the two quotients L' = N_L/D_L, R' = N_R/D_R over one reciprocal; pc' = L'+R', hd' = R'-L' with p = pc'/(12dx).
// The smoothness values enter as q_k = ((S_k + eps)/eps)^2 = (c13 t^2 + c4 b^2 + 1)^2 with t a second difference and b the
// first-difference part.  The RIGHT-biased q's of cell i are the LEFT-biased q's of cell i+1 in reverse order -- term by term the same
// expressions on the same operands (r3(i) = l1(i+1): second difference t1, first-difference part t1 + 2(u3 - u2); r2(i) = l2(i+1):
// t2 and +-(u4 - u2), squared; r1(i) = l3(i+1): t3 and t3 - 2(u4 - u3)) -- so a kernel that walks a line forms three q's per cell
// instead of six and the results equal the cell-by-cell form bit for bit (upwind_cd<HJ_WENO5>, which the direct kernels call).
template <typename T> struct WenoLine { T u[6], t[4]; };
template <typename T>
__device__ __forceinline__ WenoLine<T> weno5_line(const T* v) {
    WenoLine<T> ln;
#pragma unroll
    for (int j = 0; j < 6; ++j) ln.u[j] = v[j + 1] - v[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) ln.t[j] = (ln.u[j] + ln.u[j + 2]) - T(2) * ln.u[j + 1];
    return ln;
}
template <typename T>
__device__ __forceinline__ T weno5_q(T t, T b, const WenoK<T>& wk) {
    T q = wk.c13 * (t * t) + (wk.c4 * (b * b) + T(1));
    return q * q;
}
// left-biased q's of the cell in the middle of the line (v[3]): l1, l2, l3
template <typename T>
__device__ __forceinline__ void weno5_left_q(const WenoLine<T>& ln, const WenoK<T>& wk, T* lq) {
    const T s1 = ln.u[2] - ln.u[1], s2 = ln.u[3] - ln.u[2];
    lq[0] = weno5_q(ln.t[0], ln.t[0] + T(2) * s1, wk);
    lq[1] = weno5_q(ln.t[1], ln.u[1] - ln.u[3], wk);
    lq[2] = weno5_q(ln.t[2], ln.t[2] - T(2) * s2, wk);
}
// right-biased q's: r1, r2, r3  ( = l3, l2, l1 of the next cell of the line)
template <typename T>
__device__ __forceinline__ void weno5_right_q(const WenoLine<T>& ln, const WenoK<T>& wk, T* rq) {
    const T s2 = ln.u[3] - ln.u[2], s3 = ln.u[4] - ln.u[3];
    rq[0] = weno5_q(ln.t[3], ln.t[3] - T(2) * s3, wk);
    rq[1] = weno5_q(ln.t[2], ln.u[4] - ln.u[2], wk);
    rq[2] = weno5_q(ln.t[1], ln.t[1] + T(2) * s2, wk);
}
template <typename T> struct WenoCand;
template <typename T> __device__ __forceinline__ WenoCand<T> weno5_candidates(const WenoLine<T>& ln);
template <typename T> __device__ __forceinline__ void weno5_combine_cand(const WenoCand<T>& cd, const T* lq, const T* rq, T& pc, T& hd);
template <typename T>
__device__ __forceinline__ void weno5_combine(const WenoLine<T>& ln, const T* lq, const T* rq, T& pc, T& hd) {
    // candidates (x6): left phi1..3 = F1,F2,F3; right psi1 = G1, psi2 = F3, psi3 = F2;  weights x10: (.1,.6,.3)/q_k^2 multiplied through by
    // q1^2 q2^2 q3^2;  one reciprocal for both quotients (products stay < 1e62 in fp64): v_rcp_f64 (relative error 4.6e-8 on gfx950) + ONE
    // Newton step = 2.2e-15 (tools/probes/rcp_probe.hip, profiles/r04_weno5_smoothness_terms.txt); the scheme is checked at 1e-11
    weno5_combine_cand(weno5_candidates(ln), lq, rq, pc, hd);
}
// the same in two halves, for a caller that obtains the right-biased values late (from another lane, through LDS): the four candidate
// values first, so that the line itself need not stay in registers
template <typename T> struct WenoCand { T F1, F2, F3, G1; };
template <typename T>
__device__ __forceinline__ WenoCand<T> weno5_candidates(const WenoLine<T>& ln) {
    const T* u = ln.u;
    WenoCand<T> c;
    c.F1 = T(2) * u[0] + (T(-7) * u[1] + T(11) * u[2]);
    c.F2 = T(5) * u[2] + (T(2) * u[3] - u[1]);
    c.F3 = T(2) * u[2] + (T(5) * u[3] - u[4]);
    c.G1 = T(2) * u[5] + (T(-7) * u[4] + T(11) * u[3]);
    return c;
}
template <typename T>
__device__ __forceinline__ void weno5_combine_cand(const WenoCand<T>& cd, const T* lq, const T* rq, T& pc, T& hd) {
    const T F1 = cd.F1, F2 = cd.F2, F3 = cd.F3, G1 = cd.G1;
    const T A1 = lq[1] * lq[2], A2 = T(6) * (lq[0] * lq[2]), A3 = T(3) * (lq[0] * lq[1]);
    const T B1 = rq[1] * rq[2], B2 = T(6) * (rq[0] * rq[2]), B3 = T(3) * (rq[0] * rq[1]);
    const T NL = A1 * F1 + (A2 * F2 + A3 * F3), DL = A1 + (A2 + A3);
    const T NR = B1 * G1 + (B2 * F3 + B3 * F2), DR = B1 + (B2 + B3);
    if constexpr (sizeof(T) == 8) {
        const T den = DL * DR;
        T rc = __builtin_amdgcn_rcp(den);
        rc = rc + rc * (T(1) - den * rc);
        const T a = NL * DR, b = NR * DL;
        pc = (a + b) * rc;
        hd = (b - a) * rc;
    } else {
        const T Lq = NL / DL, Rq = NR / DR;
        pc = Lq + Rq;
        hd = Rq - Lq;
    }
}
// a cell of a line that is walked cell by cell (the axis-0 march): carry[] holds its left-biased q's on entry and those of the NEXT cell
// on return (its own right-biased ones, reversed)
template <typename T>
__device__ __forceinline__ void weno5_cd_carry(const T* v, const WenoK<T>& wk, T* carry, T& pc, T& hd) {
    const WenoLine<T> ln = weno5_line(v);
    T rq[3];
    weno5_right_q(ln, wk, rq);
    weno5_combine(ln, carry, rq, pc, hd);
    carry[0] = rq[2]; carry[1] = rq[1]; carry[2] = rq[0];
}
// two adjacent cells of a line (w[0..7], the cells at w[3] and w[4]): nine q's instead of twelve
template <typename T>
__device__ __forceinline__ void weno5_cd_pair(const T* w, const WenoK<T>& wk, T& pc0, T& hd0, T& pc1, T& hd1) {
    const WenoLine<T> l0 = weno5_line(w), l1 = weno5_line(w + 1);
    T lq0[3], m[3], rq1[3];
    weno5_left_q(l0, wk, lq0);
    weno5_right_q(l0, wk, m);              // right of the first cell = left of the second, reversed
    weno5_right_q(l1, wk, rq1);
    weno5_combine(l0, lq0, m, pc0, hd0);
    const T lq1[3] = {m[2], m[1], m[0]};
    weno5_combine(l1, lq1, rq1, pc1, hd1);
}

template <int SCHEME, typename T>
__device__ __forceinline__ void upwind_cd(const T* v, const T* K, T eps, const WenoK<T>& wk, T& pc, T& hd) {
    if constexpr (SCHEME == HJ_WENO5_ASSHIPPED) {
        // in terms of phi: with D_k = phi[i+k]-phi[i-k], S_k = phi[i+k]+phi[i-k]
        //   pc = (45 D1 - 9 D2 + D3)/(60dx)            (6th-order central difference)
        //   hd = (S3 - 6 S2 + 15 S1 - 20 phi_i)/(60dx)  (scaled 6th difference)
        // returned WITHOUT the 1/(60dx) factor (see the scaling note at the Hamiltonians):
        // the coefficients are compile-time constants.
        const T D1 = v[4] - v[2], D2 = v[5] - v[1], D3 = v[6] - v[0];
        const T S1 = v[4] + v[2], S2 = v[5] + v[1], S3 = v[6] + v[0];
        pc = T(45) * D1 + (T(-9) * D2 + D3);
        hd = T(15) * S1 + (T(-6) * S2 + (S3 + T(-20) * v[3]));
    } else if constexpr (SCHEME == HJ_WENO5) {
        // Intended WENO5 (O&F 3.32-3.41; ENO3bHelper.py:135-160) on UNDIVIDED differences: weno5_* above (the six smoothness
        // values of a cell formed here; the tiled kernels share them between neighbouring cells of a line)
        const WenoLine<T> ln = weno5_line(v);
        T lq[3], rq[3];
        weno5_left_q(ln, wk, lq);
        weno5_right_q(ln, wk, rq);
        weno5_combine(ln, lq, rq, pc, hd);
    } else if constexpr ((SCHEME == HJ_ENO3 || SCHEME == HJ_ENO2) && HJ_ENO_EXACT) {
        // Bit-faithful ENO (round 3; the drop-in default).  The reference chooses its stencil by comparing
        // |D2| / |D3| of the DIVIDED-difference tables D1 = dxInv*(g[j+1]-g[j]), D2 = (0.5*dxInv)*(dD1),
        // D3 = ((1/3)*dxInv)*(dD2) (ENO3aHelper.py:78-88, upwind_first_eno3a.py:105-128, upwind_first_eno2.py:80-84,
        // 136-141).  On exactly symmetric data those moduli tie, and which side wins a tie depends on how the
        // operands were ROUNDED: comparing the undivided differences (the lean branch below) is the same test in
        // exact arithmetic but resolves ties differently.  Here the tables are formed in the reference's operation
        // order with contraction off, so every comparison sees NumPy's operands bit for bit; the selected candidate
        // is then formed in the reference's order as well (ENO3aHelper.py:140-187: D1 + coeff*D2, then += coeff*D3),
        // also uncontracted: derivL / derivR equal the reference's bitwise on finite data.  Selection comes first
        // (an unselected candidate is never formed): see DESIGN.md section 2 for non-finite data.
        // Returns the true centred costate and half jump (sc[d] = 1).
#pragma clang fp contract(off)
        DD<T> t;
        dd_tables(v, K, t);
        const T dx = K[3], dx2 = K[4], tdx2 = K[5];
        const bool sL0 = t_abs(t.D2[1]) < t_abs(t.D2[2]), sL1 = t_abs(t.D2[2]) < t_abs(t.D2[3]);   // strict '<': ties go right
        T L, R;
        if constexpr (SCHEME == HJ_ENO2) {
            const T eL = dx * (sL0 ? t.D2[1] : t.D2[2]);
            const T eR = dx * (sL1 ? t.D2[2] : t.D2[3]);
            L = t.D1[2] + eL;
            R = t.D1[3] - eR;
        } else {
            const bool sT0 = t_abs(t.D3[0]) < t_abs(t.D3[1]), sT1 = t_abs(t.D3[1]) < t_abs(t.D3[2]);
            const bool sT2 = t_abs(t.D3[2]) < t_abs(t.D3[3]);
            // left:  LL = sT0 & sL0 -> D2[1], +2dx^2 D3[0];  M = (sT1 & !sL0) | (!sT0 & sL0): written as the reference
            // forms it, going left on D2 then right on D3 -> D2[1], +2dx^2 D3[1];  RR = !sT1 & !sL0 -> D2[2], -dx^2 D3[2]
            const bool lRR = !sL0 && !sT1;
            const T l2 = dx * (lRR ? t.D2[2] : t.D2[1]);
            const T l3a = tdx2 * ((sL0 && sT0) ? t.D3[0] : t.D3[1]);
            const T l3b = dx2 * t.D3[2];
            const T lb = t.D1[2] + l2;
            L = lRR ? (lb - l3b) : (lb + l3a);
            // right: LL -> D2[2], -dx^2 D3[1];  M -> D2[2], -dx^2 D3[2];  RR -> D2[3], +2dx^2 D3[3]
            const bool rRR = !sL1 && !sT2;
            const T r2 = dx * (rRR ? t.D2[3] : t.D2[2]);
            const T r3a = dx2 * ((sL1 && sT1) ? t.D3[1] : t.D3[2]);
            const T r3b = tdx2 * t.D3[3];
            const T rb = t.D1[3] - r2;
            R = rRR ? (rb + r3b) : (rb - r3a);
        }
        pc = T(0.5) * (L + R);
        hd = T(0.5) * (R - L);
    } else if constexpr (base_scheme(SCHEME) == HJ_ENO3) {
        // lean ENO3 (HJ_ENO3_FAST, or -DHJ_ENO_EXACT=0; rounds 1-2): selectors on the undivided differences
        // ENO3 (upwind_first_eno3a.py:105-141, ENO3aHelper.py:116-189) on UNDIVIDED differences
        // u_j = v[j+1]-v[j], s_j = u_{j+1}-u_j, t_j = s_{j+1}-s_j  (D1 = u/dx, D2 = s/(2dx^2), D3 = t/(6dx^3);
        // the |D2| / |D3| comparisons are comparisons of |s| / |t|), select FIRST, then form the one
        // candidate that was chosen:  dx*L = u2 + s/2 + {t/3 | -t/6},  dx*R = u3 - s/2 + {-t/6 | t/3}.
        // Returns pc' = L'+R', hd' = R'-L' with p = pc'/(2dx)  (sc[d] = 1/(2dx)).
        const T u0 = v[1] - v[0], u1 = v[2] - v[1], u2 = v[3] - v[2];
        const T u3 = v[4] - v[3], u4 = v[5] - v[4], u5 = v[6] - v[5];
        const T s0 = u1 - u0, s1 = u2 - u1, s2 = u3 - u2, s3 = u4 - u3, s4 = u5 - u4;
        const T t0 = s1 - s0, t1 = s2 - s1, t2 = s3 - s2, t3 = s4 - s3;
        const bool sL0 = t_abs(s1) < t_abs(s2), sL1 = t_abs(s2) < t_abs(s3);          // strict '<': ties go right
        const bool sT0 = t_abs(t0) < t_abs(t1), sT1 = t_abs(t1) < t_abs(t2), sT2 = t_abs(t2) < t_abs(t3);
        const T third = T(1) / T(3), sixth = T(1) / T(6);
        // the three corrections per side, then TWO levels of selection (a 64-bit select is two
        // v_cndmask): the middle candidate is written with the left second difference on the left side
        // and its mirror on the right side, as ENO3aHelper.py:132-168 does
        const T h1 = T(0.5) * s1, h2 = T(0.5) * s2, h3 = T(0.5) * s3;
        const T qL0 = third * t0 + h1, qL1 = third * t1 + h1, qL2 = h2 - sixth * t2;
        const T qR0 = -sixth * t1 - h2, qR1 = -sixth * t2 - h2, qR2 = third * t3 - h3;
        // left:  sL0 ? (sT0 ? c0 : c1) : (sT1 ? c1 : c2);   right: sL1 ? (sT1 ? c0 : c1) : (sT2 ? c1 : c2)
        const T Lq = u2 + (sL0 ? (sT0 ? qL0 : qL1) : (sT1 ? qL1 : qL2));
        const T Rq = u3 + (sL1 ? (sT1 ? qR0 : qR1) : (sT2 ? qR1 : qR2));
        pc = Lq + Rq;
        hd = Rq - Lq;
    } else if constexpr (base_scheme(SCHEME) == HJ_ENO2) {
        // lean ENO2 (upwind_first_eno2.py:97-148): dx*L = u2 + sel(|s1|<|s2|, s1, s2)/2, dx*R = u3 - sel(|s2|<|s3|, s2, s3)/2
        const T u1 = v[2] - v[1], u2 = v[3] - v[2], u3 = v[4] - v[3], u4 = v[5] - v[4];
        const T s1 = u2 - u1, s2 = u3 - u2, s3 = u4 - u3;
        const T Lq = u2 + T(0.5) * ((t_abs(s1) < t_abs(s2)) ? s1 : s2);
        const T Rq = u3 + T(-0.5) * ((t_abs(s2) < t_abs(s3)) ? s2 : s3);
        pc = Lq + Rq;
        hd = Rq - Lq;
    } else {
        T L, R;
        upwind<SCHEME, T>(v, K, eps, L, R);
        pc = T(0.5) * (L + R);
        hd = T(0.5) * (R - L);
    }
}

// ------------------------------------------------------------------------------------------
// Native Hamiltonians.  eval(): idx[d] = node index per dim, p[d] = centred costate
// 0.5*(derivL+derivR) (term_lax_friedrich.py:108); returns H and alpha[d] = partialFunc(.., d).
#ifndef HJ_PAR_SLOTS
#define HJ_PAR_SLOTS 8
#endif
template <typename T> struct HamTables {
    const T* coord[HJ_MAX_DIM];  // grid.vs[d]
    const T* aux[4];             // Hamiltonian-specific 1-D tables
    T par[HJ_PAR_SLOTS];         // the Hamiltonian's parameters (built-in systems: up to 4; run-time expressions: up to 8)
    // Hamiltonians whose alpha depends on the costate RANGE (round 5; artificial_diss_glf.py:80-99 hands partialFunc derivMin / derivMax):
    // 2*ND order-preserving keys written by the range pass of the substep (MODE 3 of the tiled kernels): [d] = key(max(derivL_d, derivR_d)
    // over the grid), [ND + d] = key(-min(...)); null for everybody else
    const unsigned long long* range;
    // ... and the LOCAL Lax-Friedrichs variants for such Hamiltonians (round 5, late; diss_local_laxfried.py:84-121, diss_localsq_laxfried.py:
    // 87-104): 0 the global range above for every dimension (GLF); 1 (LLF) alpha_i is evaluated with the range of dimension i replaced by
    // the NODE's own [min(p_i^-, p_i^+), max(p_i^-, p_i^+)]; 2 (LLLF) with the node's own range in EVERY dimension (no range pass at all).
    // The step bound is then 1 / max_x sum_i alpha_i(x) / dx_i (lf_local_bound_term below).  Ignored by Hamiltonians that do not read the range.
    int local_mode;
};
// value of an order-preserving key (max_key below)
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// Interface (all three systems):
//   Cell  cell(P, idx, sc)  -- constants of a grid column (depend on idx[1..] only); the fused kernel
//                              evaluates this once per block, outside the axis-0 march
//   Plane plane(P, i0, sc)  -- wave-uniform values of one axis-0 plane
//   eval(P, cell, plane, sc, q, H, alpha_s)
// Scaling: the derivative stencils may hand over UNSCALED costates q with p_d = sc[d]*q_d (the
// as-shipped WENO5 does, sc[d] = 1/(60 dx_d), so that its stencil coefficients are compile-time
// constants instead of 18 runtime SGPR pairs).  cell/plane fold sc into their constants; eval
// returns H(x, p) exactly and alpha_s[d] = sc[d]*alpha_d (what multiplies the unscaled half-jump).
// The kernel divides the max of alpha_s by sc[d] before publishing it.  Other schemes pass sc = 1.
template <typename T> struct HamDubinsRel {
    static constexpr int ND = 3;
    static constexpr int ID = HJ_HAM_DUBINS_REL;
    // H = p1(v_e - v_p cos x3) - p2 v_p sin x3 - w|p1 x2 - p2 x1 - p3| + w|p3|   (:83-88)
    // alpha = { |v_e - v_p cos x3| + |w x2|, |v_p sin x3| + |w x1|, w_e + w_p }      (:106-111)
    // alpha[d] varies along the march (axis 0) only for the dims in this mask; the others are
    // column constants and their max is taken once, outside the plane loop
    static constexpr unsigned PLANE_DEP = 0x2;
    // with p_d = sc_d q_d:  a' = sc0 a, b' = sc1 b, x1' = sc0 x1, x0' = sc1 x0
    struct Cell { T a, b, x1, alpha0; };      // a', b', x1', sc0*alpha0
    struct Plane { T x0, awx0; };             // x0', sc1*|w x0|
    // cell_raw issues the table loads, cell_fin does the arithmetic: the fused kernels put their own loads between
    struct Raw { T c, s, x1; };
    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>& P, const int* idx) {
        return Raw{P.aux[0][idx[2]], P.aux[1][idx[2]], P.coord[1][idx[1]]};
    }
    // the next cell along the CONTIGUOUS axis (pair kernel): what does not depend on that axis is taken over from the first
    // cell, so that the compiler sees ONE value and shares everything derived from it between the two cells of a pair
    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>& P, const int* idx, const Raw& first) {
        return Raw{P.aux[0][idx[2]], P.aux[1][idx[2]], first.x1};
    }
    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>& P, const Raw& r, const T* sc) {
        // contraction off (once per thread, in the setup): v_e - v_p cos x3 rounded as NumPy rounds it
        // (dubins_relative.py:84,108), so that alpha -- and with it stepBound and deltaT -- is the reference's bit for bit
#pragma clang fp contract(off)
        Cell c;
        const T a = P.par[0] - P.par[1] * r.c;
        const T b = P.par[1] * r.s;
        const T x1 = r.x1;
        c.alpha0 = sc[0] * (t_abs(a) + t_abs(P.par[2] * x1));
        c.a = sc[0] * a;
        c.b = sc[1] * b;
        c.x1 = sc[0] * x1;
        return c;
    }
    __device__ static __forceinline__ Cell cell(const HamTables<T>& P, const int* idx, const T* sc) {
        return cell_fin(P, cell_raw(P, idx), sc);
    }
    __device__ static __forceinline__ Plane plane(const HamTables<T>& P, int i0, const T* sc) {
        Plane u;
        const T x0 = P.coord[0][i0];
        u.x0 = sc[1] * x0;
        u.awx0 = sc[1] * t_abs(P.par[2] * x0);
        return u;
    }
    // NP: the reference's expression (dubins_relative.py:87-88) operation by operation, no FMA (sc = 1 there)
    template <bool NP = false>
    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane& u,
                                                const T* sc, const T* q, T& H, T* alpha) {
        const T w = P.par[2];
        const T p2 = sc[2] * q[2];
        if constexpr (NP) {
#pragma clang fp contract(off)
            H = q[0] * c.a - q[1] * c.b - w * t_abs(q[0] * c.x1 - q[1] * u.x0 - p2) + w * t_abs(p2);
        } else {
            H = q[0] * c.a - q[1] * c.b - w * t_abs(q[0] * c.x1 - q[1] * u.x0 - p2) + w * t_abs(p2);
        }
        alpha[0] = c.alpha0;
        alpha[1] = t_abs(c.b) + u.awx0;
        alpha[2] = sc[2] * P.par[3];
    }
};

template <typename T> struct HamDoubleIntegrator {
    static constexpr int ND = 2;
    static constexpr int ID = HJ_HAM_DOUBLE_INTEGRATOR;
    // H = -(p1 x2 - |p2| u)  (:71-74); alpha = { |x2|, |u| }  (:84-89)
    static constexpr unsigned PLANE_DEP = 0x0;
    struct Cell { T x1, alpha0; };            // sc0*x2, sc0*|x2|
    struct Plane { int unused; };
    struct Raw { T x1; };
    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>& P, const int* idx) { return Raw{P.coord[1][idx[1]]}; }
    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>& P, const int* idx, const Raw&) { return cell_raw(P, idx); }
    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>&, const Raw& r, const T* sc) {
        Cell c;
        c.x1 = sc[0] * r.x1;
        c.alpha0 = sc[0] * t_abs(r.x1);
        return c;
    }
    __device__ static __forceinline__ Cell cell(const HamTables<T>& P, const int* idx, const T* sc) {
        return cell_fin(P, cell_raw(P, idx), sc);
    }
    __device__ static __forceinline__ Plane plane(const HamTables<T>&, int, const T*) { return Plane{0}; }
    // NP: -(p1*x2 - |p2|*u) operation by operation (double_integrator.py:73-74)
    template <bool NP = false>
    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane&,
                                                const T* sc, const T* q, T& H, T* alpha) {
        const T us = sc[1] * P.par[0];
        if constexpr (NP) {
#pragma clang fp contract(off)
            H = -(q[0] * c.x1 - t_abs(q[1]) * us);
        } else {
            H = -(q[0] * c.x1 - t_abs(q[1]) * us);
        }
        alpha[0] = c.alpha0;
        alpha[1] = t_abs(us);
    }
};

template <typename T> struct HamDoublePendulum {
    static constexpr int ND = 4;
    static constexpr int ID = HJ_HAM_DOUBLE_PENDULUM;
    // state (th1, w1, th2, w2); drift f of the frictionless double pendulum with unit masses and
    // lengths, g = 9.8 (dynamics as in the reference's Tests/double_pendulum.py:29-51);
    // H = sum p_i f_i + u(|p2|+|p4|), alpha_i = |f_i| + u*[i in {1,3}].  aux = sin/cos tables.
    //
    // Round 5: the drift FACTORED by what it depends on.  With sd = sin(th2-th1), cd = cos(th2-th1), den = 2 - cd^2:
    //     f1 =  a*w1^2 + b1 + c*w2^2        a  = sd*cd/den,  c = sd/den,  b1 = g*(s2*cd - 2 s1)/den
    //     f3 = -a*w2^2 + b3 - 2c*w1^2                                     b3 = 2g*(s1*cd - s2)/den
    // a, b1, c, b3 depend on (th1, th2) only -- one ROW of the table below per (axis-0 plane, axis-2 index) -- and a cell is
    // left with four FMAs.  Until round 4 every cell of every plane recomputed sd, cd, den, a division and two ten-term
    // polynomials (a fifth of the C5 loop's vector instructions).  row() + eval_row() is the one arithmetic of the system:
    // eval() composes them per cell (direct / one-cell-per-lane / generic pair kernels), the 4-D pair kernel (hj_fused4v.h)
    // evaluates row() once per (plane, tile row) into LDS.  The costate scales sc[1], sc[3] are folded into the row.
    static constexpr unsigned PLANE_DEP = 0xA;
    struct Cell { T w1, w2, s2, c2; };
    struct Plane { T s1, c1; };
    using Raw = Cell;
    struct Row { T a1, b1, c1, a3, b3, c3; };          // f1s = sc1*f1 = a1*w1^2 + b1 + c1*w2^2;  f3s = sc3*f3 = a3*w2^2 + b3 + c3*w1^2
    struct TCell { T w1s, w1q, w2s, w2q; };            // sc0*w1, w1^2, sc2*w2, w2^2
    static constexpr int ROW_AXIS = 2;                 // the plane axis a Row varies along (besides axis 0)
    __device__ static __forceinline__ Row row(T s1, T c1, T s2, T c2, const T* sc) {
        const T G = T(9.8);
        const T sd = s2 * c1 - c2 * s1;  // sin(th2-th1)
        const T cd = c2 * c1 + s2 * s1;  // cos(th2-th1)
        const T den = T(2) - cd * cd;
        const T rden = T(1) / den;
        const T a = (sd * cd) * rden, c = sd * rden;
        const T b1 = (G * (s2 * cd - T(2) * s1)) * rden;
        const T b3 = (T(2) * G * (s1 * cd - s2)) * rden;
        Row r;
        r.a1 = sc[1] * a; r.b1 = sc[1] * b1; r.c1 = sc[1] * c;
        r.a3 = -(sc[3] * a); r.b3 = sc[3] * b3; r.c3 = T(-2) * (sc[3] * c);
        return r;
    }
    // the row of axis-0 plane i0 and axis-ROW_AXIS index irow
    __device__ static __forceinline__ Row row_at(const HamTables<T>& P, int i0, int irow, const T* sc) {
        return row(P.aux[0][i0], P.aux[1][i0], P.aux[2][irow], P.aux[3][irow], sc);
    }
    __device__ static __forceinline__ TCell tcell(const Cell& c, const T* sc) {
        return TCell{sc[0] * c.w1, c.w1 * c.w1, sc[2] * c.w2, c.w2 * c.w2};
    }
    // q: unscaled costates (p_d = sc[d] q_d); alpha comes back in the stencil's scale (sc[d]*alpha_d)
    __device__ static __forceinline__ void eval_row(const HamTables<T>& P, const TCell& c, const Row& r, const T* sc, const T* q, T& H, T* alpha) {
        const T us1 = sc[1] * P.par[0], us3 = sc[3] * P.par[0];
        const T f1s = r.a1 * c.w1q + (r.c1 * c.w2q + r.b1);
        const T f3s = r.a3 * c.w2q + (r.c3 * c.w1q + r.b3);
        H = q[0] * c.w1s + (q[1] * f1s + (q[2] * c.w2s + (q[3] * f3s + (us1 * t_abs(q[1]) + us3 * t_abs(q[3])))));
        alpha[0] = t_abs(c.w1s);
        alpha[1] = t_abs(f1s) + us1;
        alpha[2] = t_abs(c.w2s);
        alpha[3] = t_abs(f3s) + us3;
    }
    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>& P, const int* idx) {
        Cell c;
        c.w1 = P.coord[1][idx[1]];
        c.w2 = P.coord[3][idx[3]];
        c.s2 = P.aux[2][idx[2]];
        c.c2 = P.aux[3][idx[2]];
        return c;
    }
    // next cell along axis 3: only w2 changes -- the row is then computed once per PAIR in eval() (common subexpressions of
    // the two inlined calls)
    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>& P, const int* idx, const Raw& first) {
        Cell c = first;
        c.w2 = P.coord[3][idx[3]];
        return c;
    }
    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>&, const Raw& r, const T*) { return r; }
    __device__ static __forceinline__ Cell cell(const HamTables<T>& P, const int* idx, const T* sc) {
        return cell_fin(P, cell_raw(P, idx), sc);
    }
    __device__ static __forceinline__ Plane plane(const HamTables<T>& P, int i0, const T*) {
        Plane u;
        u.s1 = P.aux[0][i0];
        u.c1 = P.aux[1][i0];
        return u;
    }
    template <bool NP = false>      // (build-defined system: there is no reference expression order to follow)
    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane& pl,
                                                const T* sc, const T* q, T& H, T* alpha) {
        eval_row(P, tcell(c, sc), row(pl.s1, pl.c1, c.s2, c.c2, sc), sc, q, H, alpha);
    }
};
template <typename...> struct hj_void { typedef void type; };
// ---- TRANSPOSED MARCH (round 6).  A thin axis-0 slab (65 planes of 513 x 513 at N = 8) starves the axis-0 march: 133 tile columns x 3
// chunks = 399 workgroups on 256 CUs, 6 warm-up planes per 21-plane chunk (profiles/r05_thin_slab.txt).  Hamiltonian types that declare
// XPOSED describe the SAME system to a kernel whose axes (march, tile axis 1, rest) are the grid's (1, 0, rest): the slab axis becomes a tile
// axis, the march runs along the long axis 1.  Everything such a type sees is in KERNEL order (idx, sc, q, alpha); it produces the very values
// of its twin, and the dissipation sum / the published bound keys stay in GRID order (ham_gaxis) so that the results are bitwise equal.
template <typename H, typename = void> struct ham_xp { static constexpr bool value = false; };
template <typename H> struct ham_xp<H, typename hj_void<decltype(H::XPOSED)>::type> { static constexpr bool value = H::XPOSED; };
// kernel axis <-> grid axis (the swap of 0 and 1 is its own inverse)
template <typename H> __host__ __device__ constexpr int ham_gaxis(int k) { return ham_xp<H>::value ? (k == 0 ? 1 : (k == 1 ? 0 : k)) : k; }

template <typename T> struct HamDubinsRelX {
    static constexpr int ND = 3;
    static constexpr int ID = HJ_HAM_DUBINS_REL;
    static constexpr bool XPOSED = true;
    // kernel axis 1 = grid axis 0: alpha_x = |v_e - v_p cos x3| + |w x2| varies along the march (x2 is the march coordinate now);
    // kernel axis 0 = grid axis 1: alpha_y = |v_p sin x3| + |w x1| is a column constant
    static constexpr unsigned PLANE_DEP = 0x2;
    // sg0 = sc[1], sg1 = sc[0] (the scales of grid axes 0 and 1):  a' = sg0 a, b' = sg1 b, x0' = sg1 x0;  alpha1 = |b'| + sg1 |w x0|
    struct Cell { T a, b, x0, absa, alpha1; };
    struct Plane { T x1, awx1; };             // sg0 x1, |w x1|
    struct Raw { T c, s, x0; };
    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>& P, const int* idx) {
        return Raw{P.aux[0][idx[2]], P.aux[1][idx[2]], P.coord[0][idx[1]]};
    }
    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>& P, const int* idx, const Raw& first) {
        return Raw{P.aux[0][idx[2]], P.aux[1][idx[2]], first.x0};
    }
    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>& P, const Raw& r, const T* sc) {
        // every value as HamDubinsRel forms it (cell_fin / plane there), one rounding per operation
#pragma clang fp contract(off)
        Cell c;
        const T a = P.par[0] - P.par[1] * r.c;
        const T b = P.par[1] * r.s;
        c.absa = t_abs(a);
        c.a = sc[1] * a;
        c.b = sc[0] * b;
        c.x0 = sc[0] * r.x0;
        const T wx0 = P.par[2] * r.x0;
        const T awx0 = sc[0] * t_abs(wx0);
        c.alpha1 = t_abs(c.b) + awx0;
        return c;
    }
    __device__ static __forceinline__ Cell cell(const HamTables<T>& P, const int* idx, const T* sc) {
        return cell_fin(P, cell_raw(P, idx), sc);
    }
    __device__ static __forceinline__ Plane plane(const HamTables<T>& P, int im, const T* sc) {
#pragma clang fp contract(off)
        Plane u;
        const T x1 = P.coord[1][im];
        u.x1 = sc[1] * x1;
        const T wx1 = P.par[2] * x1;
        u.awx1 = t_abs(wx1);
        return u;
    }
    // q, alpha: KERNEL order (q[0] belongs to grid axis 1, q[1] to grid axis 0); the expression is HamDubinsRel::eval's, term by term
    template <bool NP = false>
    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane& u,
                                                const T* sc, const T* q, T& H, T* alpha) {
        const T w = P.par[2];
        const T p2 = sc[2] * q[2];
        if constexpr (NP) {
#pragma clang fp contract(off)
            H = q[1] * c.a - q[0] * c.b - w * t_abs(q[1] * u.x1 - q[0] * c.x0 - p2) + w * t_abs(p2);
        } else {
            H = q[1] * c.a - q[0] * c.b - w * t_abs(q[1] * u.x1 - q[0] * c.x0 - p2) + w * t_abs(p2);
        }
        {
#pragma clang fp contract(off)
            const T sum = c.absa + u.awx1;
            alpha[1] = sc[1] * sum;
        }
        alpha[0] = c.alpha1;
        alpha[2] = sc[2] * P.par[3];
    }
};

// does a Hamiltonian type factor its coefficients into rows (Row / row() / tcell() / eval_row(), as HamDoublePendulum does)?
template <typename H, typename = void> struct ham_has_rows { static constexpr bool value = false; };
template <typename H> struct ham_has_rows<H, typename hj_void<typename H::Row>::type> { static constexpr bool value = true; };

// does a Hamiltonian type read the costate range (HamUser with RANGE, hj_rtc.hip)?
template <typename H, typename = void> struct ham_reads_range { static constexpr bool value = false; };
template <typename H> struct ham_reads_range<H, typename hj_void<decltype(H::RANGE)>::type> { static constexpr bool value = H::RANGE; };

// H and alpha of one cell under the LOCAL Lax-Friedrichs rules (HamTables::local_mode 1 / 2) for a Hamiltonian that reads the costate range.
// pc / hd: centred costate and half jump in the stencil's scale (true p^-/+ = sc (pc -/+ hd)), u: the plane constants carrying the GLOBAL range.
template <bool NP, typename HAM, typename T>
__device__ __forceinline__ void lf_eval_local(const HamTables<T>& P, const typename HAM::Cell& c, const typename HAM::Plane& u,
                                              const T* sc, const T* pc, const T* hd, T& H, T* alpha) {
    constexpr int ND = HAM::ND;
    T lo[ND], hi[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const T a = sc[d] * (pc[d] - hd[d]), b = sc[d] * (pc[d] + hd[d]);
        lo[d] = a < b ? a : b;
        hi[d] = a < b ? b : a;
    }
    if (P.local_mode == 2) {
        typename HAM::Plane v = u;
#pragma unroll
        for (int d = 0; d < ND; ++d) { v.dmin[d] = lo[d]; v.dmax[d] = hi[d]; }
        HAM::template eval<NP>(P, c, v, sc, pc, H, alpha);
    } else {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            typename HAM::Plane v = u;
            v.dmin[i] = lo[i]; v.dmax[i] = hi[i];
            T Hi, ai[ND];
            HAM::template eval<NP>(P, c, v, sc, pc, Hi, ai);
            alpha[i] = ai[i];
            if (i == 0) H = Hi;
        }
    }
}

// ---- the tail of one cell's substep, shared by every kernel (fused, pair, stage-fused, direct)
// ydot = -(H - sum_d hd_d alpha_d): termLaxFriedrichs / artificialDissipationGLF (term_lax_friedrich.py:111-128,
// artificial_diss_glf.py:94-100: `diss += 0.5*derivDiff[i]*alpha`, hd = 0.5*(derivR - derivL)); alpha[] comes back
template <bool NP, typename HAM, typename T>
__device__ __forceinline__ T lf_ydot(const HamTables<T>& P, const typename HAM::Cell& c, const typename HAM::Plane& u,
                                     const T* sc, const T* pc, const T* hd, T* alpha) {
    T H;
    if constexpr (ham_reads_range<HAM>::value) {
        if (P.local_mode != 0) lf_eval_local<NP, HAM>(P, c, u, sc, pc, hd, H, alpha);
        else HAM::template eval<NP>(P, c, u, sc, pc, H, alpha);
    } else {
        HAM::template eval<NP>(P, c, u, sc, pc, H, alpha);
    }
    if constexpr (NP) {
#pragma clang fp contract(off)
        T diss = T(0);
#pragma unroll
        for (int g = 0; g < HAM::ND; ++g) diss = diss + hd[ham_gaxis<HAM>(g)] * alpha[ham_gaxis<HAM>(g)];      // grid order (ham_xp)
        return -(H - diss);
    } else {
        T diss = T(0);
#pragma unroll
        for (int g = 0; g < HAM::ND; ++g) diss += hd[ham_gaxis<HAM>(g)] * alpha[ham_gaxis<HAM>(g)];              // grid order (ham_xp)
        return -(H - diss);
    }
}
// the same with the Hamiltonian's coefficients handed over as a row (Hamiltonians that factor them: ham_has_rows)
template <bool NP, typename HAM, typename T>
__device__ __forceinline__ T lf_ydot_row(const HamTables<T>& P, const typename HAM::TCell& c, const typename HAM::Row& r,
                                         const T* sc, const T* pc, const T* hd, T* alpha) {
    T H;
    HAM::eval_row(P, c, r, sc, pc, H, alpha);
    if constexpr (NP) {
#pragma clang fp contract(off)
        T diss = T(0);
#pragma unroll
        for (int g = 0; g < HAM::ND; ++g) diss = diss + hd[ham_gaxis<HAM>(g)] * alpha[ham_gaxis<HAM>(g)];      // grid order (ham_xp)
        return -(H - diss);
    } else {
        T diss = T(0);
#pragma unroll
        for (int g = 0; g < HAM::ND; ++g) diss += hd[ham_gaxis<HAM>(g)] * alpha[ham_gaxis<HAM>(g)];              // grid order (ham_xp)
        return -(H - diss);
    }
}
// the stage expression of odeCFLn.  Contracted form (default): ca*y0 + cb*(y + dt*ydot) with (ca, cb) = (0,1), (3/4,1/4),
// (1/3,2/3), (1/2,1/2).  NP: the reference's own expressions, operation by operation -- y + dt*ydot (ode_cfl_3.py:151,184,
// 226), 0.25*(3*y0 + y2) (:193), (1/3)*(y0 + 2*y32) (:241), 0.5*(y0 + y2) (ode_cfl_2.py:201)
template <bool NP, typename T>
__device__ __forceinline__ T rk_stage_out(int stage, T ca, T cb, T dt, T y0, T y, T ydot) {
    if constexpr (NP) {
#pragma clang fp contract(off)
        const T step = dt * ydot;
        const T e = y + step;
        if (stage == HJ_STAGE_RK3_HALF) { const T a = T(3) * y0; return T(0.25) * (a + e); }
        if (stage == HJ_STAGE_RK3_FULL) { const T a = T(2) * e; return (T(1) / T(3)) * (y0 + a); }
        if (stage == HJ_STAGE_RK2_FULL) return T(0.5) * (y0 + e);
        return e;
    } else {
        return ca * y0 + cb * (y + dt * ydot);
    }
}

// ---- buffer (SRD) addressing: wave-uniform 128-bit descriptor in SGPRs + per-lane 32-bit byte
// offset, hardware range check (out-of-range loads return 0, stores are dropped)
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const T* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, /*stride*/ 0, (int)bytes, 0x00020000);
}
// off: per-lane byte offset (VGPR); soff: wave-uniform byte offset (SGPR), e.g. the plane;
// AUX: cache policy (gfx940+ encoding: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0>
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store(double v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    using V = decltype(__builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(V, v), r, off, soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
    using V = decltype(__builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(V, v), r, off, soff, AUX);
}

// order-preserving map double -> uint64 so atomicMax on the key is max on the value
__device__ __forceinline__ unsigned long long max_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
// atomicMax of a workgroup's maximum into a launch-wide key -- after LOOKING: the keys only grow, so a value that does not exceed what an
// agent-scope load shows can be dropped (a stale load costs an atomic, never a result).  Hundreds of workgroups finishing together used
// to queue hundreds of same-address atomics (~12 ns each, profiles/r05_range_path.txt) at the very end of a launch; on the shipped
// systems most workgroups carry the same maxima and now skip theirs (round 5: the kept CFL reduction cost 3-8 % of the 201^3 launch).
__device__ __forceinline__ void key_max(unsigned long long* p, double v) {
    const unsigned long long k = max_key(v);
#if defined(HJ_KEY_MAX_BLIND)        // A/B knob (tune builds): the round-4 form, an atomic whatever the key holds
    atomicMax(p, k);
#else
    if (k > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, k);
#endif
}

// Wavefront max / min of a double, every lane receives the result.  DPP data moves (row_shr 1,2,4,8 inside the four
// 16-lane rows, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 ends up with the
// reduction of all 64 lanes) instead of six ds_bpermute round trips through the LDS crossbar: the reductions sit in
// every workgroup's epilogue, which nothing overlaps.  Lanes without a source keep the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v, double ident) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <bool MAX>
__device__ __forceinline__ double wave_reduce(double v) {
    const double ident = MAX ? -__builtin_huge_val() : __builtin_huge_val();
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : fmin(a, b); };
    v = op(v, dpp_move<0x111, 0xF>(v, ident));      // row_shr:1
    v = op(v, dpp_move<0x112, 0xF>(v, ident));      // row_shr:2
    v = op(v, dpp_move<0x114, 0xF>(v, ident));      // row_shr:4
    v = op(v, dpp_move<0x118, 0xF>(v, ident));      // row_shr:8   -> lane 15 of each row holds the row's result
    v = op(v, dpp_move<0x142, 0xA>(v, ident));      // row_bcast:15 into rows 1, 3
    v = op(v, dpp_move<0x143, 0xC>(v, ident));      // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's result
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) { return wave_reduce<true>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<false>(v); }

}  // namespace hj
