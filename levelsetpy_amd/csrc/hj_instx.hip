// TRANSPOSED-MARCH launches (round 6): the substep kernels instantiated with the XPOSED twin of a Hamiltonian type (hj_device.h, ham_xp) --
// the march runs along grid axis 1, the slab axis 0 is a TILE axis of which the launch computes a window of planes.  What it is for: the thin
// slabs of the 8-GPU configurations (C4: 64-65 planes of 513 x 513), which the axis-0 march cuts into 399 short workgroups on 256 CUs
// (profiles/r05_thin_slab.txt).  Same per-cell arithmetic in the same order: bitwise the results of the axis-0 march.
// Compiled once per (dtype, Hamiltonian) with -DHJ_INST_T / -DHJ_INST_HAM like hj_inst.hip.  gfx950 only.
#include <mutex>
#include "hj_host.h"
#include "hj_fused.h"
#include "hj_fusedv.h"
#include "hj_fused4v.h"
#include "hj_launch.h"

namespace hjh {

template <typename HAM> struct xposed_of;
template <typename T> struct xposed_of<hj::HamDubinsRel<T>> { using type = hj::HamDubinsRelX<T>; };

// FusedArgs of a transposed 3-D launch: every per-axis field in KERNEL order (march = grid axis 1, tile axis 1 = grid axis 0)
template <typename T>
int fill_fused_args_xp3(hj_ctx* c, const SubstepCall& s, const Tiling& t, int scheme, FusedArgs<T, 3>& A, unsigned& grid_blocks) {
    EdgePlan ep;
    int rc = fill_fused_args<T, 3>(c, s, t, ep, scheme, true, A, grid_blocks);      // stage, Hamiltonian tables, chunk / block counts, LDS ring
    if (rc) return rc;
    const int g_of[3] = {1, 0, 2};
    const long long N0 = c->N[0], N1 = c->N[1], N2 = c->N[2];
    for (int k = 0; k < 3; ++k) {
        const int g = g_of[k];
        A.inv_dx[k] = (T)(1.0 / c->dx[g]);
        A.n[k] = (int)c->N[g];
        A.bc[k] = c->bc[g];
        A.km[k] = c->tz[g] ? T(-1) : T(1);
        fill_stencil_constants<T>(c->dx[g], A.K[k]);
        A.sc[k] = scheme_scale<T>(scheme, c->dx[g]);
        A.E[k] = t.E[k];
        A.ntile[k] = t.ntile[k];
    }
    A.stride0 = N2;                       // one step of the march = one row
    A.pstride[0] = 0;
    A.pstride[1] = (int)(N1 * N2);        // one step along tile axis 1 = one axis-0 plane
    A.pstride[2] = 1;
    A.halo_lo = A.halo_hi = 0;            // the march axis is never a slab axis
    A.xh_lo = c->halo_lo;
    A.xh_hi = c->halo_hi;
    A.xwin0 = (int)s.p0;
    A.xwin1 = (int)s.p1;
    A.xbase = (int)std::min<int64_t>(0, c->halo_lo ? s.p0 - HJ_STENCIL : 0);
    const long long hi = c->halo_hi ? std::max<long long>(s.p1 + HJ_STENCIL, N0) : N0;
    A.xspan = (unsigned)((hi - A.xbase - 1) * N1 * N2 * (long long)sizeof(T));
    A.plane_begin = 0;
    A.plane_end = (int)N1;
    A.plane_begin2 = A.plane_end2 = 0;
    A.npairs = 0;
    return HJ_OK;
}

template <typename T, typename HAMX, int SCHEME, int NT, int R, int KH, int OCC, int MODE>
int launch_xp_mode(hj_ctx* c, const SubstepCall& s, int nbuf) {
    auto kern = fused_pair_kernel<T, HAMX, SCHEME, NT, R, KH, OCC, MODE>;
    const long long N0 = c->N[0], N1 = c->N[1], N2 = c->N[2];
    const int64_t dims[4] = {N1, s.p1 - s.p0, N2, 1};
    const KernelCfg kp{NT, R, KH};
    Tiling t = make_tiling_dims(c, kp, 2, nbuf, dims);
    if (!t.ok) return HJ_XP_FALLBACK;
    const auto key = std::make_pair(reinterpret_cast<const void*>(kern), t.lds_bytes);
    auto it = c->occ_cache.find(key);
    int occ_blocks = it != c->occ_cache.end() ? it->second : 0;
    if (it == c->occ_cache.end()) {
        if (c->dry) occ_blocks = std::max(1, std::min(OCC * 256 / NT, (int)((size_t)(160 * 1024) / std::max<size_t>(1, t.lds_bytes))));
        else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_blocks, key.first, NT, t.lds_bytes) != hipSuccess || occ_blocks < 1) occ_blocks = 1;
        if (c->dry != 2) c->occ_cache.emplace(key, occ_blocks);     // (2: a planning look from a live context -- its estimate is not the device's answer)
    }
    // the span of a buffer descriptor: the rows of the window and its halo (fixed) + a chunk of the march and 3 rows either side
    const long long xbase = std::min<int64_t>(0, c->halo_lo ? s.p0 - HJ_STENCIL : 0);
    const long long hi = c->halo_hi ? std::max<long long>(s.p1 + HJ_STENCIL, N0) : N0;
    const double fixed = (double)(hi - xbase - 1) * (double)(N1 * N2) * (double)sizeof(T);
    if (fixed > 4.0e9) return HJ_XP_FALLBACK;
    choose_chunks(c, t, 0, N1, occ_blocks, 0, (double)N2 * (double)sizeof(T), fixed);
    if (!t.ok) return HJ_XP_FALLBACK;
    t.nchunks1 = t.nchunks;
    c->last_plan.ntiles = t.ntiles; c->last_plan.nchunks = t.nchunks; c->last_plan.nblocks = t.nblocks; c->last_plan.threads = NT;
    c->last_plan.wg_per_cu = occ_blocks; c->last_plan.lds_bytes = t.lds_bytes;
    c->last_kernel = "fused_pair_kernel (march along axis 1)";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = d < 3 ? t.E[d] : 0;
    c->last_nbuf = nbuf;
    c->last_nbase = 2;
    if (c->dry) return HJ_OK;
    if (c->debug) {
        fprintf(stderr, "[hj] transposed pair tiling NT=%d R=%d KH=%d OCC=%d window=[%lld,%lld) E=(%d,%d) pitch=%d ntiles=%d chunk=%d nchunks=%d blocks=%d wg/CU=%d lds=%zu\n",
                NT, R, KH, OCC, (long long)s.p0, (long long)s.p1, t.E[1], t.E[2], t.lpitch, t.ntiles, t.chunk, t.nchunks, t.nblocks, occ_blocks, t.lds_bytes);
        c->debug = 0;
    }
    FusedArgs<T, 3> A;
    memset(&A, 0, sizeof(A));
    A.max_d1sq = (const T*)(c->weno_src ? c->weno_src : c->weno_vals);
    A.bound = s.bound;
    unsigned grid_blocks = 0;
    {
        const int rc_fill = fill_fused_args_xp3<T>(c, s, t, SCHEME, A, grid_blocks);
        if (rc_fill) return rc_fill;
    }
    if (t.lds_bytes > 64 * 1024) {
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> granted_by_kernel;
        std::lock_guard<std::mutex> lock(mu);
        size_t& granted = granted_by_kernel[std::make_pair(c->device, reinterpret_cast<const void*>(kern))];
        if (granted < t.lds_bytes) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
            granted = t.lds_bytes;
        }
    }
    if (c->launch_stop) {
        hipExtLaunchKernelGGL(kern, dim3(grid_blocks), dim3(NT), (unsigned)t.lds_bytes, call_stream(c, s), nullptr, c->launch_stop, 0,
                              (const T*)s.y, (const T*)s.y0, (T*)s.out, A);
        c->launch_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(NT), t.lds_bytes, call_stream(c, s), (const T*)s.y, (const T*)s.y0, (T*)s.out, A);
    }
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

template <typename T, typename HAM, int SCHEME>
int launch_xp_cfg(hj_ctx* c, const SubstepCall& s) {
    using HAMX = typename xposed_of<HAM>::type;
    static_assert(HAM::ND == 3, "transposed march: 3-D grids (4-D: hj_fused4v.h)");
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0 && !c->no_plain;
    const int mode = plain ? (s.stage == HJ_STAGE_EULER ? 1 : 2) : 0;
    if constexpr (light_cfg(SCHEME, 3)) {
        // 512 threads x 2 pairs with the halo ring parked in LDS: the shape of the headline kernel
        const int nbuf = c->pair_ring == 0 ? 2 : 2 + c->pair_ah;
        if (mode == 1) return launch_xp_mode<T, HAMX, SCHEME, 512, 2, 2, 2, 1>(c, s, nbuf);
        if (mode == 2) return launch_xp_mode<T, HAMX, SCHEME, 512, 2, 2, 2, 2>(c, s, nbuf);
        return launch_xp_mode<T, HAMX, SCHEME, 512, 2, 2, 2, 0>(c, s, nbuf);
    } else {
        if (mode == 1) return launch_xp_mode<T, HAMX, SCHEME, 256, 1, 2, 2, 1>(c, s, 2);
        if (mode == 2) return launch_xp_mode<T, HAMX, SCHEME, 256, 1, 2, 2, 2>(c, s, 2);
        return launch_xp_mode<T, HAMX, SCHEME, 256, 1, 2, 2, 0>(c, s, 2);
    }
}

// HJ_OK: launched (or planned, dry contexts); HJ_XP_FALLBACK: this call has no transposed form -- the caller takes the ordinary launch
template <typename T, typename HAM>
int launch_xp(hj_ctx* c, const SubstepCall& s) {
    const int64_t N0 = c->N[0];
    if (c->ndim != 3 || s.q1 > s.q0 || s.gated || s.range_only || s.bound_pass || s.term || (s.scheme == HJ_WENO5 && (s.want_eps || s.eps_nrows > 0)) || s.dt_dev ||
        c->timing_dump || s.p1 - s.p0 < 1 || (s.p0 < 0 && !c->halo_lo) || (s.p1 > N0 && !c->halo_hi) || c->N[2] < 8 || c->N[1] < 8 ||
        c->total / N0 * (N0 + 2 * HJ_STENCIL * 4) >= (1ll << 31))
        return HJ_XP_FALLBACK;
    switch (s.scheme) {
        case HJ_ENO2: return launch_xp_cfg<T, HAM, HJ_ENO2>(c, s);
        case HJ_ENO3: return launch_xp_cfg<T, HAM, HJ_ENO3>(c, s);
        case HJ_WENO5: return launch_xp_cfg<T, HAM, HJ_WENO5>(c, s);
        case HJ_WENO5_ASSHIPPED: return launch_xp_cfg<T, HAM, HJ_WENO5_ASSHIPPED>(c, s);
        case HJ_ENO2_FAST: return launch_xp_cfg<T, HAM, HJ_ENO2_FAST>(c, s);
        case HJ_ENO3_FAST: return launch_xp_cfg<T, HAM, HJ_ENO3_FAST>(c, s);
    }
    return HJ_XP_FALLBACK;
}

template int launch_xp<HJ_INST_T, hj::HJ_INST_HAM<HJ_INST_T>>(hj_ctx*, const SubstepCall&);

}  // namespace hjh
