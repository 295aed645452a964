// What every launch of a tiled substep kernel shares, whatever the Hamiltonian type: the chunk plan of the launch (second
// plane range, edge ranges of a gated slab launch) and the FusedArgs block.  Used by the instantiations of hj_inst.hip
// (built-in Hamiltonians) and by hj_rtc.hip (Hamiltonians compiled at run time with hipRTC).
#pragma once
#include "hj_host.h"
#include "hj_fused.h"

namespace hjh {

struct EdgePlan { int echunk = 0, ne[2] = {0, 0}, edge_count = 0; };

// chunk_max > 0: no chunk longer than this many planes (kernels whose LDS use grows with the chunk: hj_fused4v.h)
inline int plan_chunks(hj_ctx* c, const SubstepCall& s, Tiling& t, int occ_blocks, EdgePlan& ep, int64_t chunk_max = 0) {
        choose_chunks(c, t, s.p0, s.p1, occ_blocks, chunk_max);
        if (!t.ok) return hjh::fail(HJ_EUNSUPPORTED, "axis-0 plane too large for the tiled kernel");
        t.nchunks1 = t.nchunks;
        if (s.q1 > s.q0) {     // second range: same chunk length
            t.nchunks += (int)((s.q1 - s.q0 + t.chunk - 1) / t.chunk);
            t.nblocks = t.nchunks * t.ntiles;
            t.bpx = (t.nblocks + 7) / 8;
        }
        if (s.gated) {
            // edge chunks of HJ_STENCIL planes each, ahead of everything else (hj_fused.h: logical_block, chunk_planes)
            ep.echunk = HJ_STENCIL;
            for (int w = 0; w < 2; ++w) ep.ne[w] = (int)((s.e1[w] - s.e0[w] + ep.echunk - 1) / ep.echunk);
            const int main_blocks = t.nblocks;
            ep.edge_count = (ep.ne[0] + ep.ne[1]) * t.ntiles;
            t.nblocks = main_blocks + ep.edge_count;
            t.bpx = (main_blocks + 7) / 8;
        }
    return HJ_OK;
}

// everything of FusedArgs but the intended WENO5's epsilon fields (eps_part / eps_rows) and the debug timing buffer
template <typename T, int ND>
int fill_fused_args(hj_ctx* c, const SubstepCall& s, const Tiling& t, const EdgePlan& ep, int scheme, bool pair,
                    FusedArgs<T, ND>& A, unsigned& grid_blocks) {
    long long st = 1;
    for (int d = ND - 1; d >= 0; --d) {
        A.inv_dx[d] = (T)(1.0 / c->dx[d]);
        A.n[d] = (int)c->N[d];
        A.bc[d] = c->bc[d];
        A.km[d] = c->tz[d] ? T(-1) : T(1);
        fill_stencil_constants<T>(c->dx[d], A.K[d]);
        A.sc[d] = scheme_scale<T>(scheme, c->dx[d]);
        A.pstride[d] = (d >= 1) ? (int)st : 0;
        if (d == 0) A.stride0 = st;
        st *= c->N[d];
        A.E[d] = t.E[d];
        A.ntile[d] = t.ntile[d];
    }
    for (int d = 0; d < ND; ++d) A.tb[d] = 0;
    if constexpr (ND == 4) { if (c->tile_block[0] > 0 && c->tile_block[1] > 0) { A.tb[1] = c->tile_block[0]; A.tb[2] = c->tile_block[1]; } }
    A.halo_lo = c->halo_lo;
    A.halo_hi = c->halo_hi;
    A.ntiles = t.ntiles;
    A.lpitch = t.lpitch;
    A.chunk = t.chunk;
    A.nchunks = t.nchunks;
    A.plane_begin = (int)s.p0;
    A.plane_end = (int)s.p1;
    A.plane_begin2 = (int)s.q0;
    A.plane_end2 = (int)s.q1;
    A.nchunks1 = t.nchunks1;
    A.nblocks = t.nblocks;
    if (t.nblocks >= (1 << 22)) return hjh::fail(HJ_EUNSUPPORTED, "more than 4 M workgroups in one launch (index arithmetic of the kernels)");
    A.blocks_per_xcd = t.bpx;
    // chunks marching pairwise in opposite directions (pair kernel; only in -DHJ_MAYDOWN=1 builds, hj_fused.h): not with edge ranges or a
    // second range in the launch
    A.npairs = (HJ_MAYDOWN && pair && c->pair_dirs && !s.gated && s.q1 <= s.q0 && ep.edge_count == 0) ? t.nchunks1 / 2 : 0;
    A.echunk = ep.echunk;
    A.nchunks_e1 = ep.ne[0];
    A.nchunks_e = ep.ne[0] + ep.ne[1];
    for (int w = 0; w < 2; ++w) { A.eplane[w][0] = (int)s.e0[w]; A.eplane[w][1] = (int)s.e1[w]; }
    A.edge_count = ep.edge_count;
    A.edge_bpx = (ep.edge_count + 7) / 8;
    A.edge_blocks = 8 * A.edge_bpx;
    A.gate = (s.gated && ep.edge_count > 0) ? c->gate : nullptr;
    c->gate_posted = A.gate ? ep.edge_count : 0;
    grid_blocks = (unsigned)(A.edge_blocks + t.bpx * 8);
    A.lds_nbuf = pair ? c->last_nbuf : 2;
    A.halo_ahead = (pair && c->last_nbuf > c->last_nbase) ? c->last_nbuf - c->last_nbase : 0;
    A.stage = s.stage;
    A.ydot_only = (s.stage == HJ_STAGE_YDOT);
    A.use_y0 = (s.stage >= HJ_STAGE_RK3_HALF);
    switch (s.stage) {                          // out = ca*y0 + cb*(y + dt*ydot)
        case HJ_STAGE_RK3_HALF: A.ca = T(0.75); A.cb = T(0.25); break;           // ode_cfl_3.py:184,193
        case HJ_STAGE_RK3_FULL: A.ca = T(1.0 / 3.0); A.cb = T(2.0 / 3.0); break; // :226,241
        case HJ_STAGE_RK2_FULL: A.ca = T(0.5); A.cb = T(0.5); break;             // ode_cfl_2.py:184,201
        default: A.ca = T(0); A.cb = T(1); break;
    }
    A.dt = (T)s.dt;
    A.dt_dev = s.dt_dev;
    A.post_op = s.post_op;
    A.do_clamp = s.restrict_sign != 0;
    A.clamp_lo = s.restrict_sign > 0 ? T(0) : -std::numeric_limits<T>::infinity();
    A.clamp_hi = s.restrict_sign < 0 ? T(0) : std::numeric_limits<T>::infinity();
    fill_ham<T>(c, s.par, A.ham, s.ham);
    if (s.term) {            // a TermOp launch: coefficient array 0 rides in the y0 stream whatever the stage says
        A.term = *static_cast<const hj::TermPar<T>*>(s.term);
        A.use_y0 = A.term.arr[0] != nullptr;
        for (int d = 0; d < ND; ++d) A.sc[d] = T(1);
    }
    return HJ_OK;
}

}  // namespace hjh
