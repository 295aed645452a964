// The fused / direct RK-substep kernels of ONE (dtype, Hamiltonian) pair and their launch code.
// Compiled once per pair with -DHJ_INST_T=<double|float> -DHJ_INST_HAM=<HamDubinsRel|...> (Makefile), so
// the kernel instantiations build in parallel.  gfx950 only.
#include <mutex>
#include "hj_host.h"
#include "hj_fused.h"
#include "hj_fused12.h"
#include "hj_fusedv.h"
#include "hj_fused12v.h"
#include "hj_fused4v.h"
#include "hj_flat4v.h"
#include "hj_launch.h"

namespace hjh {

// does this launch reduce max(D1^2) of its own output for the next stage's epsilon (intended WENO5; FusedArgs::eps_part)?
// Whole-grid launches of a single domain only: the seam kernel enumerates tile and chunk seams from plane 0.
inline bool eps_producer(const hj_ctx* c, const SubstepCall& s) {
    return s.scheme == HJ_WENO5 && s.want_eps && c->eps_fuse && s.stage != HJ_STAGE_YDOT && s.p0 == 0 && s.p1 == c->N[0] &&
           s.q1 <= s.q0 && !c->halo_lo && !c->halo_hi && !s.on_aux && !c->weno_src && c->total >= c->eps_fuse_min_cells &&
           c->total < (1ll << 31);
}

// PAIR: the two-cells-per-lane kernel (hj_fusedv.h; R = pairs per thread) instead of fused_substep_kernel
template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, int PD, int MODE, bool PAIR>
auto tiled_kernel() {
    if constexpr (PAIR) return fused_pair_kernel<T, HAM, SCHEME, NT, R, KH, OCC, MODE>;
    else return fused_substep_kernel<T, HAM, SCHEME, NT, R, KH, OCC, PD, MODE>;
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, int PD, int MODE, bool PAIR = false>
int launch_tiled_mode(hj_ctx* c, const SubstepCall& s, Tiling t) {
    constexpr int ND = HAM::ND;
    EdgePlan ep;
    {
        auto kern0 = tiled_kernel<T, HAM, SCHEME, NT, R, KH, OCC, PD, MODE, PAIR>();
        const auto key = std::make_pair(reinterpret_cast<const void*>(kern0), t.lds_bytes);
        auto it = c->occ_cache.find(key);
        int occ_blocks = it != c->occ_cache.end() ? it->second : 0;
        if (it == c->occ_cache.end()) {
            // (planning without a device -- or a planning look from a live context, c->dry == 2, which must not cache its estimate: the launch
            //  bound's waves per SIMD, and the CU's 160 KB of LDS)
            if (c->dry) occ_blocks = std::max(1, std::min(OCC * 256 / NT, (int)((size_t)(160 * 1024) / std::max<size_t>(1, t.lds_bytes))));
            else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_blocks, key.first, NT, t.lds_bytes) != hipSuccess || occ_blocks < 1) occ_blocks = 1;
            if (c->dry != 2) c->occ_cache.emplace(key, occ_blocks);
        }
        {
            const int rc_plan = plan_chunks(c, s, t, occ_blocks, ep);
            if (rc_plan) return rc_plan;
        }
        c->last_plan.ntiles = t.ntiles; c->last_plan.nchunks = t.nchunks; c->last_plan.nblocks = t.nblocks; c->last_plan.threads = NT;
        c->last_plan.wg_per_cu = occ_blocks; c->last_plan.lds_bytes = t.lds_bytes;
        if (c->dry) {
            c->last_kernel = PAIR ? "fused_pair_kernel" : "fused_substep_kernel";
            c->last_E[0] = t.chunk;
            for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = d < ND ? t.E[d] : 0;
            return HJ_OK;
        }
        if (c->debug) {
            fprintf(stderr, "[hj] %stiling NT=%d R=%d KH=%d PD=%d OCC=%d E=(%d,%d,%d) pitch=%d ntiles=%d chunk=%d nchunks=%d blocks=%d wg/CU=%d lds=%zu score=%.3f\n",
                    PAIR ? "pair " : "", NT, R, KH, PD, OCC, t.E[1], c->ndim > 2 ? t.E[2] : 0, c->ndim > 3 ? t.E[3] : 0, t.lpitch, t.ntiles, t.chunk,
                    t.nchunks, t.nblocks, occ_blocks, t.lds_bytes, t.score);
            c->debug = 0;
        }
    }
    if (c->debug > 1 && s.gated) fprintf(stderr, "[hj] gated launch: %d edge workgroups + %d, chunk %d, %d tiles\n", ep.edge_count, t.nblocks - ep.edge_count, t.chunk, t.ntiles);
    FusedArgs<T, ND> A;
    memset(&A, 0, sizeof(A));
    A.max_d1sq = (const T*)(c->weno_src ? c->weno_src : c->weno_vals);
    A.bound = s.bound;
    const bool produce = SCHEME == HJ_WENO5 && eps_producer(c, s);
    if (produce) {
        if ((size_t)t.nblocks > c->eps_prod_cap) {
            if (c->eps_prod) { HIP_TRY(hipFree(c->eps_prod)); c->eps_prod = nullptr; c->eps_prod_cap = 0; }
            const size_t cap = std::max<size_t>(4096, 2 * (size_t)t.nblocks);
            HIP_TRY(hipMalloc((void**)&c->eps_prod, cap * HJ_MAX_DIM * sizeof(double)));
            c->eps_prod_cap = cap;
        }
        A.eps_part = c->eps_prod;
    }
    if (SCHEME == HJ_WENO5 && s.eps_nrows > 0) { A.eps_rows = s.eps_rows; A.eps_nrows = s.eps_nrows; }
    unsigned grid_blocks = 0;
    {
        const int rc_fill = fill_fused_args<T, ND>(c, s, t, ep, SCHEME, PAIR, A, grid_blocks);
        if (rc_fill) return rc_fill;
    }
    if (produce) A.npairs = 0;        // the output reduction of the intended WENO5 pairs planes in ascending order
    auto kern = tiled_kernel<T, HAM, SCHEME, NT, R, KH, OCC, PD, MODE, PAIR>();
    c->last_kernel = PAIR ? "fused_pair_kernel" : "fused_substep_kernel";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = d < ND ? t.E[d] : 0;
    if (t.lds_bytes > 64 * 1024) {
        // once per (device, kernel), raised but never lowered: the attribute belongs to the function, not to a context
        static std::mutex mu;
        static std::map<std::pair<int, const void*>, size_t> granted_by_kernel;
        std::lock_guard<std::mutex> lock(mu);
        size_t& granted = granted_by_kernel[std::make_pair(c->device, reinterpret_cast<const void*>(kern))];
        if (granted < t.lds_bytes) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
            granted = t.lds_bytes;
        }
    }
    const char* dump = c->timing_dump;              // HJ_TIMING_DUMP (read at ctx creation): per-workgroup start/end clocks of every launch
    unsigned long long* tbuf = nullptr;
    if (dump && *dump) {
        HIP_TRY(hipMalloc(&tbuf, (size_t)t.nblocks * 12 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(tbuf, 0, (size_t)t.nblocks * 12 * sizeof(unsigned long long), call_stream(c, s)));
        A.timing = tbuf;
    }
    if (c->launch_stop) {
        // completion signal attached to the dispatch packet itself: a separate hipEventRecord costs a
        // marker packet and ~6 us of bubble before the next kernel of the stream (slab timeline)
        hipExtLaunchKernelGGL(kern, dim3(grid_blocks), dim3(NT), (unsigned)t.lds_bytes, call_stream(c, s), nullptr, c->launch_stop, 0,
                              (const T*)s.y, (const T*)s.y0, (T*)s.out, A);
        c->launch_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(NT), t.lds_bytes, call_stream(c, s), (const T*)s.y, (const T*)s.y0,
                           (T*)s.out, A);
    }
    HIP_TRY(hipGetLastError());
    if (produce) {
        // tile / chunk seams and wrap pairs of the output + the fold of the launch's rows: HJ_EPS_ROWS rows for the next launch
        SeamArgs<T, ND> S;
        memset(&S, 0, sizeof(S));
        S.y = (const T*)s.out;
        fill_grid<T, ND>(c, S.G);
        for (int d = 0; d < ND; ++d) { S.E[d] = t.E[d]; S.ntile[d] = t.ntile[d]; }
        S.chunk = t.chunk;
        S.nchunks = t.nchunks;
        S.prod = c->eps_prod;
        S.nprod = t.nblocks;
        S.rows = c->eps_rows;
        hipLaunchKernelGGL((eps_seam_kernel<T, ND>), dim3(HJ_EPS_ROWS), dim3(1024), 0, call_stream(c, s), S);
        HIP_TRY(hipGetLastError());
        c->eps_ready = true;
    }
    if (tbuf) {
        std::vector<unsigned long long> h((size_t)t.nblocks * 12);
        HIP_TRY(hipStreamSynchronize(call_stream(c, s)));
        HIP_TRY(hipMemcpy(h.data(), tbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(tbuf));
        if (FILE* f = fopen(dump, "a")) {
            fprintf(f, "# launch nblocks=%d ntiles=%d chunk=%d stage=%d\n", t.nblocks, t.ntiles, t.chunk, s.stage);
            for (int i = 0; i < t.nblocks; ++i) {
                fprintf(f, "%d %llu %llu %llu %llu", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
                // HJ_STAMP builds: shader-clock sums of the four phases of wave 0 and of the last wave (else zeros)
                const unsigned long long* ph = h.data() + 4 * (size_t)t.nblocks + 8 * (size_t)i;
                fprintf(f, " %llu %llu %llu %llu %llu %llu %llu %llu\n", ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7]);
            }
            fclose(f);
        }
    }
    return HJ_OK;
}

// plain RK stages (no clamp, no post-step operator, not ydot-only) run the flag-free instantiations
template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, int PD, bool PAIR = false>
int launch_tiled(hj_ctx* c, const SubstepCall& s, const Tiling& t) {
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0 && !c->no_plain;
    if (plain && s.stage == HJ_STAGE_EULER) return launch_tiled_mode<T, HAM, SCHEME, NT, R, KH, OCC, PD, 1, PAIR>(c, s, t);
    if (plain) return launch_tiled_mode<T, HAM, SCHEME, NT, R, KH, OCC, PD, 2, PAIR>(c, s, t);
    return launch_tiled_mode<T, HAM, SCHEME, NT, R, KH, OCC, PD, 0, PAIR>(c, s, t);
}

// ---- the 4-D pair kernel with a compile-time tile (hj_fused4v.h; round 5): (threads, pairs per thread, E1, E2, E3, waves/SIMD hint), in
// order of preference; the first whose tile fits the grid is taken.  C5 (129^4 fp32; profiles/r05_c5_tiles.txt): rows of 66 cells (264 B)
// instead of 34 cost the CU's texture-address path a third less per byte (profiles/r05_l1_rate.txt) and halve the partial cache lines:
// 5x6x66 in 512 threads (one workgroup per CU) 0.335-0.35 of 8 TB/s with fabric traffic 1.42x algorithmic, 3x5x66 in 256 threads (two
// per CU) 0.34, 5x6x34 in 256 threads 0.30 (= the generic pair kernel on the same tile: the round-4 default, traffic 1.86x).
// Shapes with 3 and 4 waves per SIMD (768 x 2 pairs, 1024 x 1 pair at 122 VGPRs) run at the same 0.335-0.35.
#ifndef HJ_TILE4
#define HJ_TILE4(X) X(512, 2, 5, 6, 66, 2) X(256, 2, 3, 5, 66, 2) X(256, 2, 5, 6, 34, 2)
#endif
// does the fixed tile fit this grid?  A tile never exceeds an axis (the last one is shifted back inside), and no tile may begin
// or end 1 or 3 cells from an end of the contiguous axis: its halo columns are pairs (hj_fused4v.h), which must lie wholly inside
// or wholly outside the grid
inline bool tile4_fits(const hj_ctx* c, int e1, int e2, int e3) {
    if (!(c->ndim == 4 && c->N[1] >= e1 && c->N[2] >= e2 && c->N[3] >= e3 && c->total < (1ll << 31))) return false;
    const int n3 = (int)c->N[3], nt = (n3 + e3 - 1) / e3;
    for (int t = 0; t < nt; ++t) {
        const int org = std::min(t * e3, n3 - e3), rest = n3 - org - e3;
        if (org == 1 || org == 3 || rest == 1 || rest == 3) return false;
    }
    return true;
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int E3, int OCC, bool PG, int MODE>
int launch_pair4_mode(hj_ctx* c, const SubstepCall& s) {
    using G = hj::Tile4<E1, E2, E3>;
    constexpr bool ROWS = hj::ham_has_rows<HAM>::value;
    constexpr int ER = hj::RowAxis<HAM, ROWS>::value == 1 ? E1 : E2;
    auto kern = fused_pair4_kernel<T, HAM, SCHEME, NT, R, E1, E2, E3, OCC, PG, MODE>;
    Tiling t;
    memset(&t, 0, sizeof(t));
    t.ok = true;
    const int Ed[4] = {1, E1, E2, E3};
    t.ntiles = 1;
    for (int d = 0; d < HJ_MAX_DIM; ++d) { t.E[d] = 1; t.ntile[d] = 1; }
    for (int d = 1; d < 4; ++d) {
        t.E[d] = Ed[d];
        t.ntile[d] = (int)((c->N[d] + Ed[d] - 1) / Ed[d]);
        t.ntiles *= t.ntile[d];
    }
    t.lpitch = G::PITCH;
    const size_t base_lds = 512 + 2 * (size_t)G::PLANE * sizeof(T);
    // two workgroups per CU (OCC waves per SIMD of NT threads): each may take half the CU's LDS; the row table of a chunk
    // (ER rows x ROWF values per plane) has to fit in what the plane buffers leave
    const int wg_per_cu = std::max(1, OCC * 256 / NT);
    const size_t lds_cap = (size_t)(160 * 1024) / wg_per_cu - 256;
    int64_t chunk_max = 0;
    if (ROWS) {
        if (base_lds + (size_t)ER * G::ROWF * sizeof(T) * 8 > lds_cap) return hjh::fail(HJ_EUNSUPPORTED, "4-D tile leaves no LDS for the row table");
        chunk_max = (int64_t)((lds_cap - base_lds) / ((size_t)ER * G::ROWF * sizeof(T)));
    }
    EdgePlan ep;
    {
        const int rc_plan = plan_chunks(c, s, t, wg_per_cu, ep, chunk_max);
        if (rc_plan) return rc_plan;
    }
    t.lds_bytes = base_lds + (ROWS ? (size_t)t.chunk * ER * G::ROWF * sizeof(T) : 0);
    c->last_plan.ntiles = t.ntiles; c->last_plan.nchunks = t.nchunks; c->last_plan.nblocks = t.nblocks; c->last_plan.threads = NT;
    c->last_plan.wg_per_cu = wg_per_cu; c->last_plan.lds_bytes = t.lds_bytes;
    if (c->dry) {
        c->last_kernel = "fused_pair4_kernel";
        c->last_E[0] = t.chunk;
        for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = t.E[d];
        return HJ_OK;
    }
    if (c->debug) {
        fprintf(stderr, "[hj] pair4 tiling NT=%d R=%d OCC=%d E=(%d,%d,%d) pitch=%d ntiles=%d chunk=%d nchunks=%d blocks=%d lds=%zu PG=%d MODE=%d\n",
                NT, R, OCC, E1, E2, E3, G::PITCH, t.ntiles, t.chunk, t.nchunks, t.nblocks, t.lds_bytes, (int)PG, MODE);
        c->debug = 0;
    }
    FusedArgs<T, 4> A;
    memset(&A, 0, sizeof(A));
    A.bound = s.bound;
    unsigned grid_blocks = 0;
    {
        const int rc_fill = fill_fused_args<T, 4>(c, s, t, ep, SCHEME, true, A, grid_blocks);
        if (rc_fill) return rc_fill;
    }
    A.lds_nbuf = 2;
    A.halo_ahead = 0;
    A.npairs = 0;
    c->last_nbuf = 2;
    c->last_nbase = 2;
    c->last_kernel = "fused_pair4_kernel";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = t.E[d];
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

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int E3, int OCC>
int launch_pair4(hj_ctx* c, const SubstepCall& s) {
    // PG: can a halo cell of a plane axis be a ghost (an extrapolated axis among 1..3)?  All-periodic grids take the lean instantiation
    const bool pg = c->bc[1] != HJ_BC_PERIODIC || c->bc[2] != HJ_BC_PERIODIC || c->bc[3] != HJ_BC_PERIODIC;
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0 && !c->no_plain;
    const int mode = plain ? (s.stage == HJ_STAGE_EULER ? 1 : 2) : 0;
#define HJ_P4(PG_, MODE_) return launch_pair4_mode<T, HAM, SCHEME, NT, R, E1, E2, E3, OCC, PG_, MODE_>(c, s)
    if (pg) { if (mode == 1) HJ_P4(true, 1); if (mode == 2) HJ_P4(true, 2); HJ_P4(true, 0); }
    if (mode == 1) HJ_P4(false, 1);
    if (mode == 2) HJ_P4(false, 2);
    HJ_P4(false, 0);
#undef HJ_P4
}

// ---- the 4-D fp32 kernel with full-row tiles and 16-byte row loads (hj_flat4v.h; round 6): (threads, pairs per thread, E1, E2, LDS row
// pitch P3, waves/SIMD hint).  Taken ahead of the compile-time tiles above when the grid's contiguous axis fits a row of the box.
#ifndef HJ_FLAT4
#define HJ_FLAT4(X) X(512, 2, 3, 5, 140, 2)
#endif
inline bool flat4_fits(const hj_ctx* c, int nt, int r, int e1, int e2, int p3) {
    if (!(c->ndim == 4 && c->dtype == HJ_F32 && c->N[1] >= e1 && c->N[2] >= e2 && c->total < (1ll << 31))) return false;
    const long long n3 = c->N[3], halfp = (n3 + 1) / 2, ch = (n3 + 3) / 4, nr = (long long)e1 * e2, nh = 6ll * (e1 + e2);
    return n3 >= 8 && n3 <= p3 - HJ_VPAD - 4 && nr * halfp <= (long long)nt * r && nr * ch <= nt && nh * ch <= 4ll * nt;
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int P3, int OCC, bool PG, int MODE>
int launch_flat4_mode(hj_ctx* c, const SubstepCall& s) {
    using G = hj::Flat4<E1, E2, P3>;
    constexpr bool ROWS = hj::ham_has_rows<HAM>::value;
    constexpr int ER = hj::RowAxis<HAM, ROWS>::value == 1 ? E1 : E2;
    auto kern = fused_flat4_kernel<T, HAM, SCHEME, NT, R, E1, E2, P3, OCC, PG, MODE>;
    Tiling t;
    memset(&t, 0, sizeof(t));
    t.ok = true;
    const int Ed[4] = {1, E1, E2, (int)c->N[3]};
    t.ntiles = 1;
    for (int d = 0; d < HJ_MAX_DIM; ++d) { t.E[d] = 1; t.ntile[d] = 1; }
    for (int d = 1; d < 4; ++d) {
        t.E[d] = Ed[d];
        t.ntile[d] = (int)((c->N[d] + Ed[d] - 1) / Ed[d]);
        t.ntiles *= t.ntile[d];
    }
    t.lpitch = P3;
    const size_t base_lds = 512 + (2 * (size_t)G::PLANE + 2 * (size_t)G::STAGE) * sizeof(T);
    const int wg_per_cu = std::max(1, std::min(OCC * 256 / NT, (int)((size_t)(160 * 1024) / (base_lds + 4096))));
    const size_t lds_cap = (size_t)(160 * 1024) / wg_per_cu - 256;
    int64_t chunk_max = 0;
    if (ROWS) {
        if (base_lds + (size_t)ER * G::ROWF * sizeof(T) * 8 > lds_cap) return hjh::fail(HJ_EUNSUPPORTED, "4-D full-row tile leaves no LDS for the row table");
        chunk_max = (int64_t)((lds_cap - base_lds) / ((size_t)ER * G::ROWF * sizeof(T)));
    }
    EdgePlan ep;
    {
        const int rc_plan = plan_chunks(c, s, t, wg_per_cu, ep, chunk_max);
        if (rc_plan) return rc_plan;
    }
    t.lds_bytes = base_lds + (ROWS ? (size_t)t.chunk * ER * G::ROWF * sizeof(T) : 0);
    c->last_plan.ntiles = t.ntiles; c->last_plan.nchunks = t.nchunks; c->last_plan.nblocks = t.nblocks; c->last_plan.threads = NT;
    c->last_plan.wg_per_cu = wg_per_cu; c->last_plan.lds_bytes = t.lds_bytes;
    c->last_kernel = "fused_flat4_kernel";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = t.E[d];
    if (c->dry) return HJ_OK;
    if (c->debug) {
        fprintf(stderr, "[hj] flat4 tiling NT=%d R=%d OCC=%d E=(%d,%d,whole rows of %d) pitch=%d ntiles=%d chunk=%d nchunks=%d blocks=%d lds=%zu PG=%d MODE=%d\n",
                NT, R, OCC, E1, E2, (int)c->N[3], P3, t.ntiles, t.chunk, t.nchunks, t.nblocks, t.lds_bytes, (int)PG, MODE);
        c->debug = 0;
    }
    FusedArgs<T, 4> A;
    memset(&A, 0, sizeof(A));
    A.bound = s.bound;
    unsigned grid_blocks = 0;
    {
        const int rc_fill = fill_fused_args<T, 4>(c, s, t, ep, SCHEME, true, A, grid_blocks);
        if (rc_fill) return rc_fill;
    }
    A.lds_nbuf = 2;
    A.halo_ahead = 0;
    A.npairs = 0;
    c->last_nbuf = 2;
    c->last_nbase = 2;
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

template <typename T, typename HAM, int SCHEME, int NT, int R, int E1, int E2, int P3, int OCC>
int launch_flat4(hj_ctx* c, const SubstepCall& s) {
    const bool pg = c->bc[1] != HJ_BC_PERIODIC || c->bc[2] != HJ_BC_PERIODIC || c->bc[3] != HJ_BC_PERIODIC;
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0 && !c->no_plain;
    const int mode = plain ? (s.stage == HJ_STAGE_EULER ? 1 : 2) : 0;
#define HJ_F4(PG_, MODE_) return launch_flat4_mode<T, HAM, SCHEME, NT, R, E1, E2, P3, OCC, PG_, MODE_>(c, s)
    if (pg) { if (mode == 1) HJ_F4(true, 1); if (mode == 2) HJ_F4(true, 2); HJ_F4(true, 0); }
    if (mode == 1) HJ_F4(false, 1);
    if (mode == 2) HJ_F4(false, 2);
    HJ_F4(false, 0);
#undef HJ_F4
}

// (threads, PAIRS per thread, halo slots per thread, waves/SIMD hint) of the pair kernel
#ifndef HJ_CONFIGS_PAIR
#define HJ_CONFIGS_PAIR(X) X(256, 1, 2, 2) X(512, 2, 2, 2)
#endif
// 4-D grids (three plane axes: the halo cross is 1.7-2.4x the tile; 5 pair + 1 single halo slots per thread): fp32 with a light stencil only
// (cfg_built): 256 threads x 2 pairs = the 1024-cell tile of the one-cell-per-lane kernel in TWO independent workgroups per CU,
// +6 % on C5 (profiles/r03_c5_config_sweep.txt); every fp64 shape spills
#ifndef HJ_CONFIGS_PAIR_4D
#define HJ_CONFIGS_PAIR_4D(X) X(256, 2, 6, 2)
#endif

// ---- launch-time choice of the tile shape (TuneState, hj_host.h)
inline int stage_class(int stage) { return stage >= HJ_STAGE_RK3_HALF ? 2 : (stage == HJ_STAGE_EULER ? 1 : 0); }
struct TuneTrial { TuneState* ts = nullptr; int cand = -1; };

// The tiling of this launch: the static choice, or -- whole-grid launches of a single domain on grids of at least
// HJ_AUTOTUNE_MIN_MCELLS cells -- the next candidate of the tuning rotation / the shape the rotation settled on.
template <int ND>
Tiling tune_begin(hj_ctx* c, const SubstepCall& s, const KernelCfg& k, int vec, int nbuf, long long key, TuneTrial& tr) {
    bool tunable = !c->dry && c->autotune && c->total >= c->autotune_min_cells && !c->full_rows && !c->tile_cells && !s.on_aux &&
                   !c->launch_stop && !c->halo_lo && !c->halo_hi && s.p0 == 0 && s.p1 == c->N[0] && s.q1 <= s.q0 &&
                   !c->timing_dump;
    if (!tunable) return make_tiling(c, k, s.p0, s.p1, vec, nbuf);
    TuneState* ts = &c->tune[key];
    if (ts->chosen < 0) {
        // a trial ends in hipEventSynchronize: never while the stream is being captured into a graph -- such a
        // launch takes the shape chosen so far (or the best-scored one) and leaves the rotation where it is
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(call_stream(c, s), &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return ts->cand.empty() ? make_tiling(c, k, s.p0, s.p1, vec, nbuf) : ts->cand[0];
        }
    }
    if (ts->cand.empty() && ts->chosen < 0) {
        std::vector<Tiling> all;
        const Tiling b0 = make_tiling(c, k, s.p0, s.p1, vec, nbuf, &all);
        if (!b0.ok) { ts->chosen = 0; return b0; }
        // candidate 0: the best-scored shape; then the shapes with LONGER rows within 35 % of its score (every sweep of round 3
        // found the winner among those: 130-cell rows at 513^3 and 451^3, 134 / 102 at 401^3, 118 at 351^3), then two with
        // shorter rows
        constexpr int LAST = ND - 1;
        ts->cand.push_back(b0);
        for (const Tiling& t : all)
            if (t.E[LAST] > b0.E[LAST] && t.score <= 1.35 * b0.score && ts->cand.size() < 7) ts->cand.push_back(t);
        int shorter = 0;
        for (const Tiling& t : all)
            if (t.E[LAST] < b0.E[LAST] && t.score <= 1.35 * b0.score && shorter < 2) { ts->cand.push_back(t); ++shorter; }
        ts->best_ms.assign(ts->cand.size(), 1e30f);
        if (ts->cand.size() <= 1) ts->chosen = 0;
    }
    if (ts->cand.empty()) return make_tiling(c, k, s.p0, s.p1, vec, nbuf);
    if (ts->chosen >= 0) return ts->cand[ts->chosen];
    tr.ts = ts;
    c->tune_seq += 1;
    tr.cand = ts->trial % (int)ts->cand.size();
    if (!c->tune_ev[0]) {
        if (hipEventCreate(&c->tune_ev[0]) != hipSuccess || hipEventCreate(&c->tune_ev[1]) != hipSuccess) { tr.ts = nullptr; return ts->cand[0]; }
    }
    (void)hipEventRecord(c->tune_ev[0], call_stream(c, s));
    return ts->cand[tr.cand];
}

// One timed trial (the first pass only warms up); the launch itself was a normal one: the results do not depend on the tiling.
template <int ND>
int tune_end(hj_ctx* c, const SubstepCall& s, TuneTrial& tr, int rc, int scheme) {
    TuneState* ts = tr.ts;
    if (!ts || rc != HJ_OK) return rc;
    HIP_TRY(hipEventRecord(c->tune_ev[1], call_stream(c, s)));
    HIP_TRY(hipEventSynchronize(c->tune_ev[1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->tune_ev[0], c->tune_ev[1]));
    const int ncand = (int)ts->cand.size();
    if (ts->trial >= ncand) ts->best_ms[tr.cand] = std::min(ts->best_ms[tr.cand], ms);
    if (++ts->trial >= ncand * c->autotune_passes) {
        ts->chosen = (int)(std::min_element(ts->best_ms.begin(), ts->best_ms.end()) - ts->best_ms.begin());
        // the best-scored shape (candidate 0) stays unless another one beats it by more than the timing noise
        if (ts->best_ms[ts->chosen] > 0.985f * ts->best_ms[0]) ts->chosen = 0;
        if (getenv("HJ_DEBUG") || getenv("HJ_AUTOTUNE_LOG")) {
            for (int i = 0; i < ncand; ++i)
                fprintf(stderr, "[hj] autotune scheme %d stage %d: E=(%d,%d,%d) ntiles=%d score=%.3f  %.4f ms%s\n", scheme, s.stage,
                        ts->cand[i].E[1], ND > 2 ? ts->cand[i].E[2] : 0, ND > 3 ? ts->cand[i].E[3] : 0, ts->cand[i].ntiles, ts->cand[i].score,
                        ts->best_ms[i], i == ts->chosen ? "  <- chosen" : "");
        }
    }
    return rc;
}

template <typename T, typename HAM, int SCHEME>
int launch_direct(hj_ctx* c, const SubstepCall& s) {
    constexpr int ND = HAM::ND;
    if (c->dry) {
        const long long cells = (s.p1 - s.p0) * (c->total / c->N[0]);
        c->last_plan.ntiles = 0; c->last_plan.nchunks = 1; c->last_plan.threads = 256; c->last_plan.wg_per_cu = 8; c->last_plan.lds_bytes = 0;
        c->last_plan.nblocks = (int)std::min<long long>((cells + 255) / 256, 256 * 16);
        c->last_kernel = "direct_substep_kernel";
        for (int d = 0; d < HJ_MAX_DIM; ++d) c->last_E[d] = 0;
        return HJ_OK;
    }
    DirectArgs<T, ND> A;
    memset(&A, 0, sizeof(A));
    A.y = (const T*)s.y;
    A.y0 = (const T*)s.y0;
    A.out = (T*)s.out;
    A.max_d1sq = (const T*)(c->weno_src ? c->weno_src : c->weno_vals);
    if (SCHEME == HJ_WENO5 && s.eps_nrows > 0) {
        // this kernel does not fold rows: one more (small) launch turns them into the ND values
        int rc = eps_rows_to_vals(c, s.eps_rows, s.eps_nrows, call_stream(c, s));
        if (rc) return rc;
        A.max_d1sq = (const T*)c->weno_vals;
    }
    A.bound = s.bound;
    fill_grid<T, ND>(c, A.G);
    for (int d = 0; d < ND; ++d) A.sc[d] = scheme_scale<T>(SCHEME, c->dx[d]);
    if (s.p0 < 0 || s.p1 > c->N[0] || (s.q1 > s.q0 && (s.q0 < 0 || s.q1 > c->N[0])))
        return hjh::fail(HJ_EUNSUPPORTED, "pad planes need the tiled kernel");
    for (int w = 0; w < 2; ++w)
        if (s.gated && s.e1[w] > s.e0[w] && (s.e0[w] < 0 || s.e1[w] > c->N[0])) return hjh::fail(HJ_EUNSUPPORTED, "pad planes need the tiled kernel");
    const long long plane = c->total / c->N[0];
    A.cell_begin = s.p0 * plane;
    A.cell_end = s.p1 * plane;
    A.stage = s.stage;
    A.post_op = s.post_op;
    A.restrict_sign = s.restrict_sign;
    A.dt = (T)s.dt;
    fill_ham<T>(c, s.par, A.ham, s.ham);
    c->gate_posted = 0;            // the direct kernel never publishes: the caller orders the exchange with an event
    for (int pass = 0; pass < 4; ++pass) {
        if (pass == 1) {
            if (s.q1 <= s.q0) continue;
            A.cell_begin = s.q0 * plane;
            A.cell_end = s.q1 * plane;
        }
        if (pass >= 2) {               // edge ranges of a gated call
            if (!s.gated || s.e1[pass - 2] <= s.e0[pass - 2]) continue;
            A.cell_begin = s.e0[pass - 2] * plane;
            A.cell_end = s.e1[pass - 2] * plane;
        }
        const long long cells = A.cell_end - A.cell_begin;
        if (cells <= 0) continue;
        int blocks = (int)std::min<long long>((cells + 255) / 256, 256 * 16);
        c->last_kernel = "direct_substep_kernel";
        for (int d = 0; d < HJ_MAX_DIM; ++d) c->last_E[d] = 0;
        hipLaunchKernelGGL((direct_substep_kernel<T, HAM, SCHEME>), dim3(blocks), dim3(256), 0, call_stream(c, s), A);
        HIP_TRY(hipGetLastError());
    }
    return HJ_OK;
}

// ---- one cooperative launch for a whole odeCFL2 / odeCFL3 step of a SMALL grid (hj_split.h, coop_rk_kernel; round 6)
template <typename T, typename HAM, int SCHEME, int CPT>
int launch_coop_cpt(hj_ctx* c, const CoopCall& s, int nblocks) {
    constexpr int ND = HAM::ND;
    auto kern = coop_rk_kernel<T, HAM, SCHEME, CPT>;
    CoopArgs<T, ND> A;
    memset(&A, 0, sizeof(A));
    A.y = (const T*)s.y; A.s1 = (T*)s.s1; A.s2 = (T*)s.s2; A.out = (T*)s.out;
    fill_grid<T, ND>(c, A.G);
    for (int d = 0; d < ND; ++d) A.sc[d] = scheme_scale<T>(SCHEME, c->dx[d]);
    A.order = s.order; A.restrict_sign = s.restrict_sign; A.post_op = s.post_op;
    A.dt = (T)s.dt;
    fill_ham<T>(c, s.par, A.ham, s.ham);
    A.sync = (CoopSync*)c->coop_sync;
    unsigned nper[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nx = 0;
    for (int b = 0; b < nblocks; ++b) nper[b & 7]++;
    for (int x = 0; x < 8; ++x) { A.nper_xcd[x] = nper[x]; A.base_xcd[x] = c->coop_xcd[x]; if (nper[x]) ++nx; }
    A.nxcd = nx;
    A.base_all = c->coop_all;
    void* args[1] = {&A};
    // (a PLAIN launch: the grid was sized against the occupancy query above, so every workgroup is resident as under hipLaunchCooperativeKernel --
    //  which costs the host 15-19 us more per launch, MI355X_MICROARCH.md "coop-launch" -- PROVIDED nothing else holds CUs of this device:
    //  HJ_COOP=2 asks for the checked cooperative launch instead)
    if (c->coop == 2) {
        if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(nblocks), dim3(256), args, 0, c->stream) != hipSuccess) {
            (void)hipGetLastError();
            c->coop = 0;
            return HJ_XP_FALLBACK;
        }
    } else {
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), 0, c->stream, A);
        HIP_TRY(hipGetLastError());
    }
    // what this launch adds to the counters (order - 1 barriers)
    for (int x = 0; x < 8; ++x) c->coop_xcd[x] += (unsigned long long)(s.order - 1) * nper[x];
    c->coop_all += (unsigned long long)(s.order - 1) * nx;
    c->last_kernel = "coop_rk_kernel";
    for (int d = 0; d < HJ_MAX_DIM; ++d) c->last_E[d] = 0;
    return HJ_OK;
}

template <typename T, typename HAM, int SCHEME>
int launch_coop_scheme(hj_ctx* c, const CoopCall& s) {
    // resident workgroups of the 1-cell instantiation decide the shape: as many 256-thread workgroups as fit at once, then 1, 2 or 4
    // cells per thread (51^3 = 132 651 cells: 519 workgroups x 1 cell when three fit per CU)
    const long long total = c->total;
    auto cap_of = [&](const void* k) {
        auto it = c->occ_cache.find(std::make_pair(k, (size_t)0));
        if (it == c->occ_cache.end()) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, 0) != hipSuccess || nb < 1) nb = 1;
            it = c->occ_cache.emplace(std::make_pair(k, (size_t)0), nb).first;
        }
        return (long long)it->second * c->num_cus;
    };
    const long long need1 = (total + 255) / 256;
    const void* k1 = reinterpret_cast<const void*>(coop_rk_kernel<T, HAM, SCHEME, 1>);
    const void* k2 = reinterpret_cast<const void*>(coop_rk_kernel<T, HAM, SCHEME, 2>);
    const void* k4 = reinterpret_cast<const void*>(coop_rk_kernel<T, HAM, SCHEME, 4>);
    if (need1 <= cap_of(k1)) return launch_coop_cpt<T, HAM, SCHEME, 1>(c, s, (int)need1);
    const long long need2 = (total + 511) / 512;
    if (need2 <= cap_of(k2)) return launch_coop_cpt<T, HAM, SCHEME, 2>(c, s, (int)need2);
    const long long need4 = (total + 1023) / 1024;
    if (need4 <= cap_of(k4)) return launch_coop_cpt<T, HAM, SCHEME, 4>(c, s, (int)need4);
    return HJ_XP_FALLBACK;
}

template <typename T, typename HAM>
int launch_coop(hj_ctx* c, const CoopCall& s) {
    if constexpr (HAM::ND > 3) return HJ_XP_FALLBACK;
    else {
        switch (s.scheme) {
            case HJ_ENO2: return launch_coop_scheme<T, HAM, HJ_ENO2>(c, s);
            case HJ_ENO3: return launch_coop_scheme<T, HAM, HJ_ENO3>(c, s);
            case HJ_WENO5_ASSHIPPED: return launch_coop_scheme<T, HAM, HJ_WENO5_ASSHIPPED>(c, s);
            case HJ_ENO2_FAST: return launch_coop_scheme<T, HAM, HJ_ENO2_FAST>(c, s);
            case HJ_ENO3_FAST: return launch_coop_scheme<T, HAM, HJ_ENO3_FAST>(c, s);
        }
        return HJ_XP_FALLBACK;
    }
}

template <typename T, typename HAM, int SCHEME>
int launch_cfg(hj_ctx* c, const SubstepCall& s) {
#if defined(HJ_TUNE_BUILD) && HJ_TUNE_BUILD == 2
    // quick-iteration build for C5: only the fp32 pendulum with the as-shipped WENO5 is compiled tiled
    constexpr bool tiled_ok = std::is_same<T, float>::value && HAM::ID == HJ_HAM_DOUBLE_PENDULUM && SCHEME == HJ_WENO5_ASSHIPPED;
#elif defined(HJ_TUNE_BUILD)
    // quick-iteration build: only fp64 Dubins with the two WENO5 arithmetics is compiled tiled
    constexpr bool tiled_ok = std::is_same<T, double>::value && HAM::ID == HJ_HAM_DUBINS_REL &&
                              (SCHEME == HJ_WENO5 || SCHEME == HJ_WENO5_ASSHIPPED);
#else
    constexpr bool tiled_ok = true;
#endif
    if constexpr (tiled_ok) {
        // small grids (round 3): the direct kernel -- one cell per thread, every stencil load issued at once, no LDS, no
        // barrier chain -- beats a tiled launch that cannot go below ~8.5 us (a 4.5 us setup in front of a dozen plane
        // iterations on a few dozen workgroups).  HJ_DIRECT_BELOW cells (default: see hj_ctx_create).
        // (not on slabs; not when a tiled configuration was asked for explicitly: HJ_PAIR=2, HJ_PAIR_NT, HJ_NT / HJ_R)
        const bool small = c->direct_below > 0 && c->total < c->direct_below && HAM::ND <= 3 && !c->halo_lo && !c->halo_hi &&
                           c->pair != 2 && c->pair_nt <= 0 && !c->cfg_from_env && s.p0 >= 0 && s.p1 <= c->N[0] && s.q1 <= s.q0;
        if (!c->force_direct && !small) {
            const bool produce = SCHEME == HJ_WENO5 && eps_producer(c, s);      // two more LDS planes (hj_fused.h, eps_part)
            KernelCfg k = c->cfg;
            int pd = c->pd, occ = c->occ_hint;
            if (!c->cfg_from_env) {
                // round-1 sweeps (profiles/): the heavier the per-cell arithmetic, the fewer cells per
                // thread fit in the 256-VGPR budget without scratch
                if (HAM::ND == 4) { k.NT = sizeof(T) == 4 ? 1024 : 512; k.R = 1; pd = 2; occ = 2; }
                // the light stencils keep 4 cells per thread in registers on large grids; below ~12 M cells
                // (201^3 = 8.1 M: 32 k cells per CU) three small independent workgroups per CU with longer
                // chunks win (sweeps at 101^3 ... 401^3 after the deferred-ghost fix)
                else if (light_scheme(SCHEME) && c->total >= 12000000) { k.NT = 512; k.R = 4; pd = 2; occ = 2; }
                // tiny grids (<= ~135^3) run one wave per SIMD and a launch is a chain of ~10 plane
                // iterations: one cell per thread shortens every iteration (7-10 % at 51^3 ... 129^3)
                else if (light_scheme(SCHEME) && c->total < 2500000) { k.NT = 512; k.R = 1; pd = 2; occ = 4; }
                // (the flag-free stage-2/3 instantiation of the as-shipped WENO5 lands on 170 VGPRs: two workgroups
                // per CU with 34-plane chunks instead of three with 23.  Forcing 168 through the launch bound --
                // config (256,2,2,3,2) -- was measured at 151^3 ... 251^3: within +-2 % of this, not kept)
                else { k.NT = 256; k.R = 2; pd = 2; occ = 2; }
                k.KH = cfg_kh(HAM::ND, k.NT, k.R);
            }
            // 4-D: fp32 with a light stencil only (C5; round 3: 256 threads x 2 pairs, two workgroups per CU: +8 % over 1024 single cells)
            const bool pair_dim = HAM::ND <= 3 || (sizeof(T) == 4 && light_scheme(SCHEME)) || c->pair_nt > 0;
            // light stencils on 2-D / 3-D grids: from 6.5 M cells (191^3 up the (512,2) shape + ring wins; 141^3 ... 181^3 run 3-10 % faster as
            // three 256-thread workgroups per CU of the one-cell-per-lane kernel, tools/experiments/r03_run50.sh, r03_run51.sh)
            const long long pair_from = (HAM::ND <= 3 && light_cfg(SCHEME, HAM::ND)) ? 6500000 : 2500000;
            // 4-D, fp32, light stencil: the compile-time-tile kernel (hj_fused4v.h) when its tile fits the grid (HJ_PAIR4=0: the generic pair kernel)
            if constexpr (HAM::ND == 4 && sizeof(T) == 4 && light_scheme(SCHEME)) {
                // (grids with an extrapolated plane axis keep the compile-time tiles: the ghost-row instantiation of the full-row kernel holds 256 VGPRs and
                //  112-136 B of scratch and runs 7 % SLOWER than pair4 there -- 72^3 x 129: 0.262-0.270 against 0.284-0.290; HJ_FLAT4=2 takes it anyway)
                const bool flat_pays = c->flat4 == 2 || (c->bc[1] == HJ_BC_PERIODIC && c->bc[2] == HJ_BC_PERIODIC && c->bc[3] == HJ_BC_PERIODIC);
                if (c->pair != 0 && c->flat4 != 0 && flat_pays && c->pair_nt <= 0 && (c->total >= pair_from || c->pair == 2)) {
                    // full-row tiles with 16-byte row loads (hj_flat4v.h, round 6) when the contiguous axis fits a row of the box (HJ_FLAT4=0: never)
                    int kf = 0;          // (HJ_FLAT4_SEL = k: only the k-th shape of the list, for A/B runs)
#define X(NT_, R_, E1_, E2_, P3_, OCC_) if ((c->flat4_sel < 0 || c->flat4_sel == kf) && flat4_fits(c, NT_, R_, E1_, E2_, P3_)) return launch_flat4<T, HAM, SCHEME, NT_, R_, E1_, E2_, P3_, OCC_>(c, s); ++kf;
                    HJ_FLAT4(X)
#undef X
                }
                if (c->pair != 0 && c->pair4 != 0 && c->pair_nt <= 0 && (c->total >= pair_from || c->pair == 2)) {
                    // the first tile of the list that fits (HJ_TILE4_SEL = k: only the k-th, for A/B runs)
                    int k4 = 0;
#define X(NT_, R_, E1_, E2_, E3_, OCC_) if ((c->tile4_sel < 0 || c->tile4_sel == k4) && tile4_fits(c, E1_, E2_, E3_)) return launch_pair4<T, HAM, SCHEME, NT_, R_, E1_, E2_, E3_, OCC_>(c, s); ++k4;
                    HJ_TILE4(X)
#undef X
                }
            }
            if (c->pair != 0 && pair_dim && (c->total >= pair_from || c->pair_nt > 0 || c->pair == 2)) {
                // two cells per lane (hj_fusedv.h), round-2 A/B at 151^3 ... 513^3: the light stencils
                // run 2 pairs per thread in 512-thread workgroups (220-232 VGPRs against 246-256 for four single
                // cells), the heavy ones 1 pair in 256-thread workgroups
                KernelCfg kp{512, 2, 2};
                int occp = 2;
                if (!light_cfg(SCHEME, HAM::ND)) { kp.NT = 256; kp.R = 1; kp.KH = 2; }
                if (HAM::ND == 4) { kp.NT = 256; kp.R = 2; kp.KH = 6; }      // 5 pair slots + 1 single slot per thread (hj_fusedv.h, HP)
                if (c->pair_nt > 0) kp.NT = c->pair_nt;
                if (c->pair_r > 0) kp.R = c->pair_r;
                if (c->pair_kh > 0) kp.KH = c->pair_kh;
                if (c->pair_occ > 0) occp = c->pair_occ;
                // halo ring parked in LDS (5 plane buffers, hj_fusedv.h): pays for the two-pairs-per-thread configuration
                // from 201^3 up (A/B tools/experiments/r02_run32.sh, r02_run33.sh: +1 % at 201^3, +3.6 % at 513^3 with
                // 11 % fewer fetched bytes); the 256-thread configurations (several workgroups per CU) lose 1-3 %
                // (2-D / 3-D only: the 4-D instantiations are built without the parked ring -- hj_fusedv.h, AHM -- whatever HJ_PAIR_RING says)
                const bool ring = HAM::ND <= 3 && (c->pair_ring == 1 || (c->pair_ring < 0 && kp.NT == 512 && kp.R == 2 && c->total >= 6500000));
                // (HJ_TWO_PLANES builds: two planes per barrier need four buffers under the planes parked ahead -- hj_fusedv.h, TWOB)
                c->last_nbase = (HJ_TWO_PLANES && light_scheme(SCHEME) && HAM::ND <= 3 && !hj::ham_xp<HAM>::value) ? 4 : 2;
                c->last_nbuf = c->last_nbase + (ring ? c->pair_ah : 0);          // planes parked ahead + the double buffer
                const long long key = ((long long)SCHEME << 40) | ((long long)stage_class(s.stage) << 36) | (1ll << 35) |
                                      ((long long)kp.NT << 20) | ((long long)kp.R << 12) | ((long long)kp.KH << 4) | (long long)(ring ? 1 : 0) |
                                      (produce ? 2ll : 0ll);
                TuneTrial tr;
                // (HJ_WENO_LDS_SHARE builds: two planes for the epsilon producer -- kept whether it runs or not -- and three for the smoothness values
                //  the middle axis shares between lanes: hj_fusedv.h, WX)
                const int wx_planes = (HJ_WENO_LDS_SHARE && SCHEME == HJ_WENO5 && HAM::ND == 3) ? 5 : (produce ? 2 : 0);
                const Tiling tp = tune_begin<HAM::ND>(c, s, kp, 2, c->last_nbuf + wx_planes, key, tr);
                if (tp.ok) {
                    int rc_t = -12345;
#define X(NT_, R_, KH_, OCC_) if constexpr (cfg_built(SCHEME, HAM::ND, NT_, R_, true, (int)sizeof(T))) { if (rc_t == -12345 && kp.NT == NT_ && kp.R == R_ && kp.KH == KH_ && occp == OCC_) rc_t = launch_tiled<T, HAM, SCHEME, NT_, R_, KH_, OCC_, 2, true>(c, s, tp); }
                    if constexpr (HAM::ND == 4) { HJ_CONFIGS_PAIR_4D(X) }
                    else { HJ_CONFIGS_PAIR(X) }
#undef X
                    if (rc_t != -12345) return tune_end<HAM::ND>(c, s, tr, rc_t, SCHEME);
                    if (c->pair_nt > 0 || c->pair_r > 0 || c->pair_kh > 0 || c->pair_occ > 0)
                        return hjh::fail(HJ_EUNSUPPORTED, "pair-kernel configuration (%d,%d,%d,%d) requested through HJ_PAIR_* is not built for scheme %d",
                                         kp.NT, kp.R, kp.KH, occp, SCHEME);
                }
            }
            const long long key1 = ((long long)SCHEME << 40) | ((long long)stage_class(s.stage) << 36) | (produce ? (1ll << 34) : 0ll) |
                                   ((long long)k.NT << 20) | ((long long)k.R << 12) | ((long long)k.KH << 4) | (long long)pd;
            TuneTrial tr1;
            Tiling t = tune_begin<HAM::ND>(c, s, k, 1, produce ? 4 : 2, key1, tr1);
            if (t.ok) {
                int rc_t = -12345;
#define X(NT_, R_, KH_, OCC_, PD_) if constexpr (cfg_built(SCHEME, HAM::ND, NT_, R_, false, (int)sizeof(T))) { if (rc_t == -12345 && k.NT == NT_ && k.R == R_ && k.KH == KH_ && pd == PD_ && occ == OCC_) rc_t = launch_tiled<T, HAM, SCHEME, NT_, R_, KH_, OCC_, PD_>(c, s, t); }
                if constexpr (HAM::ND == 4) { HJ_CONFIGS_4D(X) }
                else { HJ_CONFIGS(X) }
#undef X
                if (rc_t != -12345) return tune_end<HAM::ND>(c, s, tr1, rc_t, SCHEME);
                if (c->cfg_from_env)
                    return hjh::fail(HJ_EUNSUPPORTED, "kernel configuration (%d,%d,%d,%d,%d) requested through HJ_NT/HJ_R/HJ_KH/HJ_OCC/HJ_PD is not built for scheme %d",
                                     k.NT, k.R, k.KH, occ, pd, SCHEME);
            }
        }
    }
    return launch_direct<T, HAM, SCHEME>(c, s);
}

extern template int launch_xp<double, HamDubinsRel<double>>(hj_ctx*, const SubstepCall&);
extern template int launch_xp<float, HamDubinsRel<float>>(hj_ctx*, const SubstepCall&);

template <typename T, typename HAM>
int launch_scheme(hj_ctx* c, const SubstepCall& s) {
    if constexpr (xp_available<T, HAM>()) {
        // the caller asked for the transposed march (thin slabs; HJ_XP=2): taken where the call has such a form, else the launch below
        if (s.xp && c->xp_mode != 0 && !c->force_direct && !c->cfg_from_env && c->pair != 0) {
            // Auto mode (HJ_XP=1).  PRIOR: the two launch plans.  Both forms are memory-system bound, so a launch costs about the cells its
            // workgroups stage -- workgroups x (chunk + 6 warm-up planes) x the tile and its 3-cell frame -- stretched where the workgroups do
            // not fill the CUs' slots evenly (x (1 + 0.4 (slots / workgroups - 1)), fitted on 19 shapes: profiles/r06_thin_slab.txt, "the auto choice over shapes");
            // transposed iff that says <= 0.975 of the axis-0 march.  The model ranks 17 of the 19 shapes correctly and is all a dry context has;
            // a LIVE context measures instead (hj_host.h, XpTrial): runs of six calls of one form between two events, the forms taking turns --
            // same bits either way -- and after HJ_XP_TRIALS runs of each the faster form is kept for the life of the context.
            bool take = true;
            hj_ctx::XpTrial* tr = nullptr;
            bool sample = false;
            if (c->xp_mode == 1 && c->dry != 2) {
                const long long key = ((long long)s.scheme << 56) ^ ((long long)(s.p0 & 0xffffff) << 24) ^ (long long)(s.p1 & 0xffffff);
                auto it = c->xp_choice.find(key);
                if (it == c->xp_choice.end()) {
                    const auto keep_plan = c->last_plan;
                    const char* keep_kernel = c->last_kernel;
                    int keep_E[HJ_MAX_DIM], keep_nbuf = c->last_nbuf, keep_nbase = c->last_nbase;
                    for (int d = 0; d < HJ_MAX_DIM; ++d) keep_E[d] = c->last_E[d];
                    const int keep_dry = c->dry;
                    c->dry = 2;
                    SubstepCall a = s;
                    a.xp = false;
                    auto cost = [&]() {
                        const double wg = std::max(1, c->last_plan.nblocks);
                        const double cap = (double)std::max(1, c->num_cus) * std::max(1, c->last_plan.wg_per_cu);
                        const double slots = cap * std::ceil(wg / cap);
                        return wg * (c->last_E[0] + 2 * HJ_STENCIL) * (double)(c->last_E[1] + 2 * HJ_STENCIL) * (double)(c->last_E[2] + 2 * HJ_STENCIL) *
                               (1.0 + 0.4 * (slots / wg - 1.0));
                    };
                    double c0 = -1, cx = -1;
                    if (launch_scheme<T, HAM>(c, a) == HJ_OK && c->last_plan.ntiles > 0) c0 = cost();
                    if (launch_xp<T, HAM>(c, s) == HJ_OK) cx = cost();
                    c->dry = keep_dry;
                    c->last_plan = keep_plan; c->last_kernel = keep_kernel; c->last_nbuf = keep_nbuf; c->last_nbase = keep_nbase;
                    for (int d = 0; d < HJ_MAX_DIM; ++d) c->last_E[d] = keep_E[d];
                    hj_ctx::XpTrial t;
                    t.prior = cx > 0 && (c0 <= 0 || cx <= 0.975 * c0);
                    if (cx <= 0) t.decided = 0;                                  // this call has no transposed form
                    else if (c0 <= 0) t.decided = 1;
                    else if (c->dry || c->xp_trials == 0) t.decided = t.prior;
                    it = c->xp_choice.emplace(key, t).first;
                }
                tr = &it->second;
                take = tr->decided >= 0 ? tr->decided == 1 : tr->prior;
                if (tr->decided < 0 && !c->dry && !c->launch_stop) {
                    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
                    if (hipStreamIsCapturing(call_stream(c, s), &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
                        (void)hipGetLastError();
                        tr->calls = 0;                                                 // (a run cut by a capture is dropped)
                    } else {
                        for (int f = 0; f < 2; ++f) {
                            if (!tr->pend[f] || hipEventQuery(tr->ev[f][1]) != hipSuccess) continue;
                            float ms = 0;
                            if (hipEventElapsedTime(&ms, tr->ev[f][0], tr->ev[f][1]) == hipSuccess && ms > 0) {
                                tr->best[f] = std::min(tr->best[f], ms);
                                tr->n[f] += 1;
                            }
                            tr->pend[f] = false;
                        }
                        (void)hipGetLastError();                                       // (hipErrorNotReady is not this launch's error)
                        if (tr->calls == 0 && tr->n[0] >= c->xp_trials && tr->n[1] >= c->xp_trials) {
                            tr->decided = tr->best[1] < tr->best[0] ? 1 : 0;
                            take = tr->decided == 1;
                            if (c->debug_xp) fprintf(stderr, "[hj] planes [%lld, %lld): %d calls of the axis-0 march %.4f ms, of the transposed march %.4f ms (plan model: %s) -> %s\n",
                                                     (long long)s.p0, (long long)s.p1, (int)hj_ctx::XP_RUN, tr->best[0], tr->best[1], tr->prior ? "transposed" : "axis-0",
                                                     take ? "transposed" : "axis-0");
                        } else if (tr->calls > 0) {
                            take = tr->form == 1;                                      // inside a run
                            sample = true;
                        } else {
                            const int f = tr->n[1] + (tr->pend[1] ? 1 : 0) < tr->n[0] + (tr->pend[0] ? 1 : 0) ? 1 : 0;      // take turns, the axis-0 march first
                            if (!tr->ev[0][0]) {
                                bool ok = true;
                                for (int i = 0; i < 4; ++i) ok = ok && hipEventCreate(&tr->ev[i / 2][i % 2]) == hipSuccess;
                                if (!ok) { (void)hipGetLastError(); tr->decided = tr->prior; }
                            }
                            if (tr->decided < 0 && tr->started >= 4 * c->xp_trials + 40) {
                                // (never settled -- the stream is captured most of the time, or the tile-shape rotation never ends: what there is)
                                tr->decided = tr->n[0] > 0 && tr->n[1] > 0 ? (tr->best[1] < tr->best[0] ? 1 : 0) : (tr->prior ? 1 : 0);
                                take = tr->decided == 1;
                            } else if (tr->decided < 0 && !tr->pend[f] && tr->n[f] < c->xp_trials + 2) {
                                if (hipEventRecord(tr->ev[f][0], call_stream(c, s)) == hipSuccess) {
                                    tr->form = f; tr->started += 1; tr->seq0 = c->tune_seq; take = f == 1; sample = true;
                                } else (void)hipGetLastError();
                            }
                        }
                    }
                }
            }
            int rc = HJ_XP_FALLBACK;
            if (take) {
                rc = launch_xp<T, HAM>(c, s);
                if (rc == HJ_XP_FALLBACK && tr) { tr->decided = 0; tr->calls = 0; sample = false; }
            }
            if (rc == HJ_XP_FALLBACK && tr) {
                SubstepCall a = s;
                a.xp = false;
                rc = launch_scheme<T, HAM>(c, a);
            }
            if (rc != HJ_XP_FALLBACK) {
                if (sample) {
                    if (rc != HJ_OK) tr->calls = 0;
                    else if (++tr->calls >= (int)hj_ctx::XP_RUN) {
                        tr->calls = 0;
                        if (c->tune_seq != tr->seq0) {}       // the axis-0 march was trying tile shapes (synchronising trials): not its steady state
                        else if (hipEventRecord(tr->ev[tr->form][1], call_stream(c, s)) == hipSuccess) tr->pend[tr->form] = true;
                        else (void)hipGetLastError();
                    }
                }
                return rc;
            }
        }
    }
    switch (s.scheme) {
        case HJ_ENO2: return launch_cfg<T, HAM, HJ_ENO2>(c, s);
        case HJ_ENO3: return launch_cfg<T, HAM, HJ_ENO3>(c, s);
        case HJ_WENO5: return launch_cfg<T, HAM, HJ_WENO5>(c, s);
        case HJ_WENO5_ASSHIPPED: return launch_cfg<T, HAM, HJ_WENO5_ASSHIPPED>(c, s);
        case HJ_ENO2_FAST: return launch_cfg<T, HAM, HJ_ENO2_FAST>(c, s);
        case HJ_ENO3_FAST: return launch_cfg<T, HAM, HJ_ENO3_FAST>(c, s);
    }
    return hjh::fail(HJ_EINVAL, "unknown scheme %d", s.scheme);
}

// ------------------------------------------------------------------------------------ stage-fused launch
// (threads, A slots per thread, H slots per thread, waves/SIMD hint)
#ifndef HJ_CONFIGS12
#define HJ_CONFIGS12(X) X(512, 4, 2, 2) X(512, 3, 2, 2) X(512, 2, 2, 2)
#endif

struct Tiling12 {
    int E[HJ_MAX_DIM], ntile[HJ_MAX_DIM];
    int ntiles, chunk, nchunks, nblocks, bpx, nA, nH;
    size_t lds_bytes;
    double score;
    bool ok;
};

inline Tiling12 make_tiling12(const hj_ctx* c, int NT, int R, int KH, size_t lds_limit) {
    const int nd = c->ndim, W = HJ_STENCIL;
    Tiling12 best;
    best.ok = false;
    best.score = 1e300;
    int n[HJ_MAX_DIM];
    for (int d = 0; d < nd; ++d) n[d] = (int)c->N[d];
    std::vector<int> cand2;                 // extents of the last axis
    for (int parts = 1; parts <= 128; ++parts) {
        int e = (n[nd - 1] + parts - 1) / parts;
        if (cand2.empty() || cand2.back() != e) cand2.push_back(e);
        if (e <= 2) break;
    }
    if (c->f12_e2 > 0) cand2.assign(1, std::min(c->f12_e2, n[nd - 1]));      // tuning knob: force the row extent
    for (int e2 : cand2) {
        if (e2 < 2) continue;
        const int e1max = nd == 3 ? n[1] : 1;
        for (int e1 = (nd == 3 ? 2 : 1); e1 <= e1max; ++e1) {
            long long T0, nA, nH, ybox, wbox, cells;
            if (nd == 3) {
                cells = (long long)e1 * e2;
                nA = cells + 2 * W * (e1 + e2);
                nH = 4 * W * (e1 + e2) + 4 * W * W;
                T0 = cells + 4 * W * (e1 + e2) + 4 * W * W;
                ybox = (long long)(e1 + 4 * W) * (e2 + 4 * W);
                wbox = (long long)(e1 + 2 * W) * (e2 + 2 * W);
            } else {
                cells = e2;
                nA = cells + 2 * W;
                nH = 4 * W;
                T0 = cells + 4 * W;
                ybox = e2 + 4 * W;
                wbox = e2 + 2 * W;
            }
            if (nA > (long long)NT * R || nH > (long long)NT * KH) continue;
            const size_t lds = 512 + (size_t)(2 * ybox + 7 * wbox) * c->esz;
            if (lds > lds_limit) continue;
            const int E[3] = {1, nd == 3 ? e1 : e2, e2};
            double waste = 1.0;
            for (int d = 1; d < nd; ++d) {
                const int ed = (nd == 3) ? E[d] : e2;
                const int nt = (n[d] + ed - 1) / ed;
                waste *= (double)nt * ed / (double)n[d];
            }
            const double row_cost = 1.0 + (48.0 / (double)c->esz) / (double)e2;
            const double util = (double)nA / (double)(((nA + NT - 1) / NT) * NT);
            // loads per output cell + evaluations per output cell (stage 1 on T1, stage 2 on T)
            const double score = ((double)T0 / cells * row_cost + 0.6 * (double)(nA + cells) / cells / util) * waste;
            if (score < best.score) {
                best.ok = true;
                best.score = score;
                best.lds_bytes = lds;
                best.nA = (int)nA;
                best.nH = (int)nH;
                best.ntiles = 1;
                best.E[0] = 1; best.ntile[0] = 1;
                for (int d = 1; d < nd; ++d) {
                    best.E[d] = (nd == 3) ? E[d] : e2;
                    best.ntile[d] = (n[d] + best.E[d] - 1) / best.E[d];
                    best.ntiles *= best.ntile[d];
                }
            }
        }
    }
    return best;
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, bool PAIR>
auto stage12_kernel() {
    if constexpr (PAIR) return fused12_pair_kernel<T, HAM, SCHEME, NT, R, KH, OCC>;
    else return fused12_kernel<T, HAM, SCHEME, NT, R, KH, OCC>;
}

template <typename T, typename HAM, int SCHEME, int NT, int R, int KH, int OCC, bool PAIR = false>
int launch_fused12(hj_ctx* c, const Stage12Call& s, Tiling12 t) {
    constexpr int ND = HAM::ND;
    auto kern = stage12_kernel<T, HAM, SCHEME, NT, R, KH, OCC, PAIR>();
    c->last_kernel = PAIR ? "fused12_pair_kernel" : "fused12_kernel";
    if (t.lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes));
    const auto key = std::make_pair(reinterpret_cast<const void*>(kern), t.lds_bytes);
    auto it = c->occ_cache.find(key);
    if (it == c->occ_cache.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, key.first, NT, t.lds_bytes) != hipSuccess || nb < 1) nb = 1;
        it = c->occ_cache.emplace(key, nb).first;
    }
    {   // axis-0 chunking: whole rounds of resident workgroups; a chunk pays 12 planes of loads and 6 of stage 1
        const int64_t planes = c->N[0];
        const int64_t capacity = (int64_t)c->num_cus * it->second;
        int64_t best_nch = 1;
        double best_cost = 1e300;
        for (int64_t nch = 1; nch <= std::max<int64_t>(1, planes / 8); ++nch) {
            const int64_t chunk = (planes + nch - 1) / nch;
            const int64_t blocks = ((planes + chunk - 1) / chunk) * t.ntiles;
            const int64_t rounds = (blocks + capacity - 1) / capacity;
            const double cost = (double)rounds * ((double)chunk + (double)c->f12_warm);
            if (cost < best_cost - 1e-9) { best_cost = cost; best_nch = nch; }
        }
        if (c->target_blocks > 0) best_nch = std::max<int64_t>(1, std::min<int64_t>(planes / 8, c->target_blocks / t.ntiles));
        t.chunk = (int)((planes + best_nch - 1) / best_nch);
        t.nchunks = (int)((planes + t.chunk - 1) / t.chunk);
        t.nblocks = t.nchunks * t.ntiles;
        t.bpx = (t.nblocks + 7) / 8;
    }
    if (c->debug) {
        fprintf(stderr, "[hj] fused12%s NT=%d R=%d KH=%d OCC=%d E=(%d,%d) nA=%d nH=%d ntiles=%d chunk=%d nchunks=%d blocks=%d wg/CU=%d lds=%zu score=%.3f\n",
                PAIR ? " pair" : "", NT, R, KH, OCC, t.E[1], ND > 2 ? t.E[2] : 0, t.nA, t.nH, t.ntiles, t.chunk, t.nchunks, t.nblocks, it->second, t.lds_bytes, t.score);
        c->debug = 0;
    }
    Fused12Args<T, ND> A;
    memset(&A, 0, sizeof(A));
    A.bound = s.bound;
    long long st = 1;
    for (int d = ND - 1; d >= 0; --d) {
        A.n[d] = (int)c->N[d];
        A.bc[d] = c->bc[d];
        A.km[d] = c->tz[d] ? T(-1) : T(1);
        fill_stencil_constants<T>(c->dx[d], A.K[d]);
        A.sc[d] = scheme_scale<T>(SCHEME, c->dx[d]);
        A.pstride[d] = (d >= 1) ? (int)st : 0;
        if (d == 0) A.stride0 = st;
        st *= c->N[d];
        A.E[d] = t.E[d];
        A.ntile[d] = t.ntile[d];
    }
    A.total_bytes = (unsigned)((size_t)c->total * c->esz);
    A.ntiles = t.ntiles;
    A.chunk = t.chunk;
    A.nchunks = t.nchunks;
    A.plane_begin = 0;
    A.plane_end = (int)c->N[0];
    A.nblocks = t.nblocks;
    A.blocks_per_xcd = t.bpx;
    A.nA = t.nA;
    A.nH = t.nH;
    A.stage2 = (s.ca == 0.5) ? HJ_STAGE_RK2_FULL : HJ_STAGE_RK3_HALF;     // hj_rk_stage12 accepts exactly these two pairs
    A.ca = (T)s.ca;
    A.cb = (T)s.cb;
    A.dt = (T)s.dt;
    fill_ham<T>(c, s.par, A.ham, s.ham);
#ifdef HJ_F12_STAMP
    const char* dump = c->timing_dump;              // diagnostic build: per-wave phase clocks of the pair kernel (tools/f12_stamps.py)
    unsigned long long* tbuf = nullptr;
    const size_t tw = (size_t)t.nblocks * (NT / 64) * 10;
    if (PAIR && dump && *dump) {
        HIP_TRY(hipMalloc(&tbuf, tw * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(tbuf, 0, tw * sizeof(unsigned long long), c->stream));
        A.timing = tbuf;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(t.bpx * 8), dim3(NT), t.lds_bytes, c->stream, (const T*)s.y, (T*)s.out, A);
    HIP_TRY(hipGetLastError());
#ifdef HJ_F12_STAMP
    if (tbuf) {
        std::vector<unsigned long long> h(tw);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(h.data(), tbuf, tw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(tbuf));
        if (FILE* f = fopen(dump, "a")) {
            fprintf(f, "# fused12_pair launch nblocks=%d waves=%d ntiles=%d chunk=%d E=(%d,%d)\n", t.nblocks, NT / 64, t.ntiles, t.chunk, t.E[1], ND > 2 ? t.E[2] : 0);
            for (size_t i = 0; i < tw / 10; ++i) {
                fprintf(f, "%zu %zu", i / (NT / 64), i % (NT / 64));
                for (int k = 0; k < 10; ++k) fprintf(f, " %llu", h[i * 10 + k]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
#endif
    return HJ_OK;
}

// ---- the two-cells-per-lane stage-fused kernel (hj_fused12v.h): (threads, A PAIR slots per thread, H slots per
// thread, waves/SIMD hint)
#ifndef HJ_CONFIGS12V
#define HJ_CONFIGS12V(X) X(512, 2, 2, 2) X(512, 3, 2, 2) X(256, 4, 4, 1)
#endif

// Tiles of the pair variant: even extent on the contiguous axis, no pair of T1 may straddle a domain edge
// (pairs start on even tile columns -4, -2, .., E2+2: the unshifted remainder of the last tile must not be 1 or
// 3 columns), LDS = 2 y boxes (E1+12) x (E2+16) + 7 y1 boxes (E1+6) x (E2+8).
inline Tiling12 make_tiling12v(const hj_ctx* c, int NT, int R, int KH, size_t lds_limit) {
    const int nd = c->ndim, W = HJ_STENCIL;
    Tiling12 best;
    best.ok = false;
    best.score = 1e300;
    int n[HJ_MAX_DIM];
    for (int d = 0; d < nd; ++d) n[d] = (int)c->N[d];
    const int nl = n[nd - 1];
    for (int e2 = 4; e2 <= std::min(nl, 1024); e2 += 2) {
        if (c->f12_e2 > 0 && e2 != std::min(c->f12_e2 & ~1, nl & ~1)) continue;
        const int nt2 = (nl + e2 - 1) / e2;
        const int rem2 = nl - (nt2 - 1) * e2;
        if (nt2 > 1 && (rem2 == 1 || rem2 == 3)) continue;
        if (nt2 == 1 && e2 != nl) continue;                   // a single tile must span the (even) axis exactly
        const int e1max = nd == 3 ? n[1] : 1;
        for (int e1 = (nd == 3 ? 2 : 1); e1 <= e1max; ++e1) {
            if (nd == 3 && c->f12_e1 > 0 && e1 != std::min(c->f12_e1, e1max)) continue;
            long long nA, nH, T0, ybox, wbox, cells;
            cells = (long long)e1 * e2;
            if (nd == 3) {
                nA = cells / 2 + 2 * W * (e2 / 2) + 4 * e1;
                nH = 4 * W * (e1 + e2) + 4 * W * W;
                T0 = cells + 4 * W * (e1 + e2) + 4 * W * W + 2 * e1;
                ybox = (long long)(e1 + 4 * W) * (e2 + 16);
                wbox = (long long)(e1 + 2 * W) * (e2 + 8);
            } else {
                nA = cells / 2 + 4;
                nH = 4 * W;
                T0 = cells + 4 * W + 2;
                ybox = e2 + 16;
                wbox = e2 + 8;
            }
            if (nA > (long long)NT * R || nH > (long long)NT * KH) continue;
            const size_t lds = 512 + (size_t)(2 * ybox + 7 * wbox) * c->esz;
            if (lds > lds_limit) continue;
            double waste = 1.0;
            {
                const int nt = (nl + e2 - 1) / e2;
                waste *= (double)nt * e2 / (double)nl;
                if (nd == 3) {
                    const int nt1 = (n[1] + e1 - 1) / e1;
                    waste *= (double)nt1 * e1 / (double)n[1];
                }
            }
            const double row_cost = 1.0 + (48.0 / (double)c->esz) / (double)e2;
            // slot evaluations per output pair: every A slot runs stage 1, the slots that hold interior pairs
            // (whole waves) run stage 2
            const long long s1 = ((nA + 63) / 64) * 64, s2 = ((cells / 2 + 63) / 64) * 64;
            const double evals = (double)(s1 + s2) / (double)(cells / 2);
            const double score = ((double)T0 / cells * row_cost + 0.6 * evals) * waste;
            if (score < best.score) {
                best.ok = true;
                best.score = score;
                best.lds_bytes = lds;
                best.nA = (int)nA;
                best.nH = (int)nH;
                best.ntiles = 1;
                best.E[0] = 1; best.ntile[0] = 1;
                for (int d = 1; d < nd; ++d) {
                    best.E[d] = (nd == 3 && d == 1) ? e1 : e2;
                    best.ntile[d] = (n[d] + best.E[d] - 1) / best.E[d];
                    best.ntiles *= best.ntile[d];
                }
            }
        }
    }
    return best;
}

template <typename T, typename HAM, int SCHEME>
int launch_stage12_cfg(hj_ctx* c, const Stage12Call& s) {
#ifdef HJ_TUNE_BUILD
    constexpr bool built = std::is_same<T, double>::value && HAM::ID == HJ_HAM_DUBINS_REL && SCHEME == HJ_WENO5_ASSHIPPED;
#else
    constexpr bool built = true;
#endif
    if constexpr (HAM::ND == 4 || SCHEME == HJ_WENO5 || !built) {
        return hjh::fail(HJ_EUNSUPPORTED, "no stage-fused kernel for this scheme / dimension");
    } else {
        if (c->f12_pair) {
            // two cells per lane (round 3): 2 pair slots per thread in 512-thread workgroups
            int NT = 512, R = 2, KH = 2;
            if (c->f12_nt > 0) NT = c->f12_nt;
            if (c->f12_r > 0) R = c->f12_r;
            if (c->f12_kh > 0) KH = c->f12_kh;
            const Tiling12 t = make_tiling12v(c, NT, R, KH, (size_t)160 * 1024 - 256);
            if (t.ok) {
                if (s.probe) return HJ_OK;
#define X(NT_, R_, KH_, OCC_) if (NT == NT_ && R == R_ && KH == KH_) return launch_fused12<T, HAM, SCHEME, NT_, R_, KH_, OCC_, true>(c, s, t);
                HJ_CONFIGS12V(X)
#undef X
                return hjh::fail(HJ_EUNSUPPORTED, "stage-fused pair configuration (%d,%d,%d) is not built", NT, R, KH);
            }
            if (c->f12_pair > 1) return hjh::fail(HJ_EUNSUPPORTED, "no stage-fused pair tiling for this grid");
        }
        // heavier per-cell arithmetic -> fewer A slots per thread fit in 256 VGPRs
        int NT = 512, R = (SCHEME == HJ_ENO3) ? 2 : (SCHEME == HJ_ENO2 ? 4 : 3), KH = 2;
        if (c->f12_pair) { NT = 512; KH = 2; }                // the env overrides describe the pair configuration
        else {
        if (c->f12_nt > 0) NT = c->f12_nt;
        if (c->f12_r > 0) R = c->f12_r;
        if (c->f12_kh > 0) KH = c->f12_kh;
        }
        const Tiling12 t = make_tiling12(c, NT, R, KH, (size_t)160 * 1024 - 256);
        if (!t.ok) return hjh::fail(HJ_EUNSUPPORTED, "no stage-fused tiling for this grid");
        if (s.probe) return HJ_OK;
        // at most the scheme's default number of A slots per thread is built: 4 (ENO2), 3 (as-shipped WENO5), 2 (ENO3)
#define X(NT_, R_, KH_, OCC_) if constexpr (R_ <= ((SCHEME == HJ_ENO3) ? 2 : (SCHEME == HJ_ENO2 ? 4 : 3)) || NT_ != 512) { if (NT == NT_ && R == R_ && KH == KH_) return launch_fused12<T, HAM, SCHEME, NT_, R_, KH_, OCC_>(c, s, t); }
        HJ_CONFIGS12(X)
#undef X
        return hjh::fail(HJ_EUNSUPPORTED, "stage-fused configuration (%d,%d,%d) is not built", NT, R, KH);
    }
}

template <typename T, typename HAM>
int launch_stage12(hj_ctx* c, const Stage12Call& s) {
    switch (s.scheme) {
        case HJ_ENO2: return launch_stage12_cfg<T, HAM, HJ_ENO2>(c, s);
        case HJ_ENO3: return launch_stage12_cfg<T, HAM, HJ_ENO3>(c, s);
        case HJ_WENO5_ASSHIPPED: return launch_stage12_cfg<T, HAM, HJ_WENO5_ASSHIPPED>(c, s);
    }
    return hjh::fail(HJ_EUNSUPPORTED, "no stage-fused kernel for scheme %d", s.scheme);
}

#ifdef HJ_INST_TERM_ND
// ---- termNormal / termReinit / termConvection through the tiled one-cell-per-lane kernel (round 4): HAM = TermOp<T, ND, KIND>,
// MODE 0 (ydot only).  256 threads x 2 cells (two to four workgroups per CU); the direct term_kernel stays for small grids,
// 4-D, fp32 and slabs (hj_api.hip, term_run).
template <typename T, int ND, int KIND, int SCHEME>
int launch_term_cfg(hj_ctx* c, const SubstepCall& s) {
    // two cells per thread; the intended WENO5 (the heaviest stencil) one: with two it spills 60-268 bytes per lane
#ifndef HJ_TERM_R
#define HJ_TERM_R 2
#endif
#ifndef HJ_TERM_OCC
#define HJ_TERM_OCC 2
#endif
    constexpr int NT = 256, R = (SCHEME == HJ_WENO5) ? 1 : HJ_TERM_R, KH = 2, OCC = HJ_TERM_OCC, PD = 2;
    const KernelCfg k{NT, R, KH};
    const Tiling t = make_tiling(c, k, s.p0, s.p1, 1, 2);
    if (!t.ok) return hjh::fail(HJ_EUNSUPPORTED, "no tiling of this grid for the tiled term kernel");
    return launch_tiled_mode<T, hj::TermOp<T, ND, KIND>, SCHEME, NT, R, KH, OCC, PD, 0, false>(c, s, t);
}
template <typename T, int ND, int KIND>
int launch_term_kind(hj_ctx* c, const SubstepCall& s) {
    switch (s.scheme) {
        case HJ_ENO2: return launch_term_cfg<T, ND, KIND, HJ_ENO2>(c, s);
        case HJ_ENO3: return launch_term_cfg<T, ND, KIND, HJ_ENO3>(c, s);
        case HJ_WENO5: return launch_term_cfg<T, ND, KIND, HJ_WENO5>(c, s);
        case HJ_WENO5_ASSHIPPED: return launch_term_cfg<T, ND, KIND, HJ_WENO5_ASSHIPPED>(c, s);
    }
    return hjh::fail(HJ_EINVAL, "unknown scheme %d", s.scheme);
}
template <typename T, int ND>
int launch_term_tiled(hj_ctx* c, int kind, const SubstepCall& s) {
    switch (kind) {
        case hj::HJ_TERM_NORMAL: return launch_term_kind<T, ND, hj::HJ_TERM_NORMAL>(c, s);
        case hj::HJ_TERM_REINIT: return launch_term_kind<T, ND, hj::HJ_TERM_REINIT>(c, s);
        case hj::HJ_TERM_CONVECTION: return launch_term_kind<T, ND, hj::HJ_TERM_CONVECTION>(c, s);
    }
    return hjh::fail(HJ_EINVAL, "unknown term %d", kind);
}
template int launch_term_tiled<HJ_INST_T, HJ_INST_TERM_ND>(hj_ctx*, int, const SubstepCall&);
#else
template int launch_scheme<HJ_INST_T, hj::HJ_INST_HAM<HJ_INST_T>>(hj_ctx*, const SubstepCall&);
template int launch_coop<HJ_INST_T, hj::HJ_INST_HAM<HJ_INST_T>>(hj_ctx*, const CoopCall&);
template int launch_stage12<HJ_INST_T, hj::HJ_INST_HAM<HJ_INST_T>>(hj_ctx*, const Stage12Call&);
#endif

}  // namespace hjh
