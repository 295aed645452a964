// C ABI of libhj_mi355x.so (see include/hj_mi355x.h).  gfx950 only.
#include <chrono>
#include <dlfcn.h>

#include "hj_host.h"
#include "hj_terms.h"

namespace hjh {

thread_local char g_err[4096] = "";      // (room for a hipRTC compiler message)

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

int alpha_partials(hj_ctx* c) {
    if (!c->alpha_part) HIP_TRY(hipMalloc((void**)&c->alpha_part, (size_t)ALPHA_BLOCKS_MAX * 8 * sizeof(double)));
    return HJ_OK;
}

// a zeroed entry of the range-key ring becomes ctx->range_keys (as next_ring() does for the CFL bounds: a memset launch per pass
// costs 4 us of launch time, the ring is zeroed once per RANGE_RING passes)
int next_range_keys(hj_ctx* c) {
    const size_t entry = RANGE_ENTRY;
    if (!c->range_ring) {
        HIP_TRY(hipMalloc((void**)&c->range_ring, sizeof(unsigned long long) * entry * RANGE_RING));
        HIP_TRY(hipMemsetAsync(c->range_ring, 0, sizeof(unsigned long long) * entry * RANGE_RING, c->stream));
        c->range_pos = 0;
    }
    if (c->range_pos >= RANGE_RING) {
        // every launch that reads or writes an entry is stream-ordered on the ctx streams: drain them, zero, go round again
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->edge_stream) HIP_TRY(hipStreamSynchronize(c->edge_stream));
        HIP_TRY(hipMemsetAsync(c->range_ring, 0, sizeof(unsigned long long) * entry * RANGE_RING, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->range_pos = 0;
    }
    c->range_keys = c->range_ring + entry * (size_t)c->range_pos++;
    return HJ_OK;
}

// ------------------------------------------------------------------------------------ tiling
// Pick tile extents E[1..nd-1] (cells <= NT*R, halo slots <= KH*NT, LDS <= limit) minimising
// (cells + halo)/cells / lane_utilisation, then the axis-0 chunking.
// vec = 2: the pair kernel (hj_fusedv.h) -- k.R counts PAIRS per thread, the extent of the last axis is even, its LDS
// rows are E + 8 cells apart (left pad 4)
static Tiling make_tiling_uncached(const hj_ctx* c, const KernelCfg& k, int64_t p0, int64_t p1, int vec, int nbuf, std::vector<Tiling>* all,
                                   const int64_t* dims = nullptr);

Tiling make_tiling(const hj_ctx* c, const KernelCfg& k, int64_t p0, int64_t p1, int vec, int nbuf, std::vector<Tiling>* all) {
    if (all) return make_tiling_uncached(c, k, p0, p1, vec, nbuf, all);
    const long long key = ((long long)k.NT << 40) | ((long long)k.R << 32) | ((long long)k.KH << 24) | ((long long)vec << 16) | (long long)nbuf;
    auto it = c->tiling_cache.find(key);
    if (it == c->tiling_cache.end()) it = c->tiling_cache.emplace(key, make_tiling_uncached(c, k, p0, p1, vec, nbuf, nullptr)).first;
    return it->second;
}

// the tiling of a TRANSPOSED launch (hj_fusedv.h, XP): `dims` are the extents in KERNEL order -- dims[1] is the length of the window on the
// slab axis, the others the grid's -- cached per window length
Tiling make_tiling_dims(const hj_ctx* c, const KernelCfg& k, int vec, int nbuf, const int64_t* dims) {
    const long long key = (1ll << 62) | ((long long)k.NT << 48) | ((long long)k.R << 44) | ((long long)k.KH << 40) | ((long long)vec << 36) | ((long long)nbuf << 32) |
                          (long long)(dims[1] & 0xffffffffll);
    auto it = c->tiling_cache.find(key);
    if (it == c->tiling_cache.end()) it = c->tiling_cache.emplace(key, make_tiling_uncached(c, k, 0, dims[0], vec, nbuf, nullptr, dims)).first;
    return it->second;
}

static Tiling make_tiling_uncached(const hj_ctx* c, const KernelCfg& k, int64_t p0, int64_t p1, int vec, int nbuf, std::vector<Tiling>* all,
                                   const int64_t* dims) {
    const int nd = c->ndim;
    std::map<int, Tiling> per_row;     // best tiling per extent of the last axis (autotuner candidates)
    // HJ_TILE_CELLS (tuning): cap the tile below what the configuration holds -- more, smaller workgroups (thin slabs)
    const int cap = c->tile_cells > 0 ? std::min(k.NT * k.R * vec, c->tile_cells) : k.NT * k.R * vec;
    Tiling best;
    best.ok = false;
    best.score = 1e300;
    int n[HJ_MAX_DIM];
    for (int d = 0; d < nd; ++d) n[d] = (int)(dims ? dims[d] : c->N[d]);
    std::vector<int> cand[HJ_MAX_DIM];
    for (int d = 1; d < nd; ++d) {
        for (int parts = 1; parts <= 64; ++parts) {
            int e = (n[d] + parts - 1) / parts;
            if (e < 1) e = 1;
            if (cand[d].empty() || cand[d].back() != e) cand[d].push_back(e);
            if (e == 1) break;
        }
    }
    if (vec == 2) {            // even extents only (a tile never exceeds the axis: the last one is shifted back)
        std::vector<int> ev;
        for (int e : cand[nd - 1]) {
            int e2 = std::min((e + 1) & ~1, n[nd - 1] & ~1);
            if (e2 >= 2 && (ev.empty() || ev.back() != e2)) ev.push_back(e2);
        }
        cand[nd - 1] = ev;
        if (ev.empty()) return best;
    }
    if (c->full_rows == 1) {   // tuning knob: the last axis is never split (contiguous tile planes)
        cand[nd - 1].assign(1, n[nd - 1]);
    } else if (c->full_rows > 1) {   // tuning knob: force the last-axis extent
        cand[nd - 1].assign(1, std::min(c->full_rows, n[nd - 1]));
    }
    int E[HJ_MAX_DIM] = {1, 1, 1, 1};
    // enumerate extents of all plane axes except the first, which takes what is left
    std::vector<int> it(HJ_MAX_DIM, 0);
    while (true) {
        long long cells = 1;
        for (int d = 2; d < nd; ++d) { E[d] = cand[d][it[d]]; cells *= E[d]; }
        if (cells <= cap) {
            E[1] = (int)std::min<long long>(n[1], std::max<long long>(1, cap / cells));
            if (vec == 2 && nd == 2) E[1] &= ~1;
            // VERTICAL PAIRS (hj_fusedv.h, HJ_VPAIR): a thread's two pair slots are the same columns of two adjacent tile rows -> an even row count
            const bool vpair = vec == 2 && nd == 3 && k.R == 2 && HJ_VPAIR;
            if (vpair) E[1] &= ~1;
            const bool no_rows = E[1] < 1;          // (vertical pairs need two rows: a row extent that leaves room for one only is no candidate)
            if (E[1] < 1) E[1] = 1;
            const bool odd_row = (vec == 2 && (E[nd - 1] & 1)) || no_rows;      // the pair kernel needs an even row extent (and, with vertical pairs, two rows at least)
            // E[1] must also be one of "n/parts" only for balance; any value works for correctness
            cells *= E[1];
            long long halo = 0, box = 1;
            for (int d = 1; d < nd; ++d) { halo += 6 * (cells / E[d]); box *= (E[d] + 6); }
            // bank-friendly row pitch (E + 32) when it fits, else the minimal E + 6
            const long long rows = box / (E[nd - 1] + 6);
            int pitch = E[nd - 1] + (vec == 2 ? 8 : 6);
            if (vec == 1 && c->lds_pad > 0 && nd >= 3 && 512 + 2 * (size_t)(rows * (E[nd - 1] + 32)) * c->esz <= c->lds_limit) pitch = E[nd - 1] + 32;
            // pair kernel (HJ_LDS_PAD=1, tuning): a 16-lane group of a ds_read_b128 that straddles two tile rows is conflict free
            // iff the row pitch in PAIRS is congruent to the row length in pairs mod 16, i.e. 32 cells of padding (round 3)
            if (vec == 2 && c->lds_pad != 0 && nd >= 3) {
                const size_t cap2 = nbuf > 2 ? (size_t)160 * 1024 - 1024 : c->lds_limit;
                if (512 + (size_t)nbuf * (size_t)(rows * (E[nd - 1] + 32)) * c->esz <= cap2) pitch = E[nd - 1] + 32;
            }
            pitch += c->lds_pitch_add;          // HJ_LDS_PITCH_ADD (tuning): extra cells of row padding
            box = rows * pitch;
            size_t lds = 512 + (size_t)nbuf * (size_t)box * c->esz;
            const size_t lds_cap = nbuf > 2 ? (size_t)160 * 1024 - 1024 : c->lds_limit;   // the ring variant may take the whole CU
            // slots the kernel has for the halo: KH single cells per thread -- or, pair kernel in 4-D (hj_fusedv.h, HP), KH - 1 pair
            // slots for the layers of the non-contiguous plane axes and one single slot for the 3 + 3 cells either side of a row
            bool halo_fits = halo <= (long long)k.KH * k.NT;
            if (vec == 2 && nd == 4) {
                const long long singles = 6 * (cells / E[nd - 1]);
                halo_fits = (halo - singles) / 2 <= (long long)(k.KH - 1) * k.NT && singles <= k.NT;
            }
            if (!odd_row && halo_fits && lds <= lds_cap) {
                double util = (double)cells / (double)(((cells + k.NT - 1) / k.NT) * k.NT);
                // cells recomputed by the shifted last tile on each axis
                double waste = 1.0;
                for (int d = 1; d < nd; ++d) {
                    int nt = (n[d] + E[d] - 1) / E[d];
                    waste *= (double)nt * E[d] / (double)n[d];
                }
                // every row of the tile (own and halo) drags in partial cache lines at both ends:
                // ~48 B per row, calibrated on the 401^3 fp64 and 129^4 fp32 tile-shape sweeps
                const double row_cost = 1.0 + (48.0 / (double)c->esz) / (double)E[nd - 1];
                double score = ((double)(cells + halo) / (double)cells) * row_cost * waste / util;
                Tiling cur;
                cur.ok = true;
                cur.score = score;
                cur.lds_bytes = lds;
                cur.lpitch = pitch;
                cur.ntiles = 1;
                for (int d = 0; d < HJ_MAX_DIM; ++d) { cur.E[d] = 1; cur.ntile[d] = 1; }
                for (int d = 1; d < nd; ++d) {
                    cur.E[d] = E[d];
                    cur.ntile[d] = (n[d] + E[d] - 1) / E[d];
                    cur.ntiles *= cur.ntile[d];
                }
                cur.chunk = cur.nchunks = cur.nchunks1 = cur.nblocks = cur.bpx = 0;
                if (score < best.score) best = cur;
                if (all) {
                    auto it = per_row.find(E[nd - 1]);
                    if (it == per_row.end() || score < it->second.score) per_row[E[nd - 1]] = cur;
                }
            }
        }
        int d = 2;
        for (; d < nd; ++d) {
            if (++it[d] < (int)cand[d].size()) break;
            it[d] = 0;
        }
        if (d >= nd) break;
    }
    if (all) {
        all->clear();
        for (auto& kv : per_row) all->push_back(kv.second);
        std::sort(all->begin(), all->end(), [](const Tiling& a, const Tiling& b) { return a.score < b.score; });
    }
    return best;
}

// Axis-0 chunking: blocks = ntiles * nchunks should fill the GPU in whole "rounds" of resident
// workgroups (capacity = CUs * workgroups per CU for this kernel), and every chunk pays 6 warm-up
// planes of loads.  Pick the chunk count that minimises  rounds * (chunk + warm-up cost).
void choose_chunks(const hj_ctx* c, Tiling& t, int64_t p0, int64_t p1, int blocks_per_cu, int64_t chunk_max, double march_bytes, double fixed_bytes) {
    const int64_t planes = p1 - p0;
    const int64_t capacity = (int64_t)c->num_cus * std::max(1, blocks_per_cu);
    // one buffer descriptor spans a chunk plus 3 planes either side: keep it below 4 GiB
    // (march_bytes > 0: a transposed launch -- bytes per step of the march and the fixed part of the span, FusedArgs::xspan)
    const double plane_bytes = march_bytes > 0 ? march_bytes : (double)(c->total / c->N[0]) * (double)c->esz;
    int64_t chunk_cap = (int64_t)((4294967295.0 - fixed_bytes) / plane_bytes) - 2 * HJ_STENCIL;
    if (chunk_max > 0) chunk_cap = std::min(chunk_cap, chunk_max);
    if (chunk_cap < 1) { t.ok = false; return; }
    int64_t best_nch = 1;
    double best_cost = 1e300;
    const int64_t max_nch = c->target_blocks > 0 ? std::max<int64_t>(1, c->target_blocks / t.ntiles)
                                                 : std::max<int64_t>(1, planes / c->min_chunk);
    for (int64_t nch = 1; nch <= std::min<int64_t>(planes, std::max<int64_t>(max_nch, 1)); ++nch) {
        const int64_t chunk = (planes + nch - 1) / nch;
        if (chunk > chunk_cap) continue;
        const int64_t nchunks = (planes + chunk - 1) / chunk;
        const int64_t blocks = nchunks * t.ntiles;
        const int64_t rounds = (blocks + capacity - 1) / capacity;
        const double cost = (double)rounds * ((double)chunk + (double)c->warmup_cost);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_nch = nch; }
    }
    if (c->target_blocks > 0) best_nch = std::max<int64_t>(1, std::min<int64_t>(planes / std::max(1, c->min_chunk), c->target_blocks / t.ntiles));
    t.chunk = (int)std::min<int64_t>((planes + best_nch - 1) / best_nch, chunk_cap);
    t.nchunks = (int)((planes + t.chunk - 1) / t.chunk);
    t.nblocks = t.nchunks * t.ntiles;
    t.bpx = (t.nblocks + 7) / 8;
}

int cfg_kh(int nd, int nt, int r) {
#define X(NT_, R_, KH_, OCC_, PD_) if (nt == NT_ && r == R_) return KH_;
    if (nd == 4) { HJ_CONFIGS_4D(X) }
    else { HJ_CONFIGS(X) }
#undef X
    return -1;
}

// one object file per (dtype, Hamiltonian): hj_inst.hip
extern template int launch_scheme<double, HamDubinsRel<double>>(hj_ctx*, const SubstepCall&);
extern template int launch_scheme<double, HamDoubleIntegrator<double>>(hj_ctx*, const SubstepCall&);
extern template int launch_scheme<double, HamDoublePendulum<double>>(hj_ctx*, const SubstepCall&);
extern template int launch_scheme<float, HamDubinsRel<float>>(hj_ctx*, const SubstepCall&);
extern template int launch_scheme<float, HamDoubleIntegrator<float>>(hj_ctx*, const SubstepCall&);
extern template int launch_scheme<float, HamDoublePendulum<float>>(hj_ctx*, const SubstepCall&);
extern template int launch_coop<double, HamDubinsRel<double>>(hj_ctx*, const CoopCall&);
extern template int launch_coop<double, HamDoubleIntegrator<double>>(hj_ctx*, const CoopCall&);
extern template int launch_coop<double, HamDoublePendulum<double>>(hj_ctx*, const CoopCall&);
extern template int launch_coop<float, HamDubinsRel<float>>(hj_ctx*, const CoopCall&);
extern template int launch_coop<float, HamDoubleIntegrator<float>>(hj_ctx*, const CoopCall&);
extern template int launch_coop<float, HamDoublePendulum<float>>(hj_ctx*, const CoopCall&);
extern template int launch_term_tiled<double, 2>(hj_ctx*, int, const SubstepCall&);
extern template int launch_term_tiled<double, 3>(hj_ctx*, int, const SubstepCall&);
extern template int launch_stage12<double, HamDubinsRel<double>>(hj_ctx*, const Stage12Call&);
extern template int launch_stage12<double, HamDoubleIntegrator<double>>(hj_ctx*, const Stage12Call&);
extern template int launch_stage12<double, HamDoublePendulum<double>>(hj_ctx*, const Stage12Call&);
extern template int launch_stage12<float, HamDubinsRel<float>>(hj_ctx*, const Stage12Call&);
extern template int launch_stage12<float, HamDoubleIntegrator<float>>(hj_ctx*, const Stage12Call&);
extern template int launch_stage12<float, HamDoublePendulum<float>>(hj_ctx*, const Stage12Call&);

// should the plane range [p0, p1) be computed by the TRANSPOSED launch (march along axis 1, hj_fusedv.h XP)?  Auto rule (HJ_XP=1), by
// measurement (profiles/r06_thin_slab.txt): a 3-D range that is thin against axis 1 AND has the GPU to itself -- a thin grid or a slab
// without neighbours: 65 x 513 x 513 runs 0.286 ms per RK3 step transposed against 0.302 on the axis-0 march (399 short workgroups in two
// rounds against 252 equal ones in one).  NOT the interior launch of a slab that exchanges halos: there the edge launch and the RCCL kernel
// share the GPU with it, and a launch whose 256 workgroups all end together leaves them no CU until its end -- the step is bound by the
// chain exchange -> edges -> exchange either way (0.363 against 0.362 ms; the timelines are in the profile).  HJ_XP=2 forces it everywhere.
bool xp_wanted(const hj_ctx* c, int64_t p0, int64_t p1) {
    if (c->xp_mode == 0 || c->ndim != 3) return false;
    if (c->xp_mode == 2) return true;
    if (c->halo_lo || c->halo_hi) return false;
    // CANDIDATES only: ranges of 8 planes up to a third of axis 1 at the pair kernel's sizes.  launch_scheme (hj_inst.hip) decides: a cost model over
    // the two launch plans (dry contexts, hj_plan_substep), and on a live context both forms timed against each other on its first calls
    const int64_t w = p1 - p0;
    const int64_t lim = c->xp_max_planes > 0 ? c->xp_max_planes : c->N[1] / 3;
    return w >= 8 && w <= lim && c->total >= 6500000;
}

}  // namespace hjh

using namespace hj;
using namespace hjh;

namespace {

int ham_ndim(int ham) {
    switch (ham) {
        case HJ_HAM_DUBINS_REL: return 3;
        case HJ_HAM_DOUBLE_INTEGRATOR: return 2;
        case HJ_HAM_DOUBLE_PENDULUM: return 4;
    }
    return user_ham_ndim(ham);       // -1 unless registered at run time (hj_rtc.hip)
}
}  // namespace

namespace hjh {
// how many entries of the caller's `par` array a Hamiltonian reads (fill_ham copies exactly these)
int ham_npar(int ham) {
    switch (ham) {
        case HJ_HAM_DUBINS_REL: return 4;
        case HJ_HAM_DOUBLE_INTEGRATOR: return 1;
        case HJ_HAM_DOUBLE_PENDULUM: return 1;
    }
    return user_ham_npar(ham);
}
}  // namespace hjh

namespace {

template <typename T>
int launch_ham(hj_ctx* c, const SubstepCall& s) {
    switch (s.ham) {
        case HJ_HAM_DUBINS_REL: return launch_scheme<T, HamDubinsRel<T>>(c, s);
        case HJ_HAM_DOUBLE_INTEGRATOR: return launch_scheme<T, HamDoubleIntegrator<T>>(c, s);
        case HJ_HAM_DOUBLE_PENDULUM: return launch_scheme<T, HamDoublePendulum<T>>(c, s);
        default: if (user_ham_valid(s.ham)) return launch_user(c, s);
    }
    return fail(HJ_EINVAL, "unknown Hamiltonian id %d", s.ham);
}

int check_ham(const hj_ctx* c, int ham, const double* par) {
    const int nd = ham_ndim(ham);
    if (nd < 0) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham);
    if (nd != c->ndim)
        return fail(HJ_EINVAL, "Hamiltonian %d is %d-dimensional but the grid has dim %d", ham, nd, c->ndim);
    if (!par) return fail(HJ_EINVAL, "ham_params is NULL");
    if (ham == HJ_HAM_DUBINS_REL && (!c->aux[0] || !c->aux[1] || c->aux_n[0] != c->N[2] || c->aux_n[1] != c->N[2]))
        return fail(HJ_ESTATE, "Dubins tables missing (internal)");
    return HJ_OK;
}

template <typename T> int upload_table(hj_ctx* c, void** dst, const double* src, int64_t n) {
    std::vector<T> tmp((size_t)n);
    for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = (T)src[i];
    if (*dst) { HIP_TRY(hipFree(*dst)); *dst = nullptr; }
    HIP_TRY(hipMalloc(dst, (size_t)n * sizeof(T)));
    // synchronous copy from a temporary: tables are set up rarely
    HIP_TRY(hipMemcpy(*dst, tmp.data(), (size_t)n * sizeof(T), hipMemcpyHostToDevice));
    return HJ_OK;
}

int upload(hj_ctx* c, void** dst, const double* src, int64_t n) {
    return c->dtype == HJ_F64 ? upload_table<double>(c, dst, src, n) : upload_table<float>(c, dst, src, n);
}

// default trig tables from the coordinates (libm); Python overrides them with NumPy's values
int default_aux(hj_ctx* c) {
    std::vector<double> t;
    auto mk = [&](int slot, int dim, double (*f)(double)) -> int {
        const auto& v = c->coord_host[dim];
        t.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) t[i] = f(v[i]);
        c->aux_n[slot] = (int64_t)v.size();
        return upload(c, &c->aux[slot], t.data(), (int64_t)v.size());
    };
    int rc = HJ_OK;
    if (c->ndim == 3) {
        if ((rc = mk(0, 2, cos))) return rc;
        if ((rc = mk(1, 2, sin))) return rc;
    } else if (c->ndim == 4) {
        if ((rc = mk(0, 0, sin))) return rc;
        if ((rc = mk(1, 0, cos))) return rc;
        if ((rc = mk(2, 2, sin))) return rc;
        if ((rc = mk(3, 2, cos))) return rc;
    }
    return rc;
}

unsigned long long* next_ring(hj_ctx* c, int user_slot, int* rc) {
    *rc = HJ_OK;
    if (c->ring_pos >= RING_SLOTS) {
        // wrap-around (once per RING_SLOTS launches): kernels on EITHER stream may still be reducing into
        // their slots, so drain both before the whole ring is zeroed, and finish the memset before any
        // stream can launch into it again
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && c->edge_stream) e = hipStreamSynchronize(c->edge_stream);
        // What the user slots hold SURVIVES the reset (round 6): a sequence of launches whose bounds are read together -- the stages of
        // one hj_rk_step, a caller's substeps before its hj_read_step_bound -- may straddle the wrap; zeroing the entry of stage 2 while
        // stage 3 was being enqueued lost its bound ("bound slot holds no reduction", once per 2048 launches at the unlucky alignment).
        // The live entries (final: the streams are drained) are parked, the ring zeroed, and they move to its first entries.
        int live[HJ_BOUND_SLOTS], nlive = 0;
        const size_t eb = sizeof(unsigned long long) * HJ_MAX_DIM;
        if (e == hipSuccess && !c->ring_keep) e = hipMalloc((void**)&c->ring_keep, eb * HJ_BOUND_SLOTS);
        for (int i = 0; i < HJ_BOUND_SLOTS && e == hipSuccess; ++i)
            if (c->slot_ring[i] >= 0) {
                e = hipMemcpyAsync(c->ring_keep + (size_t)nlive * HJ_MAX_DIM, c->ring + (size_t)c->slot_ring[i] * HJ_MAX_DIM, eb, hipMemcpyDeviceToDevice, c->stream);
                live[nlive++] = i;
            }
        if (e == hipSuccess) e = hipMemsetAsync(c->ring, 0, sizeof(unsigned long long) * RING_SLOTS * HJ_MAX_DIM, c->stream);
        if (e == hipSuccess && nlive) e = hipMemcpyAsync(c->ring, c->ring_keep, eb * nlive, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { *rc = fail(HJ_EHIP, "bound ring reset: %s", hipGetErrorString(e)); return nullptr; }
        for (int i = 0; i < HJ_BOUND_SLOTS; ++i) c->slot_ring[i] = -1;
        for (int k = 0; k < nlive; ++k) c->slot_ring[live[k]] = k;
        c->ring_pos = nlive;
    }
    const int pos = c->ring_pos++;
    if (user_slot >= 0 && user_slot < HJ_BOUND_SLOTS) c->slot_ring[user_slot] = pos;
    return c->ring + (size_t)pos * HJ_MAX_DIM;
}

double key_to_double(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    memcpy(&v, &b, 8);
    return v;
}

// max(D1^2) per dim of `y` -> `out` (ndim values of the ctx dtype, device), stream-ordered
int weno_eps_to(hj_ctx* c, const void* y, void* out) {
    const int64_t S = c->total / c->N[0];
    const int bx = (int)((S + 255) / 256);
    // enough workgroups to fill the GPU a few times over, chunks of at least 8 planes
    // (round 2: folding the partials in the last workgroup to arrive -- one launch instead of two -- was
    // measured and is no faster: 2000 ticket atomics on one word cost 23 us, and with 320 workgroups the
    // serial tail of the finisher eats the launch it saves: 29.4 us against 22 + 5.7)
    int by = (int)std::max<int64_t>(1, std::min<int64_t>((c->N[0] + 7) / 8, (256 * 8 + bx - 1) / bx));
    const int chunk = (int)((c->N[0] + by - 1) / by);
    by = (int)((c->N[0] + chunk - 1) / chunk);
    const int nblocks = bx * by;
    if (nblocks > c->partials_cap) {
        if (c->partials) { HIP_TRY(hipFree(c->partials)); c->partials = nullptr; }
        HIP_TRY(hipMalloc((void**)&c->partials, sizeof(double) * HJ_MAX_DIM * (size_t)nblocks));
        c->partials_cap = nblocks;
    }
#define HJ_MAXD1(T, ND)                                                                          \
    {                                                                                            \
        GridArgs<T, ND> G;                                                                       \
        fill_grid<T, ND>(c, G);                                                                  \
        hipLaunchKernelGGL((max_d1sq_kernel<T, ND>), dim3(bx, by), dim3(256), 0, c->stream,      \
                           (const T*)y, G, c->partials, chunk);                                  \
        hipLaunchKernelGGL((partials_to_values_kernel<T>), dim3(1), dim3(256), 0, c->stream,     \
                           c->partials, nblocks, (T*)out, c->ndim);                              \
    }
    if (c->dtype == HJ_F64) {
        if (c->ndim == 2) HJ_MAXD1(double, 2) else if (c->ndim == 3) HJ_MAXD1(double, 3) else HJ_MAXD1(double, 4)
    } else {
        if (c->ndim == 2) HJ_MAXD1(float, 2) else if (c->ndim == 3) HJ_MAXD1(float, 3) else HJ_MAXD1(float, 4)
    }
#undef HJ_MAXD1
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

int weno_eps_pass(hj_ctx* c, const void* y) { return weno_eps_to(c, y, c->weno_vals); }

// The pre-pass as ONE launch (round 3): at most 512 workgroups of 1024 threads leave their rows in c->partials and the
// consuming substep kernel folds them in its prologue (FusedArgs::eps_rows), like the rows of eps_seam_kernel.
// *nrows = 0: the grid has more than 512 K columns per plane -- the caller takes the two-launch form.
// (one helper for the launch code and for hj_rk_plan's launch count: ADVICE r03)
static inline bool weno_rows_fit(const hj_ctx* c) { return (c->total / c->N[0] + 1023) / 1024 <= 512; }

int weno_eps_rows(hj_ctx* c, const void* y, int* nrows) {
    *nrows = 0;
    const int64_t S = c->total / c->N[0];
    const int64_t bx = (S + 1023) / 1024;
    if (!weno_rows_fit(c)) return HJ_OK;
    int by = (int)std::max<int64_t>(1, std::min<int64_t>((c->N[0] + 7) / 8, 512 / bx));
    const int chunk = (int)((c->N[0] + by - 1) / by);
    by = (int)((c->N[0] + chunk - 1) / chunk);
    const int nblocks = (int)bx * by;
    if (nblocks > c->partials_cap) {
        if (c->partials) { HIP_TRY(hipFree(c->partials)); c->partials = nullptr; }
        HIP_TRY(hipMalloc((void**)&c->partials, sizeof(double) * HJ_MAX_DIM * (size_t)std::max(nblocks, 4096)));
        c->partials_cap = std::max(nblocks, 4096);
    }
#define HJ_MAXD1R(T, ND)                                                                                   \
    {                                                                                                      \
        GridArgs<T, ND> G;                                                                                 \
        fill_grid<T, ND>(c, G);                                                                            \
        hipLaunchKernelGGL((max_d1sq_kernel<T, ND, 1024>), dim3((unsigned)bx, by), dim3(1024), 0, c->stream, \
                           (const T*)y, G, c->partials, chunk);                                            \
    }
    if (c->dtype == HJ_F64) {
        if (c->ndim == 2) HJ_MAXD1R(double, 2) else if (c->ndim == 3) HJ_MAXD1R(double, 3) else HJ_MAXD1R(double, 4)
    } else {
        if (c->ndim == 2) HJ_MAXD1R(float, 2) else if (c->ndim == 3) HJ_MAXD1R(float, 3) else HJ_MAXD1R(float, 4)
    }
#undef HJ_MAXD1R
    HIP_TRY(hipGetLastError());
    *nrows = nblocks;
    return HJ_OK;
}



// kept for call sites that used the two-step form: the values are already in place
int keys_to_vals(hj_ctx* c, void* out) {
    if (out != c->weno_vals)
        HIP_TRY(hipMemcpyAsync(out, c->weno_vals, c->esz * (size_t)c->ndim, hipMemcpyDeviceToDevice, c->stream));
    return HJ_OK;
}

int do_substep(hj_ctx* c, SubstepCall& s, int user_slot) {
    int rc = check_ham(c, s.ham, s.par);
    if (rc) return rc;
    if (s.scheme < 0 || s.scheme > HJ_ENO3_FAST) return fail(HJ_EINVAL, "unknown scheme %d", s.scheme);
    {   // pad planes may be computed too when the tables cover them (deep-halo stepper)
        const int64_t lo = c->halo_lo ? -(int64_t)c->pad0 : 0, hi = c->N[0] + (c->halo_hi ? c->pad0 : 0);
        if (s.p0 < lo || s.p1 > hi || s.p0 > s.p1 || (s.q1 > s.q0 && (s.q0 < lo || s.q1 > hi)))
            return fail(HJ_EINVAL, "bad plane range [%lld,%lld)", (long long)s.p0, (long long)s.p1);
    }
    if (!s.y || !s.out) return fail(HJ_EINVAL, "null array argument");
    if (s.y == s.out) return fail(HJ_EINVAL, "out must not alias the stencil input y");
    if (s.stage >= HJ_STAGE_RK3_HALF && !s.y0) return fail(HJ_EINVAL, "stage %d needs y0", s.stage);
    if (s.stage < 0 || s.stage > HJ_STAGE_RK2_FULL) return fail(HJ_EINVAL, "unknown stage %d", s.stage);
    for (int d = 0; d < c->ndim; ++d)
        if (c->N[d] < HJ_STENCIL)
            return fail(HJ_EINVAL, "grid too small along dim %d (N=%lld)", d, (long long)c->N[d]);
    if (s.p0 == s.p1) return HJ_OK;
    if (s.scheme == HJ_WENO5 && !c->weno_src) {
        // inside hj_rk_step the previous launch reduced max(D1^2) of this launch's input itself (eps_ready): no pre-pass
        if (s.eps_from_prev && c->eps_ready && c->eps_fuse) { s.eps_rows = c->eps_rows; s.eps_nrows = HJ_EPS_ROWS; }
        else {
            int nrows = 0;
            if (c->eps_fuse && !s.on_aux && (rc = weno_eps_rows(c, s.y, &nrows))) return rc;
            if (nrows > 0) { s.eps_rows = c->partials; s.eps_nrows = nrows; }
            else {
                if ((rc = weno_eps_pass(c, s.y))) return rc;
                if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
            }
        }
        if (s.want_eps && c->eps_fuse && !c->eps_rows)
            HIP_TRY(hipMalloc((void**)&c->eps_rows, sizeof(double) * HJ_MAX_DIM * HJ_EPS_ROWS));
    }
    c->eps_ready = false;          // whatever the rows described is about to be (or may have been) overwritten
    // user_slot < 0: nobody will read this launch's CFL bound (hj_rk_step and the slab steppers take dt from the static bound):
    // the kernel skips its reduction and the contended atomics, and the key ring does not advance
    s.bound = nullptr;
    if (user_slot >= 0 || c->keep_bounds) {
        s.bound = next_ring(c, user_slot, &rc);
        if (rc) return rc;
    }
    if (c->xp_mode == 2) s.xp = true;          // HJ_XP=2 (tests): every launch that has a transposed form takes it
    else if (!s.xp) s.xp = xp_wanted(c, s.p0, s.p1);      // thin grids / slabs without neighbours (the rule above)
    if (c->xp_mode == 0) s.xp = false;
    return c->dtype == HJ_F64 ? launch_ham<double>(c, s) : launch_ham<float>(c, s);
}

// hj_rk_step on a SMALL grid (what launch_cfg sends to direct_substep_kernel): the whole step as one cooperative launch (coop_rk_kernel).
// HJ_XP_FALLBACK: not applicable -- the caller issues the stages one launch each.
template <typename T> int coop_ham(hj_ctx* c, const CoopCall& s) {
    switch (s.ham) {
        case HJ_HAM_DUBINS_REL: return launch_coop<T, HamDubinsRel<T>>(c, s);
        case HJ_HAM_DOUBLE_INTEGRATOR: return launch_coop<T, HamDoubleIntegrator<T>>(c, s);
        case HJ_HAM_DOUBLE_PENDULUM: return launch_coop<T, HamDoublePendulum<T>>(c, s);
    }
    return HJ_XP_FALLBACK;
}
bool coop_applies(const hj_ctx* c, int order, int scheme, int ham) {
    if (!c->coop || order < 2 || scheme == HJ_WENO5 || c->dry || c->keep_bounds || c->halo_lo || c->halo_hi || c->ndim > 3) return false;
    // (only where the ordinary step is `order` launches of the direct kernel: launch_cfg's small-grid rule)
    if (!(c->direct_below > 0 && c->total < c->direct_below) || c->force_direct || c->pair == 2 || c->pair_nt > 0 || c->cfg_from_env || c->timing_dump)
        return false;
    return ham == HJ_HAM_DUBINS_REL || ham == HJ_HAM_DOUBLE_INTEGRATOR || ham == HJ_HAM_DOUBLE_PENDULUM;
}
int try_coop_step(hj_ctx* c, int order, int scheme, int ham, const double* par, double dt, int restrict_sign, const void* y_in, void* y_out,
                  void* work0, void* work1) {
    if (!coop_applies(c, order, scheme, ham)) return HJ_XP_FALLBACK;
    if (order == 3 && (work1 == y_out || work1 == y_in || work0 == y_in)) return HJ_XP_FALLBACK;
    if (c->coop_ok < 0) {
        int v = 0;
        c->coop_ok = (hipDeviceGetAttribute(&v, hipDeviceAttributeCooperativeLaunch, c->device) == hipSuccess && v) ? 1 : 0;
    }
    if (!c->coop_ok) return HJ_XP_FALLBACK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;           // (a cooperative launch inside a stream capture: not tried)
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return HJ_XP_FALLBACK; }
    int rc = check_ham(c, ham, par);
    if (rc) return rc;
    if (!c->coop_sync) {
        HIP_TRY(hipMalloc(&c->coop_sync, sizeof(CoopSync)));
        HIP_TRY(hipMemset(c->coop_sync, 0, sizeof(CoopSync)));
    }
    CoopCall s{scheme, ham, order, restrict_sign, c->post_step_op, par, dt, y_in, order == 2 ? work0 : work0, order == 2 ? y_out : work1, y_out};
    c->eps_ready = false;
    return c->dtype == HJ_F64 ? coop_ham<double>(c, s) : coop_ham<float>(c, s);
}

template <typename T> int stage12_ham(hj_ctx* c, const Stage12Call& s) {
    switch (s.ham) {
        case HJ_HAM_DUBINS_REL: return launch_stage12<T, HamDubinsRel<T>>(c, s);
        case HJ_HAM_DOUBLE_INTEGRATOR: return launch_stage12<T, HamDoubleIntegrator<T>>(c, s);
        case HJ_HAM_DOUBLE_PENDULUM: return launch_stage12<T, HamDoublePendulum<T>>(c, s);
    }
    return fail(HJ_EINVAL, "unknown Hamiltonian id %d", s.ham);
}

// Two RK stages in one launch (hj_fused12.h).  probe: only report whether the kernel can take this grid.
int do_stage12(hj_ctx* c, Stage12Call& s, int user_slot) {
    int rc = check_ham(c, s.ham, s.par);
    if (rc) return rc;
    if (c->ndim > 3 || s.scheme == HJ_WENO5 || s.scheme < 0 || s.scheme > 3)
        return fail(HJ_EUNSUPPORTED, "no stage-fused kernel for this scheme / dimension");
    if (c->halo_lo || c->halo_hi) return fail(HJ_EUNSUPPORTED, "the stage-fused kernel does not take slab halos");
    if (np_order(s.scheme) && !s.probe && !((s.ca == 0.75 && s.cb == 0.25) || (s.ca == 0.5 && s.cb == 0.5)))
        return fail(HJ_EUNSUPPORTED, "ENO schemes evaluate the reference's own stage expressions: (ca, cb) must be (3/4, 1/4) or (1/2, 1/2)");
    if ((double)c->total * (double)c->esz >= 4294967295.0) return fail(HJ_EUNSUPPORTED, "array of 4 GiB or more");
    if (c->N[0] < 8) return fail(HJ_EUNSUPPORTED, "fewer than 8 planes");
    for (int d = 1; d < c->ndim; ++d)
        if (c->N[d] < 4) return fail(HJ_EUNSUPPORTED, "fewer than 4 nodes along dim %d", d);
    if (!s.probe) {
        if (!s.y || !s.out) return fail(HJ_EINVAL, "null array argument");
        if (s.y == s.out) return fail(HJ_EINVAL, "out must not alias the stencil input y");
        c->eps_ready = false;
        s.bound = nullptr;
        if (user_slot >= 0 || c->keep_bounds) {
            s.bound = next_ring(c, user_slot, &rc);
            if (rc) return rc;
        }
    }
    return c->dtype == HJ_F64 ? stage12_ham<double>(c, s) : stage12_ham<float>(c, s);
}

// does hj_rk_step fuse the first two stages on this ctx?  (cached per scheme / Hamiltonian)
bool use_stage12(hj_ctx* c, int order, int scheme, int ham, const double* par, int restrict_sign) {
    if (order < 2 || c->fuse12 == 0 || restrict_sign != 0 || ham >= HJ_HAM_USER_BASE || scheme > HJ_WENO5_ASSHIPPED) return false;
    if (order == 2 && c->post_step_op != 0) return false;     // the fused post-step operator rides on the LAST stage
    if (c->fuse12 < 0) {
        // auto: the fusion trades HBM traffic for redundant ring / warm-up work; it pays once the arrays are
        // far beyond the 256 MB Infinity Cache (3-D) or whenever the ring is a sliver of the tile (2-D)
        const double bytes = (double)c->total * (double)c->esz;
        if (c->ndim == 3 && bytes < 300e6) return false;
        if (c->ndim == 2 && bytes < 32e6) return false;
    }
    const int key = scheme * 8 + ham;
    auto it = c->f12_ok.find(key);
    if (it == c->f12_ok.end()) {
        Stage12Call s{scheme, ham, par, 0.0, 0.0, 0.0, nullptr, nullptr, nullptr, true};
        it = c->f12_ok.emplace(key, do_stage12(c, s, -1) == HJ_OK).first;
    }
    return it->second;
}

int read_ring(hj_ctx* c, int pos, double* sb, double* amax) {
    unsigned long long k[HJ_MAX_DIM];
    HIP_TRY(hipMemcpyAsync(k, c->ring + (size_t)pos * HJ_MAX_DIM, sizeof(k), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double inv = 0.0;
    for (int d = 0; d < c->ndim; ++d) {
        if (k[d] == 0) return fail(HJ_ESTATE, "bound slot holds no reduction for dim %d", d);
        // alpha arrives as the ctx dtype; the reference divides by grid.dx in fp64 (artificial_diss_glf.py:107)
        const double a = key_to_double(k[d]);
        if (amax) amax[d] = a;
        inv += a / c->dx[d];
    }
    if (sb) *sb = 1.0 / inv;
    return HJ_OK;
}

}  // namespace

int hjh::eps_rows_to_vals(hj_ctx* c, const double* rows, int nrows, hipStream_t stream) {
    if (c->dtype == HJ_F64)
        hipLaunchKernelGGL((partials_to_values_kernel<double>), dim3(1), dim3(256), 0, stream, rows, nrows, (double*)c->weno_vals, c->ndim);
    else
        hipLaunchKernelGGL((partials_to_values_kernel<float>), dim3(1), dim3(256), 0, stream, rows, nrows, (float*)c->weno_vals, c->ndim);
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

static int dim_view(const hj_ctx* c, int dim, long long* outer, long long* inner) {
    long long o = 1, i = 1;
    for (int d = 0; d < dim; ++d) o *= c->N[d];
    for (int d = dim + 1; d < c->ndim; ++d) i *= c->N[d];
    *outer = o; *inner = i;
    return 0;
}

template <typename T> static DimView<T> make_view(const hj_ctx* c, int dim, bool with_halo) {
    DimView<T> V;
    dim_view(c, dim, &V.outer, &V.inner);
    V.n = (int)c->N[dim];
    V.bc = c->bc[dim];
    V.halo_lo = (with_halo && dim == 0) ? c->halo_lo : 0;
    V.halo_hi = (with_halo && dim == 0) ? c->halo_hi : 0;
    V.km = c->tz[dim] ? T(-1) : T(1);
    fill_stencil_constants<T>(c->dx[dim], V.K);
    return V;
}

template <typename T> static int upwind_launch(hj_ctx* c, int scheme, int dim, const void* phi, void* dL, void* dR, unsigned long long* keys, const T* epsv) {
    const int blocks = (int)std::min<int64_t>((c->total + 255) / 256, 256 * 8);
    DimView<T> V = make_view<T>(c, dim, true);
    T eps = T(0);
    if (scheme == HJ_WENO5) {
        // tiny D2H of one value: the split path is the host-synchronous compatibility path anyway
        T m;
        HIP_TRY(hipMemcpyAsync(&m, epsv + dim, sizeof(T), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        eps = T(1e-6) * m + Lim<T>::tiny;
    }
#define HJ_UPW(S) hipLaunchKernelGGL((upwind_kernel<T, S>), dim3(blocks), dim3(256), 0, c->stream, (const T*)phi, (T*)dL, (T*)dR, V, eps, keys)
    switch (scheme) {
        case HJ_ENO2: HJ_UPW(HJ_ENO2); break;
        case HJ_ENO3: HJ_UPW(HJ_ENO3); break;
        case HJ_WENO5: HJ_UPW(HJ_WENO5); break;
        case HJ_WENO5_ASSHIPPED: HJ_UPW(HJ_WENO5_ASSHIPPED); break;
        default: return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    }
#undef HJ_UPW
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

template <typename T, int ND>
static int upwind_all_launch(hj_ctx* c, int scheme, const void* y, void* const* dL, void* const* dR, unsigned long long* keys, const T* epsv) {
    UpwindAllArgs<T, ND> A;
    memset(&A, 0, sizeof(A));
    hjh::fill_grid<T, ND>(c, A.G);
    for (int d = 0; d < ND; ++d) { A.dL[d] = (T*)dL[d]; A.dR[d] = (T*)dR[d]; }
    A.max_d1sq = epsv;
    A.keys = keys;
    const int blocks = (int)std::min<int64_t>((c->total + 255) / 256, 256 * 16);
#define HJ_UPA(S) hipLaunchKernelGGL((upwind_all_kernel<T, ND, S>), dim3(blocks), dim3(256), 0, c->stream, (const T*)y, A)
    switch (scheme) {
        case HJ_ENO2: HJ_UPA(HJ_ENO2); break;
        case HJ_ENO3: HJ_UPA(HJ_ENO3); break;
        case HJ_WENO5: HJ_UPA(HJ_WENO5); break;
        case HJ_WENO5_ASSHIPPED: HJ_UPA(HJ_WENO5_ASSHIPPED); break;
        default: return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    }
#undef HJ_UPA
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}
template <typename T>
static int upwind_all_dispatch(hj_ctx* c, int scheme, const void* y, void* const* dL, void* const* dR, unsigned long long* keys, const T* epsv) {
    if (c->ndim == 2) return upwind_all_launch<T, 2>(c, scheme, y, dL, dR, keys, epsv);
    if (c->ndim == 3) return upwind_all_launch<T, 3>(c, scheme, y, dL, dR, keys, epsv);
    return upwind_all_launch<T, 4>(c, scheme, y, dL, dR, keys, epsv);
}

// ------------------------------------------------------------------------------------ RCCL (dlopen)
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;

int rccl_load(const char* path) {
    if (g_rccl.handle) return HJ_OK;
    const char* names[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(HJ_ESTATE, "cannot dlopen RCCL: %s", dlerror());
#define HJ_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                     \
    if (!g_rccl.field) return fail(HJ_ESTATE, "RCCL symbol %s missing", name);
    HJ_SYM(GetUniqueId, "ncclGetUniqueId")
    HJ_SYM(CommInitRank, "ncclCommInitRank")
    HJ_SYM(CommDestroy, "ncclCommDestroy")
    HJ_SYM(GroupStart, "ncclGroupStart")
    HJ_SYM(GroupEnd, "ncclGroupEnd")
    HJ_SYM(Send, "ncclSend")
    HJ_SYM(Recv, "ncclRecv")
    HJ_SYM(AllReduce, "ncclAllReduce")
    HJ_SYM(GetErrorString, "ncclGetErrorString")
    HJ_SYM(CommCount, "ncclCommCount")
    HJ_SYM(CommUserRank, "ncclCommUserRank")
#undef HJ_SYM
    g_rccl.handle = h;
    return HJ_OK;
}

#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(HJ_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));         \
    } while (0)

// post the halo sends/receives of `buf` (first interior plane) on `st`.  Order per peer is fixed so
// that lo == hi (two ranks, periodic axis, or the single-rank self ring) pairs up: sends
// [low planes -> lo, high planes -> hi], receives [hi pad <- hi, lo pad <- lo].
int post_halo(hj_ctx* c, void* buf, hipStream_t st, int depth = HJ_STENCIL) {
    if (c->external_exchange) return HJ_OK;      // the caller moves the planes (hj_comm_init_external)
    if (!c->comm) return fail(HJ_ESTATE, "hj_comm_init has not been called");
    const size_t plane = (size_t)(c->total / c->N[0]);
    const size_t cnt = plane * (size_t)depth;
    const ncclDataType_t dt = c->dtype == HJ_F64 ? ncclDouble : ncclFloat;
    char* b = (char*)buf;
    const size_t pb = plane * c->esz;
    const int64_t n = c->N[0];
    NCCL_TRY(g_rccl.GroupStart());
    if (c->lo_rank >= 0) NCCL_TRY(g_rccl.Send(b, cnt, dt, c->lo_rank, c->comm, st));
    if (c->hi_rank >= 0) NCCL_TRY(g_rccl.Send(b + (size_t)(n - depth) * pb, cnt, dt, c->hi_rank, c->comm, st));
    if (c->hi_rank >= 0) NCCL_TRY(g_rccl.Recv(b + (size_t)n * pb, cnt, dt, c->hi_rank, c->comm, st));
    if (c->lo_rank >= 0) NCCL_TRY(g_rccl.Recv(b - (size_t)depth * pb, cnt, dt, c->lo_rank, c->comm, st));
    NCCL_TRY(g_rccl.GroupEnd());
    return HJ_OK;
}

// One substep on a slab.  Streams: `main` (ctx stream) runs the interior planes, edge streams A/B run
// the low/high edge plane ranges, the comm stream the RCCL exchange of the freshly written edges.
// Dependencies (s = this substep, s-1 = the previous one, possibly of the previous RK step):
//   edges_s     need  interior_{s-1}, edges_{s-1}, comm_{s-1}   (they read y's pads)
//   comm_s      needs edges_s
//   interior_s  needs interior_{s-1} (stream order), edges_{s-1}  -- NOT comm_{s-1}: it never reads pads
// so the exchange of substep s-1 overlaps the interior of substep s as well, and the only serial chain
// is comm -> edges -> comm.  hj_slab_join() makes the ctx stream wait for everything outstanding.
int slab_substep(hj_ctx* c, int scheme, int ham, const double* par, int stage, double dt, int rs,
                 const void* y, const void* y0, void* out) {
    int rc;
    if (user_ham_dynamic(ham))
        return fail(HJ_EUNSUPPORTED, "a range-dependent alpha needs the costate range of ALL ranks before a substep: use dist.SlabIntegrator (dynamic=True), not the native stepper");
    const int64_t n = c->N[0];
    hipStream_t main = c->stream;
    const bool talk = (c->lo_rank >= 0 || c->hi_rank >= 0);
    if (scheme == HJ_WENO5) {
        // global max(D1^2) per dim: needs y complete incl. pads -> join first; then all-reduce(MAX)
        if (talk && c->slab_pending) {
            HIP_TRY(hipStreamWaitEvent(main, c->ev_edge, 0));
            HIP_TRY(hipStreamWaitEvent(main, c->ev_comm, 0));
        }
        if ((rc = weno_eps_pass(c, y))) return rc;
        if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
        if (c->comm_size > 1)
            NCCL_TRY(g_rccl.AllReduce(c->weno_vals, c->weno_vals, (size_t)c->ndim,
                                      c->dtype == HJ_F64 ? ncclDouble : ncclFloat, ncclMax, c->comm, main));
        c->weno_src = c->weno_vals;
    }
    const int64_t lo_e = c->halo_lo ? std::min<int64_t>(HJ_STENCIL, n) : 0;
    const int64_t hi_b = c->halo_hi ? std::max<int64_t>(n - HJ_STENCIL, lo_e) : n;
    if (!talk) {
        SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, 0, n};
        s.xp = xp_wanted(c, 0, n);
        rc = do_substep(c, s, -1);
        if (scheme == HJ_WENO5) c->weno_src = nullptr;
        return rc;
    }
    if (c->slab_gated && hi_b > lo_e && c->gate) {
        // Round 4 (HJ_SLAB_SCHEDULE=gated, opt-in): edges_s and interior_s are ONE launch.
        // The edge chunks are its first workgroups (dispatched first, spread over the XCDs); each adds 1 to ctx->gate when its
        // planes are in memory, and the exchange stream waits for that count (hipStreamWaitValue64 on plain device memory:
        // the gated kernel starts 1.5 us after the last publication, profiles/r04_stream_gate_probe.txt) -- no separate 25 us
        // edge launch, no event hop, and the short edge workgroups fill the CUs the interior's last round leaves idle
        // (133 tiles x 2 chunks = 266 workgroups on 256 CUs was 2 rounds for 1.04 rounds of work).
        if (c->slab_pending) HIP_TRY(hipStreamWaitEvent(main, c->ev_comm, 0));       // y's pads: the exchange of substep s-1
        SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, lo_e, hi_b};
        s.gated = true;
        if (lo_e > 0) { s.e0[0] = 0; s.e1[0] = lo_e; }
        if (hi_b < n) { const int w = lo_e > 0 ? 1 : 0; s.e0[w] = hi_b; s.e1[w] = n; }
        if ((rc = do_substep(c, s, -1))) return rc;
        HIP_TRY(hipEventRecord(c->ev_edge, main));            // what hj_slab_join waits for besides the exchange
        if (c->gate_posted > 0) {
            c->gate_count += (unsigned long long)c->gate_posted;
            HIP_TRY(hipStreamWaitValue64(c->comm_stream, c->gate, c->gate_count, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        } else {
            HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_edge, 0));   // the launch could not gate (direct kernel)
        }
        if ((rc = post_halo(c, out, c->comm_stream))) return rc;
        HIP_TRY(hipEventRecord(c->ev_comm, c->comm_stream));
        c->slab_pending = 1;
        if (scheme == HJ_WENO5) c->weno_src = nullptr;
        return HJ_OK;
    }
    if (c->slab_serial) {
        // Round 3 (HJ_SLAB_SCHEDULE=serial, default): edges_s and interior_s share the ctx stream, edges first; only the
        // exchange runs beside them.  The "overlap" schedule below lets interior_s start as soon as interior_{s-1} is done,
        // so that edges_s -- gated by the exchange of substep s-1 -- find every CU taken by interior_s and need 45-60 us
        // for three planes a free GPU computes in ~10: its critical path is comm -> edges -> comm at 110-120 us per substep
        // of a 65-plane slab (profiles/r03_thin_slab_timeline.txt).  Here edges_s run alone right after interior_{s-1}
        // (the exchange of substep s-1 finished long before, under that interior), and the exchange of substep s hides
        // under interior_s: the critical path is edges + interior on one stream.
        if (c->slab_pending) HIP_TRY(hipStreamWaitEvent(main, c->ev_comm, 0));
        {
            SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, 0, 0};
            if (lo_e > 0) { s.p0 = 0; s.p1 = lo_e; if (hi_b < n) { s.q0 = hi_b; s.q1 = n; } }
            else { s.p0 = hi_b; s.p1 = n; }
            if ((rc = do_substep(c, s, -1))) return rc;
        }
        HIP_TRY(hipEventRecord(c->ev_edge, main));
        HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_edge, 0));
        if ((rc = post_halo(c, out, c->comm_stream))) return rc;
        HIP_TRY(hipEventRecord(c->ev_comm, c->comm_stream));
        c->slab_pending = 1;
        if (hi_b > lo_e) {
            SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, lo_e, hi_b};
            s.xp = xp_wanted(c, lo_e, hi_b);
            if ((rc = do_substep(c, s, -1))) return rc;
        }
        if (scheme == HJ_WENO5) c->weno_src = nullptr;
        return HJ_OK;
    }
    // everything launched on main so far (interior_{s-1}, eps pass) gates the edges
    HIP_TRY(hipEventRecord(c->ev_start, main));
    HIP_TRY(hipStreamWaitEvent(c->edge_stream, c->ev_start, 0));
    if (c->slab_pending && c->comm_stream != c->edge_stream) HIP_TRY(hipStreamWaitEvent(c->edge_stream, c->ev_comm, 0));
    // interior_s reads the edges written by substep s-1
    if (c->slab_pending) HIP_TRY(hipStreamWaitEvent(main, c->ev_edge, 0));
    // HJ_SLAB_SCHEDULE=overlap2 (experiment): interior_s also waits for the exchange of substep s-1, so that edges_s (launched
    // first, high-priority stream) find the CUs free instead of taken by an interior launch that started 20-30 us earlier
    if (c->slab_pending && c->slab_overlap2) HIP_TRY(hipStreamWaitEvent(main, c->ev_comm, 0));
    {   // both edge ranges in ONE launch on the edge stream
        SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, 0, 0};
        if (lo_e > 0) { s.p0 = 0; s.p1 = lo_e; if (hi_b < n) { s.q0 = hi_b; s.q1 = n; } }
        else { s.p0 = hi_b; s.p1 = n; }
        s.on_aux = true;
        if ((rc = do_substep(c, s, -1))) return rc;
    }
    HIP_TRY(hipEventRecord(c->ev_edge, c->edge_stream));
    if (c->comm_stream != c->edge_stream) HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_edge, 0));
    if ((rc = post_halo(c, out, c->comm_stream))) return rc;
    HIP_TRY(hipEventRecord(c->ev_comm, c->comm_stream));
    c->slab_pending = 1;
    if (hi_b > lo_e) {
        SubstepCall s{scheme, ham, stage, rs, par, dt, y, y0, out, nullptr, lo_e, hi_b};
        s.xp = xp_wanted(c, lo_e, hi_b);
        if ((rc = do_substep(c, s, -1))) return rc;
    }
    if (scheme == HJ_WENO5) c->weno_src = nullptr;
    return HJ_OK;
}

int slab_streams_create(hj_ctx* c) {
    // the exchange and the edge planes sit on the critical path: highest priority, so their (few)
    // workgroups are dispatched ahead of the interior kernel's when both are ready
    int lo_p = 0, hi_p = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
    // ONE auxiliary stream carries the chain  edges -> exchange -> edges -> ...  in stream order: every
    // cross-stream event hop costs 10-15 us of dispatch latency on this platform
    // (profiles/r01_slab_timeline.txt)
    HIP_TRY(hipStreamCreateWithPriority(&c->edge_stream, hipStreamNonBlocking, hi_p));
    c->comm_stream = c->edge_stream;
    HIP_TRY(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_edge, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_edge2, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreateWithFlags(&c->ev_int[i], hipEventDisableTiming));
    if (!c->gate) {
        // the counter of the gated schedule: plain device memory (signal memory opened the gate 33 us late in the probe)
        int can = 0;
        (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
        if (can) {
            HIP_TRY(hipMalloc((void**)&c->gate, sizeof(unsigned long long)));
            HIP_TRY(hipMemset(c->gate, 0, sizeof(unsigned long long)));
            c->gate_count = 0;
        }
    }
    return HJ_OK;
}

// One odeCFL{1,2,3} step on a slab with ONE exchange per step ("deep halo").  The pads are
// D = 3*order planes deep.  Stage s (1..order) computes, on every side that has a neighbour, 3*(order-s)
// planes BEYOND the slab (redundantly with the neighbour: 3 % more plane updates for RK3 at 201 planes),
// so stage s+1 finds its stencil inputs locally and only the final result is exchanged (D planes).
//   main stream:  interior_s = [3s, n-3s)              needs interior_{s-1} only (stream order)
//   aux stream:   edges_s    = [-3(order-s), 3s) and [n-3s, n+3(order-s))
//                            needs edges_{s-1} (stream order) and interior_{s-1} (event)
//                 after the last stage: exchange of the D edge planes of y_out
// Across steps: interior_1 of the next step needs edges_order of this one (event); edges_1 of the next
// step needs the exchange (stream order).  The two streams never wait for each other inside a step
// except for the (already satisfied) interior events, and the exchange overlaps interior_order of this
// step and interior_1 of the next.  Requires a static dt (all native Hamiltonians) and n >= 6*order.
int slab_step_deep(hj_ctx* c, int order, int scheme, int ham, const double* par, double dt, int rs,
                   const void* cur, void* y_out, void* w0, void* w1) {
    int rc;
    const int64_t n = c->N[0];
    const int W = HJ_STENCIL, D = W * order;
    const bool lo = c->halo_lo, hi = c->halo_hi;
    hipStream_t main = c->stream, aux = c->edge_stream;
    if (!lo && !hi) {   // a single slab without neighbours: plain substeps on the ctx stream
        const void* src1[3] = {cur, w0, w1};
        void* dst1[3] = {order == 1 ? y_out : w0, order == 2 ? y_out : w1, y_out};
        const int kind1[3] = {HJ_STAGE_EULER, order == 2 ? HJ_STAGE_RK2_FULL : HJ_STAGE_RK3_HALF, HJ_STAGE_RK3_FULL};
        for (int st = 1; st <= order; ++st) {
            SubstepCall s{scheme, ham, kind1[st - 1], rs, par, dt, src1[st - 1], st == 1 ? nullptr : cur, dst1[st - 1], nullptr, 0, n};
            s.xp = xp_wanted(c, 0, n);
            if ((rc = do_substep(c, s, -1))) return rc;
        }
        return HJ_OK;
    }
    // (as in slab_substep: every launch of such a Hamiltonian would reduce the range of ITS planes -- a rank-local range, silently different from
    //  the undivided grid's)
    if (user_ham_dynamic(ham))
        return fail(HJ_EUNSUPPORTED, "a range-dependent alpha needs the costate range of ALL ranks before a substep: use dist.SlabIntegrator (dynamic=True), not the native stepper");
    if (!aux) return fail(HJ_ESTATE, "hj_comm_init / hj_comm_init_external has not been called");
    if (c->pad0 < D) return fail(HJ_ESTATE, "axis-0 tables cover %d pad planes, order %d needs %d (hj_ctx_set_axis0_pad)", c->pad0, order, D);
    if (n < 2 * D) return fail(HJ_EUNSUPPORTED, "slab of %lld planes is too thin for the deep-halo stepper (needs %d)", (long long)n, 2 * D);
    // the intended WENO5's epsilon is 1e-6 * max(D1^2) over the WHOLE grid, per stage (upwind_first_weno5a.py:69-70): with the library's own
    // communicator it is all-reduced below; an external transport (hj_comm_init_external) moves planes only -- a rank-local epsilon would
    // differ from the undivided run by ~1e-6 without a word (found by tests/fuzz_slabs.py, round 5)
    if (scheme == HJ_WENO5 && c->external_exchange && c->comm_size > 1)
        return fail(HJ_EUNSUPPORTED, "the intended WENO5 on slabs with an external transport: its epsilon needs an all-reduce per stage "
                                     "(use the library's communicator, or dist.SlabIntegrator, which all-reduces it)");
    const void* src[3] = {cur, w0, w1};
    void* dst[3] = {order == 1 ? y_out : w0, order == 2 ? y_out : w1, y_out};
    const int kind[3] = {HJ_STAGE_EULER, order == 2 ? HJ_STAGE_RK2_FULL : HJ_STAGE_RK3_HALF, HJ_STAGE_RK3_FULL};
    bool edge_signalled = false;
    if (!c->slab_pending) {     // first step after set-up: aux joins whatever main did to the buffers
        HIP_TRY(hipEventRecord(c->ev_start, main));
        HIP_TRY(hipStreamWaitEvent(aux, c->ev_start, 0));
    } else {
        HIP_TRY(hipStreamWaitEvent(main, c->ev_edge, 0));   // edges_order of the previous step
    }
    for (int st = 1; st <= order; ++st) {
        const int ext = W * (order - st);
        const int64_t i0 = lo ? W * st : 0, i1 = hi ? n - W * st : n;
        const void* y0 = st == 1 ? nullptr : cur;
        if (scheme == HJ_WENO5) {
            // global max(D1^2) of this stage's input: needs the edges of the previous stage -> join,
            // reduce over the slab, all-reduce, and hand the result to both streams
            if (st > 1 || c->slab_pending) {
                HIP_TRY(hipEventRecord(c->ev_edge2, aux));
                HIP_TRY(hipStreamWaitEvent(main, c->ev_edge2, 0));
            }
            if ((rc = weno_eps_pass(c, src[st - 1]))) return rc;
            if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
            if (c->comm_size > 1 && c->comm)
                NCCL_TRY(g_rccl.AllReduce(c->weno_vals, c->weno_vals, (size_t)c->ndim,
                                          c->dtype == HJ_F64 ? ncclDouble : ncclFloat, ncclMax, c->comm, main));
            c->weno_src = c->weno_vals;
            HIP_TRY(hipEventRecord(c->ev_start, main));
            HIP_TRY(hipStreamWaitEvent(aux, c->ev_start, 0));
        }
        if (i1 > i0) {
            SubstepCall s{scheme, ham, kind[st - 1], rs, par, dt, src[st - 1], y0, dst[st - 1], nullptr, i0, i1};
            s.xp = xp_wanted(c, i0, i1);
            c->launch_stop = c->ext_events ? c->ev_int[st - 1] : nullptr;
            rc = do_substep(c, s, -1);
            const bool signalled = c->ext_events && c->launch_stop == nullptr;   // consumed by the tiled launch
            c->launch_stop = nullptr;
            if (rc) { c->weno_src = nullptr; return rc; }
            if (!signalled) HIP_TRY(hipEventRecord(c->ev_int[st - 1], main));
        } else {
            HIP_TRY(hipEventRecord(c->ev_int[st - 1], main));
        }
        if (st > 1) HIP_TRY(hipStreamWaitEvent(aux, c->ev_int[st - 2], 0));
        {
            SubstepCall s{scheme, ham, kind[st - 1], rs, par, dt, src[st - 1], y0, dst[st - 1], nullptr, 0, 0};
            if (lo) { s.p0 = -ext; s.p1 = W * st; if (hi) { s.q0 = n - W * st; s.q1 = n + ext; } }
            else { s.p0 = n - W * st; s.p1 = n + ext; }
            s.on_aux = true;
            c->launch_stop = (c->ext_events && st == order) ? c->ev_edge : nullptr;
            rc = do_substep(c, s, -1);
            edge_signalled = c->ext_events && st == order && c->launch_stop == nullptr;
            c->launch_stop = nullptr;
            if (rc) { c->weno_src = nullptr; return rc; }
        }
        if (scheme == HJ_WENO5) c->weno_src = nullptr;
    }
    if (!edge_signalled) HIP_TRY(hipEventRecord(c->ev_edge, aux));
    if ((rc = post_halo(c, y_out, aux, D))) return rc;
    HIP_TRY(hipEventRecord(c->ev_comm, aux));
    c->slab_pending = 1;
    return HJ_OK;
}

int slab_join(hj_ctx* c) {
    if (c->slab_pending) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_edge, 0));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        c->slab_pending = 0;
    }
    return HJ_OK;
}

template <typename T>
static int split_end_launch(hj_ctx* c, const void* const* dL, const void* const* dR, const void* const* alpha,
                            const double* alpha_s, const void* ham, void* out, unsigned long long* keys) {
    SplitEndArgs<T> A;
    memset(&A, 0, sizeof(A));
    for (int d = 0; d < c->ndim; ++d) {
        A.dL[d] = (const T*)dL[d];
        A.dR[d] = (const T*)dR[d];
        A.alpha[d] = alpha ? (const T*)alpha[d] : nullptr;
        A.alpha_s[d] = alpha_s ? (T)alpha_s[d] : T(0);
    }
    A.ham = (const T*)ham;
    A.out = (T*)out;
    A.keys = keys;
    A.n = c->total;
    A.nd = c->ndim;
    const int blocks = (int)std::min<int64_t>((c->total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL((lf_split_end_kernel<T>), dim3(blocks), dim3(256), 0, c->stream, A);
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}


// =========================================================================================== ABI
extern "C" {

const char* hj_last_error(void) { return hjh::g_err; }
const char* hj_last_kernel(hj_ctx* c) { return c ? c->last_kernel : ""; }
int hj_last_launch(hj_ctx* c, int* nbuf, int* ahead) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    const bool pair = strcmp(c->last_kernel, "fused_pair_kernel") == 0;
    if (nbuf) *nbuf = pair ? c->last_nbuf : 2;
    if (ahead) *ahead = (pair && c->last_nbuf > 2) ? c->last_nbuf - 2 : 0;
    return HJ_OK;
}
int hj_last_tile(hj_ctx* c, int* e) {
    if (!c || !e) return fail(HJ_EINVAL, "null argument");
    for (int d = 0; d < HJ_MAX_DIM; ++d) e[d] = c->last_E[d];
    return HJ_OK;
}
const char* hj_version(void) { return "hj_mi355x 0.1 (gfx950)"; }

// dry_cus > 0: a host-only context for hj_plan_substep (no HIP call at all; the GPU is taken to have dry_cus compute units)
static int ctx_create_impl(hj_ctx** out, int ndim, const int64_t* N, const double* xmin, const double* dx,
                           const int* bc, const int* toward_zero, int dtype, int device, int dry_cus) {
    if (!out || !N || !xmin || !dx || !bc) return fail(HJ_EINVAL, "null argument");
    if (ndim < 2 || ndim > HJ_MAX_DIM) return fail(HJ_EUNSUPPORTED, "grid.dim must be 2..4, got %d", ndim);
    if (dtype != HJ_F64 && dtype != HJ_F32) return fail(HJ_EINVAL, "unknown dtype %d", dtype);
    int64_t total = 1;
    for (int d = 0; d < ndim; ++d) {
        if (N[d] < 1) return fail(HJ_EINVAL, "number of grid cells must be strictly positive");
        if (!(dx[d] > 0)) return fail(HJ_EINVAL, "grid cell size dx must be strictly positive");
        if (bc[d] != HJ_BC_EXTRAPOLATE && bc[d] != HJ_BC_PERIODIC) return fail(HJ_EINVAL, "unknown bc %d", bc[d]);
        total *= N[d];
    }
    if (total / N[0] >= (1ll << 31)) return fail(HJ_EUNSUPPORTED, "axis-0 plane exceeds 2^31 cells");
    if (!dry_cus) HIP_TRY(hipSetDevice(device));
    hj_ctx* c = new hj_ctx();
    c->ndim = ndim; c->dtype = dtype; c->device = device;
    c->dry = dry_cus > 0;
    c->esz = dtype == HJ_F64 ? 8 : 4;
    c->total = total;
    c->halo_lo = c->halo_hi = 0;
    c->stream = nullptr;
    for (int d = 0; d < HJ_MAX_DIM; ++d) { c->coord[d] = nullptr; c->N[d] = 1; c->bc[d] = 0; c->tz[d] = 0; c->dx[d] = 1; c->xmin[d] = 0; }
    for (int s = 0; s < 4; ++s) { c->aux[s] = nullptr; c->aux_n[s] = 0; }
    c->ring = nullptr; c->keys = nullptr; c->weno_vals = nullptr; c->weno_src = nullptr; c->flag = nullptr;
    c->partials = nullptr; c->partials_cap = 0;
    c->sb_valid = false; c->internal_slot = 0; c->diss_local = 0; c->sb_local = 0; c->post_step_op = 0;
    c->post_arr[0] = c->post_arr[1] = nullptr; c->post_arr_op[0] = c->post_arr_op[1] = 0;
    c->comm = nullptr; c->comm_rank = 0; c->comm_size = 1; c->lo_rank = c->hi_rank = -1;
    c->comm_stream = c->edge_stream = nullptr;
    c->ev_start = c->ev_edge = c->ev_edge2 = c->ev_comm = nullptr; c->slab_pending = 0;
    c->ev_int[0] = c->ev_int[1] = c->ev_int[2] = nullptr;
    c->launch_stop = nullptr; c->ext_events = env_int("HJ_EXT_EVENTS", 1);
    c->external_exchange = 0; c->pad0 = 0; c->coord0_ext = nullptr; c->aux_ext[0] = c->aux_ext[1] = nullptr;
    for (int d = 0; d < ndim; ++d) {
        c->N[d] = N[d]; c->xmin[d] = xmin[d]; c->dx[d] = dx[d]; c->bc[d] = bc[d];
        c->tz[d] = toward_zero ? (toward_zero[d] != 0) : 0;
    }
    c->cfg_from_env = (getenv("HJ_NT") != nullptr) || (getenv("HJ_R") != nullptr);
    c->cfg.NT = env_int("HJ_NT", 512);   // defaults: best of the round-1 sweep (profiles/r01_cfgsweep.txt)
    c->cfg.R = env_int("HJ_R", 4);
    c->cfg.KH = 0;
    c->force_direct = env_int("HJ_FORCE_DIRECT", 0);
    c->debug = env_int("HJ_DEBUG", 0);
    c->debug_xp = c->debug;
    c->full_rows = env_int("HJ_FULL_ROWS", 0);
    // bank-friendly LDS row pitch: -1 (default) the pair kernel only (pitch in pairs congruent to the row length mod 16:
    // +1 % at 201^3, +1-1.5 % for ENO3 / intended WENO5, nothing lost elsewhere, round 3), 0 never, 1 both kernels (the
    // one-cell-per-lane kernel gained +0.5 % only in round 1)
    c->lds_pad = env_int("HJ_LDS_PAD", -1);
    c->target_blocks = env_int("HJ_TARGET_BLOCKS", 0);   // 0 = choose from the GPU's capacity
    {
        int ncu = dry_cus;
        if (!dry_cus && (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu < 1)) ncu = 256;
        c->num_cus = ncu;
    }
    // shortest chunk a workgroup may get.  A launch never lasts less than ~9 us (kernel-argument, table and queue
    // round trips, one plane, the reduction's atomics: measured with 1-plane chunks at 51^3), so below ~0.6 M
    // cells shorter chunks on more CUs win a little (51^3: 66 -> 102 workgroups, +12 %); above, 4 planes
    c->min_chunk = std::max(1, env_int("HJ_MIN_CHUNK", total < 600000 ? 3 : 4));
    // planes of loads a chunk pays before its first result: 2*HJ_STENCIL = 6 (HJ_WARMUP_COST: sweep knob)
    c->warmup_cost = std::max(0, env_int("HJ_WARMUP_COST", 2 * HJ_STENCIL));
    c->lds_limit = (size_t)env_int("HJ_LDS_LIMIT", 64 * 1024);
    c->no_plain = env_int("HJ_NO_PLAIN", 0);
    c->fuse12 = env_int("HJ_FUSE12", 0);          // 0 never (default: measured slower), 1 whenever a tiling exists, -1 by grid size
    c->f12_nt = env_int("HJ_F12_NT", 0);
    c->f12_r = env_int("HJ_F12_R", 0);
    c->f12_kh = env_int("HJ_F12_KH", 0);
    c->f12_warm = env_int("HJ_F12_WARM", 9);
    c->f12_e2 = env_int("HJ_F12_E2", 0);
    c->tile_cells = env_int("HJ_TILE_CELLS", 0);
    // chunks of a tile column marching pairwise in opposite directions (hj_fusedv.h, "PAIRED CHUNKS"): only in -DHJ_MAYDOWN=1 builds
    // (hj_fused.h: measured in round 4 and compiled out -- the support costs every launch more than pairing gains); there -1 (default)
    // = on for 3-D grids below 10 M cells, 0 never, 1 always.  Ignored by the default build.
    c->pair_dirs = env_int("HJ_PAIR_DIRS", -1);
    if (c->pair_dirs < 0) c->pair_dirs = (ndim == 3 && total < 10000000) ? 1 : 0;
    c->term_tiled_from = (long long)env_int("HJ_TERM_TILED_FROM", 1000000);     // cells; negative: never (the direct term_kernel everywhere)
    c->tile_block[0] = env_int("HJ_TB1", 4);      // 4 x 4 positions x the 4 axis-3 tiles = the 64 workgroups an XCD holds (C5)
    c->tile_block[1] = env_int("HJ_TB2", 4);
    // launch-time choice of the tile shape on grids of >= HJ_AUTOTUNE_MIN_CELLS cells (hj_inst.hip, tuned_tiling)
    c->autotune = env_int("HJ_AUTOTUNE", 1);
    c->autotune_min_cells = (long long)env_int("HJ_AUTOTUNE_MIN_MCELLS", 40) * 1000000ll;
    c->autotune_passes = std::max(2, env_int("HJ_AUTOTUNE_PASSES", 6));
    // 3-D grids below ~52^3 run the direct kernel (round 5, same-box A/B at 31^3 ... 101^3, profiles/r05_small_grids.txt: 16.2 against 23.9 us per
    // RK3 step at 31^3, 18.2 / 24.0 at 41^3, 21.5 / 22.9 at 51^3, 30.6 / 24.9 at 65^3: a tiled launch cannot go below ~7.7 us -- 4 us of
    // prologue in front of a dozen plane iterations --, a launch of one thread per cell with every stencil load issued at once can); bitwise
    // the same results.  The test suite sets HJ_DIRECT_BELOW=0: its small grids are there to exercise the tiled kernels.
    c->direct_below = env_int("HJ_DIRECT_BELOW", ndim == 3 ? 140000 : 0);
    c->f12_pair = env_int("HJ_F12_PAIR", 1);
    c->f12_e1 = env_int("HJ_F12_E1", 0);
    c->pair = env_int("HJ_PAIR", 1);          // two-cells-per-lane kernel on grids of >= 6.5 M cells (light stencils) / 2.5 M (heavy stencils, fp32 4-D) (0: scalar kernel everywhere, 2: pair kernel whatever the size)
    c->tile4_sel = env_int("HJ_TILE4_SEL", -1);
    c->pair4 = env_int("HJ_PAIR4", 1);        // 4-D fp32 light stencils: the compile-time-tile kernel (hj_fused4v.h); 0: the generic pair kernel
    c->flat4 = env_int("HJ_FLAT4", 1);
    c->flat4_sel = env_int("HJ_FLAT4_SEL", -1);
    c->coop = env_int("HJ_COOP", 0);       // opt-in: measured SLOWER than the stage launches above ~40^3 (profiles/r06_small_grids.txt)
    c->pair_nt = env_int("HJ_PAIR_NT", 0);
    c->pair_r = env_int("HJ_PAIR_R", 0);
    c->pair_kh = env_int("HJ_PAIR_KH", 0);
    c->pair_occ = env_int("HJ_PAIR_OCC", 0);
    // planes the ring is parked ahead: 3 aligns it exactly with the neighbours' own loads (best from 251^3 up); on the
    // 201^3-class grids 2 is 0.7 % faster (one plane less to fetch synchronously in the setup; same-box A/B r02_run43.sh)
    c->pair_ah = std::max(1, std::min(3, env_int("HJ_PAIR_AH", c->total < 12000000 ? 2 : 3)));
    c->xp_mode = std::max(0, std::min(2, env_int("HJ_XP", 1)));          // transposed march (hj_instx.hip): 0 never, 1 thin slab windows, 2 wherever it exists
    c->xp_max_planes = env_int("HJ_XP_MAX_PLANES", 0);
    c->xp_trials = std::max(0, std::min(16, env_int("HJ_XP_TRIALS", 2)));
    c->lds_pitch_add = env_int("HJ_LDS_PITCH_ADD", 0) & ~1;
    {
        // per-substep slab schedule: "serial" or "overlap" (rounds 1-2); default by slab thickness -- on the single-GPU self
        // ring (profiles/r03_slab_schedule_ab.txt) serial wins on the 257-plane slab (+4 %), ties at 129 planes and loses on
        // the 65-plane slab (-13 %: the exchange kernel finds every CU's LDS taken by the interior and starts late)
        const char* sch = getenv("HJ_SLAB_SCHEDULE");
        c->slab_serial = sch ? !strcmp(sch, "serial") : (N[0] >= 192);
        c->slab_overlap2 = sch && !strcmp(sch, "overlap2");
        // round 4: "gated" -- edges and interior in ONE launch, the exchange gated on a counter the edge workgroups publish.
        // Opt-in: on the self ring it wins on the 257-plane slab (0.503 against 0.492 serial / 0.481 overlap) and loses on the
        // 65-plane one (0.338 against 0.379 overlap): there the launch itself is the problem -- 133 tiles x 3 chunks of 20 + 6
        // warm-up planes plus 266 edge workgroups of 3 + 6 planes are 96 plane loads per tile column for 65 planes of output
        // (profiles/r04_thin_slab_gated.txt)
        c->slab_gated = sch && !strcmp(sch, "gated");
    }
    c->eps_fuse_min_cells = (long long)env_int("HJ_EPS_FUSE_MIN_CELLS", 2000000);
    c->eps_fuse = env_int("HJ_EPS_FUSE", 1);         // 0: the intended WENO5 always runs its two-launch epsilon pre-pass
    {
        const char* td = getenv("HJ_TIMING_DUMP");
        if (td && *td) { c->timing_dump_path = td; c->timing_dump = c->timing_dump_path.c_str(); }
    }
    c->keep_bounds = env_int("HJ_KEEP_BOUNDS", 0);   // 1: every launch reduces its CFL bound, read or not (round-2 behaviour; A/B)
    c->pair_ring = env_int("HJ_PAIR_RING", -1);  // halo ring parked in LDS 3 planes ahead: 0 never, 1 always, -1 (default) with the (512,2) configuration
    c->cfg.KH = cfg_kh(ndim, c->cfg.NT, c->cfg.R);
    if (getenv("HJ_KH")) c->cfg.KH = env_int("HJ_KH", c->cfg.KH);
    c->pd = env_int("HJ_PD", 2);
    c->occ_hint = env_int("HJ_OCC", -1);
    if (c->occ_hint < 0) {   // default waves/SIMD hint = first table entry of this (NT, R, PD)
#define X(NT_, R_, KH_, OCC_, PD_) if (c->occ_hint < 0 && c->cfg.NT == NT_ && c->cfg.R == R_ && c->pd == PD_) c->occ_hint = OCC_;
        if (ndim == 4) { HJ_CONFIGS_4D(X) }
        else { HJ_CONFIGS(X) }
#undef X
    }
    if (c->cfg.KH < 0 && c->cfg_from_env) {
        const int nt = c->cfg.NT, r = c->cfg.R;
        delete c;
        return fail(HJ_EINVAL, "unsupported HJ_NT/HJ_R combination %d/%d", nt, r);
    }
    *out = c;
    if (c->dry) return HJ_OK;
    int rc = HJ_OK;
    auto bail = [&](int code) { hj_ctx_destroy(c); *out = nullptr; return code; };
    for (int d = 0; d < ndim; ++d) {
        c->coord_host[d].resize((size_t)N[d]);
        for (int64_t i = 0; i < N[d]; ++i) c->coord_host[d][(size_t)i] = xmin[d] + (double)i * dx[d];
        if ((rc = upload(c, &c->coord[d], c->coord_host[d].data(), N[d]))) return bail(rc);
    }
    if ((rc = default_aux(c))) return bail(rc);
    hipError_t e;
    if ((e = hipMalloc((void**)&c->ring, sizeof(unsigned long long) * RING_SLOTS * HJ_MAX_DIM)) != hipSuccess ||
        (e = hipMalloc((void**)&c->keys, sizeof(unsigned long long) * 40)) != hipSuccess ||
        (e = hipMalloc(&c->weno_vals, 8 * HJ_MAX_DIM)) != hipSuccess ||
        (e = hipMalloc((void**)&c->flag, sizeof(int))) != hipSuccess)
        return bail(fail(HJ_EHIP, "hipMalloc: %s", hipGetErrorString(e)));
    c->ring_pos = RING_SLOTS;  // forces the first use to zero the ring
    for (int i = 0; i < HJ_BOUND_SLOTS; ++i) c->slot_ring[i] = -1;
    return HJ_OK;
}

int hj_ctx_create(hj_ctx** out, int ndim, const int64_t* N, const double* xmin, const double* dx,
                  const int* bc, const int* toward_zero, int dtype, int device) {
    return ctx_create_impl(out, ndim, N, xmin, dx, bc, toward_zero, dtype, device, 0);
}

// The launch plan of one substep over planes [p0, p1) of an N-cell grid, made WITHOUT a device (bench.py --plan-only: what every rank of
// an N-GPU run will launch, before the node is there).  Same code as the real launch up to the point where the kernel would be enqueued.
// out[12] = {threads per workgroup, workgroups, tiles, chunks, chunk length (planes), tile extents E1..E3, LDS bytes, workgroups per CU,
// slab pads honoured (1/0), 0}; kernel_name (cap bytes) receives the kernel's name.  Built-in Hamiltonians only.
int hj_plan_substep(int ndim, const int64_t* N, const int* bc, int dtype, int scheme, int ham, int stage, int64_t p0, int64_t p1,
                    int halo_lo, int halo_hi, int num_cus, int64_t* out, char* kernel_name, int cap) {
    if (!N || !bc || !out) return fail(HJ_EINVAL, "null argument");
    if (ham_ndim(ham) != ndim || user_ham_valid(ham)) return fail(HJ_EUNSUPPORTED, "hj_plan_substep plans the built-in Hamiltonians (id %d, dim %d)", ham, ndim);
    if (scheme < HJ_ENO2 || scheme > HJ_ENO3_FAST) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    double xmin[HJ_MAX_DIM] = {0, 0, 0, 0}, dx[HJ_MAX_DIM] = {1, 1, 1, 1};
    hj_ctx* c = nullptr;
    int rc = ctx_create_impl(&c, ndim, N, xmin, dx, bc, nullptr, dtype, -1, num_cus > 0 ? num_cus : 256);
    if (rc) return rc;
    c->halo_lo = halo_lo != 0; c->halo_hi = halo_hi != 0;
    const double par[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    SubstepCall s{};
    s.scheme = scheme; s.ham = ham; s.stage = stage; s.restrict_sign = 0; s.par = par; s.dt = 0.0;
    s.y = s.y0 = nullptr; s.out = nullptr; s.bound = nullptr; s.p0 = p0; s.p1 = p1;
    s.xp = xp_wanted(c, p0, p1);            // the launch form the library would take for this range (thin ranges without neighbours: the transposed march)
    rc = c->dtype == HJ_F64 ? launch_ham<double>(c, s) : launch_ham<float>(c, s);
    if (rc == HJ_OK) {
        out[0] = c->last_plan.threads; out[1] = c->last_plan.nblocks; out[2] = c->last_plan.ntiles; out[3] = c->last_plan.nchunks;
        out[4] = c->last_E[0]; out[5] = c->last_E[1]; out[6] = c->last_E[2]; out[7] = c->last_E[3];
        out[8] = (int64_t)c->last_plan.lds_bytes; out[9] = c->last_plan.wg_per_cu; out[10] = (c->halo_lo || c->halo_hi) ? 1 : 0; out[11] = 0;
        if (kernel_name && cap > 0) { strncpy(kernel_name, c->last_kernel, (size_t)cap - 1); kernel_name[cap - 1] = 0; }
    }
    delete c;
    return rc;
}

void hj_ctx_destroy(hj_ctx* c) {
    if (c && c->range_ring) (void)hipFree(c->range_ring);
    if (c && c->host_words) (void)hipHostFree(c->host_words);
    if (c && c->dt_dev) (void)hipFree(c->dt_dev);
    if (c && c->alpha_part) (void)hipFree(c->alpha_part);
    if (!c) return;
    (void)hj_comm_destroy(c);
    for (int d = 0; d < HJ_MAX_DIM; ++d) if (c->coord[d]) (void)hipFree(c->coord[d]);
    for (int s = 0; s < 4; ++s) if (c->aux[s]) (void)hipFree(c->aux[s]);
    if (c->coord0_ext) (void)hipFree(c->coord0_ext);
    for (int s = 0; s < 2; ++s) if (c->aux_ext[s]) (void)hipFree(c->aux_ext[s]);
    if (c->ring) (void)hipFree(c->ring);
    if (c->ring_keep) (void)hipFree(c->ring_keep);
    if (c->keys) (void)hipFree(c->keys);
    if (c->weno_vals) (void)hipFree(c->weno_vals);
    if (c->flag) (void)hipFree(c->flag);
    if (c->partials) (void)hipFree(c->partials);
    for (int i = 0; i < 2; ++i) if (c->tune_ev[i]) (void)hipEventDestroy(c->tune_ev[i]);
    if (c->ev_bounds) (void)hipEventDestroy(c->ev_bounds);
    for (auto& kv : c->xp_choice) for (int i = 0; i < 4; ++i) if (kv.second.ev[i / 2][i % 2]) (void)hipEventDestroy(kv.second.ev[i / 2][i % 2]);
    if (c->coop_sync) (void)hipFree(c->coop_sync);
    if (c->eps_prod) (void)hipFree(c->eps_prod);
    if (c->eps_rows) (void)hipFree(c->eps_rows);
    delete c;
}

int hj_ctx_set_stream(hj_ctx* c, void* s) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    c->stream = (hipStream_t)s;
    ++c->state_gen;
    return HJ_OK;
}

int hj_ctx_set_coords(hj_ctx* c, int dim, const double* vs) {
    if (!c || !vs) return fail(HJ_EINVAL, "null argument");
    if (dim < 0 || dim >= c->ndim) return fail(HJ_EINVAL, "Illegal dim parameter");
    c->coord_host[dim].assign(vs, vs + c->N[dim]);
    c->sb_valid = false;
    int rc = upload(c, &c->coord[dim], vs, c->N[dim]);
    if (rc) return rc;
    return default_aux(c);
}

int hj_ctx_set_aux(hj_ctx* c, int slot, const double* tab, int64_t n) {
    if (!c || !tab) return fail(HJ_EINVAL, "null argument");
    if (slot < 0 || slot >= 4 || n <= 0) return fail(HJ_EINVAL, "bad aux slot/size");
    c->aux_n[slot] = n;
    c->sb_valid = false;
    return upload(c, &c->aux[slot], tab, n);
}

int hj_ctx_set_slab(hj_ctx* c, int lo, int hi) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    c->halo_lo = lo != 0;
    c->halo_hi = hi != 0;
    return HJ_OK;
}

int hj_ghost(hj_ctx* c, int dim, int width, const void* in, void* out) {
    if (!c || !in || !out) return fail(HJ_EINVAL, "null argument");
    if (dim < 0 || dim >= c->ndim) return fail(HJ_EINVAL, "Illegal dim parameter");
    if (width <= 0 || width > c->N[dim]) return fail(HJ_EINVAL, "Illegal width parameter");
    if (c->bc[dim] == HJ_BC_EXTRAPOLATE && c->N[dim] < 2) return fail(HJ_EINVAL, "extrapolation needs two nodes");
    const long long total = c->total / c->N[dim] * (c->N[dim] + 2 * width);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    if (c->dtype == HJ_F64)
        hipLaunchKernelGGL((ghost_kernel<double>), dim3(blocks), dim3(256), 0, c->stream, (const double*)in, (double*)out, make_view<double>(c, dim, false), width);
    else
        hipLaunchKernelGGL((ghost_kernel<float>), dim3(blocks), dim3(256), 0, c->stream, (const float*)in, (float*)out, make_view<float>(c, dim, false), width);
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

int hj_upwind(hj_ctx* c, int scheme, int dim, const void* phi, void* dL, void* dR, double* mm) {
    if (!c || !phi || !dL || !dR) return fail(HJ_EINVAL, "null argument");
    if (dim < 0 || dim >= c->ndim) return fail(HJ_EINVAL, "Illegal dim parameter");
    scheme = base_scheme(scheme);          // (the fast ENO variants exist for the substep kernels only)
    if (scheme < 0 || scheme > 3) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    if (c->N[dim] < HJ_STENCIL) return fail(HJ_EINVAL, "grid too small along dim %d (N=%lld)", dim, (long long)c->N[dim]);
    int rc;
    const void* epsv = c->weno_src ? c->weno_src : c->weno_vals;
    if (scheme == HJ_WENO5 && !c->weno_src) {
        if ((rc = weno_eps_pass(c, phi))) return rc;
        if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
    }
    unsigned long long* keys = nullptr;
    if (mm) {
        keys = c->keys + 32;
        HIP_TRY(hipMemsetAsync(keys, 0, 4 * sizeof(unsigned long long), c->stream));
    }
    rc = c->dtype == HJ_F64 ? upwind_launch<double>(c, scheme, dim, phi, dL, dR, keys, (const double*)epsv)
                            : upwind_launch<float>(c, scheme, dim, phi, dL, dR, keys, (const float*)epsv);
    if (rc) return rc;
    if (mm) {
        unsigned long long k[4];
        HIP_TRY(hipMemcpyAsync(k, keys, sizeof(k), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        mm[0] = -key_to_double(k[0]); mm[1] = key_to_double(k[1]);
        mm[2] = -key_to_double(k[2]); mm[3] = key_to_double(k[3]);
    }
    return HJ_OK;
}

int hj_lf_split_begin(hj_ctx* c, int scheme, const void* y, void* const* dL, void* const* dR, double* mm) {
    if (!c || !y || !dL || !dR) return fail(HJ_EINVAL, "null argument");
    scheme = base_scheme(scheme);          // (the fast ENO variants exist for the substep kernels only)
    if (scheme < 0 || scheme > 3) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    for (int d = 0; d < c->ndim; ++d) {
        if (!dL[d] || !dR[d]) return fail(HJ_EINVAL, "null derivative array for dim %d", d);
        if (c->N[d] < HJ_STENCIL) return fail(HJ_EINVAL, "grid too small along dim %d (N=%lld)", d, (long long)c->N[d]);
    }
    int rc;
    const void* epsv = c->weno_src ? c->weno_src : c->weno_vals;
    if (scheme == HJ_WENO5 && !c->weno_src) {
        if ((rc = weno_eps_pass(c, y))) return rc;
        if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
    }
    // Round 4: ONE launch for all dimensions (upwind_all_kernel: the stencils of a cell gathered once); ONE host synchronisation
    // fetches the 4*ndim reductions.  HJ_UPWIND_ALL=0: one launch per dimension (rounds 1-3; A/B)
    unsigned long long* keys = c->keys + 8;        // [8, 8 + 4*HJ_MAX_DIM)
    if (mm) HIP_TRY(hipMemsetAsync(keys, 0, 4 * HJ_MAX_DIM * sizeof(unsigned long long), c->stream));
    static const int one_launch = env_int("HJ_UPWIND_ALL", 1);
    if (one_launch) {
        rc = c->dtype == HJ_F64 ? upwind_all_dispatch<double>(c, scheme, y, dL, dR, mm ? keys : nullptr, (const double*)epsv)
                                : upwind_all_dispatch<float>(c, scheme, y, dL, dR, mm ? keys : nullptr, (const float*)epsv);
        if (rc) return rc;
    } else
    for (int d = 0; d < c->ndim; ++d) {
        unsigned long long* k = mm ? keys + 4 * d : nullptr;
        rc = c->dtype == HJ_F64 ? upwind_launch<double>(c, scheme, d, y, dL[d], dR[d], k, (const double*)epsv)
                                : upwind_launch<float>(c, scheme, d, y, dL[d], dR[d], k, (const float*)epsv);
        if (rc) return rc;
    }
    if (mm) {
        unsigned long long k[4 * HJ_MAX_DIM];
        HIP_TRY(hipMemcpyAsync(k, keys, sizeof(k), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int d = 0; d < c->ndim; ++d) {
            mm[4 * d + 0] = -key_to_double(k[4 * d + 0]); mm[4 * d + 1] = key_to_double(k[4 * d + 1]);
            mm[4 * d + 2] = -key_to_double(k[4 * d + 2]); mm[4 * d + 3] = key_to_double(k[4 * d + 3]);
        }
    }
    return HJ_OK;
}

int hj_lf_split_end(hj_ctx* c, const void* const* dL, const void* const* dR, const void* const* alpha,
                    const double* alpha_s, const void* ham, void* out, double* sb, double* amax) {
    if (!c || !dL || !dR || !out) return fail(HJ_EINVAL, "null argument");
    bool any_arr = false;
    for (int d = 0; d < c->ndim; ++d) {
        if (!dL[d] || !dR[d]) return fail(HJ_EINVAL, "null derivative array for dim %d", d);
        if (!(alpha && alpha[d]) && !alpha_s) return fail(HJ_EINVAL, "alpha of dim %d is neither an array nor a scalar", d);
        any_arr = any_arr || (alpha && alpha[d]);
    }
    unsigned long long* keys = c->keys + 24;       // [24, 28)
    if (any_arr) HIP_TRY(hipMemsetAsync(keys, 0, HJ_MAX_DIM * sizeof(unsigned long long), c->stream));
    int rc = c->dtype == HJ_F64 ? split_end_launch<double>(c, dL, dR, alpha, alpha_s, ham, out, keys)
                                : split_end_launch<float>(c, dL, dR, alpha, alpha_s, ham, out, keys);
    if (rc) return rc;
    if (sb || amax) {
        unsigned long long k[HJ_MAX_DIM] = {0, 0, 0, 0};
        if (any_arr) {
            HIP_TRY(hipMemcpyAsync(k, keys, sizeof(k), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        // stepBound = 1 / sum_d max(alpha_d)/dx_d   (artificial_diss_glf.py:107-109)
        double inv = 0.0;
        for (int d = 0; d < c->ndim; ++d) {
            double a;
            if (alpha && alpha[d]) {
                if (k[d] == 0) return fail(HJ_ESTATE, "no reduction for the alpha array of dim %d", d);
                a = key_to_double(k[d]);
            } else {
                a = alpha_s[d];
            }
            if (amax) amax[d] = a;
            inv += a / c->dx[d];
        }
        if (sb) *sb = 1.0 / inv;
    }
    return HJ_OK;
}

// ---- termNormal / termReinit / termConvection: one launch each (hj_terms.h)
extern "C++" {
namespace {
template <typename T>
void fill_term_par(const hj_ctx* c, const void* const* arr, const double* scal, int order, TermPar<T>& P) {
    double mdx = 0.0;
    for (int d = 0; d < HJ_MAX_DIM; ++d) {
        const bool in = d < c->ndim;
        P.arr[d] = (in && arr) ? (const T*)arr[d] : nullptr;
        P.scal[d] = (in && scal) ? (T)scal[d] : T(0);
        P.dx_inv[d] = in ? (T)(1.0 / c->dx[d]) : T(0);
        P.dx[d] = in ? (T)c->dx[d] : T(1);
        if (in) mdx = std::max(mdx, c->dx[d]);
    }
    P.max_dx = (T)mdx;
    P.subcell_order = order;
    const double e = (double)std::numeric_limits<double>::epsilon();
    P.small2 = (T)((1e6 * e) * (1e6 * e));     // robust_small_epsilon^2, term_reinit.py:128,274
    P.tiny = (T)e;                             // term_reinit.py:206
}
template <typename T, int ND>
int term_launch_nd(hj_ctx* c, int kind, int scheme, const void* y, const void* const* arr, const double* scal, int order,
                   void* out, unsigned long long* keys) {
    TermArgs<T, ND> A;
    memset(&A, 0, sizeof(A));
    A.y = (const T*)y;
    A.out = (T*)out;
    hjh::fill_grid<T, ND>(c, A.G);
    A.max_d1sq = (const T*)(c->weno_src ? c->weno_src : c->weno_vals);
    fill_term_par<T>(c, arr, scal, order, A.P);
    A.keys = keys;
    const int blocks = (int)std::min<int64_t>((c->total + 255) / 256, 256 * 16);
#define HJ_TK(S, K) hipLaunchKernelGGL((term_kernel<T, ND, S, K>), dim3(blocks), dim3(256), 0, c->stream, A)
#define HJ_TS(K)                                                                       \
    switch (scheme) {                                                                  \
        case HJ_ENO2: HJ_TK(HJ_ENO2, K); break;                                        \
        case HJ_ENO3: HJ_TK(HJ_ENO3, K); break;                                        \
        case HJ_WENO5: HJ_TK(HJ_WENO5, K); break;                                      \
        default: HJ_TK(HJ_WENO5_ASSHIPPED, K); break;                                  \
    }
    if (kind == HJ_TERM_NORMAL) { HJ_TS(HJ_TERM_NORMAL) }
    else if (kind == HJ_TERM_REINIT) { HJ_TS(HJ_TERM_REINIT) }
    else { HJ_TS(HJ_TERM_CONVECTION) }
#undef HJ_TS
#undef HJ_TK
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

// launches the term and fetches its ND + 1 maxima (one host synchronisation): k[d] < 0 = no cell contributed
int term_run(hj_ctx* c, int kind, int scheme, const void* y, const void* const* arr, const double* scal, int order,
             void* out, double* k) {
    if (!c || !y || !out) return fail(HJ_EINVAL, "null argument");
    scheme = base_scheme(scheme);          // (the fast ENO variants exist for the substep kernels only)
    if (scheme < 0 || scheme > 3) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    if (y == out) return fail(HJ_EINVAL, "out must not alias the stencil input y");
    for (int d = 0; d < c->ndim; ++d)
        if (c->N[d] < HJ_STENCIL) return fail(HJ_EINVAL, "grid too small along dim %d (N=%lld)", d, (long long)c->N[d]);
    int rc;
    if (scheme == HJ_WENO5 && !c->weno_src) {
        if ((rc = weno_eps_pass(c, y))) return rc;
        if ((rc = keys_to_vals(c, c->weno_vals))) return rc;
    }
    unsigned long long* keys = c->keys + 8;        // scratch keys of the split path: [8, 8 + HJ_MAX_DIM + 1)
    HIP_TRY(hipMemsetAsync(keys, 0, (HJ_MAX_DIM + 1) * sizeof(unsigned long long), c->stream));
    // Round 4: on fp64 2-D / 3-D grids of HJ_TERM_TILED_FROM cells or more (default 1 M) the term runs through the tiled substep
    // kernel (LDS-staged stencils, register queue along axis 0; TermOp of hj_termop.h) -- the same cell arithmetic as the direct
    // term_kernel (term_cell), 3-5x its speed at 201^3.  Slabs, fp32, 4-D and small grids keep the direct kernel.
    const bool tiled = c->dtype == HJ_F64 && (c->ndim == 2 || c->ndim == 3) && !c->halo_lo && !c->halo_hi && !c->force_direct &&
                       c->term_tiled_from >= 0 && c->total >= c->term_tiled_from;
    bool tiled_ran = false;
    if (tiled) {
        TermPar<double> P;
        fill_term_par<double>(c, arr, scal, order, P);
        SubstepCall s{scheme, 0, HJ_STAGE_YDOT, 0, nullptr, 0.0, y, arr ? arr[0] : nullptr, out, keys, 0, c->N[0]};
        s.term = &P;
        rc = c->ndim == 2 ? launch_term_tiled<double, 2>(c, kind, s) : launch_term_tiled<double, 3>(c, kind, s);
        tiled_ran = rc != HJ_EUNSUPPORTED;         // a grid the tiled kernel has no tiling for: the direct kernel below
    }
    if (!tiled_ran) {
#define HJ_TN(T_, ND_) rc = term_launch_nd<T_, ND_>(c, kind, scheme, y, arr, scal, order, out, keys)
        if (c->dtype == HJ_F64) { if (c->ndim == 2) HJ_TN(double, 2); else if (c->ndim == 3) HJ_TN(double, 3); else HJ_TN(double, 4); }
        else { if (c->ndim == 2) HJ_TN(float, 2); else if (c->ndim == 3) HJ_TN(float, 3); else HJ_TN(float, 4); }
#undef HJ_TN
        c->last_kernel = "term_kernel";
    }
    if (rc) return rc;
    unsigned long long h[HJ_MAX_DIM + 1];
    HIP_TRY(hipMemcpyAsync(h, keys, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int d = 0; d <= c->ndim; ++d) k[d] = h[d] ? key_to_double(h[d]) : -1.0;
    if (tiled_ran && kind == HJ_TERM_NORMAL) { k[c->ndim] = k[0]; k[0] = -1.0; }     // the tiled kernel carries termNormal's maximum in slot 0
    return HJ_OK;
}
}  // namespace
}  // extern "C++"

int hj_term_normal(hj_ctx* c, int scheme, const void* y, const void* speed, double speed_scalar, void* ydot, double* step_bound) {
    const void* arr[HJ_MAX_DIM] = {speed, nullptr, nullptr, nullptr};
    const double scal[HJ_MAX_DIM] = {speed_scalar, 0, 0, 0};
    double k[HJ_MAX_DIM + 1];
    int rc = term_run(c, HJ_TERM_NORMAL, scheme, y, arr, scal, 0, ydot, k);
    if (rc) return rc;
    // stepBound = 1 / max over nodes with |grad phi| > 0 of (sum_i |a| |p_i| / dx_i) / |grad phi|   (term_normal.py:162-164)
    if (step_bound) *step_bound = k[c->ndim] > 0.0 ? 1.0 / k[c->ndim] : std::numeric_limits<double>::infinity();
    return HJ_OK;
}

int hj_term_reinit(hj_ctx* c, int scheme, const void* y, const void* initial, int subcell_order, void* ydot, double* step_bound) {
    if (!initial) return fail(HJ_EINVAL, "null initial array");
    if (subcell_order != 0 && subcell_order != 1) return fail(HJ_EINVAL, "Reinit subcell fix order of accuracy %d not supported", subcell_order);
    const void* arr[HJ_MAX_DIM] = {initial, nullptr, nullptr, nullptr};
    double k[HJ_MAX_DIM + 1];
    int rc = term_run(c, HJ_TERM_REINIT, scheme, y, arr, nullptr, subcell_order, ydot, k);
    if (rc) return rc;
    double inv = 0.0;                               // sum_i max|S p_i / |p|| / dx_i   (term_reinit.py:217,305)
    for (int d = 0; d < c->ndim; ++d) inv += (k[d] > 0.0 ? k[d] : 0.0) / c->dx[d];
    if (step_bound) *step_bound = inv > 0.0 ? 1.0 / inv : std::numeric_limits<double>::infinity();
    return HJ_OK;
}

int hj_term_convection(hj_ctx* c, int scheme, const void* y, const void* const* velocity, const double* velocity_scalar,
                       void* ydot, double* step_bound) {
    if (!velocity && !velocity_scalar) return fail(HJ_EINVAL, "velocity is neither arrays nor scalars");
    if (c)
        for (int d = 0; d < c->ndim; ++d)
            if (!(velocity && velocity[d]) && !velocity_scalar) return fail(HJ_EINVAL, "velocity of dim %d is neither an array nor a scalar", d);
    double k[HJ_MAX_DIM + 1];
    int rc = term_run(c, HJ_TERM_CONVECTION, scheme, y, velocity, velocity_scalar, 0, ydot, k);
    if (rc) return rc;
    double inv = 0.0;                               // sum_i max|v_i| / dx_i   (term_convection.py:175-177)
    for (int d = 0; d < c->ndim; ++d) inv += (k[d] > 0.0 ? k[d] : 0.0) / c->dx[d];
    if (step_bound) *step_bound = inv > 0.0 ? 1.0 / inv : std::numeric_limits<double>::infinity();
    return HJ_OK;
}

int hj_rk_combine(hj_ctx* c, int mode, double dt, const void* x0, const void* y, const void* z, void* out, int64_t n) {
    if (!c || !y || !z || !out) return fail(HJ_EINVAL, "null argument");
    if (mode < 1 || mode > 4) return fail(HJ_EINVAL, "unknown combine mode %d", mode);
    if (mode >= 2 && !x0) return fail(HJ_EINVAL, "mode %d needs x0", mode);
    if (n <= 0) return HJ_OK;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    if (c->dtype == HJ_F64)
        hipLaunchKernelGGL((rk_combine_kernel<double>), dim3(blocks), dim3(256), 0, c->stream, mode, dt, (const double*)x0,
                           (const double*)y, (const double*)z, (double*)out, (long long)n);
    else
        hipLaunchKernelGGL((rk_combine_kernel<float>), dim3(blocks), dim3(256), 0, c->stream, mode, (float)dt, (const float*)x0,
                           (const float*)y, (const float*)z, (float*)out, (long long)n);
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

int hj_rk_substep(hj_ctx* c, int scheme, int ham, const double* par, double t, int stage, double dt,
                  int restrict_sign, const void* y, const void* y0, void* out, int bound_slot,
                  int64_t p0, int64_t p1) {
    (void)t;
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (bound_slot < 0 || bound_slot >= HJ_BOUND_SLOTS) return fail(HJ_EINVAL, "bound_slot out of range");
    SubstepCall s{scheme, ham, stage, restrict_sign, par, dt, y, y0, out, nullptr, p0, p1};
    // a plane range of a SLAB (dist.HipSlabBackend drives the per-substep schedule through this entry): thin ranges march along axis 1
    if (c->halo_lo || c->halo_hi) s.xp = xp_wanted(c, p0, p1);
    return do_substep(c, s, bound_slot);
}

int hj_read_step_bound(hj_ctx* c, int slot, double* sb, double* amax) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (slot < 0 || slot >= HJ_BOUND_SLOTS || c->slot_ring[slot] < 0)
        return fail(HJ_ESTATE, "bound slot %d holds no result", slot);
    return read_ring(c, c->slot_ring[slot], sb, amax);
}

int hj_lf_term(hj_ctx* c, int scheme, int ham, const double* par, double t, int restrict_sign,
               const void* y, void* ydot, double* sb) {
    (void)t;
    if (!c) return fail(HJ_EINVAL, "null ctx");
    const int slot = HJ_BOUND_SLOTS - 1;
    SubstepCall s{scheme, ham, HJ_STAGE_YDOT, restrict_sign, par, 0.0, y, nullptr, ydot, nullptr, 0, c->N[0]};
    int rc = do_substep(c, s, slot);
    if (rc) return rc;
    if (sb) {
        // local LF variants: alpha of a native Hamiltonian does not depend on the data, so the bound
        // 1/max_x sum_d alpha_d(x)/dx_d is a static property of the grid
        // (a range-reading run-time Hamiltonian has no static bound: its launch reduced max_x sum_d alpha_d / dx_d into the slot, in the
        //  slot's generic form -- hj_fusedv.h, local_lf)
        if (c->diss_local && !user_ham_dynamic(ham)) return hj_static_step_bound(c, ham, par, sb, nullptr);
        return read_ring(c, c->slot_ring[slot], sb, nullptr);
    }
    return HJ_OK;
}

int hj_static_step_bound(hj_ctx* c, int ham, const double* par, double* sb, double* amax) {
    if (!c || !sb) return fail(HJ_EINVAL, "null argument");
    int rc = check_ham(c, ham, par);
    if (rc) return rc;
    const int np = ham_npar(ham);
    if (c->sb_valid && c->sb_ham == ham && memcmp(c->sb_par, par, sizeof(double) * np) == 0) {
        *sb = c->diss_local ? c->sb_local : c->sb_val;
        if (amax) for (int d = 0; d < c->ndim; ++d) amax[d] = c->sb_alpha[d];
        return HJ_OK;
    }
    HIP_TRY(hipMemsetAsync(c->keys, 0, 8 * sizeof(unsigned long long), c->stream));       // (keys[6]: the kernel's workgroup counter)
    if ((rc = alpha_partials(c))) return rc;
    const int blocks = (int)std::min<int64_t>((c->total + 255) / 256, 256 * 2);
#define HJ_AB(T, HAM)                                                                            \
    {                                                                                            \
        GridArgs<T, HAM<T>::ND> G;                                                               \
        fill_grid<T, HAM<T>::ND>(c, G);                                                          \
        HamTables<T> P;                                                                          \
        fill_ham<T>(c, par, P, ham);                                                             \
        DxArgs DX;                                                                               \
        for (int d = 0; d < HJ_MAX_DIM; ++d) DX.dx[d] = c->dx[d];                                \
        hipLaunchKernelGGL((alpha_bound_kernel<T, HAM<T>>), dim3(blocks), dim3(256), 0,          \
                           c->stream, G, P, c->keys, DX, c->alpha_part, c->keys + 6,             \
                           (unsigned long long*)nullptr, 0ull, DtArgs{0, 0, 0, nullptr});        \
    }
    if (ham >= HJ_HAM_USER_BASE) {
        if ((rc = user_alpha_bound(c, ham, par, c->keys, c->keys + 6))) return rc;
    } else if (c->dtype == HJ_F64) {
        if (ham == HJ_HAM_DUBINS_REL) HJ_AB(double, HamDubinsRel)
        else if (ham == HJ_HAM_DOUBLE_INTEGRATOR) HJ_AB(double, HamDoubleIntegrator)
        else HJ_AB(double, HamDoublePendulum)
    } else {
        if (ham == HJ_HAM_DUBINS_REL) HJ_AB(float, HamDubinsRel)
        else if (ham == HJ_HAM_DOUBLE_INTEGRATOR) HJ_AB(float, HamDoubleIntegrator)
        else HJ_AB(float, HamDoublePendulum)
    }
#undef HJ_AB
    HIP_TRY(hipGetLastError());
    unsigned long long k[HJ_MAX_DIM + 1];
    HIP_TRY(hipMemcpyAsync(k, c->keys, sizeof(k), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double inv = 0.0;
    for (int d = 0; d < c->ndim; ++d) {
        c->sb_alpha[d] = key_to_double(k[d]);
        inv += c->sb_alpha[d] / c->dx[d];
    }
    c->sb_val = 1.0 / inv;
    c->sb_local = 1.0 / key_to_double(k[HJ_MAX_DIM]);   // 1 / max_x sum_d alpha_d(x)/dx_d
    c->sb_ham = ham;
    memset(c->sb_par, 0, sizeof(c->sb_par));
    memcpy(c->sb_par, par, sizeof(double) * np);
    c->sb_valid = true;
    *sb = c->diss_local ? c->sb_local : c->sb_val;
    if (amax) for (int d = 0; d < c->ndim; ++d) amax[d] = c->sb_alpha[d];
    return HJ_OK;
}

// how often the per-call state of the ctx (stream, dissipation kind, post-step operators) has been written: a host layer that caches
// "the ctx is already set up for this call" compares it with the value it saw after its own writes (another user of the same ctx may
// have written since: ADVICE r04)
unsigned long long hj_ctx_state_generation(hj_ctx* c) { return c ? c->state_gen : 0ull; }

int hj_ctx_set_post_step(hj_ctx* c, int op) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (op < 0 || op > 2) return fail(HJ_EINVAL, "unknown post-step operator %d", op);
    c->post_step_op = op;
    ++c->state_gen;
    return HJ_OK;
}

int hj_ctx_set_post_arrays(hj_ctx* c, int op_a, const void* a, int op_b, const void* b) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if ((a && (op_a < 1 || op_a > 3)) || (b && (op_b < 1 || op_b > 3))) return fail(HJ_EINVAL, "unknown operator");
    c->post_arr[0] = a; c->post_arr_op[0] = a ? op_a : 0;
    c->post_arr[1] = b; c->post_arr_op[1] = b ? op_b : 0;
    ++c->state_gen;
    return HJ_OK;
}

int hj_ctx_set_dissipation(hj_ctx* c, int kind) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (kind != HJ_DISS_GLF && kind != HJ_DISS_LLF && kind != HJ_DISS_LLLF) return fail(HJ_EINVAL, "unknown dissipation kind %d", kind);
    c->diss_local = (kind != HJ_DISS_GLF);
    c->diss_kind = kind;
    ++c->state_gen;
    return HJ_OK;
}

// One odeCFLn step for a Hamiltonian whose alpha depends on the data (the costate range; hj_rtc.hip, HJ_HAM_RANGE).  The reference's
// first schemeFunc call yields ydot AND the stepBound that fixes deltaT (ode_cfl_3.py:129-151).  alpha(x, range) does not depend on a
// node's own costate, so the bound of the first stage follows from the range alone: range pass of y, max(alpha) over the grid (the
// kernel of hj_static_step_bound, a few microseconds), ONE host read, and the first stage is an ordinary fused Euler launch with
// deltaT known -- no ydot array, no separate y + deltaT*ydot pass.  The later stages are range pass + fused substep with the
// in-kernel max(alpha) kept; their bounds are read once at the end of the step for the reference's CFL warning (hj_rk_last_bounds).
// host_words[8..16) -> last_bounds[1..] / prev_bounds (the caller has made sure the copies have completed)
static int decode_stage_bounds(hj_ctx* c) {
    const int nst = c->stage_bounds_pending;
    c->stage_bounds_pending = 0;
    for (int st = 0; st < nst; ++st) {
        double inv = 0.0;
        for (int d = 0; d < c->ndim; ++d) {
            const unsigned long long k = c->host_words[8 + st * HJ_MAX_DIM + d];
            if (k == 0) return fail(HJ_ESTATE, "bound slot holds no reduction for dim %d", d);
            inv += key_to_double(k) / c->dx[d];
        }
        c->prev_bounds[1 + st] = 1.0 / inv;
    }
    c->prev_bounds[0] = c->last_bounds[0];
    c->prev_bounds_n = 1 + nst;
    c->prev_bounds_dt = c->stage_bounds_dt;
    c->prev_bounds_new = true;
    return HJ_OK;
}

static int rk_step_dynamic(hj_ctx* c, int order, int scheme, int ham, const double* par, double t0, double tf, double factor_cfl,
                           double max_step, int restrict_sign, const void* y_in, void* y_out, void* work0, void* work1,
                           double* t_out, double* dt_out) {
    const int kind = c->diss_kind;        // GLF: range pass + bound kernel; LLF: range pass + bound pass; LLLF: bound pass only
    if (c->range_src) return fail(HJ_ESTATE, "hj_rk_step with an external range source: step the slab through hj_rk_substep (dist.SlabIntegrator)");
    const int64_t n0 = c->N[0];
    int rc;
    const int slot2 = HJ_BOUND_SLOTS - 2, slot3 = HJ_BOUND_SLOTS - 3;
    if (kind != HJ_DISS_LLLF) {   // the range of y_in -> ctx->range_keys (a fresh entry of the ring)
        SubstepCall r{scheme, ham, HJ_STAGE_YDOT, 0, par, 0.0, y_in, nullptr, y_out, nullptr, 0, n0};
        r.range_only = true;
        if ((rc = do_substep(c, r, -1))) return rc;
    }
    double sb1 = 0, dt_step = 0;
    {   // stepBound = 1 / sum_d max_x alpha_d(x, range) / dx_d   (artificial_diss_glf.py:101-109)
        // Round 5: the bound kernel's keys and workgroup counter are part of the (pre-zeroed) ring entry of the range pass -- no memset
        // launch --, and its last workgroup writes the result into page-locked host memory that this thread polls: no copy launch, no
        // stream synchronisation (the step's only host <-> device round trip: 31 -> ~12 us of idle GPU at 201^3)
        if (!c->host_words) {
            HIP_TRY(hipHostMalloc((void**)&c->host_words, 16 * sizeof(unsigned long long), hipHostMallocCoherent));
            memset(c->host_words, 0, 16 * sizeof(unsigned long long));
            HIP_TRY(hipMalloc((void**)&c->dt_dev, 8 * sizeof(double)));
        }
        const unsigned long long seq = ++c->host_seq;
        volatile unsigned long long* hw = c->host_words;
        // ... and deltaT = min(factorCFL * stepBound, tspan[1] - t, maxStep) (ode_cfl_3.py:142) is formed ON the device too: the first
        // stage is enqueued behind the bound kernel before this thread knows the value (FusedArgs::dt_dev) -- the GPU never idles for the
        // host; the host reads the very same bits from host_words[5..6] for its own bookkeeping and the later stages' arguments
        const DtArgs dta{factor_cfl, tf - t0, max_step, c->dt_dev};
        if (kind == HJ_DISS_GLF) {
            if ((rc = user_alpha_bound(c, ham, par, c->range_keys + RANGE_ALPHA_AT, c->range_keys + RANGE_DONE_AT, true, c->host_words, seq, &dta))) return rc;
        } else {
            // local variants: alpha depends on the node's own costates -- the bound is a maximum over the STENCIL results: a pass of the
            // substep kernel that stores nothing (MODE 3 with a bound slot), then one thread turns the slot into deltaT
            const int slot1 = HJ_BOUND_SLOTS - 4;
            SubstepCall b{scheme, ham, HJ_STAGE_YDOT, 0, par, 0.0, y_in, nullptr, y_out, nullptr, 0, n0};
            b.bound_pass = true;
            b.range_ready = true;
            if ((rc = do_substep(c, b, slot1))) return rc;
            DxArgs DX;
            for (int d = 0; d < HJ_MAX_DIM; ++d) DX.dx[d] = c->dx[d];
            hipLaunchKernelGGL((bound_to_dt_kernel<0>), dim3(1), dim3(64), 0, c->stream, c->ring + (size_t)c->slot_ring[slot1] * HJ_MAX_DIM, c->ndim, DX, dta,
                               c->host_words, seq);
            HIP_TRY(hipGetLastError());
        }
        {
            SubstepCall a{scheme, ham, HJ_STAGE_EULER, restrict_sign, par, std::numeric_limits<double>::quiet_NaN(), y_in, nullptr,
                          order == 1 ? y_out : work0, nullptr, 0, n0};
            a.range_ready = true;
            a.dt_dev = c->dt_dev;
            if (order == 1) a.post_op = c->post_step_op;
            if ((rc = do_substep(c, a, -1))) return rc;
        }
        const auto t_poll = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(c->host_words + 7, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 1023u) == 0) {
                // (a failed launch or a lost device must not hang the caller: after 5 s ask the stream)
                if (std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(5)) {
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    if (__atomic_load_n(c->host_words + 7, __ATOMIC_ACQUIRE) != seq) return fail(HJ_EHIP, "the bound kernel never published its result");
                }
            }
            __builtin_ia32_pause();
        }
        unsigned long long bits_sb = hw[5], bits_dt = hw[6];
        memcpy(&sb1, &bits_sb, sizeof(double));
        memcpy(&dt_step, &bits_dt, sizeof(double));
        // the previous step's later-stage bounds (copied asynchronously at its end, ahead of everything this step enqueued) have landed
        // -- on the stream they were issued on; the caller may have re-bound the ctx to another stream since (hj_ctx_set_stream): the event says so
        if (c->stage_bounds_pending) {
            if (c->ev_bounds) HIP_TRY(hipEventSynchronize(c->ev_bounds));
            if ((rc = decode_stage_bounds(c))) return rc;
        }
    }
    const double dt = dt_step;
    c->last_bounds[0] = sb1;
    c->last_bounds_n = 1;
    void* first = order == 1 ? y_out : work0;
    double t = t0 + dt;
    if (order >= 2) {
        const int stage2 = order == 2 ? HJ_STAGE_RK2_FULL : HJ_STAGE_RK3_HALF;
        void* out2 = order == 2 ? y_out : work1;
        SubstepCall b{scheme, ham, stage2, restrict_sign, par, dt, first, y_in, out2, nullptr, 0, n0};
        if (order == 2) b.post_op = c->post_step_op;
        if ((rc = do_substep(c, b, slot2))) return rc;
        const double t1 = t0 + dt, t2 = t1 + dt;
        t = 0.5 * (t0 + t2);
        if (order == 3) {
            SubstepCall d{scheme, ham, HJ_STAGE_RK3_FULL, restrict_sign, par, dt, work1, y_in, y_out, nullptr, 0, n0};
            d.post_op = c->post_step_op;
            if ((rc = do_substep(c, d, slot3))) return rc;
            const double tHalf = 0.25 * (3 * t0 + t2);
            t = (1.0 / 3.0) * (t0 + 2 * (tHalf + dt));
        }
        // the later stages' bounds (the reference warns when deltaT exceeds them, ode_cfl_3.py:173-175, 215-217): copied into page-locked
        // host memory WITHOUT waiting -- the step returns while its last stage is still running; whoever wants them (hj_rk_last_bounds)
        // waits then, and the next step decodes them for hj_rk_prev_bounds at its own synchronisation point
        const int pos2 = c->slot_ring[slot2], pos3 = order == 3 ? c->slot_ring[slot3] : pos2;
        if (order == 3 && pos3 == pos2 + 1) {            // consecutive ring entries: one copy
            HIP_TRY(hipMemcpyAsync(c->host_words + 8, c->ring + (size_t)pos2 * HJ_MAX_DIM, 2 * HJ_MAX_DIM * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        } else {
            HIP_TRY(hipMemcpyAsync(c->host_words + 8, c->ring + (size_t)pos2 * HJ_MAX_DIM, HJ_MAX_DIM * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            if (order == 3)
                HIP_TRY(hipMemcpyAsync(c->host_words + 12, c->ring + (size_t)pos3 * HJ_MAX_DIM, HJ_MAX_DIM * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        }
        if (!c->ev_bounds) HIP_TRY(hipEventCreateWithFlags(&c->ev_bounds, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_bounds, c->stream));
        c->stage_bounds_pending = order - 1;
        c->stage_bounds_dt = dt;
    }
    for (int k = 0; k < 2; ++k) {
        if (!c->post_arr[k]) continue;
        const int op = c->post_arr_op[k] == 1 ? HJ_OP_MIN : (c->post_arr_op[k] == 2 ? HJ_OP_MAX : HJ_OP_MAX_NEG);
        if ((rc = hj_minmax_with(c, op, y_out, c->post_arr[k], c->total))) return rc;
    }
    if (t_out) *t_out = t;
    if (dt_out) *dt_out = dt;
    return HJ_OK;
}

int hj_rk_last_bounds(hj_ctx* c, double* sb, int* n) {
    if (!c || !sb || !n) return fail(HJ_EINVAL, "null argument");
    if (c->stage_bounds_pending) {          // the last step's later stages may still be running: this call waits for them
        if (c->ev_bounds) HIP_TRY(hipEventSynchronize(c->ev_bounds));      // (the stream the copies went out on, whatever the ctx is bound to now)
        else HIP_TRY(hipStreamSynchronize(c->stream));
        const int nst = c->stage_bounds_pending;
        int rc = decode_stage_bounds(c);
        if (rc) return rc;
        for (int st = 0; st < nst; ++st) c->last_bounds[1 + st] = c->prev_bounds[1 + st];
        c->last_bounds_n = 1 + nst;
        c->prev_bounds_new = false;         // handed out here: hj_rk_prev_bounds does not report them a second time
    }
    *n = c->last_bounds_n;
    for (int k = 0; k < c->last_bounds_n; ++k) sb[k] = c->last_bounds[k];
    return HJ_OK;
}

int hj_rk_prev_bounds(hj_ctx* c, double* sb, int* n, double* dt) {
    if (!c || !sb || !n || !dt) return fail(HJ_EINVAL, "null argument");
    *n = 0;
    if (!c->prev_bounds_new) return HJ_OK;
    c->prev_bounds_new = false;
    *n = c->prev_bounds_n;
    *dt = c->prev_bounds_dt;
    for (int k = 0; k < c->prev_bounds_n; ++k) sb[k] = c->prev_bounds[k];
    return HJ_OK;
}

int hj_ctx_set_range_source(hj_ctx* c, const void* keys_dev) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    c->range_src = (const unsigned long long*)keys_dev;
    return HJ_OK;
}

int hj_range_alpha_max(hj_ctx* c, int ham, const double* par, double* amax) {
    if (!c || !amax) return fail(HJ_EINVAL, "null argument");
    if (!user_ham_dynamic(ham)) return fail(HJ_EINVAL, "Hamiltonian %d does not read the costate range", ham);
    if (!c->range_src && !c->range_keys) return fail(HJ_ESTATE, "no range yet: hj_range_pass + hj_ctx_set_range_source first");
    int rc = check_ham(c, ham, par);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->keys, 0, 8 * sizeof(unsigned long long), c->stream));
    if ((rc = user_alpha_bound(c, ham, par, c->keys, c->keys + 6, true))) return rc;
    unsigned long long k[HJ_MAX_DIM];
    HIP_TRY(hipMemcpyAsync(k, c->keys, sizeof(k), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int d = 0; d < c->ndim; ++d) amax[d] = key_to_double(k[d]);
    return HJ_OK;
}

int hj_range_pass(hj_ctx* c, int scheme, int ham, const double* par, const void* y, void* keys_dev) {
    if (!c || !y || !keys_dev) return fail(HJ_EINVAL, "null argument");
    if (!user_ham_dynamic(ham)) return fail(HJ_EINVAL, "Hamiltonian %d does not read the costate range", ham);
    SubstepCall s{scheme, ham, HJ_STAGE_YDOT, 0, par, 0.0, y, nullptr, keys_dev /* unused as an array */, nullptr, 0, c->N[0]};
    s.range_only = true;
    s.range_out = (unsigned long long*)keys_dev;
    return do_substep(c, s, -1);
}

int hj_bound_pass(hj_ctx* c, int scheme, int ham, const double* par, const void* y, double* sb_host) {
    if (!c || !y || !sb_host) return fail(HJ_EINVAL, "null argument");
    if (!user_ham_dynamic(ham)) return fail(HJ_EINVAL, "Hamiltonian %d does not read the costate range", ham);
    if (c->diss_kind == HJ_DISS_GLF) return fail(HJ_ESTATE, "the bound pass belongs to the local Lax-Friedrichs kinds (hj_ctx_set_dissipation); the global one has hj_range_alpha_max");
    int rc;
    if (c->diss_kind != HJ_DISS_LLLF && !c->range_src && (c->halo_lo || c->halo_hi))
        return fail(HJ_ESTATE, "LLF on a slab reads the range of the WHOLE grid in the other dimensions: hj_range_pass, reduce over the ranks, hj_ctx_set_range_source first");
    if (c->diss_kind != HJ_DISS_LLLF && !c->range_src) {     // LLF reads the grid-wide range in the other dimensions: this ctx's own, unless told
        SubstepCall r{scheme, ham, HJ_STAGE_YDOT, 0, par, 0.0, y, nullptr, c->keys /* unused as an array */, nullptr, 0, c->N[0]};
        r.range_only = true;
        if ((rc = do_substep(c, r, -1))) return rc;
    }
    const int slot = HJ_BOUND_SLOTS - 4;
    SubstepCall b{scheme, ham, HJ_STAGE_YDOT, 0, par, 0.0, y, nullptr, c->keys /* nothing is stored */, nullptr, 0, c->N[0]};
    b.bound_pass = true;
    b.range_ready = true;
    if ((rc = do_substep(c, b, slot))) return rc;
    return read_ring(c, c->slot_ring[slot], sb_host, nullptr);
}

int hj_rk_step(hj_ctx* c, int order, int scheme, int ham, const double* par, double t0, double tf,
               double factor_cfl, double max_step, int restrict_sign, const void* y_in, void* y_out,
               void* work0, void* work1, double* t_out, double* dt_out) {
    if (!c || !y_in || !y_out) return fail(HJ_EINVAL, "null argument");
    if (order < 1 || order > 3) return fail(HJ_EINVAL, "order must be 1, 2 or 3");
    if (order >= 2 && !work0) return fail(HJ_EINVAL, "work0 required for order >= 2");
    if (order == 3 && !work1) return fail(HJ_EINVAL, "work1 required for order 3");
    if (y_out == y_in) return fail(HJ_EINVAL, "y_out must not alias y_in");
    if (user_ham_dynamic(ham)) return rk_step_dynamic(c, order, scheme, ham, par, t0, tf, factor_cfl, max_step, restrict_sign, y_in, y_out, work0, work1, t_out, dt_out);
    double sb;
    int rc = hj_static_step_bound(c, ham, par, &sb, nullptr);
    if (rc) return rc;
    // deltaT = min(factorCFL*stepBound, tspan[1]-t, maxStep)  (ode_cfl_3.py:142)
    const double dt = std::min(std::min(factor_cfl * sb, tf - t0), max_step);
    const int64_t n0 = c->N[0];
    auto slot = [&]() { return -1; };      // dt comes from the static bound above: no launch of a step needs its own
    double t = t0;
    const bool fuse = use_stage12(c, order, scheme, ham, par, restrict_sign);
    // intended WENO5: every stage's launch reduces max(D1^2) of its output for the next stage's epsilon (no pre-pass between
    // the stages); across steps only inside hj_rk_integrate, where nobody else touches the state (eps_chain_*)
    const bool chain_in = c->eps_chain_in, chain_out = c->eps_chain_out && !c->post_arr[0] && !c->post_arr[1];
    const int rc_coop = (order >= 2 && !fuse) ? try_coop_step(c, order, scheme, ham, par, dt, restrict_sign, y_in, y_out, work0, work1) : HJ_XP_FALLBACK;
    if (rc_coop != HJ_XP_FALLBACK && rc_coop != HJ_OK) return rc_coop;
    if (rc_coop == HJ_OK) {
        // small grid: the whole step was ONE cooperative launch (hj_split.h, coop_rk_kernel); the times as below
        const double t1 = t0 + dt, t2 = t1 + dt;
        if (order == 2) t = 0.5 * (t0 + t2);
        else { const double tHalf = 0.25 * (3 * t0 + t2); t = (1.0 / 3.0) * (t0 + 2 * (tHalf + dt)); }
    } else if (order == 1) {
        SubstepCall s{scheme, ham, HJ_STAGE_EULER, restrict_sign, par, dt, y_in, nullptr, y_out, nullptr, 0, n0};
        s.post_op = c->post_step_op;
        s.eps_from_prev = chain_in; s.want_eps = chain_out;
        if ((rc = do_substep(c, s, slot()))) return rc;
        t = t0 + dt;
    } else if (order == 2 && fuse) {
        // both stages in one launch: y_out = (y + (y1 + dt L(y1)))/2 with y1 = y + dt L(y)  (ode_cfl_2.py:151-201)
        Stage12Call a{scheme, ham, par, dt, 0.5, 0.5, y_in, y_out, nullptr};
        if ((rc = do_stage12(c, a, slot()))) return rc;
        const double t1 = t0 + dt, t2 = t1 + dt;
        t = 0.5 * (t0 + t2);
    } else if (order == 3 && fuse) {
        // stages 1+2 in one launch (y1 never reaches HBM), then the third stage  (ode_cfl_3.py:151-241)
        Stage12Call a{scheme, ham, par, dt, 0.75, 0.25, y_in, work1, nullptr};
        if ((rc = do_stage12(c, a, slot()))) return rc;
        SubstepCall d{scheme, ham, HJ_STAGE_RK3_FULL, restrict_sign, par, dt, work1, y_in, y_out, nullptr, 0, n0};
        d.post_op = c->post_step_op;
        if ((rc = do_substep(c, d, slot()))) return rc;
        const double t1 = t0 + dt, t2 = t1 + dt;
        const double tHalf = 0.25 * (3 * t0 + t2);
        const double tThreeHalf = tHalf + dt;
        t = (1.0 / 3.0) * (t0 + 2 * tThreeHalf);
    } else if (order == 2) {
        SubstepCall a{scheme, ham, HJ_STAGE_EULER, restrict_sign, par, dt, y_in, nullptr, work0, nullptr, 0, n0};
        a.eps_from_prev = chain_in; a.want_eps = true;
        if ((rc = do_substep(c, a, slot()))) return rc;
        SubstepCall b{scheme, ham, HJ_STAGE_RK2_FULL, restrict_sign, par, dt, work0, y_in, y_out, nullptr, 0, n0};
        b.post_op = c->post_step_op;
        b.eps_from_prev = true; b.want_eps = chain_out;
        if ((rc = do_substep(c, b, slot()))) return rc;
        const double t1 = t0 + dt, t2 = t1 + dt;
        t = 0.5 * (t0 + t2);  // ode_cfl_2.py:200
    } else {
        SubstepCall a{scheme, ham, HJ_STAGE_EULER, restrict_sign, par, dt, y_in, nullptr, work0, nullptr, 0, n0};
        a.eps_from_prev = chain_in; a.want_eps = true;
        if ((rc = do_substep(c, a, slot()))) return rc;
        SubstepCall b{scheme, ham, HJ_STAGE_RK3_HALF, restrict_sign, par, dt, work0, y_in, work1, nullptr, 0, n0};
        b.eps_from_prev = true; b.want_eps = true;
        if ((rc = do_substep(c, b, slot()))) return rc;
        SubstepCall d{scheme, ham, HJ_STAGE_RK3_FULL, restrict_sign, par, dt, work1, y_in, y_out, nullptr, 0, n0};
        d.post_op = c->post_step_op;
        d.eps_from_prev = true; d.want_eps = chain_out;
        if ((rc = do_substep(c, d, slot()))) return rc;
        const double t1 = t0 + dt, t2 = t1 + dt;
        const double tHalf = 0.25 * (3 * t0 + t2);       // ode_cfl_3.py:188
        const double tThreeHalf = tHalf + dt;            // :221
        t = (1.0 / 3.0) * (t0 + 2 * tThreeHalf);         // :236
    }
    // post-step operators against caller arrays (hj_ctx_set_post_arrays): separate elementwise launches,
    // so that the fused kernel carries nothing for them (in-kernel variants cost the plain path 2-5 %
    // at 401^3 through extra registers / split scheduling blocks)
    for (int k = 0; k < 2; ++k) {
        if (!c->post_arr[k]) continue;
        const int op = c->post_arr_op[k] == 1 ? HJ_OP_MIN : (c->post_arr_op[k] == 2 ? HJ_OP_MAX : HJ_OP_MAX_NEG);
        if ((rc = hj_minmax_with(c, op, y_out, c->post_arr[k], c->total))) return rc;
    }
    if (t_out) *t_out = t;
    if (dt_out) *dt_out = dt;
    return HJ_OK;
}

int hj_rk_stage12(hj_ctx* c, int scheme, int ham, const double* par, double dt, double ca, double cb,
                  const void* y, void* out, int bound_slot) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (bound_slot < 0 || bound_slot >= HJ_BOUND_SLOTS) return fail(HJ_EINVAL, "bound_slot out of range");
    Stage12Call s{scheme, ham, par, dt, ca, cb, y, out, nullptr};
    return do_stage12(c, s, bound_slot);
}

int hj_rk_plan(hj_ctx* c, int order, int scheme, int ham, const double* par, int restrict_sign, int* launches,
               int* stage_fused) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (order < 1 || order > 3) return fail(HJ_EINVAL, "order must be 1, 2 or 3");
    int rc = check_ham(c, ham, par);
    if (rc) return rc;
    const bool f = use_stage12(c, order, scheme, ham, par, restrict_sign);
    if (launches) {
        *launches = f ? order - 1 : order;
        if (!f && coop_applies(c, order, scheme, ham) && c->coop_ok != 0) *launches = 1;      // small grids: one cooperative launch (coop_rk_kernel)
        if (scheme == HJ_WENO5) {
            // the epsilon pre-pass (2 launches) in front of every stage -- or, when the tiled kernels reduce max(D1^2) of their
            // own output (HJ_EPS_FUSE, whole-grid launches of a single domain), in front of the first stage only, with one
            // seam launch behind each of the other stages' producers
            const bool fused_eps = c->eps_fuse && !c->force_direct && !(c->direct_below > 0 && c->total < c->direct_below) &&
                                   !c->halo_lo && !c->halo_hi && !c->weno_src && c->total >= c->eps_fuse_min_cells && c->total < (1ll << 31);
            // launches of one epsilon pre-pass, decided exactly as do_substep / weno_eps_rows decide it: ONE launch whose rows the
            // consumer folds when the plane has at most 512 K columns (and HJ_EPS_FUSE is on), else the two-launch form
            const int prepass = (c->eps_fuse && weno_rows_fit(c)) ? 1 : 2;
            // hj_rk_step with fused epsilon: a pre-pass in front of the first stage, then `order` substep launches with a seam
            // launch behind each producer (all stages but the last); without: a pre-pass in front of every stage
            *launches = fused_eps ? prepass + order + (order - 1) : order * (1 + prepass);
        }
    }
    if (stage_fused) *stage_fused = f ? 1 : 0;
    return HJ_OK;
}

int hj_rk_integrate(hj_ctx* c, int order, int scheme, int ham, const double* par, double t0, double tf,
                    double factor_cfl, double max_step, int restrict_sign, const void* y_in, void* buf_a,
                    void* buf_b, void* work, int64_t max_steps, double stop_tol, double* t_out,
                    int64_t* steps_out, int* result_in) {
    if (!c || !y_in || !buf_a || !buf_b) return fail(HJ_EINVAL, "null argument");
    if (order >= 2 && !work) return fail(HJ_EINVAL, "work buffer required for order >= 2");
    if (buf_a == buf_b || buf_a == y_in || buf_b == y_in || work == y_in || work == buf_a || work == buf_b)
        return fail(HJ_EINVAL, "buffers must be distinct");
    // the loop of odeCFLn (ode_cfl_3.py:125): while tf - t >= small*|tf|, small = 100*eps (:81)
    const double small = 100.0 * 2.220446049250313e-16;
    const void* cur = y_in;
    void* outs[2] = {buf_a, buf_b};
    int which = 0;          // 0: y_in holds the state, 1: buf_a, 2: buf_b
    double t = t0;
    int64_t steps = 0;
    // stop_tol < 0: the integrators' own test; stop_tol >= 0: HJIPDE_solve's `while tNow < tau[i] - small`
    auto more = [&]() { return stop_tol < 0 ? (tf - t >= small * std::fabs(tf)) : (t < tf - stop_tol); };
    c->eps_ready = false;
    while (more() && (max_steps <= 0 || steps < max_steps)) {
        void* nxt = outs[steps & 1];
        double tn = t, dt = 0;
        // the state stays inside this call between the steps: the last launch of a step reduces max(D1^2) for the first
        // launch of the next one (intended WENO5)
        c->eps_chain_in = steps > 0;
        c->eps_chain_out = true;
        // RK3: the first stage buffer doubles as the output; RK2: `work` is the first stage buffer
        int rc = hj_rk_step(c, order, scheme, ham, par, t, tf, factor_cfl, max_step, restrict_sign, cur, nxt,
                            order == 3 ? nxt : work, work, &tn, &dt);
        c->eps_chain_in = c->eps_chain_out = false;
        if (rc) return rc;
        if (!(tn > t)) return fail(HJ_ESTATE, "time step underflow at t=%g (dt=%g)", t, dt);
        cur = nxt;
        which = 1 + (int)(steps & 1);
        t = tn;
        ++steps;
    }
    c->eps_ready = false;
    if (t_out) *t_out = t;
    if (steps_out) *steps_out = steps;
    if (result_in) *result_in = which;
    return HJ_OK;
}

int hj_max_d1sq(hj_ctx* c, const void* y, void* out_dev) {
    if (!c || !y || !out_dev) return fail(HJ_EINVAL, "null argument");
    return weno_eps_to(c, y, out_dev);
}

int hj_ctx_set_weno_eps_source(hj_ctx* c, const void* src) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    c->weno_src = src;
    return HJ_OK;
}

int hj_minmax_with(hj_ctx* c, int op, void* y, const void* other, int64_t n) {
    if (!c || !y || !other) return fail(HJ_EINVAL, "null argument");
    if (op < HJ_OP_MIN || op > HJ_OP_MAX_NEG) return fail(HJ_EINVAL, "unknown op %d", op);
    if (n <= 0) return HJ_OK;
    c->eps_ready = false;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    if (c->dtype == HJ_F64)
        hipLaunchKernelGGL((minmax_kernel<double>), dim3(blocks), dim3(256), 0, c->stream, (double*)y, (const double*)other, (long long)n, op);
    else
        hipLaunchKernelGGL((minmax_kernel<float>), dim3(blocks), dim3(256), 0, c->stream, (float*)y, (const float*)other, (long long)n, op);
    HIP_TRY(hipGetLastError());
    return HJ_OK;
}

int hj_any_nan(hj_ctx* c, const void* y, int64_t n, int* has) {
    if (!c || !y || !has) return fail(HJ_EINVAL, "null argument");
    HIP_TRY(hipMemsetAsync(c->flag, 0, sizeof(int), c->stream));
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    if (n > 0) {
        if (c->dtype == HJ_F64)
            hipLaunchKernelGGL((any_nan_kernel<double>), dim3(blocks), dim3(256), 0, c->stream, (const double*)y, (long long)n, c->flag);
        else
            hipLaunchKernelGGL((any_nan_kernel<float>), dim3(blocks), dim3(256), 0, c->stream, (const float*)y, (long long)n, c->flag);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(has, c->flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_comm_unique_id(const char* rccl_path, void* uid) {
    if (!uid) return fail(HJ_EINVAL, "null argument");
    int rc = rccl_load(rccl_path);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    NCCL_TRY(g_rccl.GetUniqueId((ncclUniqueId*)uid));
    return HJ_OK;
}

int hj_comm_init(hj_ctx* c, const char* rccl_path, int rank, int nranks, const void* uid, int lo, int hi) {
    if (!c || !uid) return fail(HJ_EINVAL, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(HJ_EINVAL, "bad rank/nranks");
    if (lo >= nranks || hi >= nranks) return fail(HJ_EINVAL, "neighbour rank out of range");
    int rc = rccl_load(rccl_path);
    if (rc) return rc;
    if (c->comm) (void)hj_comm_destroy(c);
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, uid, sizeof(id));
    NCCL_TRY(g_rccl.CommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank; c->comm_size = nranks; c->lo_rank = lo; c->hi_rank = hi;
    c->external_exchange = 0;
    return slab_streams_create(c);
}

int hj_comm_init_external(hj_ctx* c, int rank, int nranks, int lo, int hi) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(HJ_EINVAL, "bad rank/nranks");
    if (lo >= nranks || hi >= nranks) return fail(HJ_EINVAL, "neighbour rank out of range");
    if (c->comm || c->edge_stream) (void)hj_comm_destroy(c);
    HIP_TRY(hipSetDevice(c->device));
    c->comm = nullptr;
    c->comm_rank = rank; c->comm_size = nranks; c->lo_rank = lo; c->hi_rank = hi;
    c->external_exchange = 1;
    return slab_streams_create(c);
}

int hj_comm_destroy(hj_ctx* c) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    // sends/receives or edge kernels may still be in flight (also when reached from hj_ctx_destroy at
    // interpreter teardown): drain the auxiliary stream and the ctx stream before anything is destroyed
    if (c->edge_stream) {
        (void)slab_join(c);
        (void)hipStreamSynchronize(c->edge_stream);
        (void)hipStreamSynchronize(c->stream);
    }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    if (c->edge_stream) (void)hipStreamDestroy(c->edge_stream);   // comm_stream aliases it
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_edge) (void)hipEventDestroy(c->ev_edge);
    if (c->ev_edge2) (void)hipEventDestroy(c->ev_edge2);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    for (int i = 0; i < 3; ++i) { if (c->ev_int[i]) (void)hipEventDestroy(c->ev_int[i]); c->ev_int[i] = nullptr; }
    c->comm_stream = c->edge_stream = nullptr;
    c->ev_start = c->ev_edge = c->ev_edge2 = c->ev_comm = nullptr;
    if (c->gate) { (void)hipFree(c->gate); c->gate = nullptr; c->gate_count = 0; }
    c->slab_pending = 0;
    c->external_exchange = 0;
    return HJ_OK;
}

int hj_comm_info(hj_ctx* c, int* rank, int* nranks, int* lo, int* hi) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    int r = c->comm_rank, n = c->comm_size;
    if (c->comm) {
        NCCL_TRY(g_rccl.CommCount(c->comm, &n));
        NCCL_TRY(g_rccl.CommUserRank(c->comm, &r));
    }
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    if (lo) *lo = c->lo_rank;
    if (hi) *hi = c->hi_rank;
    return HJ_OK;
}

int hj_halo_exchange(hj_ctx* c, void* buf) {
    if (!c || !buf) return fail(HJ_EINVAL, "null argument");
    if (c->lo_rank < 0 && c->hi_rank < 0) return HJ_OK;
    int rc = slab_join(c);
    if (rc) return rc;
    return post_halo(c, buf, c->stream);
}

int hj_halo_exchange_depth(hj_ctx* c, void* buf, int depth) {
    if (!c || !buf) return fail(HJ_EINVAL, "null argument");
    if (depth < 1 || depth > c->N[0]) return fail(HJ_EINVAL, "bad halo depth %d", depth);
    if (c->lo_rank < 0 && c->hi_rank < 0) return HJ_OK;
    int rc = slab_join(c);
    if (rc) return rc;
    return post_halo(c, buf, c->stream, depth);
}

int hj_slab_join(hj_ctx* c) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    return slab_join(c);
}

int hj_ctx_set_axis0_pad(hj_ctx* c, int pad, const double* vs0_ext, const double* aux0_ext, const double* aux1_ext) {
    if (!c || !vs0_ext) return fail(HJ_EINVAL, "null argument");
    if (pad < 0 || pad > 64) return fail(HJ_EINVAL, "bad pad %d", pad);
    const int64_t n = c->N[0] + 2 * (int64_t)pad;
    int rc;
    if ((rc = upload(c, &c->coord0_ext, vs0_ext, n))) return rc;
    const double* tabs[2] = {aux0_ext, aux1_ext};
    for (int s = 0; s < 2; ++s) {
        if (tabs[s]) { if ((rc = upload(c, &c->aux_ext[s], tabs[s], n))) return rc; }
        else if (c->aux_ext[s]) { HIP_TRY(hipFree(c->aux_ext[s])); c->aux_ext[s] = nullptr; }
    }
    c->pad0 = pad;
    return HJ_OK;
}

int hj_slab_rk_step_deep(hj_ctx* c, int order, int scheme, int ham, const double* par, double dt, int rs,
                         const void* cur, void* y_out, void* w0, void* w1) {
    if (!c || !cur || !y_out) return fail(HJ_EINVAL, "null argument");
    if (order < 1 || order > 3) return fail(HJ_EINVAL, "order must be 1, 2 or 3");
    if (order >= 2 && !w0) return fail(HJ_EINVAL, "work0 required for order >= 2");
    if (order == 3 && !w1) return fail(HJ_EINVAL, "work1 required for order 3");
    return slab_step_deep(c, order, scheme, ham, par, dt, rs, cur, y_out, w0, w1);
}

int hj_slab_rk_step(hj_ctx* c, int order, int scheme, int ham, const double* par, double dt, int rs,
                    const void* cur, void* y_out, void* w0, void* w1) {
    if (!c || !cur || !y_out) return fail(HJ_EINVAL, "null argument");
    if (order < 1 || order > 3) return fail(HJ_EINVAL, "order must be 1, 2 or 3");
    if (order >= 2 && !w0) return fail(HJ_EINVAL, "work0 required for order >= 2");
    if (order == 3 && !w1) return fail(HJ_EINVAL, "work1 required for order 3");
    if ((c->lo_rank >= 0 || c->hi_rank >= 0) && !c->comm && !c->external_exchange) return fail(HJ_ESTATE, "hj_comm_init has not been called");
    int rc;
    if (order == 1) return slab_substep(c, scheme, ham, par, HJ_STAGE_EULER, dt, rs, cur, nullptr, y_out);
    if ((rc = slab_substep(c, scheme, ham, par, HJ_STAGE_EULER, dt, rs, cur, nullptr, w0))) return rc;
    if (order == 2) return slab_substep(c, scheme, ham, par, HJ_STAGE_RK2_FULL, dt, rs, w0, cur, y_out);
    if ((rc = slab_substep(c, scheme, ham, par, HJ_STAGE_RK3_HALF, dt, rs, w0, cur, w1))) return rc;
    return slab_substep(c, scheme, ham, par, HJ_STAGE_RK3_FULL, dt, rs, w1, cur, y_out);
}

int hj_sync(hj_ctx* c) {
    if (!c) return fail(HJ_EINVAL, "null ctx");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HJ_OK;
}

}  // extern "C"
