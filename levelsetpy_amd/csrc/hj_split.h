// Split-path and helper kernels (gfx950): the pieces of the path that the reference exposes as
// separate callables (ghost padding, one-dimension upwind derivatives with their global
// min/max), the 'maxOverGrid' WENO epsilon reduction, the HJIPDE_solve post-step operators,
// and a direct (untiled) form of the fused substep used for shapes the tiled kernel does not
// map (4-D grids) and as an independent cross-check of it.
#pragma once
#include "hj_device.h"

namespace hj {

template <typename T> struct DimView {
    long long outer, inner;  // product of sizes before / after `dim`
    int n;                   // size along dim
    int bc;
    int halo_lo, halo_hi;    // only meaningful for dim 0
    T km;
    T K[HJ_NK];
};

// value of phi at position j (may be outside [0,n)) on the line through `line` (element 0 of the
// line), element stride `s`
template <typename T>
__device__ __forceinline__ T line_value(const T* line, long long s, int j, int n, int bc, T km,
                                        int halo_lo, int halo_hi) {
    if (j >= 0 && j < n) return line[j * s];
    if (j < 0) {
        if (halo_lo) return line[j * s];
        if (bc == HJ_BC_PERIODIC) return line[(j + n) * s];
        return ghost_value(line[0], line[s], T(-j) * km);
    }
    if (halo_hi) return line[j * s];
    if (bc == HJ_BC_PERIODIC) return line[(j - n) * s];
    return ghost_value(line[(n - 1) * s], line[(n - 2) * s], T(j - n + 1) * km);
}

// ---- dataOut = grid.bdry[dim](dataIn, dim, width, ghostData): one thread per output element
template <typename T>
__global__ void ghost_kernel(const T* __restrict__ in, T* __restrict__ out, DimView<T> V, int width) {
    const long long no = (long long)V.n + 2 * width;
    const long long total = V.outer * no * V.inner;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long in_i = t % V.inner;
        const long long r = t / V.inner;
        const int j = (int)(r % no) - width;
        const long long o = r / no;
        const T* line = in + o * V.n * V.inner + in_i;
        out[t] = line_value(line, V.inner, j, V.n, V.bc, V.km, 0, 0);
    }
}

// ---- derivL, derivR = upwindFirst*(grid, data, dim) + {min L, max L, min R, max R}
// keys[0..3]: atomicMax keys of {-min L, max L, -min R, max R}
template <typename T, int SCHEME>
__global__ __launch_bounds__(256) void upwind_kernel(const T* __restrict__ phi, T* __restrict__ dL,
                                                     T* __restrict__ dR, DimView<T> V, T eps,
                                                     unsigned long long* keys) {
    const long long total = V.outer * V.n * V.inner;
    double m[4] = {-1e300, -1e300, -1e300, -1e300};
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        // position on the line (32-bit divisions when the grid allows: a 64-bit one costs ~4x as much) and the seven values,
        // every load issued unconditionally (round 4; as gather_stencils below: the branches of line_value() in front of each
        // load serialised them), ghost values fixed up afterwards by the waves that touch an extrapolated edge
        int i;
        if (total < (1ll << 31)) {
            const unsigned r = (unsigned)t / (unsigned)V.inner;
            i = (int)(r % (unsigned)V.n);
        } else {
            i = (int)((t / V.inner) % V.n);
        }
        const T* pc0 = phi + t;
        const int n = V.n;
        const bool per = V.bc == HJ_BC_PERIODIC;
        bool ghost = false;
        T v[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            int j = i + k - 3;
            if (j < 0 && !V.halo_lo) {
                if (per) j += n; else { j = 0; ghost = true; }
            } else if (j >= n && !V.halo_hi) {
                if (per) j -= n; else { j = n - 1; ghost = true; }
            }
            v[k] = pc0[(long long)(j - i) * V.inner];
        }
        if (__any(ghost ? 1 : 0)) {
            const T* line = pc0 - (long long)i * V.inner;
            if (!per && i < HJ_STENCIL && !V.halo_lo) {
                const T e = line[0], in = line[V.inner];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (i + k - 3 < 0) v[k] = ghost_value(e, in, T(3 - k - i) * V.km);
            }
            if (!per && i + HJ_STENCIL >= n && !V.halo_hi) {
                const T e = line[(long long)(n - 1) * V.inner], in = line[(long long)(n - 2) * V.inner];
#pragma unroll
                for (int k = 4; k < 7; ++k)
                    if (i + k - 3 >= n) v[k] = ghost_value(e, in, T(i + k - 3 - n + 1) * V.km);
            }
        }
        T L, R;
        upwind<SCHEME, T>(v, V.K, eps, L, R);
        dL[t] = L;
        dR[t] = R;
        m[0] = fmax(m[0], -(double)L); m[1] = fmax(m[1], (double)L);
        m[2] = fmax(m[2], -(double)R); m[3] = fmax(m[3], (double)R);
    }
    if (keys) {
        __shared__ double red[4][4];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double w = wave_max(m[k]);
            if (lane == 0) red[wv][k] = w;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const int k = threadIdx.x;
            const double w = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
            if (w > -1e299) key_max(keys + k, w);
        }
    }
}

// ---- grid description shared by the whole-grid helper kernels
template <typename T, int ND> struct GridArgs {
    int n[ND];
    int bc[ND];
    int halo_lo, halo_hi;
    T km[ND], inv_dx[ND];
    T K[ND][HJ_NK];
    long long stride[ND];
    long long total;
};

template <typename T, int ND>
__device__ __forceinline__ void decode(const GridArgs<T, ND>& G, long long t, int* idx) {
    if (G.total < (1ll << 31)) {       // 32-bit divisions: ~4x cheaper than 64-bit ones (uniform branch)
        unsigned u = (unsigned)t;
#pragma unroll
        for (int d = ND - 1; d >= 0; --d) {
            const unsigned q = u / (unsigned)G.n[d];
            idx[d] = (int)(u - q * (unsigned)G.n[d]);
            u = q;
        }
        return;
    }
#pragma unroll
    for (int d = ND - 1; d >= 0; --d) {
        const long long q = t / G.n[d];
        idx[d] = (int)(t - q * G.n[d]);
        t = q;
    }
}

// ---- max over the unstripped D1 table of D1^2 per dim (upwind_first_weno5a.py:153-156).
// Ghost-to-ghost differences repeat the first interior one (extrapolation) or interior ones
// (periodic), so the max runs over forward differences of interior cells, the periodic wrap
// pair, and -- on a slab face -- the pair reaching into the lower halo plane.
// Plane march: a thread owns one column (fixed indices on axes 1..ND-1, decoded once) and walks a
// chunk of axis-0 planes, carrying the next plane's value in a register: per cell one streamed load
// plus ND-1 neighbour loads that adjacent lanes / rows also issue (L1/L2 hits), no divisions in the
// loop.  Every workgroup leaves its ND partial maxima in partials[block][d] (no contended atomics);
// partials_to_values_kernel folds them.
template <typename T, int ND, int NT = 256>
__global__ __launch_bounds__(NT) void max_d1sq_kernel(const T* __restrict__ y, GridArgs<T, ND> G,
                                                      double* __restrict__ partials, int chunk) {
    double m[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) m[d] = 0.0;
    const long long S = G.stride[0];
    const long long col = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const int pb = blockIdx.y * chunk, pe = min(pb + chunk, G.n[0]);
    if (col < S && pb < pe) {
        int idx[ND];
        {
            unsigned u = (unsigned)col;      // plane size < 2^31 (checked at ctx creation)
#pragma unroll
            for (int d = ND - 1; d >= 1; --d) {
                const unsigned q = u / (unsigned)G.n[d];
                idx[d] = (int)(u - q * (unsigned)G.n[d]);
                u = q;
            }
        }
        long long off[ND];       // offset of the forward neighbour on axes >= 1; 0 = none
#pragma unroll
        for (int d = 1; d < ND; ++d) {
            off[d] = 0;
            if (idx[d] + 1 < G.n[d]) off[d] = G.stride[d];
            else if (G.bc[d] == HJ_BC_PERIODIC) off[d] = -(long long)(G.n[d] - 1) * G.stride[d];
        }
        const T* p0 = y + col;
        T c = p0[(long long)pb * S];
        if (pb == 0 && G.halo_lo) {      // the pair reaching into the lower halo plane
            const T D1 = G.inv_dx[0] * (c - p0[-S]);
            m[0] = fmax(m[0], (double)(D1 * D1));
        }
        // every load is unconditional (a missing neighbour is the cell itself: D1 = 0 cannot raise the
        // max) and U planes are requested before any is used: U*ND loads in flight per thread instead of
        // a full memory round trip per plane
        constexpr int U = 4;
        const int n0 = G.n[0];
        for (int p = pb; p < pe; p += U) {
            T nx[U], nb[ND][U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int pp = min(p + k, pe - 1);
                int pn = pp;                                   // forward neighbour plane on axis 0
                if (pp + 1 < n0 || G.halo_hi) pn = pp + 1;
                else if (G.bc[0] == HJ_BC_PERIODIC) pn = 0;
                const T* row = p0 + (long long)pp * S;
                nx[k] = p0[(long long)pn * S];
#pragma unroll
                for (int d = 1; d < ND; ++d) nb[d][k] = row[off[d]];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (p + k < pe) {
                    {
                        const T D1 = G.inv_dx[0] * (nx[k] - c);
                        m[0] = fmax(m[0], (double)(D1 * D1));
                    }
#pragma unroll
                    for (int d = 1; d < ND; ++d) {
                        const T D1 = G.inv_dx[d] * (nb[d][k] - c);
                        m[d] = fmax(m[d], (double)(D1 * D1));
                    }
                    if (p + k + 1 < pe) c = nx[k];     // inside the chunk the forward plane is the next own plane
                }
            }
        }
    }
    __shared__ double red[NT / 64][ND];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if (threadIdx.x < ND) {
        const int d = threadIdx.x;
        double w = red[0][d];
#pragma unroll
        for (int k = 1; k < NT / 64; ++k) w = fmax(w, red[k][d]);
        partials[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * HJ_MAX_DIM + d] = w;
    }
}

// ---- the pairs a tiled launch cannot see when it reduces max(D1^2) of its own output (FusedArgs::eps_part, hj_fused.h):
// pairs that straddle two tiles (plane axes), two chunks (axis 0), and the periodic wrap pair of every periodic axis.
// The same launch folds the producer's per-workgroup rows: workgroup g leaves ONE row (HJ_MAX_DIM doubles) in rows[g],
// gridDim.x rows in all, which the next substep kernel folds in its prologue (no atomics, no second launch).
template <typename T, int ND> struct SeamArgs {
    const T* y;
    GridArgs<T, ND> G;
    int E[ND], ntile[ND];        // tiling of the producing launch (plane axes)
    int chunk, nchunks;          // its axis-0 chunks: [k*chunk, (k+1)*chunk)
    const double* prod;          // its rows
    int nprod;
    double* rows;
};

template <typename T, int ND>
__global__ __launch_bounds__(1024) void eps_seam_kernel(const SeamArgs<T, ND> A) {
    // m[d]: largest |forward difference| on axis d (the producer's rows hold the same quantity); D1^2 = (inv_dx*m)^2 at the end
    double m[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) m[d] = 0.0;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long gtid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    for (long long i = gtid; i < A.nprod; i += nthreads)
#pragma unroll
        for (int d = 0; d < ND; ++d) m[d] = fmax(m[d], A.prod[(size_t)i * HJ_MAX_DIM + d]);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int n = A.G.n[d];
        const int nint = (d == 0 ? A.nchunks : A.ntile[d]) - 1;            // interior seams
        const int nface = nint + (A.G.bc[d] == HJ_BC_PERIODIC ? 1 : 0);
        const unsigned per_face = (unsigned)(A.G.total / n);               // < 2^31: checked by the host
        const long long items = (long long)nface * per_face;
        // offsets of pair `it` (the cell and its forward neighbour); false if the seam coincides with the end of the axis
        auto locate = [&](long long it, long long& off, long long& nxt) -> bool {
            if (it >= items) return false;
            const unsigned f = (unsigned)(it / per_face);
            unsigned c = (unsigned)(it - (long long)f * per_face);
            int i;
            long long step = A.G.stride[d];
            if ((int)f < nint) {
                i = (d == 0) ? ((int)f + 1) * A.chunk - 1 : min((int)f * A.E[d], n - A.E[d]) + A.E[d] - 1;
                if (i + 1 >= n) return false;
            } else {
                i = n - 1;
                step = -(long long)(n - 1) * A.G.stride[d];
            }
            off = (long long)i * A.G.stride[d];
#pragma unroll
            for (int e = ND - 1; e >= 0; --e) {
                if (e == d) continue;
                const unsigned q = c / (unsigned)A.G.n[e];
                off += (long long)(c - q * (unsigned)A.G.n[e]) * A.G.stride[e];
                c = q;
            }
            nxt = off + step;
            return true;
        };
        constexpr int U = 4;               // pairs in flight per thread
        for (long long it = gtid; it < items; it += U * nthreads) {
            T a[U], b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                long long off = 0, nxt = 0;
                const bool ok = locate(it + u * nthreads, off, nxt);
                a[u] = ok ? A.y[off] : T(0);
                b[u] = ok ? A.y[nxt] : T(0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) m[d] = fmax(m[d], (double)t_abs(b[u] - a[u]));
        }
    }
    __shared__ double red[16][ND];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < ND) {
        const int d = threadIdx.x;
        double w = red[0][d];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) w = fmax(w, red[k][d]);
        const T D1 = A.G.inv_dx[d] * (T)w;                                   // upwind_first_weno5a.py:153-156 on the largest |difference|
        A.rows[(size_t)blockIdx.x * HJ_MAX_DIM + d] = (double)(D1 * D1);
    }
}

// max over the per-workgroup partials -> ND values of dtype T (one workgroup)
template <typename T>
__global__ __launch_bounds__(256) void partials_to_values_kernel(const double* __restrict__ partials, int nblocks,
                                                                 T* __restrict__ out, int nd) {
    __shared__ double red[4][HJ_MAX_DIM];
    double m[HJ_MAX_DIM] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
#pragma unroll
        for (int d = 0; d < HJ_MAX_DIM; ++d)
            if (d < nd) m[d] = fmax(m[d], partials[(size_t)b * HJ_MAX_DIM + d]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < HJ_MAX_DIM; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < nd) {
        const int d = threadIdx.x;
        out[d] = (T)fmax(fmax(red[0][d], red[1][d]), fmax(red[2][d], red[3][d]));
    }
}

// keys -> values of dtype T (single thread)
template <typename T>
__global__ void keys_to_values_kernel(const unsigned long long* keys, T* out, int n) {
    if (threadIdx.x < n) {
        unsigned long long k = keys[threadIdx.x];
        unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
        out[threadIdx.x] = (T)__longlong_as_double((long long)b);
    }
}

// ---- max alpha per dim with p = 0: stepBound of a Hamiltonian whose alpha ignores the data -- and, after a range pass, of one whose alpha
// reads the costate range (hj_rtc.hip): then the kernel sits between the range pass and the first stage of EVERY step.
// Round 5: (1) a thread owns one in-plane cell and a run of planes (the column constants -- for the Dubins-like systems cos / sin of the
// heading -- and the index decode are paid once per run, not once per node); (2) NO same-address atomics for the maxima: 512 workgroups x
// (ND+1) atomicMax on one cache line were the whole kernel (23.7 us at 201^3 = 2048 atomics x ~12 ns; profiles/r05_range_path.txt) --
// every workgroup stores its partial maxima, counts itself with ONE atomic add, and the last one reduces the partials; (3) that last
// workgroup also hands the result to the host through page-locked memory (host_out[0..HJ_MAX_DIM] = the keys, host_out[7] = seq with
// release semantics): the host polls it instead of launching a copy and waiting for the stream (host_out == nullptr: keys only).
// `done` must be zero at launch; `partials`: gridDim.x * 8 doubles.
struct DxArgs { double dx[HJ_MAX_DIM]; };
// deltaT on the device (hj_rk_step with a range-dependent alpha): min(factorCFL * stepBound, tspan[1] - t, maxStep), ode_cfl_3.py:142,
// written to *dt_dev for the stage kernels already enqueued behind this one (FusedArgs::dt_dev) and -- as bits -- to host_out[5] (the
// stepBound) and host_out[6] (deltaT): the host uses THESE values, so both sides step with the same number.  dt_dev == nullptr: off.
struct DtArgs { double factor, span, max_step; double* dt_dev; };
template <typename T, typename HAM>
__global__ __launch_bounds__(1024) void alpha_bound_kernel(GridArgs<T, HAM::ND> G, HamTables<T> P,
                                                          unsigned long long* keys, DxArgs DX, double* partials,
                                                          unsigned long long* done, unsigned long long* host_out, unsigned long long seq,
                                                          DtArgs DT) {
    constexpr int ND = HAM::ND;
    const double* dx = DX.dx;
    // m[0..ND): max alpha_d (global LF);  m[ND]: max over nodes of sum_d alpha_d/dx_d (local LF variants)
    double m[ND + 1];
#pragma unroll
    for (int d = 0; d <= ND; ++d) m[d] = -1e300;
    const long long n0 = G.n[0], plane_cells = G.total / n0;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    // runs of planes per in-plane cell: as many as the launch has threads for (at least 1, at most one plane per run)
    const long long nrun = min(n0, max(1ll, nthreads / plane_cells));
    const long long items = plane_cells * nrun;
    T one[ND], p[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { one[d] = T(1); p[d] = T(0); }
    for (long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x; w < items; w += nthreads) {
        const long long run = w / plane_cells;
        long long ip = w - run * plane_cells;
        int idx[ND];
        idx[0] = 0;
#pragma unroll
        for (int d = ND - 1; d >= 1; --d) {
            const long long q = ip / G.n[d];
            idx[d] = (int)(ip - q * G.n[d]);
            ip = q;
        }
        const auto cc = HAM::cell(P, idx, one);
        const int pa = (int)(run * n0 / nrun), pb = (int)((run + 1) * n0 / nrun);
#pragma unroll 4
        for (int pl = pa; pl < pb; ++pl) {
            T H, a[ND];
            HAM::eval(P, cc, HAM::plane(P, pl, one), one, p, H, a);
            double inv = 0.0;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                m[d] = fmax(m[d], (double)a[d]);
                inv += (double)a[d] / dx[d];
            }
            m[ND] = fmax(m[ND], inv);
        }
    }
    __shared__ double red[16][ND + 1];       // up to 1024 threads (the loop is a chain of dependent loads per plane: occupancy hides it)
    __shared__ int last_flag;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d <= ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if (threadIdx.x <= ND) {
        const int d = threadIdx.x;
        double w = red[0][d];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) w = fmax(w, red[k][d]);
        __hip_atomic_store(partials + (size_t)blockIdx.x * 8 + d, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: leaves the XCD's L2
    }
    // the partial stores come from lanes 0..ND of wave 0.  No fence: __threadfence() is an L2 write-back + invalidate per WAVE that executes
    // it (3.5 us; with every wave fencing the kernel's time grew with its thread count: 287 us at 1 M threads, r05_run24).  The MI355X
    // guide's hand-off recipe for sc1 stores: the storing wave waits for its stores, ONE lane adds to the counter with an agent-scope atomic, the workgroup whose add
    // came last reads with sc1 loads)
    if (wv == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            // RELEASE at agent scope: ONE fence per workgroup -- the wave that stored the partials orders them before its
            // count, by the HIP memory model rather than by what the sc1 stores happen to do on gfx950 (the s_waitcnt above stays: it is free)
            const unsigned long long before = __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = before == (unsigned long long)gridDim.x - 1ull;
        }
    }
    __syncthreads();
    if (!last_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ... and the last workgroup acquires before it reads the others' partials
#pragma unroll
    for (int d = 0; d <= ND; ++d) m[d] = -1e300;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x) {
#pragma unroll
        for (int d = 0; d <= ND; ++d)
            m[d] = fmax(m[d], __hip_atomic_load(partials + (size_t)b * 8 + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d <= ND; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the sum goes to keys[HJ_MAX_DIM], whatever ND is; a dimension nobody evaluated keeps key 0
#pragma unroll
        for (int d = 0; d <= ND; ++d) {
            double w = red[0][d];
            for (int kk = 1; kk < (int)(blockDim.x >> 6); ++kk) w = fmax(w, red[kk][d]);
            const unsigned long long k = w > -1e299 ? max_key(w) : 0ull;
            keys[d == ND ? HJ_MAX_DIM : d] = k;
            if (host_out != nullptr) host_out[d == ND ? HJ_MAX_DIM : d] = k;
        }
        if (DT.dt_dev != nullptr) {
            double inv = 0.0;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double w = red[0][d];
                for (int kk = 1; kk < (int)(blockDim.x >> 6); ++kk) w = fmax(w, red[kk][d]);
                inv += w / dx[d];
            }
            const double sb = 1.0 / inv;
            const double dtv = fmin(fmin(DT.factor * sb, DT.span), DT.max_step);
            __hip_atomic_store(DT.dt_dev, dtv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (host_out != nullptr) {
                host_out[5] = (unsigned long long)__double_as_longlong(sb);
                host_out[6] = (unsigned long long)__double_as_longlong(dtv);
            }
        }
        if (host_out != nullptr) {
            __threadfence_system();
            __hip_atomic_store(host_out + 7, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// deltaT from a reduced bound slot (hj_rk_step, local Lax-Friedrichs variants of a range-reading Hamiltonian: the bound pass of the substep
// kernel left max_x sum_d alpha_d / dx_d in the slot's generic form): stepBound = 1 / sum_d key_d / dx_d, deltaT as DtArgs says, both to
// the device word the first stage reads and -- as bits -- to the page-locked words the host polls (see alpha_bound_kernel)
template <int UNUSED>      // (a template: the header is included by several translation units)
__global__ void bound_to_dt_kernel(const unsigned long long* keys, int nd, DxArgs DX, DtArgs DT, unsigned long long* host_out, unsigned long long seq) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double inv = 0.0;
    for (int d = 0; d < nd; ++d) inv += key_value(__hip_atomic_load(keys + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) / DX.dx[d];
    const double sb = 1.0 / inv;
    const double dtv = fmin(fmin(DT.factor * sb, DT.span), DT.max_step);
    __hip_atomic_store(DT.dt_dev, dtv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    host_out[5] = (unsigned long long)__double_as_longlong(sb);
    host_out[6] = (unsigned long long)__double_as_longlong(dtv);
    __threadfence_system();
    __hip_atomic_store(host_out + 7, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- direct (untiled) fused substep: one thread per cell, stencil neighbours straight from
// global memory (L1/L2 absorb the reuse).  Same arithmetic as fused_substep_kernel.
template <typename T, int ND> struct DirectArgs {
    const T* y;
    const T* y0;
    T* out;
    const T* max_d1sq;
    unsigned long long* bound;
    GridArgs<T, ND> G;
    long long cell_begin, cell_end;   // linear cell range (whole axis-0 planes)
    int stage, restrict_sign, post_op;
    T dt;
    T sc[ND];                     // costate scale (see hj_device.h): 1/(60dx) as-shipped WENO5, else 1
    HamTables<T> ham;
};

// The 7-point stencils of one cell along every dimension, straight from global memory, every load issued unconditionally
// (round 3; shared by direct_substep_kernel and, from round 4, term_kernel -- which went through line_value(), whose boundary
// branches sit in front of each of the 6*ND neighbour loads and serialise them: 223-325 us per term at 201^3).  The neighbour
// offsets are formed with selects (periodic: the wrapped cell; extrapolated: the edge cell, fixed up afterwards), all loads
// go out back to back, and only waves that touch an extrapolated edge run the ghost fix-up (two more loads per ghost value).
template <typename T, int ND>
__device__ __forceinline__ void gather_stencils(const GridArgs<T, ND>& G, const T* pc0, const int* idx, T centre, T (&v)[ND][7]) {
    bool ghost = false;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int n = G.n[d], i = idx[d];
        const bool per = G.bc[d] == HJ_BC_PERIODIC;
        const bool hlo = d == 0 && G.halo_lo, hhi = d == 0 && G.halo_hi;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (k == 3) { v[d][k] = centre; continue; }
            int j = i + k - 3;
            if (j < 0 && !hlo) {
                if (per) j += n; else { j = 0; ghost = true; }
            } else if (j >= n && !hhi) {
                if (per) j -= n; else { j = n - 1; ghost = true; }
            }
            v[d][k] = pc0[(long long)(j - i) * G.stride[d]];
        }
    }
    if (__any(ghost ? 1 : 0)) {
        // ghost cells of an extrapolated boundary: edge + k*slope from the edge cell and its inner neighbour
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int n = G.n[d], i = idx[d];
            if (G.bc[d] == HJ_BC_PERIODIC) continue;
            const bool hlo = d == 0 && G.halo_lo, hhi = d == 0 && G.halo_hi;
            const T* line = pc0 - (long long)i * G.stride[d];
            if (i < HJ_STENCIL && !hlo) {
                const T e = line[0], in = line[G.stride[d]];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (i + k - 3 < 0) v[d][k] = ghost_value(e, in, T(3 - k - i) * G.km[d]);
            }
            if (i + HJ_STENCIL >= n && !hhi) {
                const T e = line[(long long)(n - 1) * G.stride[d]], in = line[(long long)(n - 2) * G.stride[d]];
#pragma unroll
                for (int k = 4; k < 7; ++k)
                    if (i + k - 3 >= n) v[d][k] = ghost_value(e, in, T(i + k - 3 - n + 1) * G.km[d]);
            }
        }
    }
}

// ---- derivL[d], derivR[d] for EVERY dimension in one launch (round 4; hj_lf_split_begin: the split path of termLaxFriedrichs,
// computeGradients): the stencils of a cell are gathered once (19 loads in 3-D instead of 3 x 7 in three launches that each
// stream the whole array), 2*ND arrays are written, and the 4*ND reductions {-min L, max L, -min R, max R} per dimension go to
// keys[4*d + k] as upwind_kernel leaves them.  The intended WENO5's epsilon is formed on the device (product, then sum, as the
// host formed it for upwind_kernel: no per-dimension host synchronisation any more).  Same upwind<SCHEME>, same bits.
template <typename T, int ND> struct UpwindAllArgs {
    GridArgs<T, ND> G;
    T* dL[ND];
    T* dR[ND];
    const T* max_d1sq;            // HJ_WENO5 only
    unsigned long long* keys;     // 4*ND atomicMax keys, or null
};
template <typename T, int ND, int SCHEME>
__global__ __launch_bounds__(256) void upwind_all_kernel(const T* __restrict__ y, const UpwindAllArgs<T, ND> A) {
    T eps[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        eps[d] = T(0);
        if constexpr (SCHEME == HJ_WENO5) eps[d] = weno_eps_uncontracted<T>(A.max_d1sq[d]);
    }
    double m[4 * ND];
#pragma unroll
    for (int k = 0; k < 4 * ND; ++k) m[k] = -1e300;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < A.G.total; t += (long long)gridDim.x * blockDim.x) {
        int idx[ND];
        decode<T, ND>(A.G, t, idx);
        const T* pc0 = y + t;
        T v[ND][7];
        gather_stencils<T, ND>(A.G, pc0, idx, pc0[0], v);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            T L, R;
            upwind<SCHEME, T>(v[d], A.G.K[d], eps[d], L, R);
            A.dL[d][t] = L;
            A.dR[d][t] = R;
            m[4 * d + 0] = fmax(m[4 * d + 0], -(double)L); m[4 * d + 1] = fmax(m[4 * d + 1], (double)L);
            m[4 * d + 2] = fmax(m[4 * d + 2], -(double)R); m[4 * d + 3] = fmax(m[4 * d + 3], (double)R);
        }
    }
    if (A.keys) {
        __shared__ double red[4][4 * ND];
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 4 * ND; ++k) {
            const double w = wave_max(m[k]);
            if (lane == 0) red[wv][k] = w;
        }
        __syncthreads();
        if (threadIdx.x < 4 * ND) {
            const int k = threadIdx.x;
            const double w = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
            if (w > -1e299) key_max(A.keys + k, w);
        }
    }
}

// Round 3: every stencil load is issued unconditionally.  The first version went through line_value(), whose boundary
// branches sit in front of each of the 6*ND neighbour loads: the loads were serialised behind one another (a memory round
// trip each) and the kernel took ~40 us on a 51^3 grid.  Now the neighbour offsets are formed with selects (periodic: the
// wrapped cell; extrapolated: the edge cell, fixed up afterwards), all loads go out back to back, and only waves that touch
// an extrapolated edge run the ghost fix-up (two more loads per ghost value).  Same per-cell functions and stage expressions
// as the tiled kernels: the results are theirs bit for bit.  It is also the default for SMALL grids (launch_cfg).
template <typename T, typename HAM, int SCHEME>
__global__ __launch_bounds__(256) void direct_substep_kernel(const DirectArgs<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr bool NP = np_order(SCHEME);
    double amax[ND];
    T eps[ND];
    WenoK<T> wk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        amax[d] = -1e300;
        eps[d] = T(0);
        wk[d].c13 = T(0); wk[d].c4 = T(0);
        if constexpr (SCHEME == HJ_WENO5) {
            eps[d] = T(1e-6) * A.max_d1sq[d] + Lim<T>::tiny;
            wk[d] = weno_consts<T>(eps[d], A.G.K[d]);
        }
    }
    T ca = T(0), cb = T(1);
    if (A.stage == HJ_STAGE_RK3_HALF) { ca = T(0.75); cb = T(0.25); }
    else if (A.stage == HJ_STAGE_RK3_FULL) { ca = T(1.0 / 3.0); cb = T(2.0 / 3.0); }
    else if (A.stage == HJ_STAGE_RK2_FULL) { ca = T(0.5); cb = T(0.5); }
    const bool use_y0 = A.stage >= HJ_STAGE_RK3_HALF;
    for (long long t = A.cell_begin + blockIdx.x * (long long)blockDim.x + threadIdx.x;
         t < A.cell_end; t += (long long)gridDim.x * blockDim.x) {
        int idx[ND];
        decode<T, ND>(A.G, t, idx);
        const T* pc0 = A.y + t;
        T v[ND][7];
        const T centre = pc0[0];
        const T y0v = use_y0 ? A.y0[t] : T(0);
        const typename HAM::Cell hc = HAM::cell(A.ham, idx, A.sc);
        const typename HAM::Plane hp = HAM::plane(A.ham, idx[0], A.sc);
        gather_stencils<T, ND>(A.G, pc0, idx, centre, v);
        T pc[ND], hd[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) upwind_cd<SCHEME, T>(v[d], A.G.K[d], eps[d], wk[d], pc[d], hd[d]);
        T alpha[ND];
        T ydot = lf_ydot<NP, HAM>(A.ham, hc, hp, A.sc, pc, hd, alpha);
#pragma unroll
        for (int d = 0; d < ND; ++d) amax[d] = fmax(amax[d], (double)alpha[d]);
        // termRestrictUpdate clamp, written like the tiled kernels' (a NaN stays a NaN)
        if (A.restrict_sign > 0) ydot = (ydot < T(0)) ? T(0) : ydot;
        else if (A.restrict_sign < 0) ydot = (ydot > T(0)) ? T(0) : ydot;
        T o;
        if (A.stage == HJ_STAGE_YDOT) o = ydot;
        else {
            o = rk_stage_out<NP>(A.stage, ca, cb, A.dt, y0v, centre, ydot);
            if (A.post_op) o = post_step(A.post_op, o, use_y0 ? y0v : centre);
        }
        A.out[t] = o;
    }
    __shared__ double red[4][ND];
    if (A.bound) {   // nobody reads the bound of hj_rk_step's launches (static step bound)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    #pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double w = wave_max(amax[d]);
            if (lane == 0) red[wv][d] = w;
        }
        __syncthreads();
        if (threadIdx.x < ND) {
            const int d = threadIdx.x;
            const double w = fmax(fmax(red[0][d], red[1][d]), fmax(red[2][d], red[3][d]));
            if (w > -1e299) key_max(A.bound + d, w / (double)A.sc[d]);
        }
    }
}


// ---- ONE launch for a whole odeCFL2 / odeCFL3 step on SMALL grids (round 6).
// 51^3 (BASELINE C1, every notebook of the reference: ValueFuncs/hji_solver.py:536-542 steps it one odeCFL3 call at a time) costs three
// dependent launches of direct_substep_kernel at the launch floor (7.3 us each: dispatch, argument loads, index arithmetic, table loads, the
// stencil gather, arithmetic, store -- a chain of latencies, 0.05 of 8 TB/s).  Here the stages run inside one cooperative launch, a grid-wide
// barrier between them; a thread keeps its cell's index arithmetic and Hamiltonian constants across the stages.  Same per-cell functions
// and stage expressions as direct_substep_kernel: the results are its results bit for bit.
// MEASURED (profiles/r06_small_grids.txt) and left OPT-IN (HJ_COOP=1): 51^3 30.1 us per RK3 step against 21.6 for the three launches, 41^3 21.8
// against 17.9, 31^3 15.1 against 15.5 -- a grid barrier over 519 workgroups costs more (4-6 us, MI355X_MICROARCH.md "barrier-xcd") than the
// kernel boundary it replaces (1.5-1.9 us), and hipLaunchCooperativeKernel another 15-19 us of host time per launch (75 us per step).
//
// Barrier (every workgroup is resident: hipLaunchCooperativeKernel): the stage output is stored with agent-scope (sc1, write-through)
// stores; every wave drains its stores, the workgroup meets, ONE lane releases at agent scope and adds 1 to the counter of its XCD (32 adds
// per counter, not 256 same-address ones: ~12 ns each, profiles/r05_range_path.txt); the last arriver of an XCD adds to the global
// counter, everybody polls that; after the barrier every wave acquires (invalidates the XCD-local L2 lines of the stage buffers).
// The counters only grow: `base` is what they read when the launch starts (the host keeps count; launches on a ctx are stream ordered).
struct CoopSync {
    unsigned long long xcd[8][16];      // one cache line per XCD: arrivals of its workgroups
    unsigned long long all[16];         // XCDs that are complete
};
template <typename T, int ND> struct CoopArgs {
    const T* y;                         // state at the start of the step
    T* s1;                              // y1 = y + dt L(y)                      (order 1: the result)
    T* s2;                              // order 2: the result; order 3: y_half = (3 y + y1 + dt L(y1)) / 4
    T* out;                             // order 3: the result
    GridArgs<T, ND> G;
    int order, restrict_sign, post_op;
    T dt;
    T sc[ND];
    HamTables<T> ham;
    CoopSync* sync;
    unsigned long long base_xcd[8], base_all;      // counter values when the launch starts
    unsigned nper_xcd[8];                          // workgroups on each XCD (blockIdx & 7)
    unsigned nxcd;                                 // XCDs that have a workgroup
};

__device__ __forceinline__ void coop_grid_barrier(CoopSync* S, const unsigned long long* base_xcd, unsigned long long base_all,
                                                  const unsigned* nper, unsigned nxcd, unsigned phase /* 1, 2, .. */) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stage stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned x = blockIdx.x & 7u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned long long seen = __hip_atomic_fetch_add(&S->xcd[x][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen + 1ull == base_xcd[x] + (unsigned long long)phase * nper[x])
            __hip_atomic_fetch_add(&S->all[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long want = base_all + (unsigned long long)phase * nxcd;
        while (__hip_atomic_load(&S->all[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        // ONE lane acquires for the workgroup (the invalidate is per CU; every wave doing it costs 4x: MI355X_MICROARCH.md), the others read after the barrier
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <typename T, typename HAM, int SCHEME, int CPT>
__global__ __launch_bounds__(256) void coop_rk_kernel(const CoopArgs<T, HAM::ND> A) {
    constexpr int ND = HAM::ND;
    constexpr bool NP = np_order(SCHEME);
    static_assert(SCHEME != HJ_WENO5, "the intended WENO5 needs a grid-wide reduction between the stages");
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long t0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    // this thread's cells (CPT of them at most): index, Hamiltonian constants -- kept across the stages
    int idx[CPT][ND];
    typename HAM::Cell hc[CPT];
    typename HAM::Plane hp[CPT];
    bool real[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const long long t = t0 + k * nthreads;
        real[k] = t < A.G.total;
        decode<T, ND>(A.G, real[k] ? t : 0, idx[k]);
        hc[k] = HAM::cell(A.ham, idx[k], A.sc);
        hp[k] = HAM::plane(A.ham, idx[k][0], A.sc);
    }
    T eps[ND];
    WenoK<T> wk[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) { eps[d] = T(0); wk[d].c13 = T(0); wk[d].c4 = T(0); }
    T ystart[CPT];
    for (int st = 1; st <= A.order; ++st) {
        // stage st: source, destination, stage expression (hj_launch.h: fill_fused_args)
        const T* src = st == 1 ? A.y : (st == 2 ? A.s1 : A.s2);
        T* dst = st == A.order ? (A.order == 1 ? A.s1 : (A.order == 2 ? A.s2 : A.out)) : (st == 1 ? A.s1 : A.s2);
        const int stage = st == 1 ? HJ_STAGE_EULER : (st == 2 ? (A.order == 2 ? HJ_STAGE_RK2_FULL : HJ_STAGE_RK3_HALF) : HJ_STAGE_RK3_FULL);
        T ca = T(0), cb = T(1);
        if (stage == HJ_STAGE_RK3_HALF) { ca = T(0.75); cb = T(0.25); }
        else if (stage == HJ_STAGE_RK3_FULL) { ca = T(1.0 / 3.0); cb = T(2.0 / 3.0); }
        else if (stage == HJ_STAGE_RK2_FULL) { ca = T(0.5); cb = T(0.5); }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            if (!real[k]) continue;
            const long long t = t0 + k * nthreads;
            const T* pc0 = src + t;
            T v[ND][7];
            const T centre = pc0[0];
            if (st == 1) ystart[k] = centre;
            const T y0v = st == 1 ? T(0) : ystart[k];
            gather_stencils<T, ND>(A.G, pc0, idx[k], centre, v);
            T pc[ND], hd[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) upwind_cd<SCHEME, T>(v[d], A.G.K[d], eps[d], wk[d], pc[d], hd[d]);
            T alpha[ND];
            T ydot = lf_ydot<NP, HAM>(A.ham, hc[k], hp[k], A.sc, pc, hd, alpha);
            if (A.restrict_sign > 0) ydot = (ydot < T(0)) ? T(0) : ydot;
            else if (A.restrict_sign < 0) ydot = (ydot > T(0)) ? T(0) : ydot;
            T o = rk_stage_out<NP>(stage, ca, cb, A.dt, y0v, centre, ydot);
            if (st == A.order && A.post_op) o = post_step(A.post_op, o, st == 1 ? centre : y0v);
            if (st == A.order) dst[t] = o;
            else __hip_atomic_store(dst + t, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // read by other XCDs after the barrier
        }
        if (st < A.order) coop_grid_barrier(A.sync, A.base_xcd, A.base_all, A.nper_xcd, A.nxcd, (unsigned)st);
    }
}

// ---- split-path epilogue: what termLaxFriedrichs / artificialDissipationGLF do AFTER the user's hamFunc /
// partialFunc callbacks have run (term_lax_friedrich.py:122-128, artificial_diss_glf.py:91-104), as one pass:
//     diss = sum_d (0.5*(derivR_d - derivL_d))*alpha_d          (in dimension order, like the reference's loop)
//     out  = -(ham - diss)   if ham is given (the term's ydot), else diss (the dissipation function's result)
// alpha_d is an array (alpha[d] != null) or the scalar alpha_s[d]; keys[d] collects max alpha_d of the
// array-valued ones (the reference reduces only those: artificial_diss_glf.py:101-104).
// Contraction is off: the result is bit-identical to the reference's NumPy expressions.
template <typename T> struct SplitEndArgs {
    const T* dL[HJ_MAX_DIM];
    const T* dR[HJ_MAX_DIM];
    const T* alpha[HJ_MAX_DIM];
    T alpha_s[HJ_MAX_DIM];
    const T* ham;
    T* out;
    unsigned long long* keys;
    long long n;
    int nd;
};
template <typename T>
__global__ __launch_bounds__(256) void lf_split_end_kernel(const SplitEndArgs<T> A) {
#pragma clang fp contract(off)
    double m[HJ_MAX_DIM] = {-1e300, -1e300, -1e300, -1e300};
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < A.n;
         t += (long long)gridDim.x * blockDim.x) {
        T diss = T(0);
#pragma unroll
        for (int d = 0; d < HJ_MAX_DIM; ++d) {
            if (d >= A.nd) break;
            T a = A.alpha_s[d];
            if (A.alpha[d]) { a = A.alpha[d][t]; m[d] = fmax(m[d], (double)a); }
            const T half = T(0.5) * (A.dR[d][t] - A.dL[d][t]);
            diss = diss + half * a;
        }
        A.out[t] = A.ham ? -(A.ham[t] - diss) : diss;
    }
    __shared__ double red[4][HJ_MAX_DIM];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 0; d < HJ_MAX_DIM; ++d) {
        const double w = wave_max(m[d]);
        if (lane == 0) red[wv][d] = w;
    }
    __syncthreads();
    if ((int)threadIdx.x < A.nd && A.alpha[threadIdx.x]) {
        const int d = threadIdx.x;
        const double w = fmax(fmax(red[0][d], red[1][d]), fmax(red[2][d], red[3][d]));
        if (w > -1e299) key_max(A.keys + d, w);
    }
}

// ---- the array expressions of one odeCFLn stage for an arbitrary schemeFunc (ode_cfl_3.py:151,184-193,
// 226-241; ode_cfl_2.py:151,184-201), one pass instead of 3-5 elementwise launches; contraction off, so
// the values are those of the reference's NumPy expressions:
//   1: y + dt*z      2: 0.25*(3*x0 + (y + dt*z))      3: (1/3)*(x0 + 2*(y + dt*z))      4: 0.5*(x0 + (y + dt*z))
template <typename T>
__global__ __launch_bounds__(256) void rk_combine_kernel(int mode, T dt, const T* __restrict__ x0,
                                                         const T* __restrict__ y, const T* __restrict__ z,
                                                         T* __restrict__ out, long long n) {
#pragma clang fp contract(off)
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x) {
        const T step = dt * z[t];
        const T e = y[t] + step;
        T o = e;
        if (mode == 2) { const T a = T(3) * x0[t]; o = T(0.25) * (a + e); }
        else if (mode == 3) { const T a = T(2) * e; o = (T(1) / T(3)) * (x0[t] + a); }
        else if (mode == 4) { o = T(0.5) * (x0[t] + e); }
        out[t] = o;
    }
}

// ---- HJIPDE_solve post-step operators and NaN guard
template <typename T>
__global__ void minmax_kernel(T* __restrict__ y, const T* __restrict__ o, long long n, int op) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x) {
        const T a = y[t], b = o[t];
        // NumPy minimum/maximum propagate NaN (hji_solver.py:573-580)
        T r;
        if (a != a) r = a;
        else if (b != b) r = b;
        else if (op == HJ_OP_MIN) r = a < b ? a : b;
        else if (op == HJ_OP_MAX) r = a > b ? a : b;
        else r = a > -b ? a : -b;
        y[t] = r;
    }
}

template <typename T>
__global__ void any_nan_kernel(const T* __restrict__ y, long long n, int* flag) {
    bool bad = false;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n;
         t += (long long)gridDim.x * blockDim.x) {
        const T a = y[t];
        bad = bad || (a != a);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace hj
