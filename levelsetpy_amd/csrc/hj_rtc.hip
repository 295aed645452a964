// Hamiltonians compiled at RUN time (round 4): the fused substep kernel for a user-supplied H(x, p) / alpha(x).
//
// The reference's hamFunc / partialFunc are arbitrary Python callables (ExplicitIntegration/Term/term_lax_friedrich.py:111,
// Dissipation/artificial_diss_glf.py:98); the library fuses the three systems it was built with, everything else took the
// split path (derivative kernels -> Python callbacks -> dissipation kernel: 25x the fused step at 201^3).  Here the caller
// hands over the BODY of the Hamiltonian as a device expression once (hj_ham_register); the library wraps it in the
// interface the kernels expect from a Hamiltonian type (hj_device.h: Cell / Plane / eval), compiles
// fused_pair_kernel<double, HamUser, SCHEME, ...> and alpha_bound_kernel<double, HamUser> with hipRTC for gfx950 (1-2 s per
// scheme, on first use) and launches them through the module API with the very FusedArgs block the built-in
// instantiations get (hj_launch.h).  Every consumer of a Hamiltonian id -- hj_lf_term, hj_rk_substep, hj_rk_step,
// hj_rk_integrate, the slab steppers -- then works with the new id.
//
// What the expression may use:  x[d] (node coordinates, d = 0 .. ND-1), p[d] (costates), par[k] (the ham_params of the
// call), t-independent device math (sin, cos, fabs, sqrt, fmin, fmax ...); what it must set:  H  and  alpha[d] for every d.
// alpha must not depend on p (true of every system the reference ships: dubins_relative.py:106-111,
// dubins_absolute.py:150-170, double_integrator.py:84-89, bird.py:346): the CFL bound is then a property of the grid
// (hj_static_step_bound) and a time step needs no host synchronisation.  fp64, 2-D and 3-D grids.
#include <dlfcn.h>
#include <errno.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <sstream>
#include "hj_launch.h"
#include "hj_fusedv.h"

namespace hjh {

typedef struct _hiprtcProgram* rtcProgram;
struct Rtc {
    void* handle = nullptr;
    int (*CreateProgram)(rtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*CompileProgram)(rtcProgram, int, const char**) = nullptr;
    int (*AddNameExpression)(rtcProgram, const char*) = nullptr;
    int (*GetLoweredName)(rtcProgram, const char*, const char**) = nullptr;
    int (*GetCodeSize)(rtcProgram, size_t*) = nullptr;
    int (*GetCode)(rtcProgram, char*) = nullptr;
    int (*GetProgramLogSize)(rtcProgram, size_t*) = nullptr;
    int (*GetProgramLog)(rtcProgram, char*) = nullptr;
    int (*DestroyProgram)(rtcProgram*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static Rtc g_rtc;

static int rtc_load(const char* path) {
    if (g_rtc.handle) return HJ_OK;
    const char* names[] = {path, "libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(HJ_ESTATE, "cannot dlopen hipRTC: %s", dlerror());
#define HJ_SYM(field, name)                                                     \
    *(void**)(&g_rtc.field) = dlsym(h, name);                                   \
    if (!g_rtc.field) return fail(HJ_ESTATE, "hipRTC symbol %s missing", name);
    HJ_SYM(CreateProgram, "hiprtcCreateProgram")
    HJ_SYM(CompileProgram, "hiprtcCompileProgram")
    HJ_SYM(AddNameExpression, "hiprtcAddNameExpression")
    HJ_SYM(GetLoweredName, "hiprtcGetLoweredName")
    HJ_SYM(GetCodeSize, "hiprtcGetCodeSize")
    HJ_SYM(GetCode, "hiprtcGetCode")
    HJ_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    HJ_SYM(GetProgramLog, "hiprtcGetProgramLog")
    HJ_SYM(DestroyProgram, "hiprtcDestroyProgram")
    HJ_SYM(GetErrorString, "hiprtcGetErrorString")
#undef HJ_SYM
    g_rtc.handle = h;
    return HJ_OK;
}

struct UserKernel { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; int occ = 0; size_t lds_granted = 0; };
struct UserHam {
    std::string name, body, column_body, include_dir, rtc_path;
    int ndim = 0, nparams = 0, ncol = 0;
    std::map<int, UserKernel> substep;      // key: (scheme * 4 + MODE) * 2 + (big shape ? 1 : 0)
    std::map<int, bool> big_spills;         // key: scheme * 4 + MODE -- the big shape needed scratch for this expression
    bool no_big_lds = false;                // the runtime refused > 64 KB of dynamic LDS for a module function
    UserKernel alpha;
};
static std::vector<UserHam> g_user;

static UserHam* user_of(int ham) {
    const int i = ham - HJ_HAM_USER_BASE;
    return (i >= 0 && i < (int)g_user.size()) ? &g_user[(size_t)i] : nullptr;
}
bool user_ham_valid(int ham) { return user_of(ham) != nullptr; }
int user_ham_ndim(int ham) { const UserHam* u = user_of(ham); return u ? u->ndim : -1; }
int user_ham_npar(int ham) { const UserHam* u = user_of(ham); return u ? u->nparams : 0; }

// the translation unit hipRTC compiles: the kernel headers + the Hamiltonian type around the caller's expression
static std::string user_source(const UserHam& u, int id) {
    std::ostringstream o;
    o << "#include \"hj_fusedv.h\"\n#include \"hj_split.h\"\nnamespace hj {\n"
         "template <typename T> struct HamUser {\n"
         "    static constexpr int ND = " << u.ndim << ";\n"
         "    static constexpr int ID = " << id << ";\n"
         "    static constexpr unsigned PLANE_DEP = 0xFu;     // any alpha may vary along the march\n"
         "    static constexpr int NCOL = " << (u.ncol > 0 ? u.ncol : 1) << ";\n"
         "    struct Cell { T x[ND]; T col[NCOL]; };\n    struct Plane { T x0; };\n    using Raw = Cell;\n"
         "    __device__ static __forceinline__ Raw cell_raw(const HamTables<T>& P, const int* idx) {\n"
         "        Cell c; c.x[0] = T(0);\n"
         "        for (int d = 1; d < ND; ++d) c.x[d] = P.coord[d][idx[d]];\n"
         "        for (int k = 0; k < NCOL; ++k) c.col[k] = T(0);\n        return c;\n    }\n"
         "    __device__ static __forceinline__ Raw cell_raw_next(const HamTables<T>& P, const int* idx, const Raw& first) {\n"
         "        Cell c = first; c.x[ND - 1] = P.coord[ND - 1][idx[ND - 1]]; return c;\n    }\n"
         "    // once per grid COLUMN, outside the march: the caller's column expression (col[k] from x[1..], par)\n"
         "    __device__ static __forceinline__ Cell cell_fin(const HamTables<T>& P, const Raw& r, const T*) {\n"
         "        Cell c = r;\n        T x[ND], col[NCOL];\n        x[0] = T(0);\n"
         "        for (int d = 1; d < ND; ++d) x[d] = r.x[d];\n"
         "        for (int k = 0; k < NCOL; ++k) col[k] = T(0);\n"
         "        const T* par = P.par;\n        (void)par; (void)x;\n"
         "        {\n#line 1 \"" << u.name << " (column)\"\n" << u.column_body << "\n        }\n"
         "        for (int k = 0; k < NCOL; ++k) c.col[k] = col[k];\n        return c;\n    }\n"
         "    __device__ static __forceinline__ Cell cell(const HamTables<T>& P, const int* idx, const T* sc) { return cell_fin(P, cell_raw(P, idx), sc); }\n"
         "    __device__ static __forceinline__ Plane plane(const HamTables<T>& P, int i0, const T*) { Plane u; u.x0 = P.coord[0][i0]; return u; }\n"
         "    template <bool NP = false>\n"
         "    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane& pl, const T* sc, const T* q, T& H, T* alpha) {\n"
         "        T x[ND], p[ND];\n        x[0] = pl.x0;\n"
         "        for (int d = 1; d < ND; ++d) x[d] = c.x[d];\n"
         "        for (int d = 0; d < ND; ++d) { p[d] = sc[d] * q[d]; alpha[d] = T(0); }\n"
         "        const T* par = P.par;\n        const T* col = c.col;\n        (void)col; (void)par;\n        H = T(0);\n"
         "        {\n#line 1 \"" << u.name << "\"\n" << u.body << "\n        }\n"
         "        for (int d = 0; d < ND; ++d) alpha[d] = sc[d] * alpha[d];     // the kernels carry alpha in the stencil's scale\n"
         "    }\n};\n}\n";
    return o.str();
}

// ---- code-object cache on disk: a registered expression is compiled once per (source, kernel headers, options), not once per
// process.  Directory: $HJ_RTC_CACHE ("0" / "off" disables), else $XDG_CACHE_HOME/levelsetpy_amd, else $HOME/.cache/levelsetpy_amd.
// File <key>.hjco = "HJCO1\n" + lowered kernel name + "\n" + code object; key = FNV-1a of the generated source, the name
// expression, the compile options, the TEXT of the kernel headers it includes and the HIP runtime version.  Written atomically (temporary + rename); a
// file that does not load is ignored and replaced.
static unsigned long long fnv1a(const void* data, size_t n, unsigned long long h = 1469598103934665603ull) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static bool read_file(const std::string& path, std::string& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}
static std::string cache_dir() {
    const char* e = getenv("HJ_RTC_CACHE");
    if (e && (!strcmp(e, "0") || !strcmp(e, "off"))) return "";
    std::string d;
    if (e && *e) d = e;
    else if ((e = getenv("XDG_CACHE_HOME")) && *e) d = std::string(e) + "/levelsetpy_amd";
    else if ((e = getenv("HOME")) && *e) d = std::string(e) + "/.cache/levelsetpy_amd";
    else return "";
    // mkdir -p of the last two components (the parents of a cache home exist)
    const size_t cut = d.find_last_of('/');
    if (cut != std::string::npos && cut > 0) (void)mkdir(d.substr(0, cut).c_str(), 0700);
    if (mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) return "";
    return d;
}
static unsigned long long headers_hash(const std::string& include_dir) {
    static std::map<std::string, unsigned long long> memo;
    auto it = memo.find(include_dir);
    if (it != memo.end()) return it->second;
    unsigned long long h = 1469598103934665603ull;
    const char* files[] = {"/hj_fusedv.h", "/hj_fused.h", "/hj_device.h", "/hj_split.h", "/../../include/hj_mi355x.h"};
    std::string text;
    for (const char* f : files)
        if (read_file(include_dir + f, text)) h = fnv1a(text.data(), text.size(), h);
    memo[include_dir] = h;
    return h;
}
static int g_rtc_cache_hits = 0, g_rtc_compiles = 0;

static int rtc_build(UserHam& u, int id, const std::string& name_expr, UserKernel& out) {
    const std::string src = user_source(u, id);
    const char* base_opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-DHJ_RTC=1"};
    std::string cache_file;
    {
        const std::string dir = cache_dir();
        if (!dir.empty()) {
            unsigned long long h = fnv1a(src.data(), src.size());
            h = fnv1a(name_expr.data(), name_expr.size(), h);
            for (const char* o : base_opts) h = fnv1a(o, strlen(o), h);
            const unsigned long long hh = headers_hash(u.include_dir);
            h = fnv1a(&hh, sizeof(hh), h);
            int rtv = 0;                                   // a new ROCm release compiles again
            (void)hipRuntimeGetVersion(&rtv);
            h = fnv1a(&rtv, sizeof(rtv), h);
            char nm[64];
            snprintf(nm, sizeof(nm), "/%016llx.hjco", h);
            cache_file = dir + nm;
            std::string blob;
            if (read_file(cache_file, blob) && blob.compare(0, 6, "HJCO1\n") == 0) {
                const size_t nl = blob.find('\n', 6);
                if (nl != std::string::npos && nl + 1 < blob.size()) {
                    const std::string kname = blob.substr(6, nl - 6);
                    if (hipModuleLoadData(&out.mod, blob.data() + nl + 1) == hipSuccess &&
                        hipModuleGetFunction(&out.fn, out.mod, kname.c_str()) == hipSuccess) {
                        ++g_rtc_cache_hits;
                        return HJ_OK;
                    }
                    (void)hipGetLastError();
                    out.mod = nullptr; out.fn = nullptr;
                }
            }
        }
    }
    int rc = rtc_load(u.rtc_path.empty() ? nullptr : u.rtc_path.c_str());
    if (rc) return rc;
    ++g_rtc_compiles;
    rtcProgram prog = nullptr;
    int e = g_rtc.CreateProgram(&prog, src.c_str(), "hj_user_ham.hip", 0, nullptr, nullptr);
    if (e) return fail(HJ_EHIP, "hiprtcCreateProgram: %s", g_rtc.GetErrorString(e));
    e = g_rtc.AddNameExpression(prog, name_expr.c_str());
    const std::string inc1 = "-I" + u.include_dir, inc2 = "-I" + u.include_dir + "/../../include";
    const char* opts[] = {base_opts[0], base_opts[1], base_opts[2], base_opts[3], inc1.c_str(), inc2.c_str(), base_opts[4]};
    if (!e) e = g_rtc.CompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    if (e) {
        size_t n = 0;
        std::string log;
        if (g_rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); (void)g_rtc.GetProgramLog(prog, &log[0]); }
        (void)g_rtc.DestroyProgram(&prog);
        if (log.size() > 3000) log = log.substr(0, 3000) + " ...";
        return fail(HJ_EINVAL, "the Hamiltonian '%s' does not compile (%s):\n%s", u.name.c_str(), g_rtc.GetErrorString(e), log.c_str());
    }
    const char* lowered = nullptr;
    e = g_rtc.GetLoweredName(prog, name_expr.c_str(), &lowered);
    size_t n = 0;
    if (!e) e = g_rtc.GetCodeSize(prog, &n);
    std::vector<char> code(n);
    if (!e) e = g_rtc.GetCode(prog, code.data());
    const std::string kname = lowered ? lowered : "";
    (void)g_rtc.DestroyProgram(&prog);
    if (e) return fail(HJ_EHIP, "hipRTC: %s", g_rtc.GetErrorString(e));
    HIP_TRY(hipModuleLoadData(&out.mod, code.data()));
    HIP_TRY(hipModuleGetFunction(&out.fn, out.mod, kname.c_str()));
    if (!cache_file.empty()) {
        char tmpn[64];
        snprintf(tmpn, sizeof(tmpn), ".tmp%ld", (long)getpid());
        const std::string tmp = cache_file + tmpn;
        FILE* f = fopen(tmp.c_str(), "wb");
        if (f) {
            bool ok = fwrite("HJCO1\n", 1, 6, f) == 6 && fwrite(kname.data(), 1, kname.size(), f) == kname.size() && fputc('\n', f) != EOF &&
                      fwrite(code.data(), 1, code.size(), f) == code.size();
            ok = (fclose(f) == 0) && ok;
            if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
        }
    }
    return HJ_OK;
}

// kernel-argument block of fused_pair_kernel(const T* y, const T* y0, T* out, const FusedArgs<T, ND> A)
template <typename T, int ND> struct PairKernArgs {
    const T* y;
    const T* y0;
    T* out;
    FusedArgs<T, ND> A;
};
template <typename T, int ND> struct AlphaKernArgs {
    GridArgs<T, ND> G;
    HamTables<T> P;
    unsigned long long* keys;
    DxArgs DX;
};

static int module_launch(hipFunction_t fn, unsigned grid, unsigned block, size_t lds, hipStream_t st, void* args, size_t nbytes) {
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &nbytes, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, (unsigned)lds, st, nullptr, cfg));
    return HJ_OK;
}

// (threads, pairs per thread, halo slots per thread, waves/SIMD hint) of the run-time instantiations: the two shapes of the
// built-in pair kernels.  "small": one pair per thread in 256-thread workgroups -- ~120 VGPRs are left for an arbitrary
// Hamiltonian expression; "big": two pairs per thread in 512-thread workgroups + the parked halo ring, what the built-in
// light stencils run from 6.5 M cells up -- taken for the light stencils on such grids IF the expression compiles into it
// without scratch (hipFuncGetAttribute: a spilling kernel loses more than the shape gains), else the small shape.
struct UShape { int nt, r, kh, occ; };
constexpr UShape U_SMALL{256, 1, 2, 2}, U_BIG{512, 2, 2, 2};

template <int ND>
static int launch_user_nd(hj_ctx* c, const SubstepCall& s, UserHam& u) {
    using T = double;
    // MODE 1 / 2: the flag-free instantiations of plain RK stages (hj_inst.hip, launch_tiled); 0: every run-time flag
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0;
    const int mode = plain ? (s.stage == HJ_STAGE_EULER ? 1 : 2) : 0;
    const bool light = s.scheme == HJ_WENO5_ASSHIPPED || s.scheme == HJ_ENO2;
    bool big = light && ND <= 3 && c->total >= 6500000 && c->pair != 0 && !u.big_spills[s.scheme * 4 + mode];
    UserKernel* k = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const UShape sh = big ? U_BIG : U_SMALL;
        k = &u.substep[(s.scheme * 4 + mode) * 2 + (big ? 1 : 0)];
        if (!k->fn) {
            std::ostringstream nm;
            nm << "hj::fused_pair_kernel<double, hj::HamUser<double>, " << s.scheme << ", " << sh.nt << ", " << sh.r << ", " << sh.kh << ", "
               << sh.occ << ", " << mode << ">";
            int rc = rtc_build(u, s.ham, nm.str(), *k);
            if (rc) return rc;
            if (big) {
                int scratch = 0;
                if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn) != hipSuccess) scratch = 1;
                if (scratch > 0) {                       // the expression does not fit two pairs per thread: small shape from now on
                    u.big_spills[s.scheme * 4 + mode] = true;
                    big = false;
                    continue;
                }
            }
        }
        break;
    }
    const UShape sh = big ? U_BIG : U_SMALL;
    KernelCfg kc{sh.nt, sh.r, sh.kh};
    // halo ring parked in LDS with the big shape (as the built-in launches do, hj_inst.hip)
    bool ring = big && !u.no_big_lds && (c->pair_ring == 1 || (c->pair_ring < 0 && c->total >= 6500000));
    c->last_nbuf = ring ? 2 + c->pair_ah : 2;
    Tiling t = make_tiling(c, kc, s.p0, s.p1, 2, c->last_nbuf);
    if (t.ok && t.lds_bytes > 64 * 1024 && k->lds_granted < t.lds_bytes) {
        // more than 64 KB of dynamic LDS has to be granted to the function; if this runtime refuses that for a module
        // function, the ring (5 plane buffers) is given up and the double buffer (< 64 KB) stays
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes) == hipSuccess) {
            k->lds_granted = t.lds_bytes;
        } else {
            (void)hipGetLastError();
            u.no_big_lds = true;
            ring = false;
            c->last_nbuf = 2;
            t = make_tiling(c, kc, s.p0, s.p1, 2, 2);
        }
    }
    if (!t.ok) return fail(HJ_EUNSUPPORTED, "no tiling of this grid for the run-time kernel");
    if (t.lds_bytes > 64 * 1024 && k->lds_granted < t.lds_bytes) return fail(HJ_EUNSUPPORTED, "tile of the run-time kernel needs %zu bytes of LDS", t.lds_bytes);
    if (!k->occ) {
        int nb = 0;
        if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k->fn, sh.nt, t.lds_bytes) != hipSuccess || nb < 1) nb = 1;
        k->occ = nb;
    }
    EdgePlan ep;
    int rc = plan_chunks(c, s, t, k->occ, ep);
    if (rc) return rc;
    PairKernArgs<T, ND> K;
    memset(&K, 0, sizeof(K));
    K.y = (const T*)s.y;
    K.y0 = (const T*)s.y0;
    K.out = (T*)s.out;
    K.A.max_d1sq = (const T*)(c->weno_src ? c->weno_src : c->weno_vals);
    K.A.bound = s.bound;
    if (s.scheme == HJ_WENO5 && s.eps_nrows > 0) { K.A.eps_rows = s.eps_rows; K.A.eps_nrows = s.eps_nrows; }
    unsigned grid_blocks = 0;
    if ((rc = fill_fused_args<T, ND>(c, s, t, ep, s.scheme, true, K.A, grid_blocks))) return rc;
    if (c->debug) {
        int regs = 0, scr = 0;
        (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn);
        (void)hipFuncGetAttribute(&scr, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn);
        fprintf(stderr, "[hj] run-time kernel '%s' scheme %d mode %d: shape (%d,%d,%d,%d)%s, %d VGPRs, %d B scratch, tile (%d,%d), chunk %d, %d blocks, lds %zu\n",
                u.name.c_str(), s.scheme, mode, sh.nt, sh.r, sh.kh, sh.occ, ring ? " + ring" : "", regs, scr, t.E[1], t.E[2], t.chunk, t.nblocks, t.lds_bytes);
        c->debug = 0;
    }
    c->last_kernel = "fused_pair_kernel (hipRTC)";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = d < ND ? t.E[d] : 0;
    return module_launch(k->fn, grid_blocks, sh.nt, t.lds_bytes, call_stream(c, s), &K, sizeof(K));
}

int launch_user(hj_ctx* c, const SubstepCall& s) {
    UserHam* u = user_of(s.ham);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", s.ham);
    if (c->dtype != HJ_F64) return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians are compiled for fp64 grids");
    if (c->ndim == 2) return launch_user_nd<2>(c, s, *u);
    if (c->ndim == 3) return launch_user_nd<3>(c, s, *u);
    return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: 2-D and 3-D grids");
}

template <int ND>
static int alpha_user_nd(hj_ctx* c, int ham, const double* par, unsigned long long* keys, UserHam& u) {
    using T = double;
    if (!u.alpha.fn) {
        int rc = rtc_build(u, ham, "hj::alpha_bound_kernel<double, hj::HamUser<double>>", u.alpha);
        if (rc) return rc;
    }
    AlphaKernArgs<T, ND> K;
    memset(&K, 0, sizeof(K));
    fill_grid<T, ND>(c, K.G);
    fill_ham<T>(c, par, K.P);
    K.keys = keys;
    for (int d = 0; d < HJ_MAX_DIM; ++d) K.DX.dx[d] = c->dx[d];
    const unsigned blocks = (unsigned)std::min<int64_t>((c->total + 255) / 256, 256 * 2);
    return module_launch(u.alpha.fn, blocks, 256, 0, c->stream, &K, sizeof(K));
}

int user_alpha_bound(hj_ctx* c, int ham, const double* par, unsigned long long* keys) {
    UserHam* u = user_of(ham);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham);
    if (c->dtype != HJ_F64) return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians are compiled for fp64 grids");
    if (c->ndim == 2) return alpha_user_nd<2>(c, ham, par, keys, *u);
    if (c->ndim == 3) return alpha_user_nd<3>(c, ham, par, keys, *u);
    return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: 2-D and 3-D grids");
}

}  // namespace hjh

using namespace hjh;

extern "C" {

int hj_ham_register(const char* name, int ndim, int nparams, const char* body, const char* column_body, int ncol,
                    const char* include_dir, const char* hiprtc_path, int* ham_id) {
    if (!name || !body || !include_dir || !ham_id) return fail(HJ_EINVAL, "null argument");
    // the name goes into #line directives of the generated source (compiler messages then point into the caller's text)
    for (const char* ch = name; *ch; ++ch)
        if (*ch == '"' || *ch == '\\' || (unsigned char)*ch < 32) return fail(HJ_EINVAL, "the name of a Hamiltonian must not contain quotes, backslashes or control characters");
    if (ncol < 0 || ncol > 8) return fail(HJ_EINVAL, "0..8 column values, got %d", ncol);
    if (ncol > 0 && !column_body) return fail(HJ_EINVAL, "ncol > 0 needs a column expression");
    const std::string cb = (column_body && ncol > 0) ? column_body : "";
    if (ndim < 2 || ndim > 3) return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: grid.dim must be 2 or 3, got %d", ndim);
    if (nparams < 0 || nparams > 4) return fail(HJ_EINVAL, "a Hamiltonian takes 0..4 parameters, got %d", nparams);
    for (size_t i = 0; i < g_user.size(); ++i)
        if (g_user[i].name == name && g_user[i].ndim == ndim && g_user[i].nparams == nparams && g_user[i].body == body &&
            g_user[i].column_body == cb && g_user[i].ncol == ncol) {
            *ham_id = HJ_HAM_USER_BASE + (int)i;        // registering the same expression again: the same id, nothing recompiled
            return HJ_OK;
        }
    UserHam u;
    u.name = name;
    u.body = body;
    u.column_body = cb;
    u.ncol = ncol;
    u.include_dir = include_dir;
    u.rtc_path = hiprtc_path ? hiprtc_path : "";
    u.ndim = ndim;
    u.nparams = nparams;
    g_user.push_back(u);
    *ham_id = HJ_HAM_USER_BASE + (int)g_user.size() - 1;
    return HJ_OK;
}

int hj_ham_info(int ham_id, int* ndim, int* nparams, int* kernels_built) {
    const UserHam* u = user_of(ham_id);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham_id);
    if (ndim) *ndim = u->ndim;
    if (nparams) *nparams = u->nparams;
    if (kernels_built) {
        int n = u->alpha.fn ? 1 : 0;
        for (const auto& kv : u->substep) n += kv.second.fn ? 1 : 0;
        *kernels_built = n;
    }
    return HJ_OK;
}

int hj_ham_cache_stats(int* compiled, int* loaded_from_cache) {
    if (compiled) *compiled = g_rtc_compiles;
    if (loaded_from_cache) *loaded_from_cache = g_rtc_cache_hits;
    return HJ_OK;
}

// compile without launching (needs no GPU: hipRTC cross-compiles for gfx950) -- the check a registration can run up front
int hj_ham_compile_check(int ham_id, int scheme) {
    UserHam* u = user_of(ham_id);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham_id);
    if (scheme < 0 || scheme > 3) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    int rc = rtc_load(u->rtc_path.empty() ? nullptr : u->rtc_path.c_str());
    if (rc) return rc;
    std::ostringstream nm;
    nm << "hj::fused_pair_kernel<double, hj::HamUser<double>, " << scheme << ", " << U_SMALL.nt << ", " << U_SMALL.r << ", " << U_SMALL.kh << ", "
       << U_SMALL.occ << ", 0>";
    const std::string src = user_source(*u, ham_id);
    rtcProgram prog = nullptr;
    int e = g_rtc.CreateProgram(&prog, src.c_str(), "hj_user_ham.hip", 0, nullptr, nullptr);
    if (e) return fail(HJ_EHIP, "hiprtcCreateProgram: %s", g_rtc.GetErrorString(e));
    e = g_rtc.AddNameExpression(prog, nm.str().c_str());
    if (!e) e = g_rtc.AddNameExpression(prog, "hj::alpha_bound_kernel<double, hj::HamUser<double>>");
    const std::string inc1 = "-I" + u->include_dir, inc2 = "-I" + u->include_dir + "/../../include";
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", inc1.c_str(), inc2.c_str(), "-DHJ_RTC=1"};
    if (!e) e = g_rtc.CompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    std::string log;
    if (e) {
        size_t n = 0;
        if (g_rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); (void)g_rtc.GetProgramLog(prog, &log[0]); }
        if (log.size() > 3000) log = log.substr(0, 3000) + " ...";
    }
    (void)g_rtc.DestroyProgram(&prog);
    if (e) return fail(HJ_EINVAL, "the Hamiltonian '%s' does not compile (%s):\n%s", u->name.c_str(), g_rtc.GetErrorString(e), log.c_str());
    return HJ_OK;
}

}  // extern "C"
