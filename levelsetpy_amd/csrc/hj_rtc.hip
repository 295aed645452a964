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
// alpha may not depend on p itself -- but (round 5, flag HJ_HAM_RANGE) it may depend on the costate RANGE the reference hands to
// partialFunc: dmin[d] / dmax[d] = derivMin / derivMax of artificial_diss_glf.py:80-88.  Without the flag (every system the reference
// ships: dubins_relative.py:106-111, dubins_absolute.py:150-170, double_integrator.py:84-89, bird.py:346) the CFL bound is a
// property of the grid (hj_static_step_bound) and a time step needs no host synchronisation; with it a substep is TWO launches
// (the range pass -- MODE 3 of the tiled kernel: derivL / derivR of every cell reduced to 2*ND minima / maxima, nothing written --
// then the substep with the range visible to the expression and the in-kernel max(alpha) reduction kept) and hj_rk_step takes
// dt from the first stage's reduced bound, as ode_cfl_3.py:142 does.  fp64 and fp32; 2-D, 3-D and 4-D grids.
#include <dlfcn.h>
#include <errno.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>
#include <deque>
#include <mutex>
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

// a module is loaded on ONE device (hipModuleLoadData binds it to the device that is current): kernels are kept per device
// (ADVICE r04), their occupancy per dynamic-LDS size
struct UserKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    std::map<size_t, int> occ;              // dynamic LDS bytes -> resident workgroups per CU
    size_t lds_granted = 0;
};
struct UserHam {
    std::string name, body, column_body, include_dir, rtc_path;
    int ndim = 0, nparams = 0, ncol = 0, flags = 0;
    // key: ((((device * 2 + fp32) * 4 + scheme) * 4 + MODE) * 4 + shape)    shape: 0 small pair, 1 big pair, 2 one cell per lane (4-D fp64)
    std::map<long long, UserKernel> substep;
    std::map<int, bool> big_spills;         // key: (fp32 * 8 + scheme) * 4 + MODE -- the big shape needed scratch for this expression
    bool no_big_lds = false;                // the runtime refused > 64 KB of dynamic LDS for a module function
    std::map<int, UserKernel> alpha;        // key: device * 2 + fp32
};
// Registered Hamiltonians live in a deque (registration never moves an element another thread may be launching) and every entry point that
// looks at or extends the per-Hamiltonian kernel tables takes g_rtc_mu: kernels are compiled lazily at the first launch, and several host
// threads driving their own contexts (virtual ranks in one process: tests/fuzz_slabs.py found the race in round 5 -- eight threads met in
// the first compile of the same kernel) must see either no kernel or a complete one.  The launches themselves are asynchronous: the lock is
// held for microseconds once the kernels exist.
static std::deque<UserHam> g_user;
static std::recursive_mutex g_rtc_mu;
#define HJ_RTC_LOCK std::lock_guard<std::recursive_mutex> hj_rtc_guard(g_rtc_mu)

static UserHam* user_of(int ham) {
    const int i = ham - HJ_HAM_USER_BASE;
    return (i >= 0 && i < (int)g_user.size()) ? &g_user[(size_t)i] : nullptr;
}
bool user_ham_valid(int ham) { HJ_RTC_LOCK; return user_of(ham) != nullptr; }
bool user_ham_dynamic(int ham) { HJ_RTC_LOCK; const UserHam* u = user_of(ham); return u && (u->flags & HJ_HAM_RANGE); }
int user_ham_ndim(int ham) { HJ_RTC_LOCK; const UserHam* u = user_of(ham); return u ? u->ndim : -1; }
int user_ham_npar(int ham) { HJ_RTC_LOCK; const UserHam* u = user_of(ham); return u ? u->nparams : 0; }

// the translation unit hipRTC compiles: the kernel headers + the Hamiltonian type around the caller's expression
static std::string user_source(const UserHam& u, int id) {
    const bool rng = (u.flags & HJ_HAM_RANGE) != 0;
    std::ostringstream o;
    o << "#include \"hj_fusedv.h\"\n#include \"hj_fused4v.h\"\n#include \"hj_split.h\"\nnamespace hj {\n"
         "template <typename T> struct HamUser {\n"
         "    static constexpr int ND = " << u.ndim << ";\n"
         "    static constexpr int ID = " << id << ";\n"
         "    static constexpr unsigned PLANE_DEP = 0xFu;     // any alpha may vary along the march\n"
         "    static constexpr int NCOL = " << (u.ncol > 0 ? u.ncol : 1) << ";\n"
         "    static constexpr bool RANGE = " << (rng ? "true" : "false") << ";\n"
         "    static constexpr bool DT_DEV = RANGE;       // the stage kernels may take deltaT from device memory (FusedArgs::dt_dev)\n"
         "    struct Cell { T x[ND]; T col[NCOL]; };\n"
         "    struct Plane { T x0; T dmin[RANGE ? ND : 1], dmax[RANGE ? ND : 1]; };\n    using Raw = Cell;\n"
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
         "    // per axis-0 plane: its coordinate and (RANGE) the costate range the range pass left in P.range (wave-uniform)\n"
         "    __device__ static __forceinline__ Plane plane(const HamTables<T>& P, int i0, const T*) {\n"
         "        Plane u; u.x0 = P.coord[0][i0];\n"
         "        u.dmin[0] = T(0); u.dmax[0] = T(0);\n"
         "        if constexpr (RANGE) {\n"
         "            for (int d = 0; d < ND; ++d) { u.dmax[d] = (T)key_value(P.range[d]); u.dmin[d] = (T)(-key_value(P.range[ND + d])); }\n"
         "        }\n        return u;\n    }\n"
         "    template <bool NP = false>\n"
         "    __device__ static __forceinline__ void eval(const HamTables<T>& P, const Cell& c, const Plane& pl, const T* sc, const T* q, T& H, T* alpha) {\n"
         "        T x[ND], p[ND];\n        x[0] = pl.x0;\n"
         "        for (int d = 1; d < ND; ++d) x[d] = c.x[d];\n"
         "        for (int d = 0; d < ND; ++d) { p[d] = sc[d] * q[d]; alpha[d] = T(0); }\n"
         "        const T* par = P.par;\n        const T* col = c.col;\n        const T* dmin = pl.dmin;\n        const T* dmax = pl.dmax;\n"
         "        (void)col; (void)par; (void)dmin; (void)dmax;\n        H = T(0);\n"
         "        {\n#line 1 \"" << u.name << "\"\n" << u.body << "\n        }\n"
         "        for (int d = 0; d < ND; ++d) alpha[d] = sc[d] * alpha[d];     // the kernels carry alpha in the stencil's scale\n"
         "    }\n};\n}\n";
    return o.str();
}

// ---- code-object cache on disk: a registered expression is compiled once per (source, kernel headers, options), not once per
// process.  Directory: $HJ_RTC_CACHE ("0" / "off" disables), else $XDG_CACHE_HOME/levelsetpy_amd, else $HOME/.cache/levelsetpy_amd;
// it must belong to the user and not be writable by group or others (a planted file would be code run on the GPU), else the cache
// is off.  File <key>.hjco = "HJCO2\n" + lowered kernel name + "\n" + 16 hex digits: size of the code object + "\n" + 16 hex digits:
// FNV-1a of (key material, code object) + "\n" + code object; key = two FNV-1a hashes (128 bits) of the generated source, the name
// expression, the compile options, the TEXT of every kernel header the source includes, sizeof(FusedArgs) (the kernel-argument ABI)
// and the HIP runtime version.  A header that cannot be read turns the cache off (the key would not cover it).  Written atomically
// (temporary + rename); a file whose size or checksum does not match is ignored and replaced.
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
    struct stat st;
    if (stat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) return "";
    return d;
}
// hash of the text of every header the generated source can include; ok = false if one of them cannot be read
static unsigned long long headers_hash(const std::string& include_dir, unsigned long long seed, bool& ok) {
    static std::map<std::string, std::pair<unsigned long long, bool>> memo;
    const std::string key = include_dir + "#" + std::to_string(seed);
    auto it = memo.find(key);
    if (it != memo.end()) { ok = it->second.second; return it->second.first; }
    unsigned long long h = seed;
    ok = true;
    const char* files[] = {"/hj_fusedv.h", "/hj_fused4v.h", "/hj_fused.h", "/hj_termop.h", "/hj_device.h", "/hj_split.h", "/../../include/hj_mi355x.h"};
    std::string text;
    for (const char* f : files) {
        if (read_file(include_dir + f, text)) h = fnv1a(text.data(), text.size(), h);
        else {
            // the key cannot cover this header's text: the caller keeps the disk cache OFF for this include directory (and says so once)
            ok = false;
            fprintf(stderr, "[hj] run-time kernels: cannot read %s%s -- the on-disk code-object cache is disabled\n", include_dir.c_str(), f);
        }
    }
    memo[key] = std::make_pair(h, ok);
    return h;
}
static int g_rtc_cache_hits = 0, g_rtc_compiles = 0;

static int rtc_build(UserHam& u, int id, const std::string& name_expr, UserKernel& out) {
    const std::string src = user_source(u, id);
    const char* base_opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-DHJ_RTC=1"};
    std::string cache_file;
    unsigned long long key_lo = 0;
    {
        const std::string dir = cache_dir();
        if (!dir.empty()) {
            unsigned long long h[2];
            bool ok = true;
            for (int w = 0; w < 2; ++w) {
                const unsigned long long seed = w == 0 ? 1469598103934665603ull : 0x9e3779b97f4a7c15ull;
                unsigned long long x = fnv1a(src.data(), src.size(), seed);
                x = fnv1a(name_expr.data(), name_expr.size(), x);
                for (const char* o : base_opts) x = fnv1a(o, strlen(o), x);
                bool okh = true;
                const unsigned long long hh = headers_hash(u.include_dir, seed, okh);
                ok = ok && okh;
                x = fnv1a(&hh, sizeof(hh), x);
                const unsigned long long abi[3] = {sizeof(FusedArgs<double, 3>), sizeof(FusedArgs<float, 4>), sizeof(HamTables<double>)};
                x = fnv1a(abi, sizeof(abi), x);
                int rtv = 0;                                   // a new ROCm release compiles again
                (void)hipRuntimeGetVersion(&rtv);
                x = fnv1a(&rtv, sizeof(rtv), x);
                h[w] = x;
            }
            if (ok) {
                key_lo = h[0];
                char nm[64];
                snprintf(nm, sizeof(nm), "/%016llx%016llx.hjco", h[0], h[1]);
                cache_file = dir + nm;
                std::string blob;
                if (read_file(cache_file, blob) && blob.compare(0, 6, "HJCO2\n") == 0) {
                    const size_t n1 = blob.find('\n', 6);
                    if (n1 != std::string::npos && blob.size() >= n1 + 1 + 17 + 17) {
                        const std::string kname = blob.substr(6, n1 - 6);
                        const unsigned long long want_size = strtoull(blob.substr(n1 + 1, 16).c_str(), nullptr, 16);
                        const unsigned long long want_sum = strtoull(blob.substr(n1 + 18, 16).c_str(), nullptr, 16);
                        const size_t off = n1 + 1 + 17 + 17;
                        if (blob[n1 + 17] == '\n' && blob[n1 + 34] == '\n' && blob.size() - off == want_size &&
                            fnv1a(blob.data() + off, blob.size() - off, key_lo) == want_sum &&
                            hipModuleLoadData(&out.mod, blob.data() + off) == hipSuccess &&
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
            char meta[64];
            snprintf(meta, sizeof(meta), "%016llx\n%016llx\n", (unsigned long long)code.size(), fnv1a(code.data(), code.size(), key_lo));
            bool ok = fwrite("HJCO2\n", 1, 6, f) == 6 && fwrite(kname.data(), 1, kname.size(), f) == kname.size() && fputc('\n', f) != EOF &&
                      fwrite(meta, 1, 34, f) == 34 && fwrite(code.data(), 1, code.size(), f) == code.size();
            ok = (fclose(f) == 0) && ok;
            if (!ok || rename(tmp.c_str(), cache_file.c_str()) != 0) (void)remove(tmp.c_str());
        }
    }
    return HJ_OK;
}

// kernel-argument block of the tiled kernels: (const T* y, const T* y0, T* out, const FusedArgs<T, ND> A)
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
    double* partials;
    unsigned long long* done;
    unsigned long long* host_out;
    unsigned long long seq;
    DtArgs DT;
};

static int module_launch(hipFunction_t fn, unsigned grid, unsigned block, size_t lds, hipStream_t st, void* args, size_t nbytes) {
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &nbytes, HIP_LAUNCH_PARAM_END};
    HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, (unsigned)lds, st, nullptr, cfg));
    return HJ_OK;
}

// Kernel shapes of the run-time instantiations -- those of the built-in systems (hj_inst.hip):
//   0 "small": pair kernel, one pair per thread in 256-thread workgroups -- ~120 VGPRs are left for an arbitrary expression
//   1 "big":   pair kernel, two pairs per thread in 512-thread workgroups + the parked halo ring, what the built-in light stencils run
//              from 6.5 M cells up -- taken for the light stencils on such 2-D / 3-D grids IF the expression compiles into it without
//              scratch (hipFuncGetAttribute: a spilling kernel loses more than the shape gains), else the small shape
//   2 "single": one cell per lane (fused_substep_kernel) -- 4-D grids: 512 threads in fp64, 1024 in fp32 (heavy stencils)
//   3 "pair4d": pair kernel on 4-D fp32 grids with a light stencil (256 threads x 2 pairs, 5 pair + 1 single halo slots)
struct UShape { int nt, r, kh, occ, pd; bool pair; };
static UShape shape_of(int shape, bool fp32) {
    switch (shape) {
        case 1: return UShape{512, 2, 2, 2, 2, true};
        case 2: return fp32 ? UShape{1024, 1, 3, 2, 2, false} : UShape{512, 1, 4, 2, 2, false};
        case 3: return UShape{256, 2, 6, 2, 2, true};
    }
    return UShape{256, 1, 2, 2, 2, true};
}
static std::string kernel_name(const UShape& sh, const char* tname, int scheme, int mode) {
    std::ostringstream nm;
    if (sh.pair)
        nm << "hj::fused_pair_kernel<" << tname << ", hj::HamUser<" << tname << ">, " << scheme << ", " << sh.nt << ", " << sh.r << ", " << sh.kh << ", "
           << sh.occ << ", " << mode << ">";
    else
        nm << "hj::fused_substep_kernel<" << tname << ", hj::HamUser<" << tname << ">, " << scheme << ", " << sh.nt << ", " << sh.r << ", " << sh.kh << ", "
           << sh.occ << ", " << sh.pd << ", " << mode << ">";
    return nm.str();
}

template <typename T, int ND>
static int launch_user_nd(hj_ctx* c, const SubstepCall& s, UserHam& u) {
    constexpr bool F32 = sizeof(T) == 4;
    const char* tname = F32 ? "float" : "double";
    HIP_TRY(hipSetDevice(c->device));        // the modules below belong to this device
    // MODE 1 / 2: the flag-free instantiations of plain RK stages (hj_inst.hip, launch_tiled); 0: every run-time flag; 3: the range pass
    const bool plain = s.stage != HJ_STAGE_YDOT && s.restrict_sign == 0 && s.post_op == 0;
    const int mode = plain ? (s.stage == HJ_STAGE_EULER ? 1 : 2) : 0;
    const bool light = light_scheme(s.scheme);
    const bool dynamic = (u.flags & HJ_HAM_RANGE) != 0;
    const int spill_key = ((F32 ? 1 : 0) * 8 + s.scheme) * 4 + mode;      // (scheme ids run to HJ_ENO3_FAST = 5: radix 8)
    int shape = 0;
    if (ND == 4) shape = (F32 && light) ? 3 : 2;
    else if (light && c->total >= 6500000 && c->pair != 0 && !u.big_spills[spill_key]) shape = 1;
    auto kernel_of = [&](int shp, int md) -> UserKernel& {
        static_assert(HJ_ENO3_FAST < 8, "kernel-table key: scheme radix");
        return u.substep[((((long long)c->device * 2 + (F32 ? 1 : 0)) * 8 + s.scheme) * 4 + md) * 4 + shp];
    };
    UserKernel* k = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        k = &kernel_of(shape, mode);
        if (!k->fn) {
            int rc = rtc_build(u, s.ham, kernel_name(shape_of(shape, F32), tname, s.scheme, mode), *k);
            if (rc) return rc;
            if (shape == 1) {
                int scratch = 0;
                if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn) != hipSuccess) scratch = 1;
                if (scratch > 0) {                       // the expression does not fit two pairs per thread: small shape from now on
                    u.big_spills[spill_key] = true;
                    shape = 0;
                    continue;
                }
            }
        }
        break;
    }
    UserKernel* kr = nullptr;                            // the range pass of the same shape
    if (dynamic) {
        kr = &kernel_of(shape, 3);
        if (!kr->fn) {
            int rc = rtc_build(u, s.ham, kernel_name(shape_of(shape, F32), tname, s.scheme, 3), *kr);
            if (rc) return rc;
        }
    }
    const UShape sh = shape_of(shape, F32);
    KernelCfg kc{sh.nt, sh.r, sh.kh};
    // halo ring parked in LDS with the big shape (as the built-in launches do, hj_inst.hip)
    bool ring = shape == 1 && !u.no_big_lds && (c->pair_ring == 1 || (c->pair_ring < 0 && c->total >= 6500000));
    c->last_nbuf = ring ? 2 + c->pair_ah : 2;
    c->last_nbase = 2;
    Tiling t = make_tiling(c, kc, s.p0, s.p1, sh.pair ? 2 : 1, c->last_nbuf);
    auto grant = [&](UserKernel* kk) -> bool {
        if (!(t.ok && t.lds_bytes > 64 * 1024 && kk->lds_granted < t.lds_bytes)) return true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kk->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds_bytes) == hipSuccess) {
            kk->lds_granted = t.lds_bytes;
            return true;
        }
        (void)hipGetLastError();
        return false;
    };
    if (!grant(k) || (kr && !grant(kr))) {
        // more than 64 KB of dynamic LDS has to be granted to the function; if this runtime refuses that for a module
        // function, the ring (5 plane buffers) is given up and the double buffer (< 64 KB) stays
        u.no_big_lds = true;
        ring = false;
        c->last_nbuf = 2;
        c->last_nbase = 2;
        t = make_tiling(c, kc, s.p0, s.p1, sh.pair ? 2 : 1, 2);
    }
    if (!t.ok) return fail(HJ_EUNSUPPORTED, "no tiling of this grid for the run-time kernel");
    if (t.lds_bytes > 64 * 1024 && k->lds_granted < t.lds_bytes) return fail(HJ_EUNSUPPORTED, "tile of the run-time kernel needs %zu bytes of LDS", t.lds_bytes);
    auto occ_of = [&](UserKernel* kk) {
        auto it = kk->occ.find(t.lds_bytes);
        if (it == kk->occ.end()) {
            int nb = 0;
            if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kk->fn, sh.nt, t.lds_bytes) != hipSuccess || nb < 1) nb = 1;
            it = kk->occ.emplace(t.lds_bytes, nb).first;
        }
        return it->second;
    };
    EdgePlan ep;
    int rc = plan_chunks(c, s, t, occ_of(k), ep);
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
    if ((rc = fill_fused_args<T, ND>(c, s, t, ep, s.scheme, sh.pair, K.A, grid_blocks))) return rc;
    if (!sh.pair) { K.A.lds_nbuf = 2; K.A.halo_ahead = 0; }
    if (c->debug) {
        int regs = 0, scr = 0;
        (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn);
        (void)hipFuncGetAttribute(&scr, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, k->fn);
        fprintf(stderr, "[hj] run-time kernel '%s' %s scheme %d mode %d: shape %d (%d,%d,%d,%d)%s%s, %d VGPRs, %d B scratch, tile (%d,%d,%d), chunk %d, %d blocks, lds %zu\n",
                u.name.c_str(), tname, s.scheme, mode, shape, sh.nt, sh.r, sh.kh, sh.occ, ring ? " + ring" : "", dynamic ? " + range pass" : "", regs, scr,
                t.E[1], ND > 2 ? t.E[2] : 0, ND > 3 ? t.E[3] : 0, t.chunk, t.nblocks, t.lds_bytes);
        c->debug = 0;
    }
    c->last_kernel = sh.pair ? "fused_pair_kernel (hipRTC)" : "fused_substep_kernel (hipRTC)";
    c->last_E[0] = t.chunk;
    for (int d = 1; d < HJ_MAX_DIM; ++d) c->last_E[d] = d < ND ? t.E[d] : 0;
    if ((s.range_only || s.bound_pass) && !dynamic) return fail(HJ_EINVAL, "'%s' does not read the costate range", u.name.c_str());
    // the local-local variant evaluates alpha with the NODE's own costate range in every dimension: no grid-wide range, no range pass
    const bool need_range = c->diss_kind != HJ_DISS_LLLF;
    if (dynamic && !s.bound_pass && (s.range_only || (need_range && !(c->range_src || s.range_ready)))) {
        // the range pass: same tiling, same arguments, MODE 3 -- derivL / derivR of every cell of [p0, p1) reduced into 2*ND keys.
        // Skipped when the caller supplies the range itself (hj_ctx_set_range_source: a slab of a decomposed grid, whose range is
        // the reduction over all ranks).
        if (s.gated) return fail(HJ_EUNSUPPORTED, "Hamiltonians with a range-dependent alpha do not run in gated slab launches");
        // a slab with neighbours: a pass of its own planes would give a rank-local range where the grid's is meant (only hj_range_pass, whose
        // result the caller reduces over the ranks, may run one)
        if (!s.range_only && (c->halo_lo || c->halo_hi))
            return fail(HJ_ESTATE, "a slab's launches of '%s' read the range of the WHOLE grid: hj_range_pass, reduce over the ranks, hj_ctx_set_range_source first", u.name.c_str());
        if (!(s.range_only && s.range_out)) {
            if ((rc = next_range_keys(c))) return rc;           // a zeroed entry of the ring: no memset launch
            K.A.ham.range = c->range_keys;                      // (fill_ham ran before the ring advanced)
        }
        unsigned long long* keys = (s.range_only && s.range_out) ? s.range_out : c->range_keys;
        PairKernArgs<T, ND> R3 = K;
        R3.out = nullptr;
        R3.A.bound = nullptr;
        R3.A.range_keys = keys;
        // HamUser::plane() decodes the range it is handed in every instantiation, MODE 3 included (the compiler may hoist those loads above the
        // branch that needs them).  hj_range_pass on a context that has no range yet (a slab's first pass, before hj_ctx_set_range_source) used to
        // hand it a null pointer: a load from address 0 once an expression kept the loads alive (found late in round 5 by the first Hamiltonian
        // whose alpha_i reads another dimension's range).  The keys being written are as good as any: this pass does not use the decoded values.
        if (!R3.A.ham.range) R3.A.ham.range = keys;
        R3.A.gate = nullptr;
        R3.A.use_y0 = 0; R3.A.ydot_only = 0; R3.A.post_op = 0; R3.A.do_clamp = 0;
        R3.A.eps_part = nullptr;
        if (s.range_only && s.range_out) HIP_TRY(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * 2 * HJ_MAX_DIM, call_stream(c, s)));
        if ((rc = module_launch(kr->fn, grid_blocks, sh.nt, t.lds_bytes, call_stream(c, s), &R3, sizeof(R3)))) return rc;
        if (s.range_only) return HJ_OK;
    }
    if (dynamic && !K.A.ham.range) {
        // (the local-local variant never runs a range pass, but HamUser::plane() decodes the keys it is handed: any zeroed ring entry will do)
        if ((rc = next_range_keys(c))) return rc;
        K.A.ham.range = c->range_keys;
    }
    if (s.bound_pass) {
        // the bound pass of the local variants (hj_rk_step, before the first stage): MODE 3 with a bound slot -- the stencils of every
        // dimension, alpha under the local rule, max_x sum_d alpha_d / dx_d reduced into the slot; nothing stored
        if (c->diss_kind == HJ_DISS_GLF || !s.bound) return fail(HJ_EINVAL, "bound pass: a local Lax-Friedrichs kind and a bound slot are needed");
        PairKernArgs<T, ND> B3 = K;
        B3.out = nullptr;
        B3.A.range_keys = nullptr;
        B3.A.gate = nullptr;
        B3.A.use_y0 = 0; B3.A.ydot_only = 0; B3.A.post_op = 0; B3.A.do_clamp = 0;
        B3.A.eps_part = nullptr;
        return module_launch(kr->fn, grid_blocks, sh.nt, t.lds_bytes, call_stream(c, s), &B3, sizeof(B3));
    }
    return module_launch(k->fn, grid_blocks, sh.nt, t.lds_bytes, call_stream(c, s), &K, sizeof(K));
}

int launch_user(hj_ctx* c, const SubstepCall& s) {
    HJ_RTC_LOCK;
    UserHam* u = user_of(s.ham);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", s.ham);
    if (c->ndim != u->ndim) return fail(HJ_EINVAL, "Hamiltonian '%s' is %d-dimensional, the grid has dim %d", u->name.c_str(), u->ndim, c->ndim);
    if (c->dtype == HJ_F64) {
        if (c->ndim == 2) return launch_user_nd<double, 2>(c, s, *u);
        if (c->ndim == 3) return launch_user_nd<double, 3>(c, s, *u);
        if (c->ndim == 4) return launch_user_nd<double, 4>(c, s, *u);
    } else {
        if (c->ndim == 2) return launch_user_nd<float, 2>(c, s, *u);
        if (c->ndim == 3) return launch_user_nd<float, 3>(c, s, *u);
        if (c->ndim == 4) return launch_user_nd<float, 4>(c, s, *u);
    }
    return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: 2-D, 3-D and 4-D grids");
}

template <typename T, int ND>
static int alpha_user_nd(hj_ctx* c, int ham, const double* par, unsigned long long* keys, UserHam& u,
                         unsigned long long* done, unsigned long long* host_out, unsigned long long seq, const DtArgs* dt) {
    constexpr bool F32 = sizeof(T) == 4;
    HIP_TRY(hipSetDevice(c->device));
    UserKernel& ak = u.alpha[c->device * 2 + (F32 ? 1 : 0)];
    if (!ak.fn) {
        const std::string tn = F32 ? "float" : "double";
        int rc = rtc_build(u, ham, "hj::alpha_bound_kernel<" + tn + ", hj::HamUser<" + tn + ">>", ak);
        if (rc) return rc;
    }
    AlphaKernArgs<T, ND> K;
    memset(&K, 0, sizeof(K));
    fill_grid<T, ND>(c, K.G);
    fill_ham<T>(c, par, K.P, ham);
    K.keys = keys;
    {
        int rc = alpha_partials(c);
        if (rc) return rc;
    }
    K.partials = c->alpha_part;
    K.done = done; K.host_out = host_out; K.seq = seq;
    if (dt) K.DT = *dt;
    for (int d = 0; d < HJ_MAX_DIM; ++d) K.DX.dx[d] = c->dx[d];
    // (one workgroup of 512 per CU -- the sweep of tools/experiments/r05_alpha_sweep.py: every further workgroup costs a fence and an atomic; a thread walks a run of planes of one in-plane cell; every workgroup costs one same-address atomic)
    const unsigned blocks = (unsigned)std::min<int64_t>((c->total + 511) / 512, std::min<int64_t>((int64_t)c->num_cus, ALPHA_BLOCKS_MAX));
    // HJ_ALPHA_BLOCKS / HJ_ALPHA_THREADS: tuning knobs (tools/experiments/r05_alpha_sweep.py)
    static const int env_blocks = getenv("HJ_ALPHA_BLOCKS") ? atoi(getenv("HJ_ALPHA_BLOCKS")) : 0;
    static const int env_threads = getenv("HJ_ALPHA_THREADS") ? atoi(getenv("HJ_ALPHA_THREADS")) : 0;
    // (alpha_bound_kernel's launch bound and its reduction scratch hold 1024 threads = 16 waves; whole waves only)
    const unsigned nthr = env_threads > 0 ? (unsigned)std::max(64, std::min(1024, env_threads & ~63)) : 512u;
    const unsigned nblk = env_blocks > 0 ? (unsigned)std::min(env_blocks, ALPHA_BLOCKS_MAX) : blocks;
    return module_launch(ak.fn, nblk, nthr, 0, c->stream, &K, sizeof(K));
}

// with_range: the caller has just run the range pass of the state in question (ctx->range_keys / range_src): max(alpha) over the grid is
// then well defined for an HJ_HAM_RANGE Hamiltonian too (alpha never depends on the node's own costate)
int user_alpha_bound(hj_ctx* c, int ham, const double* par, unsigned long long* keys, unsigned long long* done, bool with_range,
                     unsigned long long* host_out, unsigned long long seq, const DtArgs* dt) {
    HJ_RTC_LOCK;
    UserHam* u = user_of(ham);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham);
    if ((u->flags & HJ_HAM_RANGE) && !with_range)
        return fail(HJ_EUNSUPPORTED, "the alpha of '%s' depends on the costate range: its step bound is a property of the data, not of the grid", u->name.c_str());
    if (c->dtype == HJ_F64) {
        if (c->ndim == 2) return alpha_user_nd<double, 2>(c, ham, par, keys, *u, done, host_out, seq, dt);
        if (c->ndim == 3) return alpha_user_nd<double, 3>(c, ham, par, keys, *u, done, host_out, seq, dt);
        if (c->ndim == 4) return alpha_user_nd<double, 4>(c, ham, par, keys, *u, done, host_out, seq, dt);
    } else {
        if (c->ndim == 2) return alpha_user_nd<float, 2>(c, ham, par, keys, *u, done, host_out, seq, dt);
        if (c->ndim == 3) return alpha_user_nd<float, 3>(c, ham, par, keys, *u, done, host_out, seq, dt);
        if (c->ndim == 4) return alpha_user_nd<float, 4>(c, ham, par, keys, *u, done, host_out, seq, dt);
    }
    return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: 2-D, 3-D and 4-D grids");
}

}  // namespace hjh

using namespace hjh;

extern "C" {

int hj_ham_register2(const char* name, int ndim, int nparams, const char* body, const char* column_body, int ncol, int flags,
                     const char* include_dir, const char* hiprtc_path, int* ham_id) {
    if (!name || !body || !include_dir || !ham_id) return fail(HJ_EINVAL, "null argument");
    HJ_RTC_LOCK;
    // the name goes into #line directives of the generated source (compiler messages then point into the caller's text)
    for (const char* ch = name; *ch; ++ch)
        if (*ch == '"' || *ch == '\\' || (unsigned char)*ch < 32) return fail(HJ_EINVAL, "the name of a Hamiltonian must not contain quotes, backslashes or control characters");
    if (ncol < 0 || ncol > 8) return fail(HJ_EINVAL, "0..8 column values, got %d", ncol);
    if (ncol > 0 && !column_body) return fail(HJ_EINVAL, "ncol > 0 needs a column expression");
    if (flags & ~HJ_HAM_RANGE) return fail(HJ_EINVAL, "unknown flags 0x%x", flags);
    const std::string cb = (column_body && ncol > 0) ? column_body : "";
    if (ndim < 2 || ndim > 4) return fail(HJ_EUNSUPPORTED, "run-time Hamiltonians: grid.dim must be 2, 3 or 4, got %d", ndim);
    if (nparams < 0 || nparams > 8) return fail(HJ_EINVAL, "a Hamiltonian takes 0..8 parameters, got %d", nparams);
    for (size_t i = 0; i < g_user.size(); ++i)
        if (g_user[i].name == name && g_user[i].ndim == ndim && g_user[i].nparams == nparams && g_user[i].body == body &&
            g_user[i].column_body == cb && g_user[i].ncol == ncol && g_user[i].flags == flags) {
            *ham_id = HJ_HAM_USER_BASE + (int)i;        // registering the same expression again: the same id, nothing recompiled
            return HJ_OK;
        }
    UserHam u;
    u.name = name;
    u.body = body;
    u.column_body = cb;
    u.ncol = ncol;
    u.flags = flags;
    u.include_dir = include_dir;
    u.rtc_path = hiprtc_path ? hiprtc_path : "";
    u.ndim = ndim;
    u.nparams = nparams;
    g_user.push_back(u);
    *ham_id = HJ_HAM_USER_BASE + (int)g_user.size() - 1;
    return HJ_OK;
}

int hj_ham_register(const char* name, int ndim, int nparams, const char* body, const char* column_body, int ncol,
                    const char* include_dir, const char* hiprtc_path, int* ham_id) {
    return hj_ham_register2(name, ndim, nparams, body, column_body, ncol, 0, include_dir, hiprtc_path, ham_id);
}

int hj_ham_info(int ham_id, int* ndim, int* nparams, int* kernels_built) {
    HJ_RTC_LOCK;
    const UserHam* u = user_of(ham_id);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham_id);
    if (ndim) *ndim = u->ndim;
    if (nparams) *nparams = u->nparams;
    if (kernels_built) {
        int n = 0;
        for (const auto& kv : u->alpha) n += kv.second.fn ? 1 : 0;
        for (const auto& kv : u->substep) n += kv.second.fn ? 1 : 0;
        *kernels_built = n;
    }
    return HJ_OK;
}

int hj_ham_flags(int ham_id, int* flags) {
    HJ_RTC_LOCK;
    const UserHam* u = user_of(ham_id);
    if (!u || !flags) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham_id);
    *flags = u->flags;
    return HJ_OK;
}

int hj_ham_cache_stats(int* compiled, int* loaded_from_cache) {
    HJ_RTC_LOCK;
    if (compiled) *compiled = g_rtc_compiles;
    if (loaded_from_cache) *loaded_from_cache = g_rtc_cache_hits;
    return HJ_OK;
}

// compile without launching (needs no GPU: hipRTC cross-compiles for gfx950) -- the check a registration can run up front
int hj_ham_compile_check(int ham_id, int scheme) {
    HJ_RTC_LOCK;
    UserHam* u = user_of(ham_id);
    if (!u) return fail(HJ_EINVAL, "unknown Hamiltonian id %d", ham_id);
    if (scheme < 0 || scheme > HJ_ENO3_FAST) return fail(HJ_EINVAL, "unknown scheme %d", scheme);
    int rc = rtc_load(u->rtc_path.empty() ? nullptr : u->rtc_path.c_str());
    if (rc) return rc;
    const bool dynamic = (u->flags & HJ_HAM_RANGE) != 0;
    const UShape sh = shape_of(u->ndim == 4 ? 2 : 0, false);
    const std::string src = user_source(*u, ham_id);
    if (const char* dump = getenv("HJ_RTC_DUMP")) {        // the translation unit hipRTC is given, for a look at it (or an offline hipcc -S)
        if (FILE* f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    rtcProgram prog = nullptr;
    int e = g_rtc.CreateProgram(&prog, src.c_str(), "hj_user_ham.hip", 0, nullptr, nullptr);
    if (e) return fail(HJ_EHIP, "hiprtcCreateProgram: %s", g_rtc.GetErrorString(e));
    // every instantiation a step can launch, in BOTH dtypes (an expression that fails only in the float instantiation, or in the
    // bound kernel of a range-reading Hamiltonian -- hj_rk_step under GLF and hj_range_alpha_max launch it -- passed the check and failed at the
    // first step)
    for (const char* tn : {"double", "float"}) {
        if (!e) e = g_rtc.AddNameExpression(prog, kernel_name(sh, tn, scheme, 0).c_str());
        if (!e && dynamic) e = g_rtc.AddNameExpression(prog, kernel_name(sh, tn, scheme, 3).c_str());
        if (!e) e = g_rtc.AddNameExpression(prog, (std::string("hj::alpha_bound_kernel<") + tn + ", hj::HamUser<" + tn + ">>").c_str());
    }
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
