// Issue cost of the vector instructions the substep kernels are made of (round 5): v_fma_f32 against v_pk_fma_f32 (is packing
// two cells into one instruction worth anything on gfx950?), v_fma_f64, v_max_f64, v_cvt_f64_f32, s_nop, v_mov -- at 1, 2 and 4
// waves per SIMD (256 / 512 / 1024-thread workgroups, one per CU), as shader cycles per wave-instruction per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void rate_kernel(unsigned long long* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 pb = {b, b}, pc = {c, c};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = 1.0001, dc = 0.5;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 8 independent chains of v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) { // 8 independent chains of v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) { // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));)
        } else if (KIND == 3) { // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (KIND == 4) { // v_add_f32
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 5) { // v_max_f64
            REP8(asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                              "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));)
        } else if (KIND == 6) { // v_add_f64
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));)
        } else if (KIND == 7) { // v_mov_b64 (the register queue's shift)
            REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                              "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));)
        } else if (KIND == 8) { // SALU
            REP8(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                              "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1"
                              ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
        } else if (KIND == 9) { // alternating SALU / VALU
            REP8(asm volatile("s_add_u32 s20, s20, 1\n v_fma_f32 %0, %0, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %1, %1, %8, %9\n"
                              "s_add_u32 s24, s24, 1\n v_fma_f32 %2, %2, %8, %9\n s_add_u32 s26, s26, 1\n v_fma_f32 %3, %3, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                              : "s20", "s22", "s24", "s26", "scc");)
        } else if (KIND == 10) { // v_fma_f32 with an SGPR operand
            REP8(asm volatile("v_fma_f32 %0, %0, s20, %9\n v_fma_f32 %1, %1, s20, %9\n v_fma_f32 %2, %2, s20, %9\n v_fma_f32 %3, %3, s20, %9\n"
                              "v_fma_f32 %4, %4, s20, %9\n v_fma_f32 %5, %5, s20, %9\n v_fma_f32 %6, %6, s20, %9\n v_fma_f32 %7, %7, s20, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20");)
        } else if (KIND == 11) { // v_pk_fma_f32 with an SGPR-pair operand (what the stencil coefficients are)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, s[20:21], %9\n v_pk_fma_f32 %1, %1, s[20:21], %9\n v_pk_fma_f32 %2, %2, s[20:21], %9\n v_pk_fma_f32 %3, %3, s[20:21], %9\n"
                              "v_pk_fma_f32 %4, %4, s[20:21], %9\n v_pk_fma_f32 %5, %5, s[20:21], %9\n v_pk_fma_f32 %6, %6, s[20:21], %9\n v_pk_fma_f32 %7, %7, s[20:21], %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc) : "s20", "s21");)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    if (s == 123.456f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(const char* name, unsigned long long* d) {
    const int iters = 2000;
    for (int nt : {64, 256, 512, 1024}) {
        hipMemset(d, 0, 8 * (1 + 256 * 16));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(nt), 0, 0, d, iters);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(nt), 0, 0, d, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1 + 256 * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < nt / 64; ++w) v.push_back((double)h[1 + b * 16 + w]);
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        const double per_wave = med / (iters * 64.0);           // cycles per instruction as one wave sees it
        const double wps = std::max(1.0, nt / 256.0);           // waves per SIMD
        printf("%-28s %4d threads/CU (%.2g waves/SIMD): %6.2f cycles per instruction per wave, %5.2f per SIMD\n", name, nt, nt / 256.0,
               per_wave, per_wave / wps);
    }
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * (1 + 256 * 16));
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<10>("v_fma_f32 (sgpr operand)", d);
    run<11>("v_pk_fma_f32 (sgpr pair)", d);
    run<4>("v_add_f32", d);
    run<3>("v_pk_add_f32", d);
    run<2>("v_fma_f64", d);
    run<6>("v_add_f64", d);
    run<5>("v_max_f64", d);
    run<7>("v_mov_b64", d);
    run<8>("s_add_u32", d);
    run<9>("s_add_u32 / v_fma_f32 mix", d);
    return 0;
}
