// What does a vector load cost the CU's texture-address / L1 path, by access SHAPE?  (round 5)
// The 4-D kernel (C5) is not bound by bytes (8 B/clk/CU), nor by instruction issue or LDS after the round-5 rework: TA_BUSY 76 %,
// 38 TA-busy cycles per wave-instruction.  This program times L1/L2-resident loads per CU for
//   width  4 / 8 / 16 bytes per lane
//   shape  "line": 64 lanes contiguous, base 128-B aligned;  "mis": the same shifted by 8 bytes;
//          "rowsNN": lanes dealt to rows of NN bytes that start every 516 bytes (129 floats: the C5 grid), base shifted by 8 B
// with 8 waves per CU (2 per SIMD), each wave issuing batches of 8 independent loads.
// Output: cycles per wave-instruction per CU (lower = cheaper), bytes per clock per CU.
// Build: hipcc -O3 --offload-arch=gfx950 l1_rate.hip -o l1_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int W> struct Vec;
template <> struct Vec<4> { typedef float T; };
template <> struct Vec<8> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct Vec<16> { typedef float T __attribute__((ext_vector_type(4))); };

// region: bytes each WORKGROUP cycles through (8 KB: L1 hits; 2 MB: L2 hits)
template <int W>
__global__ __launch_bounds__(512) void load_kernel(const char* __restrict__ buf, unsigned long long* out, int iters, int row_bytes, int shift,
                                                   unsigned region, unsigned step) {
    typedef typename Vec<W>::T V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned off;
    if (row_bytes <= 0) off = (unsigned)lane * W + shift;
    else {
        const int lpr = row_bytes / W;          // lanes per row
        off = (unsigned)(lane / lpr) * 516u + (unsigned)(lane % lpr) * W + shift;
    }
    const char* base = buf + (size_t)blockIdx.x * region;
    unsigned pos = (unsigned)wave * 4096u;      // waves start apart
    V acc = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        V v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = *reinterpret_cast<const V*>(base + ((pos + (unsigned)k * step) & (region - 1u)) + off);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
        pos += 8 * step;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s;
    if constexpr (W == 4) s = acc; else s = acc.x;
    if (s == 123.456f) out[0] = 1;
    if (lane == 0) out[1 + blockIdx.x * 8 + wave] = t1 - t0;
}

template <int W> void run(const char* name, const char* d, unsigned long long* o, int row_bytes, int shift, unsigned region, unsigned step) {
    const int iters = 400, nt = 512;
    hipMemset(o, 0, 8 * (1 + 256 * 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(load_kernel<W>, dim3(256), dim3(nt), 0, 0, d, o, iters, row_bytes, shift, region, step);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + 256 * 8);
    hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < nt / 64; ++w) v.push_back((double)h[1 + b * 8 + w]);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2];
    const double per_instr_cu = med / (iters * 8.0) / (nt / 64);      // cycles per wave-instruction seen by the CU
    printf("%-34s W=%2d region %7u: %6.1f cycles per wave-instruction per CU, %5.1f B/clk/CU\n", name, W, region, per_instr_cu, 64.0 * W / per_instr_cu);
}

int main() {
    char* d;
    unsigned long long* o;
    const size_t bytes = (size_t)256 * (4u << 20) + (1 << 20);
    hipMalloc(&d, bytes);
    hipMemset(d, 0, bytes);
    hipMalloc(&o, 8 * (1 + 256 * 8));
    for (unsigned region : {16384u, 2097152u}) {
        const unsigned step = region == 16384u ? 2048u : 8256u;     // the waves walk the region
        run<4>("line (aligned)", d, o, 0, 0, region, step);
        run<8>("line (aligned)", d, o, 0, 0, region, step);
        run<16>("line (aligned)", d, o, 0, 0, region, step);
        run<4>("line shifted 8 B", d, o, 0, 8, region, step);
        run<8>("line shifted 8 B", d, o, 0, 8, region, step);
        run<16>("line shifted 8 B", d, o, 0, 8, region, step);
        run<4>("rows of 136 B / 516", d, o, 136, 8, region, step);
        run<8>("rows of 136 B / 516", d, o, 136, 8, region, step);
        run<8>("rows of 128 B / 516", d, o, 128, 8, region, step);
        run<16>("rows of 128 B / 516", d, o, 128, 8, region, step);
        run<8>("rows of 264 B / 516", d, o, 264, 8, region, step);
        run<16>("rows of 256 B / 516", d, o, 256, 8, region, step);
        run<8>("rows of 512 B / 516", d, o, 512, 8, region, step);
        run<4>("rows of 12 B / 516 (halo cols)", d, o, 12, 8, region, step);
        run<4>("rows of 24 B / 516", d, o, 24, 8, region, step);
    }
    return 0;
}
