// HBM bandwidth microbenchmark: 2 reads + 1 write per element (the RK stage pattern) at 8 and 16
// bytes per lane.  Build: hipcc -O3 --offload-arch=gfx950 bw.hip -o bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename V>
__global__ void triad(const V* __restrict__ a, const V* __restrict__ b, V* __restrict__ c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        V x = a[i], y = b[i];
        V o;
        if constexpr (sizeof(V) == 8) o = x * 0.75 + y * 0.25;
        else { o.x = x.x * 0.75 + y.x * 0.25; o.y = x.y * 0.75 + y.y * 0.25; }
        c[i] = o;
    }
}
template <typename V>
__global__ void copyk(const V* __restrict__ a, V* __restrict__ c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c[i] = a[i];
}
int main() {
    size_t n = (size_t)1 << 27;  // doubles: 1 GiB per array
    double *a, *b, *c;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        for (int mode = 0; mode < 4; ++mode) {
            float ms = 0;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(e0);
                if (mode == 0) triad<double><<<blocks, 256>>>(a, b, c, n);
                if (mode == 1) triad<double2><<<blocks, 256>>>((double2*)a, (double2*)b, (double2*)c, n / 2);
                if (mode == 2) copyk<double><<<blocks, 256>>>(a, c, n);
                if (mode == 3) copyk<double2><<<blocks, 256>>>((double2*)a, (double2*)c, n / 2);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (it == 0 || t < ms) ms = t;
            }
            double bytes = (mode < 2 ? 3.0 : 2.0) * n * 8;
            printf("blocks=%5d %s %2dB/lane: %.3f ms  %.2f TB/s\n", blocks, mode < 2 ? "triad" : "copy ", (mode & 1) ? 16 : 8, ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
