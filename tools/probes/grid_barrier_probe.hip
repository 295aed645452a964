// How long does a grid-wide barrier take on the MI355X?  G workgroups of NT threads, K barriers in a row, each followed by
// a small read of data another workgroup wrote before the barrier (checks the release/acquire pairing across XCDs).
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// sense-free counting barrier: every workgroup adds 1, waits until the count reaches G * (generation + 1)
__device__ inline bool grid_barrier(unsigned* count, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(count, 1u, __ATOMIC_RELEASE);                      // agent scope by default for global memory
        long long t0 = wall_clock64();
        while (__atomic_load_n(count, __ATOMIC_ACQUIRE) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ll) { atomicExch(err, 1u); ok = false; break; }   // 2 s at 100 MHz
        }
    }
    __syncthreads();
    return ok;
}

__global__ void probe(unsigned* count, unsigned* err, double* data, int K, int n_per_wg, double* sink) {
    const int G = gridDim.x, b = blockIdx.x;
    double acc = 0;
    for (int k = 0; k < K; ++k) {
        // write my slice, barrier, read the next workgroup's slice
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) data[(size_t)(k & 1) * G * n_per_wg + (size_t)b * n_per_wg + i] = (double)(k * 1000 + b);
        __threadfence();
        if (!grid_barrier(count, (unsigned)G * (unsigned)(k + 1), err)) return;
        const int nb = (b + G / 2 + 1) % G;
        for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) {
            double v = __builtin_nontemporal_load(&data[(size_t)(k & 1) * G * n_per_wg + (size_t)nb * n_per_wg + i]);
            if (v != (double)(k * 1000 + nb)) atomicExch(err, 2u);
            acc += v;
        }
    }
    if (acc == -1.0) *sink = acc;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 200;
    unsigned *count, *err; double *data, *sink;
    CK(hipMalloc(&count, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&sink, 8));
    const int NPW = 512;
    CK(hipMalloc(&data, sizeof(double) * 2 * 1024 * NPW));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt : {256, 512}) for (int G : {32, 64, 128, 256, 512}) {
        int per_cu = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, probe, nt, 0));
        if (G > per_cu * 256) continue;
        float best = 1e30f; unsigned herr = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(count, 0, 4)); CK(hipMemset(err, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(probe, dim3(G), dim3(nt), 0, 0, count, err, data, K, NPW, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            if (herr) break;
        }
        // launch-only cost: K = 0
        float l0 = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(count, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(probe, dim3(G), dim3(nt), 0, 0, count, err, data, 0, NPW, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); l0 = ms < l0 ? ms : l0;
        }
        printf("NT=%d G=%4d  K=%d barriers: %.1f us total, %.2f us per barrier+4KB exchange   (empty launch %.1f us)  err=%u\n", nt, G, K,
               best * 1e3, (best - l0) * 1e3 / K, l0 * 1e3, herr);
    }
    // back-to-back dependent empty launches on one stream: the floor of a launch-per-stage design
    {
        const int L = 300; float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(count, 0, 4));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < L; ++i) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, count, err, data, 0, NPW, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%d back-to-back empty launches (256 x 512): %.2f us per launch\n", L, best * 1e3 / L);
    }
    return 0;
}
