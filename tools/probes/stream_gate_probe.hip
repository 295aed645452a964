// Two questions behind the thin-slab schedule (round 4):
//  (1) can a stream be gated on a value a RUNNING kernel publishes -- hipStreamWaitValue64 on signal memory that a kernel's
//      workgroups add to with system-scope atomics -- and how long after the publication does the gated kernel start?
//  (2) can a few CUs be kept free for the communication kernels with a CU-masked compute stream
//      (hipExtStreamCreateWithCUMask), so that a kernel on another stream starts at once while a launch that fills every
//      unmasked CU (one 512-thread workgroup with ~120 KB of LDS per CU, like the substep kernel) is running?
// hipcc --offload-arch=gfx950 -O3 -o stream_gate_probe stream_gate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <time.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every workgroup: spin `pre` ticks (100 MHz), [publish], spin `post` ticks; stamps {start, publish, end, hw_id} per workgroup
__global__ __launch_bounds__(512) void worker(unsigned long long* flag, int publish_first_n, long long pre, long long post,
                                              unsigned long long* stamps) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) smem[0] = 1;
    const int b = blockIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < pre) __builtin_amdgcn_s_sleep(8);
    long long tp = 0;
    __syncthreads();
    if (b < publish_first_n && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                                   // system scope
        __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        tp = wall_clock64();
    }
    const long long t1 = wall_clock64();
    while (wall_clock64() - t1 < post) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        stamps[4 * b + 0] = (unsigned long long)t0;
        stamps[4 * b + 1] = (unsigned long long)tp;
        stamps[4 * b + 2] = (unsigned long long)wall_clock64();
        stamps[4 * b + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 32);
    }
}

__global__ void stamp_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)wall_clock64();
}

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CUs %d, hipDeviceAttributeCanUseStreamWaitValue %d\n", ncu, can);
    unsigned long long *flag_sig = nullptr, *flag_plain = nullptr, *stamps, *gate;
    hipError_t es = hipExtMallocWithFlags((void**)&flag_sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(es));
    CK(hipMalloc(&flag_plain, 8));
    CK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 4096));
    CK(hipMalloc(&gate, sizeof(unsigned long long) * 64));
    hipStream_t sa, sb;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    const size_t lds = 120 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(worker), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<unsigned long long> h(4 * 4096), hg(64);

    // ---------------- (1) stream gated on a value published by a running kernel
    for (int which = 0; which < 2; ++which) {
        unsigned long long* flag = which == 0 ? flag_sig : flag_plain;
        if (!flag) continue;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(flag, 0, 8));
            CK(hipMemset(gate, 0, 8 * 64));
            CK(hipDeviceSynchronize());
            const int G = ncu;                 // one workgroup per CU, first 32 publish after 20 us, all run 60 us more
            hipLaunchKernelGGL(worker, dim3(G), dim3(512), lds, sa, flag, 32, 2000ll, 6000ll, stamps);
            hipError_t ew = hipStreamWaitValue64(sb, flag, 32, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
            if (ew != hipSuccess) { printf("  hipStreamWaitValue64 on %s memory: %s\n", which == 0 ? "signal" : "plain", hipGetErrorString(ew)); (void)hipGetLastError(); CK(hipDeviceSynchronize()); break; }
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, sb, gate);
            {   // never hang the box: if the gate has not opened within 2 s, open it from the host and say so
                int spins = 0;
                while (hipStreamQuery(sb) == hipErrorNotReady && spins < 2000) { struct timespec ts = {0, 1000000}; nanosleep(&ts, nullptr); ++spins; }
                if (spins >= 2000) {
                    printf("  gate on %s memory did NOT open by itself within 2 s: released from the host\n", which == 0 ? "signal" : "plain");
                    unsigned long long big = 1000000ull;
                    hipStream_t sc;
                    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
                    CK(hipMemcpyAsync(flag, &big, 8, hipMemcpyHostToDevice, sc));
                    CK(hipStreamSynchronize(sc));
                }
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * G, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hg.data(), gate, 8 * 64, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, last_pub = 0, end = 0;
            for (int b = 0; b < G; ++b) { t0 = std::min(t0, h[4 * b]); end = std::max(end, h[4 * b + 2]); if (b < 32) last_pub = std::max(last_pub, h[4 * b + 1]); }
            printf("  gate on %s memory: last publication at %.1f us, gated kernel ran at %.1f us, worker kernel ended at %.1f us  (gated kernel %s the worker's end)\n",
                   which == 0 ? "signal" : "plain ", (last_pub - t0) / 100.0, (hg[0] - t0) / 100.0, (end - t0) / 100.0, hg[0] < end ? "BEFORE" : "after");
        }
    }

    // ---------------- (2) CU-masked compute stream: are the masked-out CUs free for another stream's kernel?
    {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0xFFFFFFFFu);
        // keep CU 0 of every group of 32 (one per XCD if CUs are numbered XCD-major) out of the compute stream
        for (int cu = 0; cu < ncu; cu += 32) mask[cu / 32] &= ~(1u << (cu % 32));
        hipStream_t sm;
        hipError_t em = hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data());
        printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(em));
        if (em == hipSuccess) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(gate, 0, 8 * 64));
                CK(hipDeviceSynchronize());
                const int G = 2 * ncu;         // two rounds of one workgroup per CU: every unmasked CU busy for ~2 x 40 us
                hipLaunchKernelGGL(worker, dim3(G), dim3(512), lds, sm, flag_plain, 0, 2000ll, 2000ll, stamps);
                hipLaunchKernelGGL(stamp_kernel, dim3(8), dim3(64), 0, sb, gate);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * G, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hg.data(), gate, 8 * 64, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, end = 0;
                std::vector<unsigned long long> ids;
                for (int b = 0; b < G; ++b) { t0 = std::min(t0, h[4 * b]); end = std::max(end, h[4 * b + 2]); ids.push_back(h[4 * b + 3]); }
                std::sort(ids.begin(), ids.end());
                const int distinct = (int)(std::unique(ids.begin(), ids.end()) - ids.begin());
                unsigned long long g0 = ~0ull, g1 = 0;
                for (int i = 0; i < 8; ++i) { g0 = std::min(g0, hg[i]); g1 = std::max(g1, hg[i]); }
                printf("  masked stream: %d workgroups on %d distinct (xcc, hw_id) places, ran %.1f us; the other stream's 8 workgroups ran at %.1f .. %.1f us after the start (%s)\n",
                       G, distinct, (end - t0) / 100.0, ((long long)g0 - (long long)t0) / 100.0, ((long long)g1 - (long long)t0) / 100.0,
                       g1 < end ? "while the masked launch was running" : "only after it");
            }
            // reference: the same on an unmasked stream
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(gate, 0, 8 * 64));
                CK(hipDeviceSynchronize());
                const int G = 2 * ncu;
                hipLaunchKernelGGL(worker, dim3(G), dim3(512), lds, sa, flag_plain, 0, 2000ll, 2000ll, stamps);
                hipLaunchKernelGGL(stamp_kernel, dim3(8), dim3(64), 0, sb, gate);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 4 * G, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hg.data(), gate, 8 * 64, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, end = 0;
                for (int b = 0; b < G; ++b) { t0 = std::min(t0, h[4 * b]); end = std::max(end, h[4 * b + 2]); }
                unsigned long long g0 = ~0ull, g1 = 0;
                for (int i = 0; i < 8; ++i) { g0 = std::min(g0, hg[i]); g1 = std::max(g1, hg[i]); }
                printf("  unmasked stream: ran %.1f us; the other stream's 8 workgroups ran at %.1f .. %.1f us after the start\n",
                       (end - t0) / 100.0, ((long long)g0 - (long long)t0) / 100.0, ((long long)g1 - (long long)t0) / 100.0);
            }
        }
    }
    return 0;
}
