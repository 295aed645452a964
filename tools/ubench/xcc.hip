// Which XCD does block b run on?  (HW_REG_XCC_ID)  Build: hipcc -O3 --offload-arch=gfx950 xcc.hip -o xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xf);
    // keep the block alive a little so all blocks of a round are co-resident
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}
int main() {
    for (int nb : {64, 252, 256, 504}) {
        int* d; (void)hipMalloc(&d, nb * sizeof(int));
        hipLaunchKernelGGL(k, dim3(nb), dim3(512), 42560, 0, d);
        std::vector<int> h(nb);
        (void)hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost);
        printf("grid=%d:", nb);
        for (int i = 0; i < 40 && i < nb; ++i) printf(" %d", h[i]);
        int bad = 0;
        for (int i = 0; i < nb; ++i) if (h[i] != h[i % 8]) ++bad;
        printf("  ... blocks not matching the b%%8 pattern: %d\n", bad);
        (void)hipFree(d);
    }
    return 0;
}
