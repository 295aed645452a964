// Accuracy of v_rcp_f64 on gfx950 and of one / two Newton steps on top of it (the intended WENO5 brings its two quotients over ONE
// reciprocal: hj_device.h, upwind_cd<HJ_WENO5>).  Prints the maximum relative error over 2^24 arguments spread over 1e-30 .. 1e30.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void probe(double* err) {
    double e0 = 0, e1 = 0, e2 = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << 24); i += gridDim.x * blockDim.x) {
        const double u = (double)i / (double)(1u << 24);
        const double x = (1.0 + u * 0.9999) * exp2(floor(200.0 * (u * 7919.0 - floor(u * 7919.0)) - 100.0));
        const double exact = 1.0 / x;
        double r = __builtin_amdgcn_rcp(x);
        e0 = fmax(e0, fabs(r - exact) / exact);
        r = r + r * (1.0 - x * r);
        e1 = fmax(e1, fabs(r - exact) / exact);
        r = r + r * (1.0 - x * r);
        e2 = fmax(e2, fabs(r - exact) / exact);
    }
    atomicMax((unsigned long long*)&err[0], __double_as_longlong(e0));
    atomicMax((unsigned long long*)&err[1], __double_as_longlong(e1));
    atomicMax((unsigned long long*)&err[2], __double_as_longlong(e2));
}
int main() {
    double* d; double h[3] = {0, 0, 0};
    (void)hipMalloc(&d, sizeof(h)); (void)hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("v_rcp_f64 max relative error %.3e; after one Newton step %.3e; after two %.3e  (2^-53 = 1.11e-16)\n", h[0], h[1], h[2]);
    return 0;
}
