// Accuracy of v_rsq_f64 (+ Newton / cubic refinements) and the latency of dependent fp64 VALU ops on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 tools/rsq_probe.hip -o tools/rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_acc(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y0 = __builtin_amdgcn_rsq(v);
    double h = 0.5 * v;
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    double e = fma(-v * y0, y0, 1.0);                      // cubic step from y0
    double yc = fma(y0 * e, fma(e, 0.375, 0.5), y0);
    double e1 = fma(-v * y1, y1, 1.0);                     // one Newton + one residual correction
    double y1c = fma(y1 * 0.5, e1, y1);
    out[5 * i] = y0; out[5 * i + 1] = y1; out[5 * i + 2] = y2; out[5 * i + 3] = yc; out[5 * i + 4] = y1c;
}

template <int KIND>
__global__ void k_lat(double* out, int iters, unsigned long long* clk) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) a = fma(a, b, 1e-12);
            if (KIND == 1) a = a * b;
            if (KIND == 2) a = __builtin_amdgcn_rsq(a);
            if (KIND == 3) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3); a = fma(__hiloint2double(hi, lo), b, 1e-12); }
            asm volatile("" : "+v"(a));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = std::ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s & 63) - 32); }
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_acc, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
    const char* names[5] = {"v_rsq_f64", "+1 Newton", "+2 Newton", "cubic step", "1 Newton + residual"};
    for (int k = 0; k < 5; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) { long double ex = 1.0L / sqrtl((long double)x[i]); long double r = fabsl((o[5 * i + k] - ex) / ex); if (r > worst) worst = r; }
        printf("%-22s max rel err %.3Le  (2^%.1Lf)\n", names[k], worst, log2l(worst));
    }
    unsigned long long* clk; hipMalloc(&clk, 8); unsigned long long c;
    const int iters = 10000;
    const char* ln[4] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_rsq_f64", "readlane x2 + fma"};
    for (int k = 0; k < 4; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            if (k == 0) hipLaunchKernelGGL(k_lat<0>, dim3(1), dim3(64), 0, 0, dout, iters, clk);
            if (k == 1) hipLaunchKernelGGL(k_lat<1>, dim3(1), dim3(64), 0, 0, dout, iters, clk);
            if (k == 2) hipLaunchKernelGGL(k_lat<2>, dim3(1), dim3(64), 0, 0, dout, iters, clk);
            if (k == 3) hipLaunchKernelGGL(k_lat<3>, dim3(1), dim3(64), 0, 0, dout, iters, clk);
            hipDeviceSynchronize();
        }
        hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("%-22s %.1f s_memtime ticks per op (one wave)\n", ln[k], (double)c / (16.0 * iters));
    }
    return 0;
}
