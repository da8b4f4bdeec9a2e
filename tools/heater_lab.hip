// heater_lab.hip -- round 5, verdict item 4 (the moments kernel's clock tax): does the clock the chip holds under the MFMA kernels
// drop over a ~1.5 ms stretch of latency-bound launches, and does a kernel that keeps the matrix pipes busy over that stretch hold
// it?  Two kernels bounded by the 100 MHz real-time counter: lab_sleep (one wave asleep) and lab_heat (nwg workgroups of four waves
// chaining v_mfma_f64_16x16x4_f64, or fp64 FMAs, on registers).  Built on the GPU box by tools/clock_lab.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void sleep_kernel(unsigned ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// mode 0: MFMA fp64; 1: fp64 FMA; out: never written unless the values misbehave (keeps the chain alive)
__global__ __launch_bounds__(256) void heat_kernel(unsigned ticks, int mode, double* out) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9;
    do {
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                acc0 = acc0 * a + b;
                acc1 = acc1 * a + b;
            }
        }
    } while (__builtin_amdgcn_s_memrealtime() - t0 < ticks);
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678) out[0] = acc0[0];
}

// duty-cycled fp64 FMAs: bursts of 64 x 2 x 4 FMAs, then s_sleep `nap` (units of 64 clocks) -- and, with burst = 0, waves that only sleep
__global__ __launch_bounds__(256) void duty_kernel(unsigned ticks, int burst, int nap, double* out) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9;
    do {
        for (int r = 0; r < burst; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0 = acc0 * a + b;
                acc1 = acc1 * a + b;
            }
        }
        for (int r = 0; r < nap; ++r) __builtin_amdgcn_s_sleep(8);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < ticks);
    if (acc0[0] + acc1[1] == 12345.678) out[0] = acc0[0];
}
extern "C" int lab_duty(double us, int nwg, int threads, int burst, int nap, double* out, void* stream) {
    hipLaunchKernelGGL(duty_kernel, dim3(nwg), dim3(threads), 0, (hipStream_t)stream, (unsigned)(us * 100.0), burst, nap, out);
    return (int)hipGetLastError();
}
extern "C" int lab_sleep(double us, void* stream) {
    hipLaunchKernelGGL(sleep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)(us * 100.0));
    return (int)hipGetLastError();
}
extern "C" int lab_heat(double us, int nwg, int mode, double* out, void* stream) {
    hipLaunchKernelGGL(heat_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (unsigned)(us * 100.0), mode, out);
    return (int)hipGetLastError();
}
