// Write-only bandwidth of an [N x 1024] fp64 matrix under different assignments of rows to workgroups (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o tools/store_pattern
//   block : workgroup b writes the 64-row block b of one 512-column tile (the K(X,Z) fill up to round 2)
//   sweep : workgroup g of G writes rows g/2, g/2 + G/2, ... of column tile g%2: the chip sweeps the matrix linearly
//   linear: plain 16-byte-per-thread linear fill (what torch.fill_ does)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256) void k_block(double* K, int64_t N) {
    const int t = threadIdx.x, m = blockIdx.y * 512 + 2 * t;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    for (int rr = 0; rr < 64; ++rr) {
        const int64_t n = n0 + rr;
        if (n >= N) break;
        v2d v = {(double)n, (double)m};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(K + n * 1024 + m));
        else *reinterpret_cast<v2d*>(K + n * 1024 + m) = v;
    }
}
template <int NT>
__global__ __launch_bounds__(256) void k_sweep(double* K, int64_t N) {
    const int t = threadIdx.x, m = (blockIdx.x & 1) * 512 + 2 * t;
    for (int64_t n = blockIdx.x >> 1; n < N; n += gridDim.x >> 1) {
        v2d v = {(double)n, (double)m};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(K + n * 1024 + m));
        else *reinterpret_cast<v2d*>(K + n * 1024 + m) = v;
    }
}
__global__ __launch_bounds__(256) void k_linear(double* K, int64_t total2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total2; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<v2d*>(K)[i] = v2d{1.0, 2.0};
}
template <typename F>
void run(const char* name, F f, double bytes) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
    const int64_t N = 1000064; double* K; hipMalloc(&K, N * 1024 * 8); const double bytes = (double)N * 1024 * 8;
    run("block (64 rows x 512 cols per WG)", [&] { hipLaunchKernelGGL(k_block<0>, dim3((N + 63) / 64, 2), dim3(256), 0, 0, K, N); }, bytes);
    run("block, non-temporal", [&] { hipLaunchKernelGGL(k_block<1>, dim3((N + 63) / 64, 2), dim3(256), 0, 0, K, N); }, bytes);
    for (int G : {512, 1024, 1536, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, 64, "sweep, G = %d workgroups", G);
        run(nm, [&] { hipLaunchKernelGGL(k_sweep<0>, dim3(G), dim3(256), 0, 0, K, N); }, bytes);
        snprintf(nm, 64, "sweep, G = %d, non-temporal", G);
        run(nm, [&] { hipLaunchKernelGGL(k_sweep<1>, dim3(G), dim3(256), 0, 0, K, N); }, bytes);
    }
    run("linear fill, 4096 workgroups", [&] { hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, K, N * 512); }, bytes);
    return 0;
}
