// mfma_peak.hip -- measures the sustained MFMA rate of gfx950 for the two instructions the E-step kernels use
// (v_mfma_f64_16x16x4_f64, v_mfma_f32_16x16x4_f32), operands in registers, no memory traffic: the practical ceiling
// the kernels in t-svgp_amd/csrc are judged against (clock under load included).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && ./tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double* out, int iters, double seed, unsigned long long* clk) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// random operands (4 A and 4 B fragments per lane from a hash): data-dependent power -> the clock the chip really holds
__global__ __launch_bounds__(256) void k_f64_rand(double* out, int iters, double seed, unsigned long long* clk) {
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    double a[4], b[4];
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 4; ++i) {
        h = h * 1664525u + 1013904223u; a[i] = ((double)(h >> 8) / 16777216.0 - 0.5) * 2.0 + seed * 1e-9;
        h = h * 1664525u + 1013904223u; b[i] = ((double)(h >> 8) / 16777216.0 - 0.5) * 2.0;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 2
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// MFMA + LDS operand reads (+ optional global streaming): NREAD ds_read_b64 per 16 MFMAs, random data in LDS.
// Models the E-step kernels' inner loop to see which clock the chip holds under that mix.
template <int NREAD, int GLOADS, int BAR = 0>
__global__ __launch_bounds__(256, 2) void k_f64_lds(double* out, const double* __restrict__ src, int iters, unsigned long long* clk) {
    __shared__ double sm[4608];
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < 4608; i += 256) { h = h * 1664525u + 1013904223u; sm[i] = ((double)(h >> 8) / 16777216.0 - 0.5); }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double* base = sm + (lane & 15) * 17 + (lane >> 4);
    const double* gp = src + ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    double gsum = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        double f[16];
#pragma unroll
        for (int i = 0; i < NREAD; ++i) f[i] = base[((it + i) & 15) * 272 + (i & 3) * 4];
        if (GLOADS) {
#pragma unroll
            for (int g = 0; g < GLOADS; ++g) { v2d_t v = *reinterpret_cast<const v2d_t*>(gp + ((size_t)(it * GLOADS + g) & 4095) * 131072); gsum += v[0] + v[1]; }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f[(i & 3) % (NREAD / 2)]), "v"(f[NREAD / 2 + (i >> 2) % (NREAD / 2)]));
        if (BAR > 0 && (it % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = gsum;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// 4 waves per SIMD: 8 accumulators per wave (32x64 wave tile: 2 A + 4 B fragment reads per 8 MFMAs), barrier every BAR iterations
template <int BAR>
__global__ __launch_bounds__(256, 4) void k_f64_lds8(double* out, int iters, unsigned long long* clk) {
    __shared__ double sm[4608];
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4d{0, 0, 0, 0};
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < 4608; i += 256) { h = h * 1664525u + 1013904223u; sm[i] = ((double)(h >> 8) / 16777216.0 - 0.5); }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double* base = sm + (lane & 15) * 17 + (lane >> 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        double f[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) f[i] = base[((it + i) & 15) * 272 + (i & 3) * 4];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(f[i & 1]), "v"(f[2 + (i >> 1)]));
        if (BAR > 0 && (it % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float seed, unsigned long long* clk) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// As k_f64_lds<8, 0, BAR>, with the accumulators in AGPRs (AG = 1) and / or the operand reads of the NEXT 16 MFMAs issued in
// front of the current ones (PIPE = 1: two register sets, loop unrolled by two): does the 7 % that eight ds_read_b64 per 16
// MFMAs cost come from the LDS latency, or from the reads' VGPR writes competing with the MFMAs' accumulator traffic?
#define MFMA16(ACC, F, CON)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"               \
                                                                 : CON(ACC[i])                                           \
                                                                 : "v"(F[i & 3]), "v"(F[4 + (i >> 2)]));
template <int AG, int PIPE, int BAR>
__global__ __launch_bounds__(256, 2) void k_f64_lds2(double* out, int iters, unsigned long long* clk) {
    __shared__ double sm[4608];
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0, 0, 0, 0};
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = threadIdx.x; i < 4608; i += 256) { h = h * 1664525u + 1013904223u; sm[i] = ((double)(h >> 8) / 16777216.0 - 0.5); }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const double* base = sm + (lane & 15) * 17 + (lane >> 4);
    double f[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = base[(i & 15) * 272 + (i & 3) * 4];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = base[((it + 1 + i) & 15) * 272 + (i & 3) * 4];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = base[((it + i) & 15) * 272 + (i & 3) * 4];
        }
        if (AG) { MFMA16(acc, f, "+a") } else { MFMA16(acc, f, "+v") }
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = base[((it + 2 + i) & 15) * 272 + (i & 3) * 4];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = base[((it + 1 + i) & 15) * 272 + (i & 3) * 4];
        }
        if (AG) { MFMA16(acc, g, "+a") } else { MFMA16(acc, g, "+v") }
        if (BAR > 0 && ((it >> 1) % BAR) == BAR - 1) __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks, int iters, int nacc, double flop_per_mfma) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* clk;
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    for (int rep = 0; rep < 3; ++rep) launch(clk);  // warm-up: let the clock settle under load
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep) launch(clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    const double waves = blocks * 4.0;
    const double flops = waves * (double)iters * nacc * flop_per_mfma * reps;
    const double mfma_per_wave = (double)iters * nacc;
    printf("%-34s blocks=%5d  %.2f TFLOP/s  | %.1f cycles per MFMA per wave, in-kernel clock %.2f GHz\n", name, blocks,
           flops / (ms * 1e-3) / 1e12, cyc / mfma_per_wave, cyc / (rt * 10.0) );
    hipFree(clk);
}

int main() {
    const int iters = 20000;
    double* od; float* of;
    hipMalloc(&od, sizeof(double) * 256 * 2048);
    hipMalloc(&of, sizeof(float) * 256 * 2048);
    for (int blocks : {256, 512, 1024}) {
        run("f64 16x16x4, 16 accumulators", [&](unsigned long long* c) { hipLaunchKernelGGL(k_f64<16>, dim3(blocks), dim3(256), 0, 0, od, iters, 0.37, c); }, blocks, iters, 16, 2048.0);
    }
    for (int blocks : {256, 512}) {
        run("f64 16x16x4, 16 acc, RANDOM operands", [&](unsigned long long* c) { hipLaunchKernelGGL(k_f64_rand, dim3(blocks), dim3(256), 0, 0, od, iters * 4, 0.37, c); }, blocks, iters * 4, 16, 2048.0);
    }
    double* src;
    hipMalloc(&src, (size_t)4096 * 131072 * 8 + (1 << 22));
    hipMemset(src, 0, (size_t)4096 * 131072 * 8 + (1 << 22));
    run("f64 + 8 ds_read_b64 /16 MFMA, 2 WG/CU", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 0>), dim3(512), dim3(256), 0, 0, od, src, iters, c); }, 512, iters, 16, 2048.0);
    run("f64 + 16 ds_read_b64 /16 MFMA, 2 WG/CU", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<16, 0>), dim3(512), dim3(256), 0, 0, od, src, iters, c); }, 512, iters, 16, 2048.0);
    run("f64 + 8 ds_read, barrier / 64 MFMA, 2WG", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 0, 4>), dim3(512), dim3(256), 0, 0, od, src, iters, c); }, 512, iters, 16, 2048.0);
    run("f64 + 8 ds_read, barrier / 64 MFMA, 1WG", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 0, 4>), dim3(256), dim3(256), 0, 0, od, src, iters, c); }, 256, iters, 16, 2048.0);
    run("f64 + 8 ds_read, no barrier, 1WG/CU", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 0, 0>), dim3(256), dim3(256), 0, 0, od, src, iters, c); }, 256, iters, 16, 2048.0);
    run("f64 + 8 ds_read, barrier / 16 MFMA, 2WG", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 0, 1>), dim3(512), dim3(256), 0, 0, od, src, iters, c); }, 512, iters, 16, 2048.0);
    run("8acc: 6 reads/8 MFMA, bar/64 MFMA, 4 waves/SIMD", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds8<8>), dim3(1024), dim3(256), 0, 0, od, iters * 2, c); }, 1024, iters * 2, 8, 2048.0);
    run("8acc: 6 reads/8 MFMA, bar/64 MFMA, 2 waves/SIMD", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds8<8>), dim3(512), dim3(256), 0, 0, od, iters * 2, c); }, 512, iters * 2, 8, 2048.0);
    run("8acc: 6 reads/8 MFMA, no barrier, 4 waves/SIMD", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds8<0>), dim3(1024), dim3(256), 0, 0, od, iters * 2, c); }, 1024, iters * 2, 8, 2048.0);
    run("f64 + 8 ds_read + 1 gload16B /16 MFMA", [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds<8, 1>), dim3(512), dim3(256), 0, 0, od, src, iters, c); }, 512, iters, 16, 2048.0);
#define RUN2(NAME, AG, PIPE, BAR) run(NAME, [&](unsigned long long* c) { hipLaunchKernelGGL((k_f64_lds2<AG, PIPE, BAR>), dim3(512), dim3(256), 0, 0, od, iters, c); }, 512, iters, 16, 2048.0)
    RUN2("8 reads/16 MFMA, acc VGPR, no barrier", 0, 0, 0);
    RUN2("8 reads/16 MFMA, acc AGPR, no barrier", 1, 0, 0);
    RUN2("  reads one body ahead, acc VGPR", 0, 1, 0);
    RUN2("  reads one body ahead, acc AGPR", 1, 1, 0);
    RUN2("8 reads/16 MFMA, VGPR, barrier/64 MFMA", 0, 0, 2);
    RUN2("8 reads/16 MFMA, AGPR, barrier/64 MFMA", 1, 0, 2);
    RUN2("  reads ahead, VGPR, barrier/64 MFMA", 0, 1, 2);
    RUN2("  reads ahead, AGPR, barrier/64 MFMA", 1, 1, 2);
    for (int blocks : {256, 512}) {
        run("f64 16x16x4, 4 accumulators", [&](unsigned long long* c) { hipLaunchKernelGGL(k_f64<4>, dim3(blocks), dim3(256), 0, 0, od, iters * 4, 0.37, c); }, blocks, iters * 4, 4, 2048.0);
    }
    for (int blocks : {256, 512, 1024}) {
        run("f32 16x16x4, 16 accumulators", [&](unsigned long long* c) { hipLaunchKernelGGL(k_f32<16>, dim3(blocks), dim3(256), 0, 0, of, iters, 0.37f, c); }, blocks, iters, 16, 2048.0);
    }
    return 0;
}
