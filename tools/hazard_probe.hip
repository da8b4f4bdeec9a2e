// hazard_probe: how long after a v_mfma_f64_16x16x4_f64 (and v_mfma_f32_16x16x4_f32) ISSUES may its A / B operand registers be
// overwritten by an LDS read whose data returns asynchronously?  (Evidence tool, round 4.)
//
// Round 3 found, with the instruction stream of panel1_kernel pinned by sched_barrier, that a ds_read whose destination is an
// A/B operand register of an MFMA issued just before it corrupted results; the fix was an ordering convention (T fragments read
// first, A fragments last, a set's registers kept occupied to the end of its k-step).  This probe measures the WINDOW so that the
// convention can be checked on the generated code (tests/test_isa_hazards.py) instead of being trusted.
//
// One wave per workgroup; the fp64 accumulators live in AGPRs as in the product kernels.  Everything between "operands ready" and "result read" is ONE asm statement, so the compiler inserts
// nothing:   [K leading MFMAs on other accumulators: the matrix pipe is busy when the target issues]
//            target:  v_mfma  accT, a, b, accT
//            [D fillers: s_nop 0 | v_mfma on other registers]
//            overwrite: ds_read_b64 a (or b), poison        (or v_mov_b32 as the VALU control)
//            s_waitcnt lgkmcnt(0); pad; result
// The result is compared with the same MFMA computed on a COPY of the operands that nothing overwrites.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/hazard_probe tools/hazard_probe.hip && tools/hazard_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(2);                                                              \
        }                                                                         \
    } while (0)

enum { OVER_A = 0, OVER_B = 1 };
enum { FILL_NOP = 0, FILL_MFMA = 1 };
enum { BY_LDS = 0, BY_VALU = 1 };

// K leading MFMAs, D fillers, WHICH operand is overwritten, FILL kind, BY what.
template <int K, int D, int WHICH, int FILL, int BY, int CHAIN = 0>
__global__ __launch_bounds__(64) void probe_f64(const double* __restrict__ in, double* __restrict__ out, int iters) {
    __shared__ double poison[64];
    const int lane = threadIdx.x;
    poison[lane] = 7.0e7 + lane;  // what the overwrite brings
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)poison + 8u * lane;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        double a = in[lane] + it, b = in[64 + lane] - it;
        const double a0 = a, b0 = b;  // the copy nothing touches
        v4d accT = {0, 0, 0, 0}, l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0}, l2 = {0, 0, 0, 0}, f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
        double fa = a0 * 0.5, fb = b0 * 0.25;  // operands of the leading / filler MFMAs (never overwritten)
        asm volatile(
            "s_nop 7\n\t"
            ".if %[chain] == 0\n\t"
            ".if %[k] > 0\n\t v_mfma_f64_16x16x4_f64 %[l0], %[fa], %[fb], %[l0]\n\t .endif\n\t"
            ".if %[k] > 1\n\t v_mfma_f64_16x16x4_f64 %[l1], %[fa], %[fb], %[l1]\n\t .endif\n\t"
            ".if %[k] > 2\n\t v_mfma_f64_16x16x4_f64 %[l2], %[fa], %[fb], %[l2]\n\t .endif\n\t"
            ".else\n\t"  // dependent chain: the target's srcC is the result of the MFMA in front of it
            ".if %[k] > 0\n\t v_mfma_f64_16x16x4_f64 %[t], %[fa], %[fb], %[t]\n\t .endif\n\t"
            ".if %[k] > 1\n\t v_mfma_f64_16x16x4_f64 %[t], %[fa], %[fb], %[t]\n\t .endif\n\t"
            ".if %[k] > 2\n\t v_mfma_f64_16x16x4_f64 %[t], %[fa], %[fb], %[t]\n\t .endif\n\t"
            ".endif\n\t"
            "v_mfma_f64_16x16x4_f64 %[t], %[a], %[b], %[t]\n\t"
            ".if %[fill] == 0\n\t .rept %[d]\n\t s_nop 0\n\t .endr\n\t .endif\n\t"
            ".if %[fill] == 1\n\t"
            "  .if %[d] > 0\n\t v_mfma_f64_16x16x4_f64 %[f0], %[fa], %[fb], %[f0]\n\t .endif\n\t"
            "  .if %[d] > 1\n\t v_mfma_f64_16x16x4_f64 %[f1], %[fa], %[fb], %[f1]\n\t .endif\n\t"
            "  .if %[d] > 2\n\t v_mfma_f64_16x16x4_f64 %[f0], %[fa], %[fb], %[f0]\n\t .endif\n\t"
            "  .if %[d] > 3\n\t v_mfma_f64_16x16x4_f64 %[f1], %[fa], %[fb], %[f1]\n\t .endif\n\t"
            ".endif\n\t"
            ".if %[by] == 0\n\t"
            "  .if %[which] == 0\n\t ds_read_b64 %[a], %[addr]\n\t .else\n\t ds_read_b64 %[b], %[addr]\n\t .endif\n\t"
            ".else\n\t"
            "  .if %[which] == 0\n\t v_mov_b64 %[a], 2.0\n\t .else\n\t v_mov_b64 %[b], 2.0\n\t .endif\n\t"
            ".endif\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
            "s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
            "s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
            : [t] "+a"(accT), [a] "+v"(a), [b] "+v"(b), [l0] "+a"(l0), [l1] "+a"(l1), [l2] "+a"(l2), [f0] "+a"(f0), [f1] "+a"(f1)
            : [fa] "v"(fa), [fb] "v"(fb), [addr] "v"(addr), [k] "n"(K), [d] "n"(D), [which] "n"(WHICH), [fill] "n"(FILL),
              [by] "n"(BY), [chain] "n"(CHAIN)
            : "memory");
        v4d ref = {0, 0, 0, 0};
        if (CHAIN)
            for (int q = 0; q < K; ++q) ref = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, ref, 0, 0, 0);
        ref = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, ref, 0, 0, 0);
        for (int j = 0; j < 4; ++j) bad += (accT[j] != ref[j]);
        // keep the side results alive
        if (l0[0] + l1[0] + l2[0] + f0[0] + f1[0] == 1.2345e300) out[1] = a + b;
    }
    // per-lane count of wrong accumulator elements
    out[2 + blockIdx.x * 64 + lane] = (double)bad;
}

template <int K, int D, int WHICH, int FILL, int BY>
__global__ __launch_bounds__(64) void probe_f32(const double* __restrict__ in, double* __restrict__ out, int iters) {
    __shared__ float poison[64];
    const int lane = threadIdx.x;
    poison[lane] = 7.0e7f + lane;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)poison + 4u * lane;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        float a = (float)in[lane] + it, b = (float)in[64 + lane] - it;
        const float a0 = a, b0 = b;
        v4f accT = {0, 0, 0, 0}, l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0}, l2 = {0, 0, 0, 0}, f0 = {0, 0, 0, 0}, f1 = {0, 0, 0, 0};
        float fa = a0 * 0.5f, fb = b0 * 0.25f;
        asm volatile(
            "s_nop 7\n\t"
            ".if %[k] > 0\n\t v_mfma_f32_16x16x4_f32 %[l0], %[fa], %[fb], %[l0]\n\t .endif\n\t"
            ".if %[k] > 1\n\t v_mfma_f32_16x16x4_f32 %[l1], %[fa], %[fb], %[l1]\n\t .endif\n\t"
            ".if %[k] > 2\n\t v_mfma_f32_16x16x4_f32 %[l2], %[fa], %[fb], %[l2]\n\t .endif\n\t"
            "v_mfma_f32_16x16x4_f32 %[t], %[a], %[b], %[t]\n\t"
            ".if %[fill] == 0\n\t .rept %[d]\n\t s_nop 0\n\t .endr\n\t .endif\n\t"
            ".if %[fill] == 1\n\t"
            "  .if %[d] > 0\n\t v_mfma_f32_16x16x4_f32 %[f0], %[fa], %[fb], %[f0]\n\t .endif\n\t"
            "  .if %[d] > 1\n\t v_mfma_f32_16x16x4_f32 %[f1], %[fa], %[fb], %[f1]\n\t .endif\n\t"
            "  .if %[d] > 2\n\t v_mfma_f32_16x16x4_f32 %[f0], %[fa], %[fb], %[f0]\n\t .endif\n\t"
            "  .if %[d] > 3\n\t v_mfma_f32_16x16x4_f32 %[f1], %[fa], %[fb], %[f1]\n\t .endif\n\t"
            ".endif\n\t"
            ".if %[by] == 0\n\t"
            "  .if %[which] == 0\n\t ds_read_b32 %[a], %[addr]\n\t .else\n\t ds_read_b32 %[b], %[addr]\n\t .endif\n\t"
            ".else\n\t"
            "  .if %[which] == 0\n\t v_mov_b32 %[a], 0x40000000\n\t .else\n\t v_mov_b32 %[b], 0x40000000\n\t .endif\n\t"
            ".endif\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
            "s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
            : [t] "+v"(accT), [a] "+v"(a), [b] "+v"(b), [l0] "+v"(l0), [l1] "+v"(l1), [l2] "+v"(l2), [f0] "+v"(f0), [f1] "+v"(f1)
            : [fa] "v"(fa), [fb] "v"(fb), [addr] "v"(addr), [k] "n"(K), [d] "n"(D), [which] "n"(WHICH), [fill] "n"(FILL),
              [by] "n"(BY)
            : "memory");
        v4f ref = {0, 0, 0, 0};
        ref = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, ref, 0, 0, 0);
        for (int j = 0; j < 4; ++j) bad += (accT[j] != ref[j]);
        if (l0[0] + l1[0] + l2[0] + f0[0] + f1[0] == 1.2345e30f) out[1] = a + b;
    }
    out[2 + blockIdx.x * 64 + lane] = (double)bad;
}

struct Result {
    long wrong;      // wrong accumulator elements over all blocks, lanes and iterations
    int lanes_hit;   // distinct lanes (0..63) that ever saw a wrong element
    int first_lane;  // lowest lane hit (or -1)
};

template <typename Kern>
Result run(Kern kern, const double* d_in, double* d_out, int blocks, int iters) {
    CHECK(hipMemset(d_out, 0, sizeof(double) * (2 + 64 * blocks)));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_in, d_out, iters);
    CHECK(hipDeviceSynchronize());
    std::vector<double> h(2 + 64 * blocks);
    CHECK(hipMemcpy(h.data(), d_out, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    Result r{0, 0, -1};
    bool hit[64] = {};
    for (int b = 0; b < blocks; ++b)
        for (int l = 0; l < 64; ++l) {
            const long w = (long)h[2 + b * 64 + l];
            r.wrong += w;
            if (w) hit[l] = true;
        }
    for (int l = 0; l < 64; ++l)
        if (hit[l]) {
            ++r.lanes_hit;
            if (r.first_lane < 0) r.first_lane = l;
        }
    return r;
}

template <int K, int WHICH, int FILL, int BY, int D, int CHAIN = 0>
void row_f64(const double* d_in, double* d_out, int blocks, int iters, char* buf, size_t n) {
    Result r = run(probe_f64<K, D, WHICH, FILL, BY, CHAIN>, d_in, d_out, blocks, iters);
    snprintf(buf + strlen(buf), n - strlen(buf), " %9ld/%-2d", r.wrong, r.lanes_hit);
}
template <int K, int WHICH, int FILL, int BY, int D>
void row_f32(const double* d_in, double* d_out, int blocks, int iters, char* buf, size_t n) {
    Result r = run(probe_f32<K, D, WHICH, FILL, BY>, d_in, d_out, blocks, iters);
    snprintf(buf + strlen(buf), n - strlen(buf), " %9ld/%-2d", r.wrong, r.lanes_hit);
}

template <int K, int WHICH, int FILL, int BY, int CHAIN = 0>
void sweep(const char* label, bool f64, const double* d_in, double* d_out, int blocks, int iters) {
    char buf[1024];
    buf[0] = 0;
#define ROW(D)                                                                    \
    if (f64) row_f64<K, WHICH, FILL, BY, D, CHAIN>(d_in, d_out, blocks, iters, buf, sizeof(buf)); \
    else row_f32<K, WHICH, FILL, BY, D>(d_in, d_out, blocks, iters, buf, sizeof(buf));
    if (FILL == FILL_NOP) {
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(6) ROW(8) ROW(12) ROW(16) ROW(24) ROW(32) ROW(48)
    } else {
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4)
    }
#undef ROW
    printf("%-64s%s\n", label, buf);
}

int main() {
    const int blocks = 256, iters = 200;  // one wave per workgroup on every CU; 256 x 200 x 256 accumulator elements per cell
    double *d_in, *d_out;
    std::vector<double> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 1.0 + 0.03125 * i;
    CHECK(hipMalloc(&d_in, sizeof(double) * 128));
    CHECK(hipMalloc(&d_out, sizeof(double) * (2 + 64 * blocks)));
    CHECK(hipMemcpy(d_in, h.data(), sizeof(double) * 128, hipMemcpyHostToDevice));
    printf("cell = wrong accumulator elements / distinct lanes hit, out of %ld elements per cell\n", (long)blocks * iters * 256);
    for (int f64 = 1; f64 >= 0; --f64) {
        printf("\n=== %s: target MFMA, then D x s_nop 0, then the overwrite of its operand ===\n",
               f64 ? "v_mfma_f64_16x16x4_f64 + ds_read_b64" : "v_mfma_f32_16x16x4_f32 + ds_read_b32");
        printf("%-64s%s\n", "D (wait states between the MFMA and the read) =",
               "         0          1          2          3          4          6          8         12         16         24         32         48");
        sweep<0, OVER_A, FILL_NOP, BY_LDS>("LDS read -> srcA, pipe idle (K = 0 leading MFMAs)", f64, d_in, d_out, blocks, iters);
        sweep<1, OVER_A, FILL_NOP, BY_LDS>("LDS read -> srcA, K = 1 leading MFMA in the pipe", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_A, FILL_NOP, BY_LDS>("LDS read -> srcA, K = 2", f64, d_in, d_out, blocks, iters);
        sweep<3, OVER_A, FILL_NOP, BY_LDS>("LDS read -> srcA, K = 3", f64, d_in, d_out, blocks, iters);
        sweep<0, OVER_B, FILL_NOP, BY_LDS>("LDS read -> srcB, K = 0", f64, d_in, d_out, blocks, iters);
        sweep<1, OVER_B, FILL_NOP, BY_LDS>("LDS read -> srcB, K = 1", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_B, FILL_NOP, BY_LDS>("LDS read -> srcB, K = 2", f64, d_in, d_out, blocks, iters);
        sweep<3, OVER_B, FILL_NOP, BY_LDS>("LDS read -> srcB, K = 3", f64, d_in, d_out, blocks, iters);
        if (f64) {
            sweep<1, OVER_A, FILL_NOP, BY_LDS, 1>("LDS read -> srcA, K = 1, DEPENDENT chain (same accumulator)", f64, d_in, d_out, blocks, iters);
            sweep<3, OVER_A, FILL_NOP, BY_LDS, 1>("LDS read -> srcA, K = 3, DEPENDENT chain", f64, d_in, d_out, blocks, iters);
            sweep<1, OVER_B, FILL_NOP, BY_LDS, 1>("LDS read -> srcB, K = 1, DEPENDENT chain", f64, d_in, d_out, blocks, iters);
            sweep<3, OVER_B, FILL_NOP, BY_LDS, 1>("LDS read -> srcB, K = 3, DEPENDENT chain", f64, d_in, d_out, blocks, iters);
        }
        sweep<0, OVER_A, FILL_NOP, BY_VALU>("control: v_mov -> srcA (VALU write), K = 0", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_A, FILL_NOP, BY_VALU>("control: v_mov -> srcA (VALU write), K = 2", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_B, FILL_NOP, BY_VALU>("control: v_mov -> srcB (VALU write), K = 2", f64, d_in, d_out, blocks, iters);
        printf("\n=== the same with D independent MFMAs (other registers) between the target and the overwrite ===\n");
        printf("%-64s%s\n", "D (MFMAs between) =", "         0          1          2          3          4");
        sweep<0, OVER_A, FILL_MFMA, BY_LDS>("LDS read -> srcA, K = 0", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_A, FILL_MFMA, BY_LDS>("LDS read -> srcA, K = 2", f64, d_in, d_out, blocks, iters);
        sweep<0, OVER_B, FILL_MFMA, BY_LDS>("LDS read -> srcB, K = 0", f64, d_in, d_out, blocks, iters);
        sweep<2, OVER_B, FILL_MFMA, BY_LDS>("LDS read -> srcB, K = 2", f64, d_in, d_out, blocks, iters);
    }
    CHECK(hipFree(d_in));
    CHECK(hipFree(d_out));
    return 0;
}
