// panel_lab.hip -- bench for loop structures of the moments kernel (fp64, upper form, fused mean) against the production
// kernel of the same process (libtsvgp_hip.so through dlopen), interleaved rounds, same random operands.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I include tools/panel_lab.hip -o tools/panel_lab -ldl
//   ./tools/panel_lab [rows] [path to libtsvgp_hip.so]
// Variant here: "pipe" -- the chunk stream of a row panel software-pipelined at k-step granularity: two fragment register
// sets (the reads of k-step s+1 issue under the MFMAs of k-step s), ONE barrier per chunk placed in front of the chunk's
// last k-step (the first fragments of the next chunk are read under those MFMAs), column tiles chained without draining the
// pipeline, line-coalesced staging loads (8 lanes per 128-byte row run).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>
#include <cstdint>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int TILE = 128, KC = 16, RS = 17, NT = 256;
#ifndef LAB_OCC
#define LAB_OCC 2
#endif
constexpr int OPS = TILE * RS;       // doubles per operand image
constexpr int BUFS = 2 * OPS;        // doubles per chunk buffer (A image + T image)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Frag { double a[2]; double b[8]; };

template <int MASK>
__device__ __forceinline__ void rd(Frag& f, const double* ap0, const double* ap1, const double* bp, const int off) {
    f.a[0] = ap0[off];
    f.a[1] = ap1[off];
#pragma unroll
    for (int n = 0; n < 8; ++n)
        if (MASK & (1 << n)) f.b[n] = bp[off + n * 16 * RS];
}
template <int MASK>
__device__ __forceinline__ void mm(v4d (&acc)[2][8], const Frag& f) {
#pragma unroll
    for (int n = 0; n < 8; ++n)
        if (MASK & (1 << n)) {
            acc[0][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[0], f.b[n], acc[0][n], 0, 0, 0);
            acc[1][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[1], f.b[n], acc[1][n], 0, 0, 0);
        }
}
// interleave: one LDS read behind each of the first NR MFMAs of the group
template <int NR>
__device__ __forceinline__ void sched_mfma_dsread() {
#ifndef LAB_NOSCHED
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#endif
}
constexpr int popc(int m) { return m ? (m & 1) + popc(m >> 1) : 0; }

struct Cursor {  // a position in the workgroup's chunk stream: column tile and k-chunk (upper form: chunks it*8 .. nchunk-1)
    int it, c;
};

template <int DUMMY = 0>
__global__ __launch_bounds__(NT, 2) void moments_pipe(const double* __restrict__ A, const double* __restrict__ Tm,
                                                      const double* __restrict__ gamma, double* __restrict__ qout,
                                                      double* __restrict__ mout, int Mp) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    __shared__ __attribute__((aligned(16))) double lds[2 * BUFS];
    __shared__ double rowq[TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntile = Mp / TILE, nchunk = Mp / KC;
    const int64_t n0 = (int64_t)blockIdx.x * TILE;

    // staging role: piece q (0..3) of a chunk image = 16 bytes of row q*32 + (t>>3), doubles 2*(t&7), 2*(t&7)+1
    const int prow = t >> 3, pseg = t & 7;
    const unsigned voff0 = (unsigned)((prow * Mp + 2 * pseg) * sizeof(double));  // byte offset inside a 32-row group
    const char* Ab = reinterpret_cast<const char*>(A + n0 * Mp);
    const size_t grp = (size_t)32 * Mp * sizeof(double);  // bytes between the row groups of consecutive pieces
    double* const lw = lds + prow * RS + 2 * pseg;

    // fragment read addresses ([row][k] images, row stride 17 doubles)
    const int lr = lane & 15, lk = lane >> 4;
    const double* const ap0 = lds + (w * 16 + lr) * RS + lk;
    const double* const ap1 = lds + ((7 - w) * 16 + lr) * RS + lk;
    const double* const bp = lds + OPS + lr * RS + lk;

    for (int j = t; j < Mp; j += NT) gsm[j] = gamma[j];

    v2d ra[4], rb[4];
    double mpart[4] = {0, 0, 0, 0};
    auto fetch = [&](const Cursor cu) {
        const unsigned vo = voff0 + (unsigned)(cu.c * KC * sizeof(double));
        const char* Tb = reinterpret_cast<const char*>(Tm + (size_t)cu.it * TILE * Mp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const v2d*>(Ab + q * grp + vo);
            rb[q] = *reinterpret_cast<const v2d*>(Tb + q * grp + vo);
        }
    };
    auto advance = [&](Cursor& cu) {  // next chunk of the stream; it == ntile: past the end
        if (++cu.c == nchunk) {
            ++cu.it;
            cu.c = cu.it * (TILE / KC);
        }
    };
    auto stage = [&](const int buf, const int c_staged, const bool with_gamma) {
        if (with_gamma) {
            const v2d g = *reinterpret_cast<const v2d*>(gsm + c_staged * KC + 2 * pseg);
#pragma unroll
            for (int q = 0; q < 4; ++q) mpart[q] += ra[q][0] * g[0] + ra[q][1] * g[1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double* p = lw + buf * BUFS + q * 32 * RS;
            p[0] = ra[q][0];
            p[1] = ra[q][1];
            p[OPS] = rb[q][0];
            p[OPS + 1] = rb[q][1];
        }
    };

    v4d acc[2][8];
    Frag f0, f1;
    Cursor cf{0, 0};  // fetch cursor: runs two chunks ahead of the compute position
    double rs_mine = 0.0;

    // ---- one chunk of the stream.  M: its column-block mask; MN: the mask of the NEXT chunk of the stream (0: none);
    // BUF: LDS buffer of this chunk (= parity of its position in the stream); GN: the next chunk belongs to column tile 0
    // (its A values also feed the mean).  On entry f0 holds the fragments of k-step 0 and the registers ra / rb the next chunk.
    auto chunk = [&](auto m_tag, auto mn_tag, auto buf_tag, auto gn_tag, const int c_next) {
        constexpr int M = decltype(m_tag)::value, MN = decltype(mn_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr bool GN = decltype(gn_tag)::value;
        constexpr int B0 = BUF * BUFS, B1 = (BUF ^ 1) * BUFS, NRD = 2 + popc(M);
        rd<M>(f1, ap0, ap1, bp, B0 + 4);
        mm<M>(acc, f0);
        sched_mfma_dsread<NRD>();
        rd<M>(f0, ap0, ap1, bp, B0 + 8);
        mm<M>(acc, f1);
        sched_mfma_dsread<NRD>();
        rd<M>(f1, ap0, ap1, bp, B0 + 12);
        if constexpr (MN != 0) stage(BUF ^ 1, c_next, GN);
        mm<M>(acc, f0);
        sched_mfma_dsread<NRD>();
        __syncthreads();
        if constexpr (MN != 0) {
            if (cf.it < ntile) {
                fetch(cf);
                advance(cf);
            }
            rd<MN>(f0, ap0, ap1, bp, B1);
        }
        mm<M>(acc, f1);
        if constexpr (MN != 0) sched_mfma_dsread<2 + popc(MN)>();
    };
#define IC(x) std::integral_constant<int, (x)>{}
#define BC(x) std::integral_constant<bool, (x)>{}

    // prologue: chunk (0, 0) staged, chunk (0, 1) in registers, first fragments read
    fetch(cf);
    advance(cf);
    __syncthreads();  // gamma in LDS
    stage(0, 0, true);
    fetch(cf);
    advance(cf);
    __syncthreads();
    rd<0x01>(f0, ap0, ap1, bp, 0);

    for (int it = 0; it < ntile; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[s][n] = v4d{0, 0, 0, 0};
        const int cd = it * (TILE / KC);
        const bool last_tile = it + 1 == ntile;
        auto diag = [&](auto gn) {
            chunk(IC(0x01), IC(0x03), IC(0), gn, cd + 1);
            chunk(IC(0x03), IC(0x07), IC(1), gn, cd + 2);
            chunk(IC(0x07), IC(0x0F), IC(0), gn, cd + 3);
            chunk(IC(0x0F), IC(0x1F), IC(1), gn, cd + 4);
            chunk(IC(0x1F), IC(0x3F), IC(0), gn, cd + 5);
            chunk(IC(0x3F), IC(0x7F), IC(1), gn, cd + 6);
            chunk(IC(0x7F), IC(0xFF), IC(0), gn, cd + 7);
            if (!last_tile) chunk(IC(0xFF), IC(0xFF), IC(1), gn, cd + 8);
            else chunk(IC(0xFF), IC(0), IC(1), BC(false), 0);
        };
        auto full = [&](auto gn) {
            // full k-tiles behind the diagonal one: (ntile - 1 - it) * 8 chunks, all but the last followed by another full one;
            // the last one is followed by the first chunk of the next column tile (mask 0x01, never a gamma chunk)
            const int c_last = nchunk - 1;
            for (int c = cd + 8; c < c_last - 1; c += 2) {
                chunk(IC(0xFF), IC(0xFF), IC(0), gn, c + 1);
                chunk(IC(0xFF), IC(0xFF), IC(1), gn, c + 2);
            }
            chunk(IC(0xFF), IC(0xFF), IC(0), gn, c_last);
            chunk(IC(0xFF), IC(0x01), IC(1), BC(false), (it + 1) * (TILE / KC));
        };
        if (it == 0) {
            diag(BC(true));
            if (!last_tile) full(BC(true));
        } else {
            diag(BC(false));
            if (!last_tile) full(BC(false));
        }
        // epilogue of the column tile: squares of the tile's entries, summed per row
        double keep = 0.0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q = 0.0;
#pragma unroll
                for (int n = 0; n < 8; ++n) q += acc[s][n][r] * acc[s][n][r];
                q += __shfl_xor(q, 1);
                q += __shfl_xor(q, 2);
                q += __shfl_xor(q, 4);
                q += __shfl_xor(q, 8);
                keep = ((lane & 7) == s * 4 + r) ? q : keep;
            }
        rs_mine += keep;
    }
    if ((lane & 15) < 8) {
        const int l8 = lane & 15;
        rowq[((l8 >> 2) == 0 ? w : 7 - w) * 16 + (lane >> 4) + 4 * (l8 & 3)] = rs_mine;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mpart[q] += __shfl_xor(mpart[q], 1);
        mpart[q] += __shfl_xor(mpart[q], 2);
        mpart[q] += __shfl_xor(mpart[q], 4);
    }
    __syncthreads();
    if (t < TILE) qout[n0 + t] = rowq[t];
    if (pseg == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mout[n0 + q * 32 + prow] = mpart[q];
    }
}


// =====================================================================================================================
// Variant "dma": the pipelined chunk stream of "pipe" with the operands moved global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write, no VALU on the way), which leaves room for the second
// fragment register set without spilling.  LDS images are [row][8 units of 16 bytes] with NO padding (a DMA instruction
// writes 1 KiB lane-linear); bank conflicts are removed by XOR-ing the unit index with f(row) = (row & 7) ^ ((row >> 3) & 1)
// on the per-lane global source address and on the fragment reads (conflict-free for ds_read_b64 over each half wave).
// The mean (column tile 0) reads the thread's own 4 x 16 bytes of the landed A image back from LDS.
// =====================================================================================================================
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int DOPS = TILE * KC;     // doubles per operand image (16 KB)
constexpr int DBUFS = 2 * DOPS;     // doubles per chunk buffer (32 KB)

template <int MASK>
__device__ __forceinline__ void rd2(Frag& f, const double* lds, const int a0, const int a1, const int b, const int imm) {
    f.a[0] = lds[a0 + imm];
    f.a[1] = lds[a1 + imm];
#pragma unroll
    for (int n = 0; n < 8; ++n)
        if (MASK & (1 << n)) f.b[n] = lds[b + imm + DOPS + n * 256];
}

template <int DUMMY = 0>
__global__ __launch_bounds__(NT, LAB_OCC) void moments_dma(const double* __restrict__ A, const double* __restrict__ Tm,
                                                     const double* __restrict__ gamma, double* __restrict__ qout,
                                                     double* __restrict__ mout, int Mp) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    __shared__ __attribute__((aligned(1024))) double lds[2 * DBUFS];
    __shared__ double rowq[TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntile = Mp / TILE, nchunk = Mp / KC;
    const int64_t n0 = (int64_t)blockIdx.x * TILE;

    // DMA role: wave w moves the 1-KiB pieces P = 4 q + w (q = 0..3) of each operand image: rows 8 P .. 8 P + 7; lane L lands
    // at unit L & 7 of row 8 P + (L >> 3) and fetches the global unit (L & 7) ^ f(row), f(row) = (L >> 3) ^ (w & 1)
    const int drow = lane >> 3;
    const int dlog = (lane & 7) ^ drow ^ (w & 1);
    const unsigned dvoff = (unsigned)((drow * Mp + 2 * dlog) * sizeof(double));
    const char* Ab = reinterpret_cast<const char*>(A + (n0 + 8 * w) * Mp);
    const size_t grp = (size_t)32 * Mp * sizeof(double);

    // fragment reads: row r of a 16-row block, element k = 4 ks + lk of the chunk -> unit ((2 ks + (lk >> 1)) ^ f), half lk & 1
    const int lr = lane & 15, lk = lane >> 4;
    const int fr = (lr & 7) ^ (lr >> 3);
    int offa0[4], offa1[4], offb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int o = lr * 16 + (((2 * ks + (lk >> 1)) ^ fr) << 1) + (lk & 1);
        offb[ks] = o;
        offa0[ks] = o + w * 256;
        offa1[ks] = o + (7 - w) * 256;
    }
    // mean (tile 0): this thread's own 16-byte units of the A image: unit t & 7 of rows q * 32 + (t >> 3)
    const int glog = (t & 7) ^ ((t >> 3) & 7) ^ (w & 1);  // the logical unit behind physical unit t & 7 of those rows

    for (int j = t; j < Mp; j += NT) gsm[j] = gamma[j];

    double mpart[4] = {0, 0, 0, 0};
    // The DMA instructions are issued from inline asm: the compiler then knows nothing about LDS writes in flight and puts no
    // "s_waitcnt vmcnt" in front of the fragment reads of OTHER buffers (with the builtin it waits for the DMA it has just
    // issued before the next ds_read of the loop: the whole memory latency, every chunk).  Ordering is by hand: vmcnt(0) +
    // barrier in front of the first read of a landed chunk.
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * 1024);
    auto dma = [&](const Cursor cu, const int buf) {
        const unsigned vo = dvoff + (unsigned)(cu.c * KC * sizeof(double));
        const char* Tb = reinterpret_cast<const char*>(Tm + ((size_t)cu.it * TILE + 8 * w) * Mp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned la = lds_base + (unsigned)((buf * DBUFS + q * 512) * sizeof(double));
            const char* ga = Ab + q * grp;
            const char* gt = Tb + q * grp;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(vo), "s"(ga) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la + (unsigned)(DOPS * sizeof(double))), "v"(vo), "s"(gt) : "memory", "m0");
        }
    };
    auto advance = [&](Cursor& cu) {
        if (++cu.c == nchunk) {
            ++cu.it;
            cu.c = cu.it * (TILE / KC);
        }
    };
    auto mean_part = [&](const int buf, const int c) {  // the A image of chunk c (landed, behind a barrier) times gamma
        const v2d g = *reinterpret_cast<const v2d*>(gsm + c * KC + 2 * glog);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v2d x = *reinterpret_cast<const v2d*>(lds + buf * DBUFS + q * 512 + t * 2);
            mpart[q] += x[0] * g[0] + x[1] * g[1];
        }
    };

    v4d acc[2][8];
    Frag f0, f1;
    Cursor cf{0, 0};
    double rs_mine = 0.0;

    // one chunk of the stream (see moments_pipe::chunk).  GC: this chunk belongs to column tile 0 (its A image feeds the mean)
    auto chunk = [&](auto m_tag, auto mn_tag, auto buf_tag, auto gc_tag, const int c_this) {
        constexpr int M = decltype(m_tag)::value, MN = decltype(mn_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr bool GC = decltype(gc_tag)::value;
        constexpr int B0 = BUF * DBUFS, B1 = (BUF ^ 1) * DBUFS, NRD = 2 + popc(M);
        rd2<M>(f1, lds, offa0[1], offa1[1], offb[1], B0);
        if constexpr (GC) mean_part(BUF, c_this);
        mm<M>(acc, f0);
        sched_mfma_dsread<NRD>();
        rd2<M>(f0, lds, offa0[2], offa1[2], offb[2], B0);
        mm<M>(acc, f1);
        sched_mfma_dsread<NRD>();
        rd2<M>(f1, lds, offa0[3], offa1[3], offb[3], B0);
        mm<M>(acc, f0);
        sched_mfma_dsread<NRD>();
        // the next chunk (DMA issued one chunk ago) has landed for this wave; the barrier makes that true for every wave and
        // says that nobody reads this chunk's buffer any more (all fragment reads above are complete: lgkmcnt(0))
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (MN != 0) {
            if (cf.it < ntile) {  // chunk c + 2 goes into the buffer this chunk just released
                dma(cf, BUF);
                advance(cf);
            }
            rd2<MN>(f0, lds, offa0[0], offa1[0], offb[0], B1);
        }
        mm<M>(acc, f1);
        if constexpr (MN != 0) sched_mfma_dsread<2 + popc(MN)>();
    };

    // prologue: chunks (0, 0) and (0, 1) on their way, first fragments read
    dma(cf, 0);
    advance(cf);
    dma(cf, 1);
    advance(cf);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    rd2<0x01>(f0, lds, offa0[0], offa1[0], offb[0], 0);

    for (int it = 0; it < ntile; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[s][n] = v4d{0, 0, 0, 0};
        const int cd = it * (TILE / KC);
        const bool last_tile = it + 1 == ntile;
        auto diag = [&](auto gc) {
            chunk(IC(0x01), IC(0x03), IC(0), gc, cd);
            chunk(IC(0x03), IC(0x07), IC(1), gc, cd + 1);
            chunk(IC(0x07), IC(0x0F), IC(0), gc, cd + 2);
            chunk(IC(0x0F), IC(0x1F), IC(1), gc, cd + 3);
            chunk(IC(0x1F), IC(0x3F), IC(0), gc, cd + 4);
            chunk(IC(0x3F), IC(0x7F), IC(1), gc, cd + 5);
            chunk(IC(0x7F), IC(0xFF), IC(0), gc, cd + 6);
            if (!last_tile) chunk(IC(0xFF), IC(0xFF), IC(1), gc, cd + 7);
            else chunk(IC(0xFF), IC(0), IC(1), gc, cd + 7);
        };
        auto full = [&](auto gc) {
            const int c_last = nchunk - 1;
            for (int c = cd + 8; c < c_last - 1; c += 2) {
                chunk(IC(0xFF), IC(0xFF), IC(0), gc, c);
                chunk(IC(0xFF), IC(0xFF), IC(1), gc, c + 1);
            }
            chunk(IC(0xFF), IC(0xFF), IC(0), gc, c_last - 1);
            chunk(IC(0xFF), IC(0x01), IC(1), gc, c_last);
        };
        if (it == 0) {
            diag(BC(true));
            if (!last_tile) full(BC(true));
        } else {
            diag(BC(false));
            if (!last_tile) full(BC(false));
        }
        double keep = 0.0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q = 0.0;
#pragma unroll
                for (int n = 0; n < 8; ++n) q += acc[s][n][r] * acc[s][n][r];
                q += __shfl_xor(q, 1);
                q += __shfl_xor(q, 2);
                q += __shfl_xor(q, 4);
                q += __shfl_xor(q, 8);
                keep = ((lane & 7) == s * 4 + r) ? q : keep;
            }
        rs_mine += keep;
    }
    if ((lane & 15) < 8) {
        const int l8 = lane & 15;
        rowq[((l8 >> 2) == 0 ? w : 7 - w) * 16 + (lane >> 4) + 4 * (l8 & 3)] = rs_mine;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mpart[q] += __shfl_xor(mpart[q], 1);
        mpart[q] += __shfl_xor(mpart[q], 2);
        mpart[q] += __shfl_xor(mpart[q], 4);
    }
    __syncthreads();
    if (t < TILE) qout[n0 + t] = rowq[t];
    if ((t & 7) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mout[n0 + q * 32 + (t >> 3)] = mpart[q];
    }
}


// =====================================================================================================================
// Variant "dma2": the dma variant with the instruction stream of a chunk laid out by hand.  Every MFMA is followed by at
// most one other instruction (a fragment read of the NEXT k-step, an LDS-DMA issue, a piece of the mean) and a
// sched_barrier(0) pins that order; the compiler only allocates registers and counts the waits.
//   k-step 0: MFMAs on set X | reads of k-step 1 -> set Y            (+ the mean's LDS reads in column tile 0)
//   k-step 1: MFMAs on set Y | reads of k-step 2 -> set X            (+ the mean's FMAs)
//   k-step 2: MFMAs on set X | reads of k-step 3 -> set Y, all in the first slots
//   s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier     (next chunk landed for everyone, this chunk's buffer free)
//   k-step 3: MFMAs on set Y | 8 LDS-DMA issues for chunk c + 2 into this chunk's buffer, reads of (c + 1, k-step 0) -> set X
// =====================================================================================================================
template <int I, int N, class F>
__device__ __forceinline__ void cfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cfor<I + 1, N>(f);
    }
}
#ifdef LAB_NOSB
#define SB()
#else
#define SB() __builtin_amdgcn_sched_barrier(0)
#endif
struct Frag2 { double v[10]; };  // v[0], v[1]: A fragments of the wave's two row blocks; v[2 + n]: T fragment of column block n

template <int DUMMY = 0>
__global__ __launch_bounds__(NT, LAB_OCC) void moments_dma2(const double* __restrict__ A, const double* __restrict__ Tm,
                                                      const double* __restrict__ gamma, double* __restrict__ qout,
                                                      double* __restrict__ mout, int Mp) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    __shared__ __attribute__((aligned(1024))) double lds[2 * DBUFS];
    __shared__ double rowq[TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntile = Mp / TILE, nchunk = Mp / KC;
    const int64_t n0 = (int64_t)blockIdx.x * TILE;

    const int drow = lane >> 3;
    const int dlog = (lane & 7) ^ drow ^ (w & 1);
    const unsigned dvoff = (unsigned)((drow * Mp + 2 * dlog) * sizeof(double));
    const char* Ab = reinterpret_cast<const char*>(A + (n0 + 8 * w) * Mp);
    const size_t grp = (size_t)32 * Mp * sizeof(double);

    const int lr = lane & 15, lk = lane >> 4;
    const int fr = (lr & 7) ^ (lr >> 3);
    int offa0[4], offa1[4], offb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int o = lr * 16 + (((2 * ks + (lk >> 1)) ^ fr) << 1) + (lk & 1);
        offb[ks] = o + DOPS;
        offa0[ks] = o + w * 256;
        offa1[ks] = o + (7 - w) * 256;
    }
    const int glog = (t & 7) ^ ((t >> 3) & 7) ^ (w & 1);

    for (int j = t; j < Mp; j += NT) gsm[j] = gamma[j];

    double mpart[4] = {0, 0, 0, 0};
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * 1024);
    // DMA piece i (0..7) of chunk `cu` into buffer `buf`: i even -> A piece q = i / 2, i odd -> T piece q
    unsigned dma_vo = 0;
    const char* dma_tb = nullptr;
    auto dma_setup = [&](const Cursor cu) __attribute__((always_inline)) {
        dma_vo = dvoff + (unsigned)(cu.c * KC * sizeof(double));
        dma_tb = reinterpret_cast<const char*>(Tm + ((size_t)cu.it * TILE + 8 * w) * Mp);
    };
    auto dma_piece = [&](auto i_tag, const int buf) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, q = I >> 1;
        const unsigned la = lds_base + (unsigned)((buf * DBUFS + q * 512 + (I & 1) * DOPS) * sizeof(double));
        const char* g = ((I & 1) ? dma_tb : Ab) + q * grp;
        const unsigned vo_ = dma_vo;  // (an asm operand alone does not capture a variable in a generic lambda)
        // the "s" constraint does not move a value the compiler keeps in vector registers: say that these are wave-uniform
        const unsigned la_u = __builtin_amdgcn_readfirstlane(la);
        const uint64_t gv = (uint64_t)(uintptr_t)g;
        // (readfirstlane returns int: without the casts to unsigned the low half is SIGN-extended into the high one -- a wild
        // address whenever bit 31 of the low half is set: the memory access fault of gpurun_out/r3c/lab2_1e6.txt)
        const uint64_t g_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gv >> 32)) << 32) |
                             (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gv);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la_u), "v"(vo_), "s"(g_u) : "memory");
    };
    auto advance = [&](Cursor& cu) __attribute__((always_inline)) {  // saturates at the last chunk of the stream (a harmless re-fetch into a dead buffer)
        if (cu.c + 1 == nchunk) {
            if (cu.it + 1 < ntile) {
                ++cu.it;
                cu.c = cu.it * (TILE / KC);
            }
        } else {
            ++cu.c;
        }
    };

    v4d acc[2][8];
    Frag2 fx, fy;
    Cursor cf{0, 0};
    double rs_mine = 0.0;

    // element E of the fragment set of k-step KS in buffer byte-offset-free form (doubles)
    auto rd1 = [&](Frag2& f, auto e_tag, auto ks_tag, const int boff) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value, KS = decltype(ks_tag)::value;
        if constexpr (E == 0) f.v[0] = lds[offa0[KS] + boff];
        else if constexpr (E == 1) f.v[1] = lds[offa1[KS] + boff];
        else f.v[E] = lds[offb[KS] + boff + (E - 2) * 256];
    };
    // MFMA number I (0 .. 2 m - 1) of a k-step on set f: column block I / 2, row block I % 2
    auto mf = [&](const Frag2& f, auto i_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, n = I >> 1, sblk = I & 1;
        acc[sblk][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
    };

    // slot i of a k-step's reads -> element of the set: the T fragments first (they were consumed early in the previous use of
    // the set), the two A fragments -- operands of that step's LAST MFMAs -- last
    auto rds = [&](Frag2& f, auto slot_tag, auto m_tag, auto ks_tag, const int boff) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value, MM = decltype(m_tag)::value;
        if constexpr (S < MM) rd1(f, IC(2 + S), ks_tag, boff);
        else rd1(f, IC(S - MM), ks_tag, boff);
    };
    // keeps the registers of a set occupied up to this point (they are not handed to the reads issued during the step)
    auto keep_set = [&](const Frag2& f, auto m_tag) __attribute__((always_inline)) {
#ifndef LAB_NOKEEP
        cfor<0, 2 + decltype(m_tag)::value>([&](auto e) __attribute__((always_inline)) {
            const double x = f.v[decltype(e)::value];  // (an asm operand alone does not capture)
            asm volatile("" ::"v"(x));
        });
#endif
    };
    v2d gx[4], gg;  // the mean's operands in flight (column tile 0)
    auto chunk = [&](auto m_tag, auto mn_tag, auto buf_tag, auto gc_tag, const int c_this) __attribute__((always_inline)) {
        constexpr int M = decltype(m_tag)::value, MN = decltype(mn_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr bool GC = decltype(gc_tag)::value;
        constexpr int m = popc(M), mn = popc(MN), NM = 2 * m, NR = 2 + m, NRN = 2 + mn;
        constexpr int B0 = BUF * DBUFS, B1 = (BUF ^ 1) * DBUFS;
        // k-step 0 (the DMA addresses of chunk c + 2 are scalar work: in front of the first MFMAs, not behind the barrier)
        if constexpr (MN != 0) {
            dma_setup(cf);
            advance(cf);
            SB();
        }
        constexpr int S0 = NR + (GC ? 5 : 0);
        cfor<0, (NM > S0 ? NM : S0)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fx, i);
            if constexpr (I < NR) rds(fy, i, IC(m), IC(1), B0);
            else if constexpr (GC && I == NR) gg = *reinterpret_cast<const v2d*>(gsm + c_this * KC + 2 * glog);
            else if constexpr (GC && I > NR && I < NR + 5) gx[I - NR - 1] = *reinterpret_cast<const v2d*>(lds + B0 + (I - NR - 1) * 512 + t * 2);
            SB();
        });
        keep_set(fx, IC(m));
        // k-step 1
        constexpr int S1 = NR + (GC ? 4 : 0);
        cfor<0, (NM > S1 ? NM : S1)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fy, i);
            if constexpr (I < NR) rds(fx, i, IC(m), IC(2), B0);
            else if constexpr (GC && I < NR + 4) mpart[I - NR] += gx[I - NR][0] * gg[0] + gx[I - NR][1] * gg[1];
            SB();
        });
        keep_set(fy, IC(m));
        // k-step 2
        cfor<0, (NM > NR ? NM : NR)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fx, i);
            if constexpr (I < NR) rds(fy, i, IC(m), IC(3), B0);
            SB();
        });
        keep_set(fx, IC(m));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        SB();
        // k-step 3
        if constexpr (MN != 0) {
            constexpr int S3 = 8 + NRN;
            cfor<0, (NM > S3 ? NM : S3)>([&](auto i) __attribute__((always_inline)) {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fy, i);
                if constexpr (I < 8) dma_piece(i, BUF);
                else if constexpr (I < S3) rds(fx, IC(I - 8), IC(mn), IC(0), B1);
                SB();
            });
        } else {
            cfor<0, NM>([&](auto i) __attribute__((always_inline)) { mf(fy, i); });
        }
        keep_set(fy, IC(m));
    };

    // prologue
    dma_setup(cf);
    cfor<0, 8>([&](auto i) __attribute__((always_inline)) { dma_piece(i, 0); });
    advance(cf);
    dma_setup(cf);
    cfor<0, 8>([&](auto i) __attribute__((always_inline)) { dma_piece(i, 1); });
    advance(cf);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    cfor<0, 3>([&](auto i) __attribute__((always_inline)) { rd1(fx, i, IC(0), 0); });  // a0, a1, b0 of the first chunk

    for (int it = 0; it < ntile; ++it) {
        const int cd = it * (TILE / KC);
        const bool last_tile = it + 1 == ntile;
        // an accumulator is zeroed in front of the diagonal chunk that first touches its column block
#ifdef LAB_EAGER_ZERO
        cfor<0, 8>([&](auto n) __attribute__((always_inline)) { acc[0][decltype(n)::value] = v4d{0, 0, 0, 0}; acc[1][decltype(n)::value] = v4d{0, 0, 0, 0}; });
#define ZACC(n_)
#else
#define ZACC(n_) { acc[0][n_] = v4d{0, 0, 0, 0}; acc[1][n_] = v4d{0, 0, 0, 0}; }
#endif
        auto diag = [&](auto gc) __attribute__((always_inline)) {
            ZACC(0) chunk(IC(0x01), IC(0x03), IC(0), gc, cd);
            ZACC(1) chunk(IC(0x03), IC(0x07), IC(1), gc, cd + 1);
            ZACC(2) chunk(IC(0x07), IC(0x0F), IC(0), gc, cd + 2);
            ZACC(3) chunk(IC(0x0F), IC(0x1F), IC(1), gc, cd + 3);
            ZACC(4) chunk(IC(0x1F), IC(0x3F), IC(0), gc, cd + 4);
            ZACC(5) chunk(IC(0x3F), IC(0x7F), IC(1), gc, cd + 5);
            ZACC(6) chunk(IC(0x7F), IC(0xFF), IC(0), gc, cd + 6);
            ZACC(7)
            if (!last_tile) chunk(IC(0xFF), IC(0xFF), IC(1), gc, cd + 7);
            else chunk(IC(0xFF), IC(0), IC(1), gc, cd + 7);
        };
        auto full = [&](auto gc) __attribute__((always_inline)) {
            const int c_last = nchunk - 1;
            for (int c = cd + 8; c < c_last - 1; c += 2) {
                chunk(IC(0xFF), IC(0xFF), IC(0), gc, c);
                chunk(IC(0xFF), IC(0xFF), IC(1), gc, c + 1);
            }
            chunk(IC(0xFF), IC(0xFF), IC(0), gc, c_last - 1);
            chunk(IC(0xFF), IC(0x01), IC(1), gc, c_last);
        };
        if (it == 0) {
            diag(BC(true));
            if (!last_tile) full(BC(true));
        } else {
            diag(BC(false));
            if (!last_tile) full(BC(false));
        }
        double keep = 0.0;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q = 0.0;
#pragma unroll
                for (int n = 0; n < 8; ++n) q += acc[s][n][r] * acc[s][n][r];
                q += __shfl_xor(q, 1);
                q += __shfl_xor(q, 2);
                q += __shfl_xor(q, 4);
                q += __shfl_xor(q, 8);
                keep = ((lane & 7) == s * 4 + r) ? q : keep;
            }
        rs_mine += keep;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the saturated re-fetches of the last two chunks
    if ((lane & 15) < 8) {
        const int l8 = lane & 15;
        rowq[((l8 >> 2) == 0 ? w : 7 - w) * 16 + (lane >> 4) + 4 * (l8 & 3)] = rs_mine;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mpart[q] += __shfl_xor(mpart[q], 1);
        mpart[q] += __shfl_xor(mpart[q], 2);
        mpart[q] += __shfl_xor(mpart[q], 4);
    }
    __syncthreads();
    if (t < TILE) qout[n0 + t] = rowq[t];
    if ((t & 7) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) mout[n0 + q * 32 + (t >> 3)] = mpart[q];
    }
}

struct Cursor3 { int pn, it, c; };  // panel, column tile, k-chunk

template <int DUMMY = 0>
__global__ __launch_bounds__(NT, 1) void moments_p1(const double* __restrict__ A, const double* __restrict__ Tm,
                                                      const double* __restrict__ gamma, double* __restrict__ qout,
                                                      double* __restrict__ mout, int Mp, int npanel) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    __shared__ __attribute__((aligned(1024))) double lds[4 * DBUFS];  // ring of four chunk buffers (128 KB): three chunks in flight
    __shared__ double rowq[TILE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntile = Mp / TILE, nchunk = Mp / KC;

    const int drow = lane >> 3;
    const int dlog = (lane & 7) ^ drow ^ (w & 1);
    const unsigned dvoff = (unsigned)((drow * Mp + 2 * dlog) * sizeof(double));
    const char* Abase = reinterpret_cast<const char*>(A + (size_t)8 * w * Mp);
    const size_t panel_bytes = (size_t)TILE * Mp * sizeof(double);
    const size_t grp = (size_t)32 * Mp * sizeof(double);

    const int lr = lane & 15, lk = lane >> 4;
    const int fr = (lr & 7) ^ (lr >> 3);
    int offa0[4], offa1[4], offb[4], offa0_hi[4], offa1_hi[4], offb_hi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int o = lr * 16 + (((2 * ks + (lk >> 1)) ^ fr) << 1) + (lk & 1);
        offb[ks] = o + DOPS;
        offa0[ks] = o + w * 256;
        offa1[ks] = o + (7 - w) * 256;
        offb_hi[ks] = offb[ks] + 2 * DBUFS;
        offa0_hi[ks] = offa0[ks] + 2 * DBUFS;
        offa1_hi[ks] = offa1[ks] + 2 * DBUFS;
    }
    const int glog = (t & 7) ^ ((t >> 3) & 7) ^ (w & 1);

    for (int j = t; j < Mp; j += NT) gsm[j] = gamma[j];

    double mpart[4] = {0, 0, 0, 0};
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * 1024);
    // DMA piece i (0..7) of chunk `cu` into buffer `buf`: i even -> A piece q = i / 2, i odd -> T piece q
    unsigned dma_vo = 0;
    const char* dma_tb = nullptr;
    const char* dma_ab = nullptr;
    auto dma_setup = [&](const Cursor3 cu) __attribute__((always_inline)) {
        dma_vo = dvoff + (unsigned)(cu.c * KC * sizeof(double));
        dma_tb = reinterpret_cast<const char*>(Tm + ((size_t)cu.it * TILE + 8 * w) * Mp);
        dma_ab = Abase + (size_t)cu.pn * panel_bytes;
    };
    auto dma_piece = [&](auto i_tag, const int buf) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, q = I >> 1;
        const unsigned la = lds_base + (unsigned)((buf * DBUFS + q * 512 + (I & 1) * DOPS) * sizeof(double));
        const char* g = ((I & 1) ? dma_tb : dma_ab) + q * grp;
        const unsigned vo_ = dma_vo;  // (an asm operand alone does not capture a variable in a generic lambda)
        // the "s" constraint does not move a value the compiler keeps in vector registers: say that these are wave-uniform
        const unsigned la_u = __builtin_amdgcn_readfirstlane(la);
        const uint64_t gv = (uint64_t)(uintptr_t)g;
        // (readfirstlane returns int: without the casts to unsigned the low half is SIGN-extended into the high one -- a wild
        // address whenever bit 31 of the low half is set: the memory access fault of gpurun_out/r3c/lab2_1e6.txt)
        const uint64_t g_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(gv >> 32)) << 32) |
                             (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gv);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la_u), "v"(vo_), "s"(g_u) : "memory");
    };
    auto advance = [&](Cursor3& cu) __attribute__((always_inline)) {  // saturates at the last chunk of the stream (a harmless re-fetch into a dead buffer)
        if (cu.c + 1 == nchunk) {
            if (cu.it + 1 < ntile) {
                ++cu.it;
                cu.c = cu.it * (TILE / KC);
            } else if (cu.pn + (int)gridDim.x < npanel) {
                cu.pn += (int)gridDim.x;
                cu.it = 0;
                cu.c = 0;
            }
        } else {
            ++cu.c;
        }
    };

    v4d acc[2][8];
    Frag2 fx, fy;
    Cursor3 cf{(int)blockIdx.x, 0, 0};
    double rs_mine = 0.0;

    // element E of the fragment set of k-step KS in buffer byte-offset-free form (doubles)
    auto rd1 = [&](Frag2& f, auto e_tag, auto ks_tag, auto buf_tag) __attribute__((always_inline)) {
        constexpr int E = decltype(e_tag)::value, KS = decltype(ks_tag)::value, RB = decltype(buf_tag)::value;
        constexpr int boff = (RB & 1) * DBUFS;
        if constexpr (RB < 2) {
            if constexpr (E == 0) f.v[0] = lds[offa0[KS] + boff];
            else if constexpr (E == 1) f.v[1] = lds[offa1[KS] + boff];
            else f.v[E] = lds[offb[KS] + boff + (E - 2) * 256];
        } else {
            if constexpr (E == 0) f.v[0] = lds[offa0_hi[KS] + boff];
            else if constexpr (E == 1) f.v[1] = lds[offa1_hi[KS] + boff];
            else f.v[E] = lds[offb_hi[KS] + boff + (E - 2) * 256];
        }
    };
    // MFMA number I (0 .. 2 m - 1) of a k-step on set f: column block I / 2, row block I % 2
    auto mf = [&](const Frag2& f, auto i_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, n = I >> 1, sblk = I & 1;
        acc[sblk][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
    };

    // slot i of a k-step's reads -> element of the set: the T fragments first (they were consumed early in the previous use of
    // the set), the two A fragments -- operands of that step's LAST MFMAs -- last
    auto rds = [&](Frag2& f, auto slot_tag, auto m_tag, auto ks_tag, auto boff) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value, MM = decltype(m_tag)::value;
        if constexpr (S < MM) rd1(f, IC(2 + S), ks_tag, boff);
        else rd1(f, IC(S - MM), ks_tag, boff);
    };
    // keeps the registers of a set occupied up to this point (they are not handed to the reads issued during the step)
    auto keep_set = [&](const Frag2& f, auto m_tag) __attribute__((always_inline)) {
#ifndef LAB_NOKEEP
        cfor<0, 2 + decltype(m_tag)::value>([&](auto e) __attribute__((always_inline)) {
            const double x = f.v[decltype(e)::value];  // (an asm operand alone does not capture)
            asm volatile("" ::"v"(x));
        });
#endif
    };
    v2d gx[4], gg;  // the mean's operands in flight (column tile 0)
    auto chunk = [&](auto m_tag, auto mn_tag, auto buf_tag, auto gc_tag, const int c_this) __attribute__((always_inline)) {
        constexpr int M = decltype(m_tag)::value, MN = decltype(mn_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr bool GC = decltype(gc_tag)::value;
        constexpr int m = popc(M), mn = popc(MN), NM = 2 * m, NR = 2 + m, NRN = 2 + mn;
        constexpr int B0 = BUF * DBUFS;
        constexpr auto RB0 = IC(BUF);
        constexpr auto RB1 = IC((BUF + 1) & 3);
        // k-step 0 (the DMA addresses of chunk c + 2 are scalar work: in front of the first MFMAs, not behind the barrier)
        if constexpr (MN != 0) {
            dma_setup(cf);
            advance(cf);
            SB();
        }
        constexpr int S0 = NR + (GC ? 5 : 0);
        cfor<0, (NM > S0 ? NM : S0)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fx, i);
            if constexpr (I < NR) rds(fy, i, IC(m), IC(1), RB0);
            else if constexpr (GC && I == NR) gg = *reinterpret_cast<const v2d*>(gsm + c_this * KC + 2 * glog);
            else if constexpr (GC && I > NR && I < NR + 5) gx[I - NR - 1] = *reinterpret_cast<const v2d*>(lds + B0 + (I - NR - 1) * 512 + t * 2);
            SB();
        });
        keep_set(fx, IC(m));
        // k-step 1
        constexpr int S1 = NR + (GC ? 4 : 0);
        cfor<0, (NM > S1 ? NM : S1)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fy, i);
            if constexpr (I < NR) rds(fx, i, IC(m), IC(2), RB0);
            else if constexpr (GC && I < NR + 4) mpart[I - NR] += gx[I - NR][0] * gg[0] + gx[I - NR][1] * gg[1];
            SB();
        });
        keep_set(fy, IC(m));
        // k-step 2
        cfor<0, (NM > NR ? NM : NR)>([&](auto i) __attribute__((always_inline)) {
            constexpr int I = decltype(i)::value;
            if constexpr (I < NM) mf(fx, i);
            if constexpr (I < NR) rds(fy, i, IC(m), IC(3), RB0);
            SB();
        });
        keep_set(fx, IC(m));
        // chunk c + 1 has landed for this wave (c + 2, c + 3: up to 16 DMA pieces may still fly)
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        SB();
        // k-step 3
        if constexpr (MN != 0) {
            constexpr int S3 = 8 + NRN;
            cfor<0, (NM > S3 ? NM : S3)>([&](auto i) __attribute__((always_inline)) {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fy, i);
                if constexpr (I < 8) dma_piece(i, BUF);
                else if constexpr (I < S3) rds(fx, IC(I - 8), IC(mn), IC(0), RB1);
                SB();
            });
        } else {
            cfor<0, NM>([&](auto i) __attribute__((always_inline)) { mf(fy, i); });
        }
        keep_set(fy, IC(m));
    };

    // prologue
    cfor<0, 4>([&](auto b) __attribute__((always_inline)) {
        dma_setup(cf);
        cfor<0, 8>([&](auto i) __attribute__((always_inline)) { dma_piece(i, decltype(b)::value); });
        advance(cf);
    });
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    cfor<0, 3>([&](auto i) __attribute__((always_inline)) { rd1(fx, i, IC(0), IC(0)); });  // a0, a1, b0 of the first chunk

    for (int pn = (int)blockIdx.x; pn < npanel; pn += (int)gridDim.x) {
        const int64_t n0 = (int64_t)pn * TILE;
        const bool last_panel = pn + (int)gridDim.x >= npanel;
        rs_mine = 0.0;
        for (int it = 0; it < ntile; ++it) {
            const int cd = it * (TILE / KC);
            const bool last_tile = it + 1 == ntile;
#define ZACC(n_) { acc[0][n_] = v4d{0, 0, 0, 0}; acc[1][n_] = v4d{0, 0, 0, 0}; }
            auto diag = [&](auto gc) __attribute__((always_inline)) {
                ZACC(0) chunk(IC(0x01), IC(0x03), IC(0), gc, cd);
                ZACC(1) chunk(IC(0x03), IC(0x07), IC(1), gc, cd + 1);
                ZACC(2) chunk(IC(0x07), IC(0x0F), IC(2), gc, cd + 2);
                ZACC(3) chunk(IC(0x0F), IC(0x1F), IC(3), gc, cd + 3);
                ZACC(4) chunk(IC(0x1F), IC(0x3F), IC(0), gc, cd + 4);
                ZACC(5) chunk(IC(0x3F), IC(0x7F), IC(1), gc, cd + 5);
                ZACC(6) chunk(IC(0x7F), IC(0xFF), IC(2), gc, cd + 6);
                ZACC(7)
                if (!last_tile) chunk(IC(0xFF), IC(0xFF), IC(3), gc, cd + 7);
                else if (!last_panel) chunk(IC(0xFF), IC(0x01), IC(3), gc, cd + 7);  // the next panel's first chunk follows
                else chunk(IC(0xFF), IC(0), IC(3), gc, cd + 7);
            };
            auto full = [&](auto gc) __attribute__((always_inline)) {
                const int c_last = nchunk - 1;
                for (int c = cd + 8; c < c_last - 3; c += 4) {
                    chunk(IC(0xFF), IC(0xFF), IC(0), gc, c);
                    chunk(IC(0xFF), IC(0xFF), IC(1), gc, c + 1);
                    chunk(IC(0xFF), IC(0xFF), IC(2), gc, c + 2);
                    chunk(IC(0xFF), IC(0xFF), IC(3), gc, c + 3);
                }
                chunk(IC(0xFF), IC(0xFF), IC(0), gc, c_last - 3);
                chunk(IC(0xFF), IC(0xFF), IC(1), gc, c_last - 2);
                chunk(IC(0xFF), IC(0xFF), IC(2), gc, c_last - 1);
                chunk(IC(0xFF), IC(0x01), IC(3), gc, c_last);  // the next column tile's first chunk follows
            };
            if (it == 0) {
                diag(BC(true));
                if (!last_tile) full(BC(true));
            } else {
                diag(BC(false));
                if (!last_tile) full(BC(false));
            }
            double keep = 0.0;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double q = 0.0;
#pragma unroll
                    for (int n = 0; n < 8; ++n) q += acc[s][n][r] * acc[s][n][r];
                    q += __shfl_xor(q, 1);
                    q += __shfl_xor(q, 2);
                    q += __shfl_xor(q, 4);
                    q += __shfl_xor(q, 8);
                    keep = ((lane & 7) == s * 4 + r) ? q : keep;
                }
            rs_mine += keep;
        }
        // end of the row panel: its outputs (the DMA ring keeps running: the next panel's first chunks are on their way)
        if ((lane & 15) < 8) {
            const int l8 = lane & 15;
            rowq[((l8 >> 2) == 0 ? w : 7 - w) * 16 + (lane >> 4) + 4 * (l8 & 3)] = rs_mine;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mpart[q] += __shfl_xor(mpart[q], 1);
            mpart[q] += __shfl_xor(mpart[q], 2);
            mpart[q] += __shfl_xor(mpart[q], 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t < TILE) qout[n0 + t] = rowq[t];
        if ((t & 7) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mout[n0 + q * 32 + (t >> 3)] = mpart[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) mpart[q] = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // rowq is free again
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the saturated re-fetches of the last chunks
}

typedef int (*moments_fn)(const double*, const double*, const double*, const double*, double, int, double, double*, double*, double*,
                          double*, double*, int32_t*, int64_t, int64_t, int, int, int, void*);

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 1000000;
    const char* libpath = argc > 2 ? argv[2] : "t-svgp_amd/csrc/libtsvgp_hip.so";
    const int M = 1024;
    const int64_t Np = (rows + 127) / 128 * 128;
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen %s failed: %s\n", libpath, dlerror()); return 1; }
    moments_fn prod = (moments_fn)dlsym(h, "tsvgp_moments_f64");
    if (!prod) { printf("no tsvgp_moments_f64\n"); return 1; }
    // optional second build of the library (an experiment build of the same sources): timed beside the first
    moments_fn alt = nullptr;
    if (argc > 3) {
        void* h2 = dlopen(argv[3], RTLD_NOW | RTLD_LOCAL);
        if (!h2) { printf("dlopen %s failed: %s\n", argv[3], dlerror()); return 1; }
        alt = (moments_fn)dlsym(h2, "tsvgp_moments_f64");
        if (alt == prod) { printf("the second library resolved to the first\n"); return 1; }
    }

    std::vector<double> hA((size_t)Np * M), hT((size_t)M * M, 0.0), hg(M), hY(Np);
    unsigned long long s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 2.0; };
    for (auto& v : hA) v = rnd() / 32;
    for (int i = 0; i < M; ++i) for (int j = i; j < M; ++j) hT[(size_t)i * M + j] = rnd() / 32;  // upper triangular
    for (auto& v : hg) v = rnd();
    for (auto& v : hY) v = rnd();
    double *A, *T, *g, *Y, *q, *m, *mean, *var, *g0, *g1, *vep;
    int32_t* npp;
    CHECK(hipMalloc(&A, hA.size() * 8)); CHECK(hipMalloc(&T, hT.size() * 8)); CHECK(hipMalloc(&g, M * 8)); CHECK(hipMalloc(&Y, Np * 8));
    CHECK(hipMalloc(&q, Np * 8)); CHECK(hipMalloc(&m, Np * 8)); CHECK(hipMalloc(&mean, Np * 8)); CHECK(hipMalloc(&var, Np * 8));
    CHECK(hipMalloc(&g0, Np * 8)); CHECK(hipMalloc(&g1, Np * 8)); CHECK(hipMalloc(&vep, (Np / 128) * 8)); CHECK(hipMalloc(&npp, (Np / 128) * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(T, hT.data(), hT.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(g, hg.data(), M * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(Y, hY.data(), Np * 8, hipMemcpyHostToDevice));

    auto run_prod = [&]() { int rc = prod(A, T, g, Y, 1e9, 1, 0.1, mean, var, g0, g1, vep, npp, rows, Np, M, 1, 1, nullptr); if (rc) { printf("prod rc %d\n", rc); exit(1); } };
    auto run_alt = [&]() { int rc = alt(A, T, g, Y, 1e9, 1, 0.1, mean, var, g0, g1, vep, npp, rows, Np, M, 1, 1, nullptr); if (rc) { printf("alt rc %d\n", rc); exit(1); } };
    auto run_dma = [&]() { hipLaunchKernelGGL((moments_dma<0>), dim3((unsigned)(Np / 128)), dim3(NT), (size_t)M * 8, 0, A, T, g, q, m, M); };
    auto run_dma2 = [&]() { hipLaunchKernelGGL((moments_dma2<0>), dim3((unsigned)(Np / 128)), dim3(NT), (size_t)M * 8, 0, A, T, g, q, m, M); };
    int p1_grid = getenv("LAB_P1_GRID") ? atoi(getenv("LAB_P1_GRID")) : 256;
    auto run_p1 = [&]() { const int np = (int)(Np / 128); hipLaunchKernelGGL((moments_p1<0>), dim3((unsigned)std::min(np, p1_grid)), dim3(NT), (size_t)M * 8, 0, A, T, g, q, m, M, np); };
    auto run_pipe = [&]() { hipLaunchKernelGGL((moments_pipe<0>), dim3((unsigned)(Np / 128)), dim3(NT), (size_t)M * 8, 0, A, T, g, q, m, M); };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, int reps) { fn(); CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };

    run_prod(); run_pipe(); CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    std::vector<double> hq(Np), hm(Np), hmean(Np), hvar(Np);
    CHECK(hipMemcpy(hq.data(), q, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hm.data(), m, Np * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hmean.data(), mean, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hvar.data(), var, Np * 8, hipMemcpyDeviceToHost));
    double eq = 0, em = 0, sq = 0, sm = 0;
    for (int64_t n = 0; n < rows; ++n) {
        eq = std::max(eq, std::fabs((1e9 - hq[n]) - hvar[n])); sq = std::max(sq, std::fabs(hq[n]));
        em = std::max(em, std::fabs(hm[n] - hmean[n])); sm = std::max(sm, std::fabs(hmean[n]));
    }
    printf("pipe rows %lld: max |q - q_prod| %.3e (max q %.3e; var = 1e9 - q carries 1e-7 of rounding)   max |mean - mean_prod| %.3e (max %.3e)\n",
           (long long)rows, eq, sq, em, sm);
    CHECK(hipMemset(q, 0, Np * 8)); CHECK(hipMemset(m, 0, Np * 8));
    run_dma(); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError());
    CHECK(hipMemcpy(hq.data(), q, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hm.data(), m, Np * 8, hipMemcpyDeviceToHost));
    eq = em = 0;
    for (int64_t n = 0; n < rows; ++n) {
        eq = std::max(eq, std::fabs((1e9 - hq[n]) - hvar[n]));
        em = std::max(em, std::fabs(hm[n] - hmean[n]));
    }
    printf("dma  rows %lld: max |q - q_prod| %.3e   max |mean - mean_prod| %.3e\n", (long long)rows, eq, em);
    CHECK(hipMemset(q, 0, Np * 8)); CHECK(hipMemset(m, 0, Np * 8));
    run_dma2(); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError());
    CHECK(hipMemcpy(hq.data(), q, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hm.data(), m, Np * 8, hipMemcpyDeviceToHost));
    eq = em = 0;
    for (int64_t n = 0; n < rows; ++n) {
        eq = std::max(eq, std::fabs((1e9 - hq[n]) - hvar[n]));
        em = std::max(em, std::fabs(hm[n] - hmean[n]));
    }
    printf("dma2 rows %lld: max |q - q_prod| %.3e   max |mean - mean_prod| %.3e\n", (long long)rows, eq, em);
    {
        int shown = 0;
        long long bad = 0;
        for (int64_t n = 0; n < rows; ++n) {
            const double d = std::fabs((1e9 - hq[n]) - hvar[n]);
            if (d > 1e-6) {
                ++bad;
                if (shown < 24) { printf("   row %lld (panel %lld, row-in-panel %lld): q %.6e  q_prod %.6e\n", (long long)n, (long long)(n / 128), (long long)(n % 128), hq[n], 1e9 - hvar[n]); ++shown; }
            }
        }
        printf("   rows off by more than 1e-6: %lld of %lld\n", bad, (long long)rows);
    }
    CHECK(hipMemset(q, 0, Np * 8)); CHECK(hipMemset(m, 0, Np * 8));
    run_p1(); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError());
    CHECK(hipMemcpy(hq.data(), q, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hm.data(), m, Np * 8, hipMemcpyDeviceToHost));
    eq = em = 0;
    for (int64_t n = 0; n < rows; ++n) {
        eq = std::max(eq, std::fabs((1e9 - hq[n]) - hvar[n]));
        em = std::max(em, std::fabs(hm[n] - hmean[n]));
    }
    printf("p1   rows %lld: max |q - q_prod| %.3e   max |mean - mean_prod| %.3e\n", (long long)rows, eq, em);
    const double flop = (double)rows * M * (M + 1) + 2.0 * rows * M;
    if (alt) {
        std::vector<double> hm2(Np), hv2(Np);
        CHECK(hipMemset(mean, 0, Np * 8)); CHECK(hipMemset(var, 0, Np * 8));
        run_alt(); CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hm2.data(), mean, Np * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hv2.data(), var, Np * 8, hipMemcpyDeviceToHost));
        double dv = 0, dm = 0;
        for (int64_t n = 0; n < rows; ++n) { dv = std::max(dv, std::fabs(hv2[n] - hvar[n])); dm = std::max(dm, std::fabs(hm2[n] - hmean[n])); }
        printf("alt library: max |var - var_prod| %.3e   max |mean - mean_prod| %.3e\n", dv, dm);
        for (int round = 0; round < 6; ++round) {
            const float tp = timeit(run_prod, 5), ta = timeit(run_alt, 5);
            printf("round %d: production %.3f ms (%.3f of 78.6)   alt %.3f ms (%.3f)\n", round, tp, flop / tp / 1e9 / 78.6, ta, flop / ta / 1e9 / 78.6);
        }
        return 0;
    }
    for (int round = 0; round < 4; ++round) {
        const float tp = timeit(run_prod, 5), td = timeit(run_dma, 5), t2 = timeit(run_dma2, 5), t3 = timeit(run_p1, 5);
        printf("round %d: production %.3f ms (%.3f of 78.6)   dma %.3f ms (%.3f)   dma2 %.3f ms (%.3f)   p1 %.3f ms (%.2f TFLOP/s, %.3f)\n", round, tp,
               flop / tp / 1e9 / 78.6, td, flop / td / 1e9 / 78.6, t2, flop / t2 / 1e9 / 78.6, t3, flop / t3 / 1e9, flop / t3 / 1e9 / 78.6);
    }
    return 0;
}
