// Experiment: how fast can one CU stage global memory into LDS, by where the bytes are served from?
//   mode 0: global_load_lds_dwordx4 (LDS DMA)      mode 1: global_load_dwordx4 + ds_write_b128      mode 2: register loads only
// Every workgroup sweeps its own region of `region` bytes over and over: 8/16 KB stays in the CU's L1, 64 KB in the XCD's L2
// (256 workgroups x 64 KB = 2 MB per XCD), 4 MB and up come from the Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/lds_dma_rate tools/exp/lds_dma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* src, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void sweep(const char* __restrict__ base, unsigned region, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* mine = base + (size_t)blockIdx.x * region;
    const unsigned mask = region - 1;
    unsigned off = wave * 1024u + lane * 16u;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(wave * 8u * 1024u);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                dma16(mine + (off & mask), lds_wave + u * 1024u);
                off += NW * 1024u;
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = *reinterpret_cast<const float4*>(mine + (off & mask));
                off += NW * 1024u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE == 1) *reinterpret_cast<float4*>(lds + lds_wave + u * 1024u + lane * 16u) = v[u];
                else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) { const float* f = reinterpret_cast<const float*>(lds); acc.x = f[threadIdx.x]; }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int MODE, int NW>
static void run(const char* d, float* sink, unsigned region, int wgs, const char* what) {
    const int iters = 1500;
    const size_t lds = NW * 8 * 1024;
    CK(hipFuncSetAttribute((const void*)sweep<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    sweep<MODE, NW><<<wgs, NW * 64, lds>>>(d, region, 50, sink);
    CK(hipEventRecord(e0));
    sweep<MODE, NW><<<wgs, NW * 64, lds>>>(d, region, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * NW * iters * 8 * 1024;
    const double per_cu = bytes / (ms * 1e-3) / 256.0;
    printf("%-22s waves %d wgs %4d region %8u B: %7.1f GB/s per CU  %6.2f TB/s chip  (%.3f ms)\n", what, NW, wgs, region, per_cu / 1e9,
           bytes / (ms * 1e-3) / 1e12, ms);
}


// Ring structure of a K loop: every step each wave issues N DMA instructions for step+S-1, after a counted wait and a
// workgroup barrier for step (nothing reads LDS).  PAT 0: contiguous 1 KB per instruction.  PAT 1: eight 128-byte segments
// per instruction, one per "pixel row" of pitch `pitch` bytes, walking K (128 B per step) then nine tap shifts over a 40-wide
// image.  PAT 2: PAT 1 with the 16-byte chunks of a segment XOR-permuted between lanes.
template <int NW, int N, int S, int PAT>
__global__ __launch_bounds__(NW * 64) void ring(const char* __restrict__ base, unsigned region, unsigned pitch, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* mine = base + (size_t)blockIdx.x * region;
    const unsigned mask = region - 1;
    const unsigned ksteps = pitch / 128u;  // K-steps per tap
    unsigned issued = 0;
    auto issue = [&](int stage) {
        const unsigned tap = (issued / ksteps) % 9u, kc = issued % ksteps;
        const unsigned shift = (tap / 3u) * 40u + tap % 3u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            unsigned off;
            if (PAT == 0) {
                off = ((issued * N + i) * NW + wave) * 1024u + lane * 16u;
            } else {
                const unsigned row = (i * NW + wave) * 8u + lane / 8u;
                unsigned chunk = lane & 7u;
                if (PAT == 2) chunk ^= (row >> 1) & 7u;
                off = (row + shift) * pitch + kc * 128u + chunk * 16u;
            }
            dma16(mine + (off & mask), (stage * N + i) * NW * 1024u + wave * 1024u);
        }
        ++issued;
    };
    for (int s = 0; s < S - 1; ++s) issue(s);
    for (int kt = 0; kt < steps; ++kt) {
        int ahead = steps - 1 - kt;
        if (ahead > S - 2) ahead = S - 2;
        if (ahead * N == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (ahead * N == N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * N) : "memory");
        if (kt + S - 1 < steps) issue((kt + S - 1) % S);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (reinterpret_cast<const float*>(lds)[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

template <int NW, int N, int S, int PAT>
static void run_ring(const char* d, float* sink, unsigned region, unsigned pitch, int wgs) {
    static_assert(S >= 2 && S <= 4, "ring depth");
    const int steps = 18 * 40;
    const size_t lds = (size_t)S * N * NW * 1024;
    CK(hipFuncSetAttribute((const void*)ring<NW, N, S, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ring<NW, N, S, PAT><<<wgs, NW * 64, lds>>>(d, region, pitch, 36, sink);
    CK(hipEventRecord(e0));
    ring<NW, N, S, PAT><<<wgs, NW * 64, lds>>>(d, region, pitch, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * NW * steps * N * 1024;
    printf("ring pat %d waves %d x %d per step (%3d KB) S %d wgs %4d region %7u pitch %4u: %7.1f GB/s per CU  %6.2f TB/s  %6.0f clk/step\n", PAT, NW, N,
           NW * N, S, wgs, region, pitch, bytes / (ms * 1e-3) / 256.0 / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / steps);
}

// Loader waves run the ring (PAT 2 gather, N instructions per wave and step); consumer waves meet them at the step's
// barrier and in between execute MIX & 1: 16 MFMAs on register operands, MIX & 2: 16 ds_read_b128 of the landed stage.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NL, int NC, int N, int S, int MIX>
__global__ __launch_bounds__((NL + NC) * 64) void ring_mix(const char* __restrict__ base, unsigned region, unsigned pitch, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* mine = base + (size_t)blockIdx.x * region;
    const unsigned mask = region - 1;
    const unsigned ksteps = pitch / 128u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(mine), 0, (int)region, 0x00020000);
    if (wave < NL) {
        if (MIX & 4) __builtin_amdgcn_s_setprio(3);
        unsigned issued = 0;
        auto issue = [&](int stage) {
            const unsigned tap = (issued / ksteps) % 9u, kc = issued % ksteps;
            const unsigned shift = (tap / 3u) * 40u + tap % 3u;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const unsigned row = (i * NL + wave) * 8u + lane / 8u;
                const unsigned chunk = (lane & 7u) ^ ((row >> 1) & 7u);
                const unsigned off = (row + shift) * pitch + kc * 128u + chunk * 16u;
                if (MIX & 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (stage * N + i) * NL * 1024u + wave * 1024u), 16,
                                                             off & mask, 0, 0, 0);
                else
                    dma16(mine + (off & mask), (stage * N + i) * NL * 1024u + wave * 1024u);
            }
            ++issued;
        };
        for (int s = 0; s < S - 1; ++s) issue(s);
        for (int kt = 0; kt < steps; ++kt) {
            int ahead = steps - 1 - kt;
            if (ahead > S - 2) ahead = S - 2;
            if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
            if (kt + S - 1 < steps) issue((kt + S - 1) % S);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        f16x8 af, bf;
        for (int e = 0; e < 8; ++e) { af[e] = (_Float16)(lane * 0.001f); bf[e] = (_Float16)(e * 0.01f); }
        const unsigned cw = wave - NL;
        for (int kt = 0; kt < steps; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const char* st = lds + (kt % S) * (N * NL * 1024);
            if (MIX & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(st + ((cw * 16 + r) * 1024 + lane * 16) % (N * NL * 1024));
                    if (MIX & 1) { if (r & 1) af = v; else bf = v; }
                    else asm volatile("" ::"v"(v));
                }
            }
            if (MIX & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[r & 3], 0, 0, 0);
                    if (MIX & 8) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        float sum = 0.f;
        for (int k = 0; k < 4; ++k) sum += acc[k][0];
        if (sum == 123.456f) sink[1] = sum;
    }
    __syncthreads();
    if (reinterpret_cast<const float*>(lds)[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

template <int NL, int NC, int N, int S, int MIX>
static void run_mix(const char* d, float* sink, unsigned region, unsigned pitch, int wgs) {
    const int steps = 18 * 40;
    const size_t lds = (size_t)S * N * NL * 1024;
    CK(hipFuncSetAttribute((const void*)ring_mix<NL, NC, N, S, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ring_mix<NL, NC, N, S, MIX><<<wgs, (NL + NC) * 64, lds>>>(d, region, pitch, 36, sink);
    CK(hipEventRecord(e0));
    ring_mix<NL, NC, N, S, MIX><<<wgs, (NL + NC) * 64, lds>>>(d, region, pitch, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * NL * steps * N * 1024;
    printf("mix %2d: %d loaders x %d per step (%3d KB) S %d + %d consumers (mfma %d, ds_read %d): %7.1f GB/s per CU  %6.0f clk/step @2.4GHz\n", MIX, NL, N,
           NL * N, S, NC, MIX & 1, (MIX >> 1) & 1, bytes / (ms * 1e-3) / 256.0 / 1e9, ms * 1e-3 * 2.4e9 / steps);
}

// MFMA issue rate with nothing else running: every wave loops over 4 independent 32x32x16 f16 accumulators.
__global__ __launch_bounds__(512) void mfma_only(int iters, float scale, float* sink) {
    const unsigned lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    f16x8 af, bf;
    for (int e = 0; e < 8; ++e) { af[e] = (_Float16)(scale * (float)((lane * 37 + e * 11) % 97 - 48) * 0.01f); bf[e] = (_Float16)(scale * (float)((lane * 13 + e * 7) % 89 - 44) * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[r & 3], 0, 0, 0);
    }
    float sum = 0.f;
    for (int k = 0; k < 4; ++k) sum += acc[k][0];
    if (sum == 123.456f) sink[1] = sum;
}
static void run_mfma(float* sink, int threads, float scale) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    mfma_only<<<256, threads>>>(100, scale, sink);
    CK(hipEventRecord(e0));
    mfma_only<<<256, threads>>>(iters, scale, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 256.0 * (threads / 64) * iters * 16 * 32768.0;
    printf("mfma only: %d waves per CU, operands %s: %7.1f TFLOP/s (%.3f ms) = %.1f clk per MFMA and SIMD @2.4GHz\n", threads / 64, scale == 0.f ? "zero" : "nonzero",
           flops / (ms * 1e-3) / 1e12, ms, ms * 1e-3 * 2.4e9 / (iters * 16.0 * (threads / 256.0)));
}

// As ring_mix, but roles by SIMD: with 16 waves placed cyclically on the 4 SIMDs, waves 0,4,8,12 share one SIMD and are
// the loaders (N instructions each per step); the 12 waves on the other three SIMDs execute MF MFMAs each per step.
template <int N, int S, int MF>
__global__ __launch_bounds__(1024) void ring_simd(const char* __restrict__ base, unsigned region, unsigned pitch, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const char* mine = base + (size_t)blockIdx.x * region;
    const unsigned mask = region - 1;
    const unsigned ksteps = pitch / 128u;
    constexpr int NL = 4;
    if ((wave & 3u) == 0u) {
        const unsigned lw = wave >> 2;
        unsigned issued = 0;
        auto issue = [&](int stage) {
            const unsigned tap = (issued / ksteps) % 9u, kc = issued % ksteps;
            const unsigned shift = (tap / 3u) * 40u + tap % 3u;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const unsigned row = (i * NL + lw) * 8u + lane / 8u;
                const unsigned chunk = (lane & 7u) ^ ((row >> 1) & 7u);
                const unsigned off = (row + shift) * pitch + kc * 128u + chunk * 16u;
                dma16(mine + (off & mask), (stage * N + i) * NL * 1024u + lw * 1024u);
            }
            ++issued;
        };
        for (int s = 0; s < S - 1; ++s) issue(s);
        for (int kt = 0; kt < steps; ++kt) {
            int ahead = steps - 1 - kt;
            if (ahead > S - 2) ahead = S - 2;
            if (ahead == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
            if (kt + S - 1 < steps) issue((kt + S - 1) % S);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k)
            for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
        f16x8 af, bf;
        for (int e = 0; e < 8; ++e) { af[e] = (_Float16)(lane * 0.001f); bf[e] = (_Float16)(e * 0.01f); }
        for (int kt = 0; kt < steps; ++kt) {
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int r = 0; r < MF; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[r & 3], 0, 0, 0);
        }
        float sum = 0.f;
        for (int k = 0; k < 4; ++k) sum += acc[k][0];
        if (sum == 123.456f) sink[1] = sum;
    }
    __syncthreads();
    if (reinterpret_cast<const float*>(lds)[threadIdx.x] == 123.456f) sink[0] = 1.f;
}
template <int N, int S, int MF>
static void run_simd(const char* d, float* sink, unsigned region, unsigned pitch, int wgs) {
    const int steps = 18 * 40;
    const size_t lds = (size_t)S * N * 4 * 1024;
    CK(hipFuncSetAttribute((const void*)ring_simd<N, S, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    ring_simd<N, S, MF><<<wgs, 1024, lds>>>(d, region, pitch, 36, sink);
    CK(hipEventRecord(e0));
    ring_simd<N, S, MF><<<wgs, 1024, lds>>>(d, region, pitch, steps, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)wgs * 4 * steps * N * 1024;
    printf("roles by SIMD: 4 loaders on one SIMD x %d per step (%3d KB) S %d + 12 waves x %d MFMAs: %7.1f GB/s per CU  %6.0f clk/step @2.4GHz\n", N, 4 * N, S, MF,
           bytes / (ms * 1e-3) / 256.0 / 1e9, ms * 1e-3 * 2.4e9 / steps);
}

int main() {
    const size_t total = (size_t)1 << 31;
    char* d; float* sink;
    CK(hipMalloc(&d, total)); CK(hipMemset(d, 1, total)); CK(hipMalloc(&sink, 64));
    const unsigned regions[] = {8u << 10, 16u << 10, 64u << 10, 1u << 20, 4u << 20};
    for (int wgs : {256, 512}) {
        for (unsigned r : regions) {
            if ((size_t)r * wgs > total) continue;
            run<0, 4>(d, sink, r, wgs, "lds-dma");
            run<1, 4>(d, sink, r, wgs, "regs+ds_write");
            run<2, 4>(d, sink, r, wgs, "regs only");
        }
    }
    for (unsigned r : regions) run<0, 8>(d, sink, r, 256, "lds-dma");
    // 256x128 tile, BK 64: 8 waves x 6 instructions = 48 KB per step; 128x128 BK 32: 4 waves x 4 = 16 KB, two workgroups per CU
    run_ring<8, 6, 2, 0>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 3, 0>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 2, 1>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 3, 1>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 2, 2>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 3, 2>(d, sink, 128u << 10, 256, 256);
    run_ring<8, 6, 2, 2>(d, sink, 256u << 10, 512, 256);
    run_ring<8, 6, 2, 2>(d, sink, 512u << 10, 1024, 256);
    run_ring<4, 4, 3, 0>(d, sink, 64u << 10, 256, 512);
    run_ring<4, 4, 3, 2>(d, sink, 64u << 10, 256, 512);
    run_ring<4, 4, 4, 2>(d, sink, 64u << 10, 256, 512);
    run_ring<4, 8, 2, 2>(d, sink, 64u << 10, 256, 512);
    run_ring<4, 8, 3, 2>(d, sink, 64u << 10, 256, 512);
    run_mfma(sink, 256, 0.f); run_mfma(sink, 256, 1.f); run_mfma(sink, 512, 0.f); run_mfma(sink, 512, 1.f); run_mfma(sink, 512, 1.f);
    run_simd<12, 2, 0>(d, sink, 128u << 10, 256, 256); run_simd<12, 2, 8>(d, sink, 128u << 10, 256, 256); run_simd<12, 2, 16>(d, sink, 128u << 10, 256, 256);
    run_simd<12, 3, 0>(d, sink, 128u << 10, 256, 256); run_simd<12, 3, 8>(d, sink, 128u << 10, 256, 256); run_simd<12, 3, 16>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 16>(d, sink, 128u << 10, 256, 256); run_mix<8, 8, 6, 3, 17>(d, sink, 128u << 10, 256, 256); run_mix<8, 8, 6, 2, 17>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 5>(d, sink, 128u << 10, 256, 256); run_mix<8, 8, 6, 3, 9>(d, sink, 128u << 10, 256, 256); run_mix<8, 8, 6, 3, 13>(d, sink, 128u << 10, 256, 256); run_mix<8, 8, 6, 2, 5>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 2, 0>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 2, 1>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 2, 2>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 2, 3>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 0>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 1>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 2>(d, sink, 128u << 10, 256, 256);
    run_mix<8, 8, 6, 3, 3>(d, sink, 128u << 10, 256, 256);
    CK(hipDeviceSynchronize());
    return 0;
}
