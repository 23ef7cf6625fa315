// Micro-benchmarks behind the GEMM design numbers quoted in DESIGN.md (gfx950 / MI355X), stand-alone (no torch):
//   mfma     pure v_mfma_f32_32x32x16_bf16 issue rate on random bf16 operands, 1 or 2 waves per SIMD, with the shader clock
//            that the run sustained (s_memtime ticks / wall time): the MFMA ceiling the GEMM fractions are priced against
//   ldsmfma  the same MFMA stream fed by ds_read_b128 fragment reads from a resident LDS tile (no global traffic, no barriers)
//            for several wave-tile shapes: what the LDS read path alone allows
//   dma      global_load_lds (16 B per lane) streaming rate per CU from an L2- / Infinity-Cache- / HBM-resident source
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/bin/ubench && tools/bin/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static void* dev_rand_bf16(size_t n, unsigned seed, float scale = 1.0f) {
    std::vector<uint16_t> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = f2bf(scale * (((s >> 8) & 0xffff) / 32768.0f - 1.0f)); }
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}

// ---- pure MFMA ---------------------------------------------------------------------------------------------------
template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_only(const bf16x8_t* src, float* sink, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 8 + i) % 4096]; b[i] = src[(threadIdx.x * 8 + 4 + i) % 4096]; }
    f32x16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
    if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- MFMA stream with a workgroup barrier every PER MFMAs (no memory at all): what one s_barrier costs the matrix pipe.
// skew > 0: workgroups with blockIdx >= gridDim / 2 (the second workgroup of a CU) start `skew` MFMAs late (out of phase).
template <int PER, int THREADS>
__global__ __launch_bounds__(THREADS) void mfma_barrier(const bf16x8_t* src, float* sink, long long* cyc, int iters, int skew) {
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 8 + i) % 4096]; b[i] = src[(threadIdx.x * 8 + 4 + i) % 4096]; }
    f32x16_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (skew > 0 && blockIdx.x >= gridDim.x / 2)
        for (int it = 0; it < skew; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < PER / 8; ++g)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(512) void barrier_only(long long* cyc, int iters) {
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- MFMA fed from LDS -------------------------------------------------------------------------------------------
// Workgroup = WW x WX waves, wave tile = FW x FX fragments of 32 x 32; LDS holds one K = 64 stage of (TW + TX) rows x 128 B in the
// GEMM kernels' layout (16-byte chunks XOR-swizzled by the row), reread for every K-tile; optional barrier per K-tile.
template <int FW, int FX, int WW, int WX, bool BARRIER>
__global__ __launch_bounds__(WW * WX * 64) void lds_mfma(const uint4* src, float* sink, int kt_total) {
    constexpr int TW = WW * FW * 32, TX = WX * FX * 32, RB = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < (TW + TX) * RB / 16; i += WW * WX * 64) ((uint4*)smem)[i] = src[i % 8192];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;
    const int frow = lane & 31, fhi = lane >> 5, fswz = (frow >> 1) & 7;
    const char* wt = smem + (ww * FW * 32 + frow) * RB;
    const char* xt = smem + TW * RB + (wx * FX * 32 + frow) * RB;
    f32x16_t acc[FW][FX];
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t wf[2][FW], xf[2][FX];
    auto load = [&](int kk, int par) {
        const int off = ((kk * 2 + fhi) ^ fswz) * 16;
#pragma unroll
        for (int i = 0; i < FW; ++i) wf[par][i] = *(const bf16x8_t*)(wt + i * 32 * RB + off);
#pragma unroll
        for (int j = 0; j < FX; ++j) xf[par][j] = *(const bf16x8_t*)(xt + j * 32 * RB + off);
    };
    for (int kt = 0; kt < kt_total; ++kt) {
        if (BARRIER) __builtin_amdgcn_s_barrier();
        load(0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) load(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
                for (int j = 0; j < FX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], xf[kk & 1][j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
}

// The same loop with the barrier moved INSIDE the K-tile: it is passed just before the last k-step's MFMAs, whose fragments are
// already in registers, and the first fragments of the next tile are read behind it - no wave waits on LDS right after a barrier.
template <int FW, int FX, int WW, int WX>
__global__ __launch_bounds__(WW * WX * 64) void lds_mfma_pipe(const uint4* src, float* sink, int kt_total) {
    constexpr int TW = WW * FW * 32, TX = WX * FX * 32, RB = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < (TW + TX) * RB / 16; i += WW * WX * 64) ((uint4*)smem)[i] = src[i % 8192];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;
    const int frow = lane & 31, fhi = lane >> 5, fswz = (frow >> 1) & 7;
    const char* wt = smem + (ww * FW * 32 + frow) * RB;
    const char* xt = smem + TW * RB + (wx * FX * 32 + frow) * RB;
    f32x16_t acc[FW][FX];
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t wf[2][FW], xf[2][FX];
    auto load = [&](int kk, int par) {
        const int off = ((kk * 2 + fhi) ^ fswz) * 16;
#pragma unroll
        for (int i = 0; i < FW; ++i) wf[par][i] = *(const bf16x8_t*)(wt + i * 32 * RB + off);
#pragma unroll
        for (int j = 0; j < FX; ++j) xf[par][j] = *(const bf16x8_t*)(xt + j * 32 * RB + off);
    };
    load(0, 0);
    for (int kt = 0; kt < kt_total; ++kt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            __builtin_amdgcn_sched_barrier(0);
            load((kk + 1) & 3, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
                for (int j = 0; j < FX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], xf[kk & 1][j], acc[i][j], 0, 0, 0);
            // one fragment read of the next k-step behind every MFMA_PER_READ MFMAs of this one (no read bursts after a barrier)
            constexpr int MPR = 1;
#pragma unroll
            for (int q = 0; q < FW + FX; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
}

// ---- LDS-DMA streaming -------------------------------------------------------------------------------------------
// Every wave streams 1-KiB pieces (global_load_lds, 16 B per lane) from `span` bytes of source (workgroup b starts at a different
// offset) into a private LDS slot ring, `DEPTH` pieces in flight per wave.
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_stream(const char* src, size_t span, int pieces, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    char* slot = smem + wave * DEPTH * 1024;
    size_t off = ((size_t)blockIdx.x * 7919 * 65536 + (size_t)wave * 1024) % span;
    const size_t step = (size_t)nw * 1024;
    for (int p = 0; p < pieces; ++p) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + off + lane * 16), LDS_PTR(slot + (p % DEPTH) * 1024), 16, 0, 0);
        off += step; if (off + 1024 > span) off = (size_t)wave * 1024;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((const float*)smem)[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

template <typename F> static double time_us(F fn, int iters, int warm) {
    for (int i = 0; i < warm; ++i) fn();
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    void* src = dev_rand_bf16(1 << 20, 11);
    float* sink; CK(hipMalloc(&sink, 64));
    long long* cyc; CK(hipMalloc(&cyc, cus * 8 * sizeof(long long)));
    printf("CUs: %d\n", cus);
    const char* only = argc > 1 ? argv[1] : "";
    if (!*only || !strcmp(only, "mfma"))
    // ---- pure MFMA: sustained for ~0.4 s per configuration so that the clock settles under power management
    {
        const int iters = 20000;
        auto run = [&](const char* name, int threads, int blocks_per_cu, auto kern, int nacc) {
            const int grid = cus * blocks_per_cu;
            double best = 1e30; long long c = 0;
            for (int rep = 0; rep < 6; ++rep) {
                double us = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, (const bf16x8_t*)src, sink, cyc, iters); }, 4, 1);
                if (us < best) best = us;
            }
            CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double flops = 2.0 * 32 * 32 * 16 * (double)nacc * iters * (threads / 64) * grid;
            printf("mfma  %-44s %8.1f us  %7.1f TF/s  | wave cycles %lld -> %.3f GHz, %.1f cycles per MFMA per SIMD\n", name, best, flops / best / 1e6,
                   c, c / best / 1e3, (double)c / ((double)nacc * iters * (threads / 256.0) * blocks_per_cu));
        };
        run("1 wave/SIMD (256 thr), 8 accumulators", 256, 1, mfma_only<8, 256>, 8);
        run("1 wave/SIMD (256 thr), 16 accumulators", 256, 1, mfma_only<16, 256>, 16);
        run("2 waves/SIMD (512 thr), 8 accumulators", 512, 1, mfma_only<8, 512>, 8);
        run("2 waves/SIMD (2 x 256 thr), 8 accumulators", 256, 2, mfma_only<8, 512>, 8);
    }
    if (!*only || !strcmp(only, "barrier")) {
        long long c = 0;
        for (int threads : {256, 512}) {
            hipLaunchKernelGGL(barrier_only, dim3(cus), dim3(threads), 0, 0, cyc, 10000);
            CK(hipDeviceSynchronize()); CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            printf("barrier-only loop, %d threads: %.1f cycles per s_barrier\n", threads, c / 10000.0);
        }
        auto run = [&](const char* name, auto kern, int per, int threads, int per_cu, int skew) {
            const int grid = cus * per_cu, iters = 4000 * 64 / per;
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) best = std::min(best, time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, (const bf16x8_t*)src, sink, cyc, iters, skew); }, 3, 1));
            CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double flops = 2.0 * 32 * 32 * 16 * (double)per * iters * (threads / 64) * grid;
            const double waves_per_simd = threads / 256.0 * per_cu;
            printf("mfma+barrier %-58s %8.1f us  %7.1f TF/s | %.0f cycles per barrier interval (MFMA time %d)\n", name, best, flops / best / 1e6,
                   (double)c / iters, (int)(per * 32 * waves_per_simd));
        };
        run("4 waves, barrier every 16 MFMAs", mfma_barrier<16, 256>, 16, 256, 1, 0);
        run("4 waves, barrier every 32 MFMAs", mfma_barrier<32, 256>, 32, 256, 1, 0);
        run("4 waves, barrier every 64 MFMAs", mfma_barrier<64, 256>, 64, 256, 1, 0);
        run("8 waves, barrier every 16 MFMAs", mfma_barrier<16, 512>, 16, 512, 1, 0);
        run("8 waves, barrier every 32 MFMAs", mfma_barrier<32, 512>, 32, 512, 1, 0);
        run("2 x 4 waves per CU, barrier every 16 MFMAs, in phase", mfma_barrier<16, 512>, 16, 256, 2, 0);
        run("2 x 4 waves per CU, barrier every 16 MFMAs, skew 1 (8 MFMAs)", mfma_barrier<16, 512>, 16, 256, 2, 1);
        run("2 x 4 waves per CU, barrier every 32 MFMAs, in phase", mfma_barrier<32, 512>, 32, 256, 2, 0);
        run("2 x 4 waves per CU, barrier every 32 MFMAs, skew 2 (16 MFMAs)", mfma_barrier<32, 512>, 32, 256, 2, 2);
        run("2 x 4 waves per CU, barrier every 64 MFMAs, skew 4 (32 MFMAs)", mfma_barrier<64, 512>, 64, 256, 2, 4);
    }
    if (!*only || !strcmp(only, "lds"))
    // ---- LDS-fed MFMA
    {
        const int kt = 4000;
        auto run = [&](const char* name, auto kern, int fw, int fx, int ww, int wx, int per_cu) {
            const size_t lds = (size_t)(ww * fw + wx * fx) * 32 * 128;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = cus * per_cu;
            double best = 1e30;
            for (int rep = 0; rep < 4; ++rep) best = std::min(best, time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(ww * wx * 64), lds, 0, (const uint4*)src, sink, kt); }, 3, 1));
            const double flops = 2.0 * 32 * 32 * 16 * fw * fx * 4.0 * kt * ww * wx * grid;
            printf("ldsmfma %-52s %8.1f us  %7.1f TF/s  (%.2f ds_read_b128 per MFMA)\n", name, best, flops / best / 1e6, (double)(fw + fx) / (fw * fx));
        };
        run("8 waves 2x4, wave 64x64  (tile 128x256)", lds_mfma<2, 2, 2, 4, false>, 2, 2, 2, 4, 1);
        run("8 waves 2x4, wave 64x64, barrier per K-tile", lds_mfma<2, 2, 2, 4, true>, 2, 2, 2, 4, 1);
        run("8 waves 2x4, wave 128x64 (tile 256x256)", lds_mfma<4, 2, 2, 4, false>, 4, 2, 2, 4, 1);
        run("8 waves 2x4, wave 128x64, barrier per K-tile", lds_mfma<4, 2, 2, 4, true>, 4, 2, 2, 4, 1);
        run("4 waves 2x2, wave 128x64 (tile 256x128), 2 WG/CU", lds_mfma<4, 2, 2, 2, false>, 4, 2, 2, 2, 2);
        run("4 waves 2x2, wave 128x64, barrier, 2 WG/CU", lds_mfma<4, 2, 2, 2, true>, 4, 2, 2, 2, 2);
        run("4 waves 2x2, wave 128x128 (tile 256x256)", lds_mfma<4, 4, 2, 2, false>, 4, 4, 2, 2, 1);
        run("4 waves 2x2, wave 128x128, barrier per K-tile", lds_mfma<4, 4, 2, 2, true>, 4, 4, 2, 2, 1);
        run("8 waves 2x4, wave 64x64, barrier inside the tile", lds_mfma_pipe<2, 2, 2, 4>, 2, 2, 2, 4, 1);
        run("8 waves 2x4, wave 128x64, barrier inside the tile", lds_mfma_pipe<4, 2, 2, 4>, 4, 2, 2, 4, 1);
        run("4 waves 2x2, wave 128x128, barrier inside the tile", lds_mfma_pipe<4, 4, 2, 2>, 4, 4, 2, 2, 1);
        run("4 waves 2x2, wave 128x64, barrier inside, 2 WG/CU", lds_mfma_pipe<4, 2, 2, 2>, 4, 2, 2, 2, 2);
        run("8 waves 1x8, wave 128x96, barrier inside the tile", lds_mfma_pipe<4, 3, 1, 8>, 4, 3, 1, 8, 1);
        run("8 waves 2x4, wave 64x192 (tile 128x768), barrier", lds_mfma<2, 6, 2, 4, true>, 2, 6, 2, 4, 1);
        run("8 waves 1x8, wave 128x96 (tile 128x768), barrier", lds_mfma<4, 3, 1, 8, true>, 4, 3, 1, 8, 1);
    }
    if (!*only || !strcmp(only, "dma"))
    // ---- LDS-DMA streaming
    {
        const size_t big = (size_t)3 << 30;
        char* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 1, big));
        auto run = [&](const char* name, auto kern, int depth, size_t span, int threads) {
            const size_t lds = (size_t)(threads / 64) * depth * 1024;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int pieces = 4096;
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) best = std::min(best, time_us([&] { hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), lds, 0, (const char*)buf, span, pieces, sink); }, 3, 1));
            const double bytes = (double)pieces * 1024 * (threads / 64) * cus;
            printf("dma   %-40s span %7.1f MB  %8.1f us  %6.2f TB/s  %5.1f B/ns/CU\n", name, span / 1048576.0, best, bytes / best / 1e6, bytes / best / 1e3 / cus);
        };
        for (size_t span : {(size_t)8 << 20, (size_t)128 << 20, big}) {
            run("8 waves, 2 pieces in flight per wave", dma_stream<2>, 2, span, 512);
            run("8 waves, 4 pieces in flight per wave", dma_stream<4>, 4, span, 512);
            run("8 waves, 8 pieces in flight per wave", dma_stream<8>, 8, span, 512);
            run("4 waves, 8 pieces in flight per wave", dma_stream<8>, 8, span, 256);
            run("4 waves, 16 pieces in flight per wave", dma_stream<16>, 16, span, 256);
        }
    }
    return 0;
}
