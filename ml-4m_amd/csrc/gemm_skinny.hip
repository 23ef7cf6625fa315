// NT GEMM for a handful of rows (gfx950):  out[m][n] = sum_k X[m][k] * W[n][k],  M <= 32  - the Linears of a decoding step.
//
// Autoregressive generation (fourm/models/generate.py autoregressive_generate; upstream generate.py:850-914) runs every Linear of the
// decoder on B <= 16 rows per token.  The tiled kernels give such a launch N / 128 workgroups (6 for a 768-wide projection): a token cost
// 135 dependent launches of ~12 us, 1.6 ms, against 49 us of weight streaming.  Here a workgroup owns 32 output
// features and the whole reduction: its 4 waves split K, each wave issues ALL its operand loads up front (one memory round trip), runs
// K / 64 MFMAs (v_mfma_f32_32x32x16_bf16, W rows on the MFMA row side, the <= 32 token rows on the column side), the partial sums meet in
// LDS and every wave finishes a quarter of the features.  N / 32 workgroups: 72 for qkv, 24 for a 768-wide projection, 938 for a 30 000-word
// head - the weights stream from HBM at the rate the whole chip pulls, and the launch is a few microseconds of latency.
// Epilogues as in gemm.hip, bit for bit: bf16 (+ bias), fp32 residual (+ bias), SwiGLU (two weights; act and optionally g | u).
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"

namespace {
using namespace fmk;

constexpr int SK_MAXKT = 16;          // k-steps of 16 per wave held in flight: K <= 4 * 16 * 16 = 1024 per pass (longer K: several passes)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(NTArgs a) {
    __shared__ float part[(EPI == EPI_SWIGLU ? 2 : 1) * 4 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fhi = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int N = a.N, K = a.K, M = a.M;
    const int n = n0 + fr < N ? n0 + fr : N - 1;          // W row of this lane (clamped: its results are dropped)
    const int m = fr < M ? fr : M - 1;                    // token row of this lane
    const bf16_t* wrow = a.W + (size_t)n * a.ldw + 8 * fhi;
    const bf16_t* w2row = EPI == EPI_SWIGLU ? a.W2 + (size_t)n * a.ldw + 8 * fhi : nullptr;
    const bf16_t* xrow = a.X + (size_t)m * a.ldx + 8 * fhi;
    f32x16_t acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
    // this wave's share of the reduction: k-steps [ks0, ks1) of 16
    const int steps = K / 16, per = (steps + 3) / 4;
    const int ks0 = wave * per, ks1 = min(steps, ks0 + per);
    for (int base = ks0; base < ks1; base += SK_MAXKT) {
        const int cnt = min(SK_MAXKT, ks1 - base);
        bf16x8_t wf[SK_MAXKT], uf[SK_MAXKT], xf[SK_MAXKT];
#pragma unroll
        for (int i = 0; i < SK_MAXKT; ++i)
            if (i < cnt) {
                const int k = (base + i) * 16;
                wf[i] = *(const bf16x8_t*)(wrow + k);
                if constexpr (EPI == EPI_SWIGLU) uf[i] = *(const bf16x8_t*)(w2row + k);
                xf[i] = *(const bf16x8_t*)(xrow + k);
            }
#pragma unroll
        for (int i = 0; i < SK_MAXKT; ++i)
            if (i < cnt) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[i], acc, 0, 0, 0);
                if constexpr (EPI == EPI_SWIGLU) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf[i], xf[i], acc2, 0, 0, 0);
            }
    }
    // partial sums -> LDS; wave w finishes accumulator registers 4w .. 4w + 3 = features n0 + 8w + 4 fhi + {0..3} of token row fr
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        part[(wave * 16 + r) * 64 + lane] = acc[r];
        if constexpr (EPI == EPI_SWIGLU) part[4 * 16 * 64 + (wave * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();
    float v[4], v2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s += part[(w * 16 + 4 * wave + e) * 64 + lane];
            if constexpr (EPI == EPI_SWIGLU) s2 += part[4 * 16 * 64 + (w * 16 + 4 * wave + e) * 64 + lane];
        }
        v[e] = s; v2[e] = s2;
    }
    if (fr >= M) return;
    const int nb = n0 + 8 * wave + 4 * fhi;                // first of this lane's 4 consecutive features
    if (nb >= N) return;
    float b[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (a.bias && nb + e < N) b[e] = bfround(a.bias[nb + e]);
        if (EPI == EPI_SWIGLU && a.bias2 && nb + e < N) b2[e] = bfround(a.bias2[nb + e]);
    }
    if constexpr (EPI == EPI_BF16) {
        bf16_t* o = (bf16_t*)a.out + (size_t)fr * a.ldo + nb;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (nb + e < N) o[e] = f2bf(v[e] + b[e]);
    } else if constexpr (EPI == EPI_RES) {
        float* o = (float*)a.out + (size_t)fr * a.ldo + nb;
        const float* rr = a.res + (size_t)fr * a.ldr + nb;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (nb + e < N) o[e] = rr[e] + bfround(v[e] + b[e]);
    } else if constexpr (EPI == EPI_SWIGLU) {
        bf16_t* ao = (bf16_t*)a.out + (size_t)fr * a.ldo + nb;
        bf16_t* gu = a.out2 ? (bf16_t*)a.out2 + (size_t)fr * a.ldo2 + nb : nullptr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (nb + e >= N) continue;
            const float g = bfround(v[e] + b[e]), u = bfround(v2[e] + b2[e]);
            ao[e] = f2bf(bfround(silu_f(g)) * u);
            if (gu) { gu[e] = f2bf(g); gu[a.Hp + e] = f2bf(u); }
        }
    }
}

}  // namespace

// Returns 1 when it took the launch, 0 when the arguments are outside what it handles, < 0 on a launch error.
int fm_launch_nt_skinny(const fmk::NTArgs& a, int epilogue, hipStream_t s) {
    using namespace fmk;
    static const bool off = [] { const char* e = getenv("FOURM_NT_SKINNY"); return e && atoi(e) == 0; }();
    if (off || a.groups || a.M > 32 || a.M < 1 || a.K % 16 != 0 || a.K < 64) return 0;
    if ((a.ldw % 8) || (a.ldx % 8) || ((((uintptr_t)a.W | (uintptr_t)a.X | (uintptr_t)a.W2)) & 15)) return 0;
    const dim3 grid((a.N + 31) / 32);
    if (epilogue == FM_EPI_BF16) hipLaunchKernelGGL(gemm_skinny_kernel<EPI_BF16>, grid, dim3(256), 0, s, a);
    else if (epilogue == FM_EPI_RESIDUAL && a.res) hipLaunchKernelGGL(gemm_skinny_kernel<EPI_RES>, grid, dim3(256), 0, s, a);
    else if (epilogue == FM_EPI_SWIGLU && a.W2) hipLaunchKernelGGL(gemm_skinny_kernel<EPI_SWIGLU>, grid, dim3(256), 0, s, a);
    else return 0;
    if (hipGetLastError() != hipSuccess) return -2;
    return 1;
}
