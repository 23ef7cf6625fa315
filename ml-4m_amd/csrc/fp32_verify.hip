// fp32 VERIFICATION path (gfx950): every floating-point kernel of the 4M step once more with fp32 activations, fp32 weights and
// no bf16 rounding anywhere.  Not the hot path: these kernels are plain (LDS-tiled FMA GEMM, one workgroup per attention row) and
// exist so that the engine's launch sequence - selection, glue, hand-written backward formulas, masks, segmentation - can be
// checked against the upstream fp32 model at fp32 tolerances (logits 1e-4, gradients 1e-3; tests/test_model_gpu.py
// test_fp32_verification_mode), which a bf16 pipeline cannot show (two bf16 pipelines differ by ~6e-3 from each other).
// Selected per model with FourM.compute_precision = "fp32" (engine.act_dtype); same C-ABI conventions as the bf16 kernels.
//
// Upstream semantics restated here in fp32: fourm/models/fm_utils.py:93-219 (LayerNorm, Mlp, GatedMlp, Attention,
// CrossAttention incl. masked_fill(-finfo(float32).max)), fm.py:573-637 (losses).
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr float NEG_FILL32 = -3.4028234663852886e38f;     // -finfo(float32).max

__device__ __forceinline__ float gelu32(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float sigm32(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------------------------------------
// out[m][n] (+)= sum_k X[m*sxm + k*sxk] * W[n*swn + k*swk]   + epilogue.  64 x 64 tile, 256 threads, 4 x 4 outputs per thread.
// ------------------------------------------------------------------------------------------------------------------------------
template <bool DUAL>
__global__ __launch_bounds__(256) void gemm_f32_kernel(fm_gemm_f32_args a) {
    __shared__ float xs[16][65], ws[16][65], w2s[DUAL ? 16 : 1][65];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* X = (const float*)a.X; const float* W = (const float*)a.W; const float* W2 = (const float*)a.W2;
    float* out = (float*)a.out;
    int M = a.M, N = a.N, K = a.K;
    long long swn = a.swn, swk = a.swk;
    if (a.groups && a.tile_group) {                       // grouped NT: the row tile picks its weight matrix
        const int g = a.tile_group[m0 / a.seg_rows];
        if (g < 0) return;
        W = (const float*)a.groups[g].W; N = a.groups[g].N; K = a.groups[g].K;
        if (a.groups[g].pad_) { swn = 1; swk = a.groups[g].ldw; } else { swn = a.groups[g].ldw; swk = 1; }     // pad_ = 1: W is read as W[k][n]
    }
    if (a.seg_start) {                                    // grouped TN: group blockIdx.z reduces its own row segment
        const int g = blockIdx.z;
        M = a.groups[g].N; out = (float*)a.groups[g].out;
        if (!out || M <= 0) return;
        const long long r0 = a.seg_start[g];
        K = a.seg_count[g];
        X += r0 * a.sxk; W += r0 * swk;
    }
    if (m0 >= M || n0 >= N) return;
    float acc[4][4] = {}, acc2[DUAL ? 4 : 1][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, kk = i & 15;            // consecutive threads walk k: contiguous when sxk == 1
            const int k = k0 + kk;
            const bool kin = k < K;
            xs[kk][r] = (kin && m0 + r < M) ? X[(long long)(m0 + r) * a.sxm + (long long)k * a.sxk] : 0.f;
            ws[kk][r] = (kin && n0 + r < N) ? W[(long long)(n0 + r) * swn + (long long)k * swk] : 0.f;
            if constexpr (DUAL) w2s[kk][r] = (kin && n0 + r < N) ? W2[(long long)(n0 + r) * swn + (long long)k * swk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float xv[4], wv[4], w2v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xv[i] = xs[kk][ty * 4 + i]; wv[i] = ws[kk][tx * 4 + i]; if constexpr (DUAL) w2v[i] = w2s[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = fmaf(xv[i], wv[j], acc[i][j]);
                    if constexpr (DUAL) acc2[i][j] = fmaf(xv[i], w2v[j], acc2[i][j]);
                }
        }
        __syncthreads();
    }
    const float* bias = (const float*)a.bias; const float* bias2 = (const float*)a.bias2; const float* res = (const float*)a.res;
    float* out2 = (float*)a.out2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] + (bias ? bias[n] : 0.f);
            float* o = out + (size_t)m * a.ldo + n;
            switch (a.epilogue) {
                case FM_EPI_GELU:
                    if (out2) out2[(size_t)m * a.ldo2 + n] = v;
                    *o = gelu32(v);
                    break;
                case FM_EPI_TANH: *o = tanhf(v); break;
                case FM_EPI_RESIDUAL: *o = res[(size_t)m * a.ldr + n] + v; break;
                case FM_EPI_SWIGLU: {
                    if constexpr (DUAL) {
                        const float u = acc2[i][j] + (bias2 ? bias2[n] : 0.f);
                        if (out2) { out2[(size_t)m * a.ldo2 + n] = v; out2[(size_t)m * a.ldo2 + a.Hp + n] = u; }
                        *o = v * sigm32(v) * u;
                    }
                    break;
                }
                default:                                   // FM_EPI_BF16 / FM_EPI_F32: plain fp32 result (+ res when given)
                    if (a.epilogue == FM_EPI_F32 && res) v += res[(size_t)m * a.ldr + n];
                    *o = a.accumulate ? *o + v : v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The NT case (both operands reduction-contiguous) on the fp32 matrix cores: v_mfma_f32_32x32x2_f32 is exact fp32 (one fmaf chain
// per output, bitwise) at the fp32 vector rate.  128(feature) x 128(row) tile, 4 waves of 64 x 64, K-step 32 through LDS
// (row stride 33 floats: the 32 lanes of a half-wave hit 32 banks).  Weight-like operand on the MFMA row side, like gemm.hip:
// a lane ends up with 4 consecutive features of one token row -> float4 epilogue.  Used by the verification path and by the
// tokenizer's post-MLP / 1x1 projection, which upstream computes with autocast disabled (vq/models/vit_models.py:494-496).
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(fm_gemm_f32_args a) {
    constexpr int T = 128, KS = 32, LDT = KS + 1;
    __shared__ float Ws[T * LDT], Xs[T * LDT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ww = wave >> 1, wx = wave & 1;
    const int n0 = blockIdx.x * T, m0 = blockIdx.y * T;
    const float* X = (const float*)a.X; const float* W = (const float*)a.W;
    const int M = a.M, N = a.N, K = a.K;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging: thread -> (row = tid / 8 + 32 p, 4 floats at column (tid % 8) * 4)
    const int lr = threadIdx.x >> 3, lc = (threadIdx.x & 7) * 4;
    float4 wreg[4], xreg[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = lr + 32 * p;
            const int n = n0 + r < N ? n0 + r : N - 1, m = m0 + r < M ? m0 + r : M - 1;
            const int k = k0 + lc;
            if (k + 3 < K) {
                wreg[p] = *(const float4*)(W + (long long)n * a.swn + k);
                xreg[p] = *(const float4*)(X + (long long)m * a.sxm + k);
            } else {
                float w4[4], x4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { w4[e] = k + e < K ? W[(long long)n * a.swn + k + e] : 0.f; x4[e] = k + e < K ? X[(long long)m * a.sxm + k + e] : 0.f; }
                wreg[p] = make_float4(w4[0], w4[1], w4[2], w4[3]); xreg[p] = make_float4(x4[0], x4[1], x4[2], x4[3]);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float* wd = Ws + (lr + 32 * p) * LDT + lc; float* xd = Xs + (lr + 32 * p) * LDT + lc;
            wd[0] = wreg[p].x; wd[1] = wreg[p].y; wd[2] = wreg[p].z; wd[3] = wreg[p].w;
            xd[0] = xreg[p].x; xd[1] = xreg[p].y; xd[2] = xreg[p].z; xd[3] = xreg[p].w;
        }
    };
    load(0);
    for (int k0 = 0; k0 < K; k0 += KS) {
        __syncthreads();                       // the previous step's readers are done
        stash();
        __syncthreads();
        if (k0 + KS < K) load(k0 + KS);        // next K-step in flight under the MFMAs
        const float* wb = Ws + (ww * 64 + (lane & 31)) * LDT + (lane >> 5);
        const float* xb = Xs + (wx * 64 + (lane & 31)) * LDT + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KS; kk += 2) {
            const float a0 = wb[kk], a1 = wb[32 * LDT + kk], b0 = xb[kk], b1 = xb[32 * LDT + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    const float* bias = (const float*)a.bias; const float* res = (const float*)a.res;
    float* out = (float*)a.out;
    const int fhi = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + wx * 64 + j * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + ww * 64 + i * 32 + 8 * g + 4 * fhi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= N) continue;
                    float v = acc[i][j][4 * g + e] + (bias ? bias[n + e] : 0.f);
                    float* o = out + (size_t)m * a.ldo + n + e;
                    if (a.epilogue == FM_EPI_TANH) v = tanhf(v);
                    else if (a.epilogue == FM_EPI_GELU) { if (a.out2) ((float*)a.out2)[(size_t)m * a.ldo2 + n + e] = v; v = gelu32(v); }
                    else if (a.epilogue == FM_EPI_RESIDUAL || (a.epilogue == FM_EPI_F32 && res)) v += res[(size_t)m * a.ldr + n + e];
                    *o = a.accumulate ? *o + v : v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// attention: one workgroup (64 threads) per (b, h, query).  Scores in LDS, softmax in fp32, blocked scores REPLACED by NEG_FILL32.
// ------------------------------------------------------------------------------------------------------------------------------
struct Attn32 {
    const float* Q; const float* K; const float* V; float* O; const float* dO; float* dQ; float* dK; float* dV;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, B, H, Nq, Nk, mask_kind, causal, zero_attn;
    float scale;
    const uint8_t* kpad; const int32_t* cs; const int16_t* modq; const int16_t* modk; const uint8_t* dense;
};

__device__ __forceinline__ bool blocked32(const Attn32& a, int b, int q, int k) {
    switch (a.mask_kind) {
        case FM_MASK_KEYPAD: return a.kpad[(size_t)b * a.Nk + k] != 0;
        case FM_MASK_DECODER: {
            bool blk = a.causal ? (k > q) : (a.cs ? k >= a.cs[(size_t)b * a.Nq + q] : false);
            if (a.modq) blk = blk || (a.modq[(size_t)b * a.Nq + q] != a.modk[(size_t)b * a.Nk + k]);
            return blk;
        }
        case FM_MASK_DENSE: return a.dense[((size_t)b * a.Nq + q) * a.Nk + k] != 0;
        default: return false;
    }
}

template <bool BWD>
__global__ __launch_bounds__(64) void attn_f32_kernel(Attn32 a) {
    extern __shared__ float sm[];                          // p[Nk] (+ ds[Nk] in the backward)
    float* p = sm; float* ds = sm + a.Nk;
    __shared__ float qrow[64], dorow[64], orow[64];
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z, t = threadIdx.x;
    const float* Qr = a.Q + ((size_t)b * a.Nq + q) * a.ldq + h * 64;
    qrow[t] = Qr[t];
    if (BWD) dorow[t] = a.dO[((size_t)b * a.Nq + q) * a.lddo + h * 64 + t];
    __syncthreads();
    float mx = -INFINITY;
    for (int k = t; k < a.Nk; k += 64) {
        const float* Kr = a.K + ((size_t)b * a.Nk + k) * a.ldk + h * 64;
        float s = 0.f;
        for (int d = 0; d < 64; ++d) s = fmaf(qrow[d], Kr[d], s);
        s *= a.scale;
        if (blocked32(a, b, q, k)) s = NEG_FILL32;
        p[k] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (a.zero_attn) mx = fmaxf(mx, 0.f);                  // softmax1: the padded zero logit (fm_utils.py:28-30)
    float sum = 0.f;
    for (int k = t; k < a.Nk; k += 64) { const float e = expf(p[k] - mx); p[k] = e; sum += e; }
    sum = wave_sum(sum);
    if (a.zero_attn) sum += expf(-mx);
    __syncthreads();
    const float inv = 1.0f / sum;
    float o = 0.f;                                          // thread t owns output dimension t
    for (int k = 0; k < a.Nk; ++k) o = fmaf(p[k] * inv, a.V[((size_t)b * a.Nk + k) * a.ldv + h * 64 + t], o);
    if (!BWD) { a.O[((size_t)b * a.Nq + q) * a.ldo + h * 64 + t] = o; return; }
    orow[t] = o;
    const float delta = wave_sum(o * dorow[t]);            // sum_d O[d] dO[d]
    __syncthreads();
    for (int k = t; k < a.Nk; k += 64) {
        const float* Vr = a.V + ((size_t)b * a.Nk + k) * a.ldv + h * 64;
        float dp = 0.f;
        for (int d = 0; d < 64; ++d) dp = fmaf(dorow[d], Vr[d], dp);
        const float pk = p[k] * inv;
        p[k] = pk;
        // masked_fill stops the gradient at blocked scores
        ds[k] = blocked32(a, b, q, k) ? 0.f : pk * (dp - delta) * a.scale;
    }
    __syncthreads();
    float dq = 0.f;
    for (int k = 0; k < a.Nk; ++k) {
        const size_t kr = ((size_t)b * a.Nk + k);
        dq = fmaf(ds[k], a.K[kr * a.ldk + h * 64 + t], dq);
        unsafeAtomicAdd(a.dK + kr * a.lddk + h * 64 + t, ds[k] * qrow[t]);
        unsafeAtomicAdd(a.dV + kr * a.lddv + h * 64 + t, p[k] * dorow[t]);
    }
    a.dQ[((size_t)b * a.Nq + q) * a.lddq + h * 64 + t] = dq;
}

// ------------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward with fp32 dy: dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w; dw, db by atomics.
// One workgroup (256 threads) per row.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* dy, int lddy, const int* dy_row_map, const float* x, int ldx, const float* w,
                                                         const float* mean, const float* rstd, const float* dres, float* dx, int lddx,
                                                         float* dx2, int lddx2, float* dw, float* db, int R, int D) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int sr = dy_row_map ? dy_row_map[r] : r;
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float g = sr >= 0 ? dy[(size_t)sr * lddy + c] * w[c] : 0.f;
        const float xh = (x[(size_t)r * ldx + c] - mu) * rs;
        s1 += g; s2 += g * xh;
    }
    const float m1 = block_sum256(s1, red) / (float)D;
    const float m2 = block_sum256(s2, red) / (float)D;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float dyv = sr >= 0 ? dy[(size_t)sr * lddy + c] : 0.f;
        const float xh = (x[(size_t)r * ldx + c] - mu) * rs;
        float v = rs * (dyv * w[c] - m1 - xh * m2);
        if (dres) v += dres[(size_t)r * lddx + c];
        dx[(size_t)r * lddx + c] = v;
        if (dx2) dx2[(size_t)r * lddx2 + c] = v;
        if (dw) unsafeAtomicAdd(dw + c, dyv * xh);
        if (db) unsafeAtomicAdd(db + c, dyv);
    }
}

// per-head LayerNorm of q / k (qk_norm models): one 64-thread workgroup per (row, head)
__global__ __launch_bounds__(64) void headnorm_f32_fwd_kernel(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, float* stats,
                                                              int H, float eps) {
    const int r = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    const float v = x[(size_t)r * ldx + h * 64 + t];
    const float mu = wave_sum(v) * (1.0f / 64.0f);
    const float dv = v - mu;
    const float rs = rsqrtf(wave_sum(dv * dv) * (1.0f / 64.0f) + eps);
    y[(size_t)r * ldy + h * 64 + t] = dv * rs * w[t] + (b ? b[t] : 0.f);
    if (t == 0) { stats[(size_t)blockIdx.x * 2] = mu; stats[(size_t)blockIdx.x * 2 + 1] = rs; }
}
__global__ __launch_bounds__(64) void headnorm_f32_bwd_kernel(const float* dy, int lddy, const float* x, int ldx, const float* w, const float* stats,
                                                              float* dx, int lddx, float* dw, float* db, int H) {
    const int r = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    const float mu = stats[(size_t)blockIdx.x * 2], rs = stats[(size_t)blockIdx.x * 2 + 1];
    const float dyv = dy[(size_t)r * lddy + h * 64 + t];
    const float xh = (x[(size_t)r * ldx + h * 64 + t] - mu) * rs;
    const float g = dyv * w[t];
    const float m1 = wave_sum(g) * (1.0f / 64.0f), m2 = wave_sum(g * xh) * (1.0f / 64.0f);
    dx[(size_t)r * lddx + h * 64 + t] = rs * (g - m1 - xh * m2);
    if (dw) unsafeAtomicAdd(dw + t, dyv * xh);
    if (db) unsafeAtomicAdd(db + t, dyv);
}

// activation backward + column sums
__global__ void swiglu_bwd_f32_kernel(const float* da, int ldda, const float* gu, int ldgu, float* dgu, int lddgu, int R, int H, int Hp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    const float g = gu[(size_t)r * ldgu + c], u = gu[(size_t)r * ldgu + Hp + c], d = da[(size_t)r * ldda + c];
    const float sg = sigm32(g);
    dgu[(size_t)r * lddgu + c] = d * u * (sg * (1.0f + g * (1.0f - sg)));      // d(silu(g) * u) / dg
    dgu[(size_t)r * lddgu + Hp + c] = d * g * sg;                               // ... / du
}
__global__ void gelu_bwd_f32_kernel(const float* dh, int lddh, const float* pre, int ldp, float* dpre, int lddp, int R, int H) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * H) return;
    const int r = (int)(i / H), c = (int)(i % H);
    const float x = pre[(size_t)r * ldp + c];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f)), pdf = 0.3989422804014327f * expf(-0.5f * x * x);
    dpre[(size_t)r * lddp + c] = dh[(size_t)r * lddh + c] * (cdf + x * pdf);
}
__global__ void colsum_f32_kernel(const float* dy, int ldy, float* db, int R, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int r = blockIdx.y; r < R; r += gridDim.y) s += dy[(size_t)r * ldy + n];
    unsafeAtomicAdd(db + n, s);
}

}  // namespace

// ---- C entry points ------------------------------------------------------------------------------------------------------------
extern "C" int fm_gemm_f32(const fm_gemm_f32_args* p, void* stream) {
    FM_CHECK_ARG(p && p->X && (p->W || p->groups) && (p->out || p->seg_start), "fm_gemm_f32: null pointer");
    FM_CHECK_ARG(p->M > 0 || p->seg_start, "fm_gemm_f32: bad shape");
    FM_CHECK_ARG(p->epilogue != FM_EPI_SWIGLU_BWD && p->epilogue != FM_EPI_GELU_BWD, "fm_gemm_f32: activation-backward epilogues are separate kernels here");
    const int maxN = p->groups && p->tile_group ? p->max_N : p->N;
    const int maxM = p->seg_start ? p->max_N : p->M;
    dim3 grid((maxN + 63) / 64, (maxM + 63) / 64, p->seg_start ? p->n_groups : 1);
    // plain NT with reduction-contiguous, 16-byte aligned operands: the fp32 matrix cores
    const bool nt = p->sxk == 1 && p->swk == 1 && !p->groups && !p->seg_start && p->epilogue != FM_EPI_SWIGLU && p->sxm % 4 == 0 &&
                    p->swn % 4 == 0 && ((((uintptr_t)p->X | (uintptr_t)p->W)) & 15) == 0;
    if (nt) {
        hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3((p->N + 127) / 128, (p->M + 127) / 128), dim3(256), 0, (hipStream_t)stream, *p);
        FM_CHECK_LAUNCH("fm_gemm_f32");
        return 0;
    }
    if (p->epilogue == FM_EPI_SWIGLU) {
        FM_CHECK_ARG(p->W2, "fm_gemm_f32: SwiGLU needs W2");
        hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, *p);
    } else hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, *p);
    FM_CHECK_LAUNCH("fm_gemm_f32");
    return 0;
}

static int fill32(Attn32& a, const fm_attn_args* p) {
    FM_CHECK_ARG(p && p->Q && p->K && p->V && p->O && p->head_dim == 64, "fm_attn_f32: bad argument");
    FM_CHECK_ARG(p->kv_batch_rows == 0 || p->kv_batch_rows == p->Nk, "fm_attn_f32: kv_batch_rows (K/V cache layout) is a bf16 forward option");
    a.Q = (const float*)p->Q; a.K = (const float*)p->K; a.V = (const float*)p->V; a.O = (float*)p->O;
    a.dO = (const float*)p->dO; a.dQ = (float*)p->dQ; a.dK = (float*)p->dK; a.dV = (float*)p->dV;
    a.ldq = p->ldq; a.ldk = p->ldk; a.ldv = p->ldv; a.ldo = p->ldo; a.lddo = p->lddo; a.lddq = p->lddq; a.lddk = p->lddk; a.lddv = p->lddv;
    a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.mask_kind = p->mask_kind; a.causal = p->causal; a.scale = p->scale; a.zero_attn = p->zero_attn;
    a.kpad = (const uint8_t*)p->kpad; a.cs = p->cs; a.modq = p->modq; a.modk = p->modk; a.dense = (const uint8_t*)p->dense;
    return 0;
}
extern "C" int fm_attn_f32_fwd(const fm_attn_args* p, void* stream) {
    Attn32 a{};
    if (int rc = fill32(a, p)) return rc;
    hipLaunchKernelGGL(attn_f32_kernel<false>, dim3(a.Nq, a.H, a.B), dim3(64), (size_t)a.Nk * 4, (hipStream_t)stream, a);
    FM_CHECK_LAUNCH("fm_attn_f32_fwd");
    return 0;
}
/* dK and dV are ACCUMULATED (atomics): the caller zeroes them */
extern "C" int fm_attn_f32_bwd(const fm_attn_args* p, void* stream) {
    Attn32 a{};
    if (int rc = fill32(a, p)) return rc;
    FM_CHECK_ARG(a.dO && a.dQ && a.dK && a.dV, "fm_attn_f32_bwd: null pointer");
    hipLaunchKernelGGL(attn_f32_kernel<true>, dim3(a.Nq, a.H, a.B), dim3(64), (size_t)a.Nk * 8, (hipStream_t)stream, a);
    FM_CHECK_LAUNCH("fm_attn_f32_bwd");
    return 0;
}

extern "C" int fm_layernorm_bwd_f32(const void* dy, int lddy, const int32_t* dy_row_map, const void* x, int ldx, const void* w, const void* mean,
                                    const void* rstd, const void* dres, void* dx, int lddx, void* dx2, int lddx2, void* dw, void* db, int R,
                                    int D, void* stream) {
    FM_CHECK_ARG(dy && x && w && mean && rstd && dx && R > 0 && D > 0, "fm_layernorm_bwd_f32: bad argument");
    hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, dy_row_map, (const float*)x, ldx,
                       (const float*)w, (const float*)mean, (const float*)rstd, (const float*)dres, (float*)dx, lddx, (float*)dx2, lddx2,
                       (float*)dw, (float*)db, R, D);
    FM_CHECK_LAUNCH("fm_layernorm_bwd_f32");
    return 0;
}
extern "C" int fm_headnorm_f32_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, void* stats, int R, int H, float eps,
                                   void* stream) {
    FM_CHECK_ARG(x && w && y && stats && R > 0 && H > 0, "fm_headnorm_f32_fwd: bad argument");
    hipLaunchKernelGGL(headnorm_f32_fwd_kernel, dim3(R * H), dim3(64), 0, (hipStream_t)stream, (const float*)x, ldx, (const float*)w, (const float*)b,
                       (float*)y, ldy, (float*)stats, H, eps);
    FM_CHECK_LAUNCH("fm_headnorm_f32_fwd");
    return 0;
}
extern "C" int fm_headnorm_f32_bwd(const void* dy, int lddy, const void* x, int ldx, const void* w, const void* stats, void* dx, int lddx, void* dw,
                                   void* db, int R, int H, void* stream) {
    FM_CHECK_ARG(dy && x && w && stats && dx && R > 0 && H > 0, "fm_headnorm_f32_bwd: bad argument");
    hipLaunchKernelGGL(headnorm_f32_bwd_kernel, dim3(R * H), dim3(64), 0, (hipStream_t)stream, (const float*)dy, lddy, (const float*)x, ldx,
                       (const float*)w, (const float*)stats, (float*)dx, lddx, (float*)dw, (float*)db, H);
    FM_CHECK_LAUNCH("fm_headnorm_f32_bwd");
    return 0;
}
extern "C" int fm_swiglu_bwd_f32(const void* da, int ldda, const void* gu, int ldgu, void* dgu, int lddgu, int R, int H, int Hp, void* stream) {
    FM_CHECK_ARG(da && gu && dgu && R > 0 && H > 0, "fm_swiglu_bwd_f32: bad argument");
    hipLaunchKernelGGL(swiglu_bwd_f32_kernel, dim3((unsigned)(((size_t)R * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)da, ldda,
                       (const float*)gu, ldgu, (float*)dgu, lddgu, R, H, Hp);
    FM_CHECK_LAUNCH("fm_swiglu_bwd_f32");
    return 0;
}
extern "C" int fm_gelu_bwd_f32(const void* dh, int lddh, const void* pre, int ldp, void* dpre, int lddp, int R, int H, void* stream) {
    FM_CHECK_ARG(dh && pre && dpre && R > 0 && H > 0, "fm_gelu_bwd_f32: bad argument");
    hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3((unsigned)(((size_t)R * H + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dh, lddh,
                       (const float*)pre, ldp, (float*)dpre, lddp, R, H);
    FM_CHECK_LAUNCH("fm_gelu_bwd_f32");
    return 0;
}
extern "C" int fm_colsum_f32(const void* dy, int ldy, void* db, int R, int N, void* stream) {
    FM_CHECK_ARG(dy && db && R > 0 && N > 0, "fm_colsum_f32: bad argument");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256, R < 64 ? R : 64), dim3(256), 0, (hipStream_t)stream, (const float*)dy, ldy, (float*)db, R, N);
    FM_CHECK_LAUNCH("fm_colsum_f32");
    return 0;
}
