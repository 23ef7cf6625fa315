// VQ tokenizer front end: conv-style patch gather and the cosine-similarity codebook search.
//
// fourm/vq/models/vit_models.py:402-405,482 (Conv2d(k=s=patch) == GEMM over (c, py, px)-ordered patches)
// fourm/vq/quantizers/quantize_lucid.py:388-407 (l2norm latents and codes, dist = z @ E^T, argmax with
// first-index tie break, quantize = embed[ind]).  Upstream materialises the (B*196, 16384) distance matrix
// (and a one-hot of the same size); here a lane owns one latent row and scans the codebook from LDS with a
// running arg-max: nothing of size rows x codes ever exists.  HBM/LDS-bound fp32 work (exact f32 FMAs in a
// fixed order: code assignment is bit-reproducible).
#include "common.h"
#include "fourm_hip.h"

namespace {

// out[(b*G + g)][c*P*P + py*P + px] = img[b][c][gy*P + py][gx*P + px]     (bf16, pad columns zero)
__global__ __launch_bounds__(256) void vq_patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int ldo, int B, int C,
                                                          int H, int W, int P) {
    const int gw = W / P, gh = H / P, F = C * P * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < B * gh * gw; r += gridDim.x * 4) {
        const int b = r / (gh * gw), g = r % (gh * gw), gy = g / gw, gx = g % gw;
        const float* base = img + (size_t)b * C * H * W;
        for (int f = lane; f < ldo; f += 64) {
            float v = 0.f;
            if (f < F) {
                const int c = f / (P * P), py = (f / P) % P, px = f % P;
                v = base[((size_t)c * H + gy * P + py) * W + gx * P + px];
            }
            out[(size_t)r * ldo + f] = f2bf(v);
        }
    }
}

// rows / max(||row||, 1e-12)   (F.normalize, p = 2)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = x[(size_t)r * ldx + d]; s += v * v; }
        s = wave_sum(s);
        const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
        for (int d = lane; d < D; d += 64) y[(size_t)r * ldy + d] = x[(size_t)r * ldx + d] * inv;
    }
}

constexpr int VQ_D = 32;          // latent dimension handled by the register-resident search
constexpr int VQ_CHUNK = 256;     // codes staged in LDS at a time (32 KB)

// grid (ceil(R/256), splits): thread = latent row, codes [split*K/splits, ...) scanned in ascending order
// code_bias (optional, K floats): the score of code c starts at code_bias[c] instead of 0 - with -|e_c|^2 / 2 the arg-max of
// <z, e_c> - |e_c|^2 / 2 is the nearest code in EUCLIDEAN distance (EuclideanCodebook, quantize_lucid.py:272-278: argmax of -(|z|^2 - 2 z e + |e|^2))
__global__ __launch_bounds__(256) void vq_search_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ En, int K,
                                                        int R, int codes_per_split, float* __restrict__ best_val, int* __restrict__ best_idx,
                                                        int normalize_latents, const float* __restrict__ code_bias) {
    __shared__ __attribute__((aligned(16))) float code[VQ_CHUNK * VQ_D];
    __shared__ float cbias[VQ_CHUNK];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int rc = r < R ? r : R - 1;
    float zr[VQ_D];
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < VQ_D; d += 4) {
        const float4 v = *(const float4*)(z + (size_t)rc * ldz + d);
        zr[d] = v.x; zr[d + 1] = v.y; zr[d + 2] = v.z; zr[d + 3] = v.w;
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (normalize_latents) {
        const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
        for (int d = 0; d < VQ_D; ++d) zr[d] *= inv;
    }
    const int c_begin = blockIdx.y * codes_per_split, c_end = min(K, c_begin + codes_per_split);
    float bv = -INFINITY;
    int bi = c_begin;
    for (int c0 = c_begin; c0 < c_end; c0 += VQ_CHUNK) {
        const int n = min(VQ_CHUNK, c_end - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < n * VQ_D / 4; i += 256) *(float4*)(code + i * 4) = *(const float4*)(En + (size_t)c0 * VQ_D + i * 4);
        if (threadIdx.x < n) cbias[threadIdx.x] = code_bias ? code_bias[c0 + threadIdx.x] : 0.f;
        __syncthreads();
        for (int c = 0; c < n; ++c) {
            const float* e = code + c * VQ_D;          // same address in every lane: LDS broadcast
            float acc = cbias[c];
#pragma unroll
            for (int d = 0; d < VQ_D; ++d) acc = fmaf(zr[d], e[d], acc);
            if (acc > bv) { bv = acc; bi = c0 + c; }   // strict: the first maximum wins (torch.argmax)
        }
    }
    if (r < R) {
        best_val[(size_t)r * gridDim.y + blockIdx.y] = bv;
        best_idx[(size_t)r * gridDim.y + blockIdx.y] = bi;
    }
}

// merge the per-split winners (ascending split order keeps the first-index tie break), emit the
// token and the quantised vector embed[token] in (B, D, h, w) layout
__global__ __launch_bounds__(256) void vq_merge_kernel(const float* __restrict__ best_val, const int* __restrict__ best_idx, int splits,
                                                       const float* __restrict__ embed, long long* __restrict__ tokens,
                                                       float* __restrict__ quant, int R, int D, int tokens_per_image) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float bv = best_val[(size_t)r * splits];
    int bi = best_idx[(size_t)r * splits];
    for (int s = 1; s < splits; ++s) {
        const float v = best_val[(size_t)r * splits + s];
        if (v > bv) { bv = v; bi = best_idx[(size_t)r * splits + s]; }
    }
    tokens[r] = bi;
    if (quant) {
        const int b = r / tokens_per_image, t = r % tokens_per_image;
        for (int d = 0; d < D; ++d) quant[((size_t)b * D + d) * tokens_per_image + t] = embed[(size_t)bi * D + d];
    }
}

// y = x + bf16->f32(t)   helpers for the fp32 tail are covered by the GEMM epilogues; nothing else here

}  // namespace

extern "C" int fm_vq_patchify(const void* img, void* out, int ld_out, int B, int C, int H, int W, int P, void* stream) {
    FM_CHECK_ARG(img && out && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ld_out >= C * P * P, "fm_vq_patchify: bad argument");
    int grid = (B * (H / P) * (W / P) + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)img, (bf16_t*)out, ld_out, B, C, H, W, P);
    FM_CHECK_LAUNCH("fm_vq_patchify");
    return 0;
}

extern "C" int fm_l2norm_rows(const void* x, int ldx, void* y, int ldy, int R, int D, void* stream) {
    FM_CHECK_ARG(x && y && R > 0 && D > 0, "fm_l2norm_rows: bad argument");
    int grid = (R + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, R, D);
    FM_CHECK_LAUNCH("fm_l2norm_rows");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Codebook training statistics and EMA update (CosineSimCodebook.forward, training branch: quantize_lucid.py:409-425)
// ------------------------------------------------------------------------------------------------
// bins[tok[r]] += 1;  sums[tok[r]][:] += l2norm(z[r])     (fp32 atomics; a wave per latent row, lane = feature)
__global__ __launch_bounds__(256) void vq_code_stats_kernel(const float* __restrict__ z, int ldz, const long long* __restrict__ tokens, int R, int D,
                                                            float* __restrict__ bins, float* __restrict__ sums, int normalize) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float v = lane < D ? z[(size_t)r * ldz + lane] : 0.f;
        const float inv = normalize ? 1.0f / fmaxf(sqrtf(wave_sum(v * v)), 1e-12f) : 1.0f;
        const long long k = tokens[r];
        if (lane < D) unsafeAtomicAdd(sums + (size_t)k * D + lane, v * inv);
        if (lane == 0) unsafeAtomicAdd(bins + k, 1.0f);
    }
}

// cluster_size = cluster_size * decay + bins * (1 - decay);  embed = embed * decay + target * (1 - decay) with
// target = l2norm(sums / bins) for codes that received latents, l2norm(embed) for the others.  A wave per code.
__global__ __launch_bounds__(256) void vq_ema_finalize_kernel(const float* __restrict__ bins, const float* __restrict__ sums, float* __restrict__ embed,
                                                              float* __restrict__ cluster, int K, int D, float decay) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float alpha = 1.0f - decay;
    for (int k = blockIdx.x * 4 + wave; k < K; k += gridDim.x * 4) {
        const float b = bins[k];
        const float e = lane < D ? embed[(size_t)k * D + lane] : 0.f;
        float t = e;
        if (b != 0.f) t = lane < D ? sums[(size_t)k * D + lane] / b : 0.f;
        const float inv = 1.0f / fmaxf(sqrtf(wave_sum(t * t)), 1e-12f);
        if (lane < D) embed[(size_t)k * D + lane] = __fmaf_rn(alpha, t * inv, e * decay);
        if (lane == 0) cluster[k] = __fmaf_rn(alpha, b, cluster[k] * decay);
    }
}

extern "C" int fm_vq_assign_bias(const void* z, int ldz, const void* codes, const void* code_bias, const void* embed, int K, int D, int R,
                                 int tokens_per_image, int normalize_latents, void* ws_val, void* ws_idx, int splits, int64_t* tokens,
                                 void* quant, void* stream);
extern "C" int fm_vq_assign(const void* z, int ldz, const void* codes_normalized, const void* embed, int K, int D, int R,
                            int tokens_per_image, int normalize_latents, void* ws_val, void* ws_idx, int splits, int64_t* tokens,
                            void* quant, void* stream) {
    return fm_vq_assign_bias(z, ldz, codes_normalized, nullptr, embed, K, D, R, tokens_per_image, normalize_latents, ws_val, ws_idx, splits, tokens, quant, stream);
}

// bias[k] = -|embed[k]|^2 / 2  (one thread per code)
__global__ void vq_code_bias_kernel(const float* __restrict__ embed, int K, int D, float* __restrict__ bias) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int d = 0; d < D; ++d) { const float v = embed[(size_t)k * D + d]; s = fmaf(v, v, s); }
    bias[k] = -0.5f * s;
}

extern "C" int fm_vq_code_bias(const void* embed, int K, int D, void* bias, void* stream) {
    FM_CHECK_ARG(embed && bias && K > 0 && D > 0, "fm_vq_code_bias: bad argument");
    hipLaunchKernelGGL(vq_code_bias_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)embed, K, D, (float*)bias);
    FM_CHECK_LAUNCH("fm_vq_code_bias");
    return 0;
}

extern "C" int fm_vq_assign_bias(const void* z, int ldz, const void* codes_normalized, const void* code_bias, const void* embed, int K, int D, int R,
                                 int tokens_per_image, int normalize_latents, void* ws_val, void* ws_idx, int splits, int64_t* tokens,
                                 void* quant, void* stream) {
    FM_CHECK_ARG(z && codes_normalized && embed && ws_val && ws_idx && tokens, "fm_vq_assign: null pointer");
    FM_CHECK_ARG(D == VQ_D, "fm_vq_assign: latent_dim=%d unsupported (this build handles %d)", D, VQ_D);
    FM_CHECK_ARG(K > 0 && R > 0 && splits > 0 && splits <= 64 && ldz % 4 == 0 && tokens_per_image > 0, "fm_vq_assign: bad shape");
    const int per = ((K + splits - 1) / splits + 3) / 4 * 4;
    dim3 grid((R + 255) / 256, splits);
    hipLaunchKernelGGL(vq_search_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, ldz, (const float*)codes_normalized, K, R, per,
                       (float*)ws_val, (int*)ws_idx, normalize_latents, (const float*)code_bias);
    hipLaunchKernelGGL(vq_merge_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)ws_val, (const int*)ws_idx, splits,
                       (const float*)embed, (long long*)tokens, (float*)quant, R, D, tokens_per_image);
    FM_CHECK_LAUNCH("fm_vq_assign");
    return 0;
}

static int vq_code_stats_impl(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, int normalize, void* stream);
extern "C" int fm_vq_code_stats(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, void* stream) {
    return vq_code_stats_impl(z, ldz, tokens, R, D, K, bins, sums, 1, stream);
}
extern "C" int fm_vq_code_stats_raw(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, void* stream) {
    return vq_code_stats_impl(z, ldz, tokens, R, D, K, bins, sums, 0, stream);
}
static int vq_code_stats_impl(const void* z, int ldz, const int64_t* tokens, int R, int D, int K, void* bins, void* sums, int normalize, void* stream) {
    FM_CHECK_ARG(z && tokens && bins && sums && R > 0 && K > 0 && D > 0 && D <= 64, "fm_vq_code_stats: bad argument (latent_dim <= 64)");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(bins, 0, (size_t)K * 4, s) != hipSuccess || hipMemsetAsync(sums, 0, (size_t)K * D * 4, s) != hipSuccess) {
        fm_set_error("fm_vq_code_stats: memset failed");
        return -2;
    }
    int grid = (R + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_code_stats_kernel, dim3(grid), dim3(256), 0, s, (const float*)z, ldz, (const long long*)tokens, R, D, (float*)bins, (float*)sums, normalize);
    FM_CHECK_LAUNCH("fm_vq_code_stats");
    return 0;
}

// EuclideanCodebook EMA (quantize_lucid.py:286-296): cluster_size and embed_avg move towards the batch counts / sums; then
// embed = embed_avg / smoothed, smoothed[k] = (cluster_size[k] + eps) / (S + K eps) * S with S = sum_k cluster_size[k] (Laplace smoothing).
// Pass 1 (a wave per code): the two EMAs and S (one atomic per wave);  pass 2: the division.
__global__ __launch_bounds__(256) void vq_ema_euclid_pass1(const float* __restrict__ bins, const float* __restrict__ sums, float* __restrict__ embed_avg,
                                                           float* __restrict__ cluster, float* __restrict__ total, int K, int D, float decay) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float alpha = 1.0f - decay;
    float part = 0.f;
    for (int k = blockIdx.x * 4 + wave; k < K; k += gridDim.x * 4) {
        if (lane < D) embed_avg[(size_t)k * D + lane] = __fmaf_rn(alpha, sums[(size_t)k * D + lane], embed_avg[(size_t)k * D + lane] * decay);
        const float c = __fmaf_rn(alpha, bins[k], cluster[k] * decay);
        if (lane == 0) { cluster[k] = c; part += c; }
    }
    if (lane == 0 && part != 0.f) unsafeAtomicAdd(total, part);
}
__global__ __launch_bounds__(256) void vq_ema_euclid_pass2(const float* __restrict__ embed_avg, const float* __restrict__ cluster, const float* __restrict__ total,
                                                           float* __restrict__ embed, int K, int D, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * D) return;
    const int k = i / D;
    const float S = *total;
    const float smoothed = (cluster[k] + eps) / (S + (float)K * eps) * S;
    embed[i] = embed_avg[i] / smoothed;
}

extern "C" int fm_vq_ema_update_euclid(const void* bins, const void* sums, void* embed, void* embed_avg, void* cluster_size, void* total_scratch,
                                       int K, int D, float decay, float eps, void* stream) {
    FM_CHECK_ARG(bins && sums && embed && embed_avg && cluster_size && total_scratch && K > 0 && D > 0 && D <= 64 && decay >= 0.f && decay <= 1.f,
                 "fm_vq_ema_update_euclid: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(total_scratch, 0, 4, s) != hipSuccess) { fm_set_error("fm_vq_ema_update_euclid: memset failed"); return -2; }
    int grid = (K + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_ema_euclid_pass1, dim3(grid), dim3(256), 0, s, (const float*)bins, (const float*)sums, (float*)embed_avg, (float*)cluster_size,
                       (float*)total_scratch, K, D, decay);
    hipLaunchKernelGGL(vq_ema_euclid_pass2, dim3((K * D + 255) / 256), dim3(256), 0, s, (const float*)embed_avg, (const float*)cluster_size,
                       (const float*)total_scratch, (float*)embed, K, D, eps);
    FM_CHECK_LAUNCH("fm_vq_ema_update_euclid");
    return 0;
}

extern "C" int fm_vq_ema_update(const void* bins, const void* sums, void* embed, void* cluster_size, int K, int D, float decay, void* stream) {
    FM_CHECK_ARG(bins && sums && embed && cluster_size && K > 0 && D > 0 && D <= 64 && decay >= 0.f && decay <= 1.f, "fm_vq_ema_update: bad argument");
    int grid = (K + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_ema_finalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)bins, (const float*)sums, (float*)embed,
                       (float*)cluster_size, K, D, decay);
    FM_CHECK_LAUNCH("fm_vq_ema_update");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Tokenizer training path: image assembly behind the decoder's output projection, the quantizer's gradient, the
// tanh "post MLP" backward.  [vq/models/vit_models.py:640-648; vq/quantizers/quantize_lucid.py:533-541]
// ------------------------------------------------------------------------------------------------
namespace {

// img[b][c][gy*P + py][gx*P + px] = rows[(b*G + g)][c*P*P + py*P + px]      ('b (nh nw) (c ph pw) -> b c (nh ph) (nw pw)')
__global__ __launch_bounds__(256) void vq_unpatchify_kernel(const float* __restrict__ rows, int ld, float* __restrict__ img, int B, int C, int H, int W, int P) {
    const int gw = W / P;
    const size_t total = (size_t)B * C * H * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % W), y = (int)((e / W) % H), c = (int)((e / ((size_t)W * H)) % C), b = (int)(e / ((size_t)W * H * C));
        const int g = (y / P) * gw + x / P, f = (c * P + y % P) * P + x % P;
        img[e] = rows[((size_t)b * (H / P) * gw + g) * ld + f];
    }
}

// Gradient that reaches the latents z (R, D) f32 through the quantizer in training mode:
//   quantize = x + (q - x).detach()          -> d z  = d quantize                      (straight through, :533-534)
//   loss += w * mse(q.detach(), x)            -> d z += g_loss * w * 2 (z - q) / (R D)  (:539-541), q = embed[token]
// dq: gradient w.r.t. the quantised rows (R, ldq) f32; also returns the VALUE of the commitment term (atomic sum of squares / (R D)).
__global__ __launch_bounds__(256) void vq_latent_grad_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ embed, const long long* __restrict__ tokens,
                                                             const float* __restrict__ dq, int lddq, const float* __restrict__ g_loss, float weight,
                                                             float* __restrict__ dz, int lddz, float* __restrict__ commit, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float gl = g_loss ? g_loss[0] : 0.f;
    const float coef = gl * weight * 2.0f / ((float)R * (float)D);
    float sq = 0.f;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        if (lane < D) {
            const float diff = z[(size_t)r * ldz + lane] - embed[(size_t)tokens[r] * D + lane];
            sq += diff * diff;
            if (dz) dz[(size_t)r * lddz + lane] = (dq ? dq[(size_t)r * lddq + lane] : 0.f) + coef * diff;
        }
    }
    if (commit) {
        sq = wave_sum(sq);
        if (lane == 0 && sq != 0.f) unsafeAtomicAdd(commit, sq * weight / ((float)R * (float)D));
    }
}

// dx = dy * (1 - t^2), t = tanh(pre) as saved by the forward (f32, row stride ld)
__global__ __launch_bounds__(256) void tanh_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ t, float* __restrict__ dx, int R, int N, int ld) {
    const size_t total = (size_t)R * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = (e / N) * ld + e % N;
        const float tv = t[i];
        dx[i] = dy[i] * (1.0f - tv * tv);
    }
}

// rows[r][:] = table[idx[r]][:]   (f32; the quantised vectors as GEMM operand rows)
__global__ __launch_bounds__(256) void embed_rows_f32_kernel(const float* __restrict__ table, const long long* __restrict__ idx, float* __restrict__ out, int ldo, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4)
        for (int d = lane; d < D; d += 64) out[(size_t)r * ldo + d] = table[(size_t)idx[r] * D + d];
}

}  // namespace

extern "C" int fm_vq_unpatchify(const void* rows, int ld_rows, void* img, int B, int C, int H, int W, int P, void* stream) {
    FM_CHECK_ARG(rows && img && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ld_rows >= C * P * P, "fm_vq_unpatchify: bad argument");
    size_t blocks = ((size_t)B * C * H * W + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(vq_unpatchify_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)rows, ld_rows, (float*)img, B, C, H, W, P);
    FM_CHECK_LAUNCH("fm_vq_unpatchify");
    return 0;
}

extern "C" int fm_vq_latent_grad(const void* z, int ldz, const void* embed, const int64_t* tokens, const void* dquant, int ld_dquant, const void* grad_loss,
                                 float commitment_weight, void* dz, int ld_dz, void* commit_value, int R, int D, void* stream) {
    FM_CHECK_ARG(z && embed && tokens && R > 0 && D > 0 && D <= 64 && (dz || commit_value), "fm_vq_latent_grad: bad argument (D <= 64)");
    int grid = (R + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_latent_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z, ldz, (const float*)embed, (const long long*)tokens,
                       (const float*)dquant, ld_dquant, (const float*)grad_loss, commitment_weight, (float*)dz, ld_dz, (float*)commit_value, R, D);
    FM_CHECK_LAUNCH("fm_vq_latent_grad");
    return 0;
}

extern "C" int fm_tanh_bwd_f32(const void* dy, const void* t, void* dx, int R, int N, int ld, void* stream) {
    FM_CHECK_ARG(dy && t && dx && R > 0 && N > 0 && ld >= N, "fm_tanh_bwd_f32: bad argument");
    size_t blocks = ((size_t)R * N + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(tanh_bwd_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)t, (float*)dx, R, N, ld);
    FM_CHECK_LAUNCH("fm_tanh_bwd_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 product on the bf16 matrix cores: x = hi + lo (two bf16: 16 of fp32's 24 significant bits), x w ~ hi_x hi_w + hi_x lo_w + lo_x hi_w
// (the lo_x lo_w term is below 2^-16 relative).  Both operands are written as three column blocks - X' = [hi | hi | lo], W' = [hi | lo | hi] -
// so that ONE bf16 NT GEMM with reduction length 3 K accumulates the three terms in fp32.  Used for the tokenizer's fp32 tail (the tanh
// post-MLP upstream runs with autocast off, vit_models.py:494-496) at inference: ~1e-5 relative instead of bf16's 4e-3, at ~4 x the rate of
// v_mfma_f32_32x32x2_f32.  apply_tanh: x <- tanh(x) first (the activation between fc1 and fc2, fused into the split of fc2's operand).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int R, int K, int weight_order,
                                                     int apply_tanh) {
    const int vec = K / 4;
    const long long total = (long long)R * vec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / vec;
        const int c = (int)(i % vec) * 4;
        const float4 v = *(const float4*)(x + (size_t)r * ldx + c);
        float f[4] = {v.x, v.y, v.z, v.w};
        float hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (apply_tanh) f[j] = tanhf(f[j]);
            hi[j] = bfround(f[j]);
            lo[j] = f[j] - hi[j];
        }
        const uint2 h = make_uint2(pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])), l = make_uint2(pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]));
        bf16_t* o = out + (size_t)r * ldo + c;
        *(uint2*)o = h;
        *(uint2*)(o + K) = weight_order ? l : h;          // activations [hi | hi | lo], weights [hi | lo | hi]
        *(uint2*)(o + 2 * K) = weight_order ? h : l;
    }
}
}  // namespace

extern "C" int fm_split3_bf16(const void* x, int ldx, void* out, int ldo, int R, int K, int weight_order, int apply_tanh, void* stream) {
    FM_CHECK_ARG(x && out && R > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= 3 * K, "fm_split3_bf16: bad argument (K %% 4 == 0, ldo >= 3 K)");
    size_t blocks = ((size_t)R * (K / 4) + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (bf16_t*)out, ldo, R, K, weight_order, apply_tanh);
    FM_CHECK_LAUNCH("fm_split3_bf16");
    return 0;
}

extern "C" int fm_embed_rows_f32(const void* table, const int64_t* idx, void* out, int ld_out, int R, int D, void* stream) {
    FM_CHECK_ARG(table && idx && out && R > 0 && D > 0 && ld_out >= D, "fm_embed_rows_f32: bad argument");
    int grid = (R + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(embed_rows_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)table, (const long long*)idx, (float*)out, ld_out, R, D);
    FM_CHECK_LAUNCH("fm_embed_rows_f32");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Input variants of the tokenizer (VQ.prepare_input, vq/vqvae.py:269-286): ImageNet-standardised pixels brought back to [-1, 1]
// (undo_std) and semantic-segmentation class maps embedded by a learned table (cls_emb), folded into the patch gather.
// ------------------------------------------------------------------------------------------------
namespace {

// out[(b*G + g)][c*P*P + py*P + px] = scale[c] * v + shift[c],  v = img[b][c][y][x]  or  cls_emb[labels[b][y][x]][c]
__global__ __launch_bounds__(256) void vq_patchify_ex_kernel(const float* __restrict__ img, const long long* __restrict__ labels, const float* __restrict__ cls_emb,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, bf16_t* __restrict__ out, int ldo,
                                                             int B, int C, int H, int W, int P) {
    const int gw = W / P, gh = H / P, F = C * P * P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < B * gh * gw; r += gridDim.x * 4) {
        const int b = r / (gh * gw), g = r % (gh * gw), gy = g / gw, gx = g % gw;
        for (int f = lane; f < ldo; f += 64) {
            float v = 0.f;
            if (f < F) {
                const int c = f / (P * P), py = (f / P) % P, px = f % P;
                const size_t pix = (size_t)(gy * P + py) * W + gx * P + px;
                v = labels ? cls_emb[(size_t)labels[(size_t)b * H * W + pix] * C + c] : img[((size_t)b * C + c) * H * W + pix];
                if (scale) v = scale[c] * v + shift[c];
            }
            out[(size_t)r * ldo + f] = f2bf(v);
        }
    }
}

// d cls_emb[labels[b][y][x]][c] += d patches[(b*G + g)][c*P*P + py*P + px]   (bf16 patch-row gradients, fp32 atomics)
__global__ __launch_bounds__(256) void vq_cls_emb_bwd_kernel(const bf16_t* __restrict__ dp, int ld, const long long* __restrict__ labels, float* __restrict__ d_emb,
                                                             int B, int C, int H, int W, int P) {
    const int gw = W / P;
    const size_t total = (size_t)B * H * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % W), y = (int)((e / W) % H), b = (int)(e / ((size_t)W * H));
        const size_t r = (size_t)b * (H / P) * gw + (y / P) * gw + x / P;
        const long long k = labels[e];
        for (int c = 0; c < C; ++c) unsafeAtomicAdd(d_emb + (size_t)k * C + c, bf2f(dp[r * ld + (c * P + y % P) * P + x % P]));
    }
}

// Training-mode quantizer with norm_latents (quantize_lucid.py:525-527, :533-541): x = l2norm(z) enters the codebook and the commitment term.
//   dx = dquant + g_loss * w * 2 (x - q) / (R D);   dz = (dx - x <x, dx>) / max(|z|, 1e-12)      (backward of F.normalize)
__global__ __launch_bounds__(256) void vq_latent_grad_norm_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ embed, const long long* __restrict__ tokens,
                                                                  const float* __restrict__ dq, int lddq, const float* __restrict__ g_loss, float weight,
                                                                  float* __restrict__ dz, int lddz, float* __restrict__ commit, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float gl = g_loss ? g_loss[0] : 0.f;
    const float coef = gl * weight * 2.0f / ((float)R * (float)D);
    float sq = 0.f;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float zv = lane < D ? z[(size_t)r * ldz + lane] : 0.f;
        const float nrm = fmaxf(sqrtf(wave_sum(zv * zv)), 1e-12f);
        const float x = zv / nrm;
        const float diff = lane < D ? x - embed[(size_t)tokens[r] * D + lane] : 0.f;
        sq += diff * diff;
        if (dz) {
            const float dx = lane < D ? (dq ? dq[(size_t)r * lddq + lane] : 0.f) + coef * diff : 0.f;
            const float dot = wave_sum(x * dx);
            if (lane < D) dz[(size_t)r * lddz + lane] = (dx - x * dot) / nrm;
        }
    }
    if (commit) {
        sq = wave_sum(sq);
        if (lane == 0 && sq != 0.f) unsafeAtomicAdd(commit, sq * weight / ((float)R * (float)D));
    }
}

}  // namespace

extern "C" int fm_vq_patchify_ex(const void* img, const int64_t* labels, const void* cls_emb, const void* scale, const void* shift, void* out, int ld_out,
                                 int B, int C, int H, int W, int P, void* stream) {
    FM_CHECK_ARG(out && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ld_out >= C * P * P, "fm_vq_patchify_ex: bad argument");
    FM_CHECK_ARG((labels && cls_emb) || (img && !labels), "fm_vq_patchify_ex: pass pixels (img) or class ids (labels + cls_emb)");
    FM_CHECK_ARG((scale == nullptr) == (shift == nullptr), "fm_vq_patchify_ex: scale and shift go together");
    int grid = (B * (H / P) * (W / P) + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_patchify_ex_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)img, (const long long*)labels, (const float*)cls_emb,
                       (const float*)scale, (const float*)shift, (bf16_t*)out, ld_out, B, C, H, W, P);
    FM_CHECK_LAUNCH("fm_vq_patchify_ex");
    return 0;
}

extern "C" int fm_vq_cls_emb_bwd(const void* d_patches, int ld, const int64_t* labels, void* d_cls_emb, int B, int C, int H, int W, int P, void* stream) {
    FM_CHECK_ARG(d_patches && labels && d_cls_emb && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0 && ld >= C * P * P, "fm_vq_cls_emb_bwd: bad argument");
    size_t blocks = ((size_t)B * H * W + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(vq_cls_emb_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_patches, ld, (const long long*)labels,
                       (float*)d_cls_emb, B, C, H, W, P);
    FM_CHECK_LAUNCH("fm_vq_cls_emb_bwd");
    return 0;
}

extern "C" int fm_vq_latent_grad_normalized(const void* z, int ldz, const void* embed, const int64_t* tokens, const void* dquant, int ld_dquant, const void* grad_loss,
                                            float commitment_weight, void* dz, int ld_dz, void* commit_value, int R, int D, void* stream) {
    FM_CHECK_ARG(z && embed && tokens && R > 0 && D > 0 && D <= 64 && (dz || commit_value), "fm_vq_latent_grad_normalized: bad argument (D <= 64)");
    int grid = (R + 3) / 4;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(vq_latent_grad_norm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)z, ldz, (const float*)embed, (const long long*)tokens,
                       (const float*)dquant, ld_dquant, (const float*)grad_loss, commitment_weight, (float*)dz, ld_dz, (float*)commit_value, R, D);
    FM_CHECK_LAUNCH("fm_vq_latent_grad_normalized");
    return 0;
}
