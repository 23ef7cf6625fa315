// Diffusion detokenizer (DiVAE decoder) on gfx950: the pieces of the conditional UNet and of the sampling loop that are not GEMMs.
//
// Replaces, for inference, fourm/vq/models/unet/unet.py (ResBlock :163-274, AttentionBlock :277-322 with QKVAttentionLegacy :345-374,
// Upsample / Downsample :103-160, PatchedUNetCondCat :693-744), nn.py:23-25 (GroupNorm32) / :120-140 (timestep_embedding) and the
// element-wise half of fourm/vq/scheduling/scheduling_{ddim,ddpm}.py (step, _threshold_sample).
//
// Layout: feature maps are row-major (B * H * W, C) bf16 ("NHWC": a pixel is a row, like a token of the transformer trunk), so
//   * a 1 x 1 convolution IS a dense NT GEMM on the rows (fm_gemm_nt: skip_connection, qkv, proj_out);
//   * a 3 x 3 convolution is fm_unet_im2col (this file: 9 shifted copies of the rows side by side, zero at the border; the nearest x2
//     up-sampling in front of Upsample.conv, the stride of Downsample.op and the channel concatenation with the skip tensor are folded
//     into the gather) followed by the same GEMM with K = 9 C: the 3 x 3 convolutions are 97 % of the decoder's FLOPs and run on the
//     MFMA kernels of csrc/gemm_nt3.hip at their long-K rate; the gather is one streaming pass (HBM-bound, 16-byte accesses);
//   * GroupNorm(32) + SiLU (+ the per-sample embedding added in front of it) is a statistics pass per (sample, group) and one
//     streaming pass, fp32 arithmetic like GroupNorm32.
// The spatial self-attention blocks have ONE head of 512 channels over 196 / 49 positions (0.3 % of the FLOPs): a plain fp32 kernel.
#include "common.h"
#include "fourm_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------------
// im2col: out[(b, oy, ox)][tap * C + c] = in[b][oy * stride + ky - pad][ox * stride + kx - pad][c]   (zero outside), C = C1 + C2
//   in = [src1 | src2] along channels; src1 is read at (y >> up1, x >> up1) (nearest x2 up-sampling when up1 = 1), src2 at
//   (y * H2 / H, x * W2 / W) (nearest: the skip tensor at the same resolution, or the 14 x 14 conditioning under the 56 x 56 grid).
// ------------------------------------------------------------------------------------------------------------------------------------
struct Im2colArgs {
    const bf16_t* src1; const bf16_t* src2; bf16_t* out;
    int ld1, ld2, ldo;
    int B, H, W;            // logical input grid (after the up-sampling of src1)
    int C1, C2, H2, W2;
    int Ho, Wo, ksize, stride, up1;
    int kpad;               // columns [ksize^2 * C, kpad) are written as zeros (GEMM reduction padding)
};

__global__ __launch_bounds__(256) void im2col_kernel(Im2colArgs a) {
    const int C = a.C1 + a.C2;
    const int vec_per_row = a.kpad / 8;
    const long long total = (long long)a.B * a.Ho * a.Wo * vec_per_row;
    const int pad = a.ksize / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i % vec_per_row);
        const long long row = i / vec_per_row;
        const int ox = (int)(row % a.Wo), oy = (int)((row / a.Wo) % a.Ho), b = (int)(row / ((long long)a.Wo * a.Ho));
        const int k = v * 8;
        uint4 val = make_uint4(0u, 0u, 0u, 0u);
        if (k < a.ksize * a.ksize * C) {
            const int tap = k / C, c = k % C;            // C1, C2 are multiples of 8: a vector never straddles taps or sources
            const int y = oy * a.stride + tap / a.ksize - pad, x = ox * a.stride + tap % a.ksize - pad;
            if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
                if (c < a.C1) {
                    const int sy = y >> a.up1, sx = x >> a.up1, sw = a.W >> a.up1, sh = a.H >> a.up1;
                    val = *(const uint4*)(a.src1 + ((size_t)(b * sh + sy) * sw + sx) * a.ld1 + c);
                } else {
                    const int sy = y * a.H2 / a.H, sx = x * a.W2 / a.W;
                    val = *(const uint4*)(a.src2 + ((size_t)(b * a.H2 + sy) * a.W2 + sx) * a.ld2 + (c - a.C1));
                }
            }
        }
        *(uint4*)(a.out + (size_t)row * a.ldo + k) = val;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// GroupNorm(G) over (B, HW, C) rows, fp32 statistics (nn.py:23-25), optional per-(sample, channel) addend in front (ResBlock: h + emb_out,
// unet.py:270) and SiLU behind (in_layers / out_layers: normalization, SiLU, conv).
// ------------------------------------------------------------------------------------------------------------------------------------
// statistics, deterministic (no atomics: two runs of the decoder must agree bit for bit - any last-bit noise in a GroupNorm is amplified to
// the bf16 rounding level by the layers behind it).  A workgroup takes 32 consecutive rows of one sample and every channel (8-byte coalesced
// loads: thread = (row lane, channel quad), fixed for all its rows), reduces per channel in registers, per group in LDS in a fixed order,
// and writes (sum, sum of squares) of its rows to partial[b][chunk][g]; gn_finalize_kernel adds the chunks in order.  B * HW / 32
// workgroups stream the map once (the first form - one workgroup per (sample, group) gathering 2-byte values at a row stride - took as
// long as the convolution GEMMs on the 56 x 56 maps).  Sums are taken about a per-group SHIFT (the group's first value of the sample):
// E[d^2] - E[d]^2 then does not cancel for maps with a large mean.
constexpr int GN_ROWS = 32;
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ add, int ld_add, int HW, int C, int G,
                                                       float* __restrict__ partial) {
    extern __shared__ float gsm[];                       // [row_lanes][C][2]
    const int chunks = (HW + GN_ROWS - 1) / GN_ROWS;
    const int b = blockIdx.x / chunks, chunk = blockIdx.x % chunks, r0 = chunk * GN_ROWS;
    const int cpg = C / G, vec = C / 4;
    const int row_lanes = 256 / vec;                     // >= 1 (C <= 1024)
    const bf16_t* xb = x + (size_t)b * HW * ldx;
    const float* ab = add ? add + (size_t)b * ld_add : nullptr;
    const int nr = min(GN_ROWS, HW - r0);
    const int rl = threadIdx.x / vec, cq = threadIdx.x % vec;
    if (rl < row_lanes) {
        const int c0 = cq * 4;
        float sh[4], as[4] = {0.f, 0.f, 0.f, 0.f}, aq[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (c0 + j) / cpg;
            sh[j] = bf2f(xb[g * cpg]) + (ab ? ab[g * cpg] : 0.f);              // row 0 of the sample, first channel of the group
            if (ab) av[j] = ab[c0 + j];
        }
        for (int r = rl; r < nr; r += row_lanes) {
            float f[4];
            unpack_bf4(*(const uint2*)(xb + (size_t)(r0 + r) * ldx + c0), f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = f[j] + av[j] - sh[j];
                as[j] += d; aq[j] += d * d;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { gsm[(rl * C + c0 + j) * 2] = as[j]; gsm[(rl * C + c0 + j) * 2 + 1] = aq[j]; }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += 256) {
        float s = 0.f, q = 0.f;
        for (int l = 0; l < row_lanes; ++l)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += gsm[(l * C + c) * 2]; q += gsm[(l * C + c) * 2 + 1]; }
        float* o = partial + (((size_t)b * chunks + chunk) * G + g) * 2;
        o[0] = s; o[1] = q;
    }
}

// stats[b][g] = (sum d, sum d^2) over the chunks, in chunk order
__global__ void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int B, int G, int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * G) return;
    const int b = i / G, g = i % G;
    float s = 0.f, q = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const float* p = partial + (((size_t)b * chunks + c) * G + g) * 2;
        s += p[0]; q += p[1];
    }
    stats[2 * i] = s; stats[2 * i + 1] = q;
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ add, int ld_add, const float* __restrict__ stats,
                                                       const float* __restrict__ w, const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy,
                                                       int B, int HW, int C, int G, int silu, float eps) {
    const int cpg = C / G, vec = C / 4;
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    const long long total = (long long)B * HW * vec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i % vec);
        const long long row = i / vec;
        const int b = (int)(row / HW), c0 = v * 4;
        const uint2 raw = *(const uint2*)(x + (size_t)row * ldx + c0);
        float f[4];
        unpack_bf4(raw, f);
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j, g = c / cpg;
            const float shift = bf2f(x[(size_t)b * HW * ldx + g * cpg]) + (add ? add[(size_t)b * ld_add + g * cpg] : 0.f);
            const float md = stats[2 * (b * G + g)] * inv_n;                             // mean - shift
            const float var = fmaxf(stats[2 * (b * G + g) + 1] * inv_n - md * md, 0.f);
            float t = (f[j] + (add ? add[(size_t)b * ld_add + c] : 0.f) - shift - md) * rsqrtf(var + eps) * w[c] + bias[c];
            if (silu) t = t / (1.0f + __expf(-t));
            o[j] = t;
        }
        *(uint2*)(y + (size_t)row * ldy + c0) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
}

// One launch for the whole GroupNorm when there are enough (sample, group) pairs to fill the chip: workgroup (g, b) owns the HW x cpg values of
// its group - three sweeps over data that stays in the L2 (sum -> mean; squared deviations -> rstd; normalise, affine, SiLU, store).  The
// three-kernel form above costs three ~11 us launches per norm, 76 norms per UNet evaluation; it stays for small batches (B * G < 64), where
// its row chunks give more workgroups.
__global__ __launch_bounds__(256) void gn_fused_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ add, int ld_add, const float* __restrict__ w,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ y, int ldy, int HW, int C, int G, int silu, float eps) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G, qpr = cpg / 4;                 // quads (4 channels = 8 bytes) per row of the group
    const int rows_per_it = 256 / qpr;                    // cpg in {4, 8, 16, 32, ...}: qpr divides 256 (checked by the launcher)
    const int rl = threadIdx.x / qpr, c0 = g * cpg + (threadIdx.x % qpr) * 4;
    const bf16_t* xb = x + (size_t)b * HW * ldx + c0;
    float av[4] = {0.f, 0.f, 0.f, 0.f};
    if (add) { const float4 t = *(const float4*)(add + (size_t)b * ld_add + c0); av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w; }
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float inv_n = 1.0f / ((float)HW * (float)cpg);
    float s = 0.f;
    for (int r = rl; r < HW; r += rows_per_it) {
        float f[4];
        unpack_bf4(*(const uint2*)(xb + (size_t)r * ldx), f);
        s += (f[0] + av[0]) + (f[1] + av[1]) + (f[2] + av[2]) + (f[3] + av[3]);
    }
    const float mean = block_sum(s) * inv_n;
    float q = 0.f;
    for (int r = rl; r < HW; r += rows_per_it) {
        float f[4];
        unpack_bf4(*(const uint2*)(xb + (size_t)r * ldx), f);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = f[j] + av[j] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(block_sum(q) * inv_n + eps);
    const float4 wv = *(const float4*)(w + c0), bv = *(const float4*)(bias + c0);
    const float ws[4] = {wv.x * rstd, wv.y * rstd, wv.z * rstd, wv.w * rstd}, bs[4] = {bv.x, bv.y, bv.z, bv.w};
    bf16_t* yb = y + (size_t)b * HW * ldy + c0;
    for (int r = rl; r < HW; r += rows_per_it) {
        float f[4], o[4];
        unpack_bf4(*(const uint2*)(xb + (size_t)r * ldx), f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (f[j] + av[j] - mean) * ws[j] + bs[j];
            if (silu) t = t / (1.0f + __expf(-t));
            o[j] = t;
        }
        *(uint2*)(yb + (size_t)r * ldy) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
}

// out = a + b (bf16 feature maps: skip_connection(x) + h, x + attention)
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b, int ldb, bf16_t* __restrict__ out, int ldo,
                                                       long long rows, int C) {
    const int vec = C / 4;
    const long long total = rows * vec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / vec;
        const int c0 = (int)(i % vec) * 4;
        float fa[4], fb[4];
        unpack_bf4(*(const uint2*)(a + (size_t)row * lda + c0), fa);
        unpack_bf4(*(const uint2*)(b + (size_t)row * ldb + c0), fb);
        *(uint2*)(out + (size_t)row * ldo + c0) = make_uint2(pack2bf(fa[0] + fb[0], fa[1] + fb[1]), pack2bf(fa[2] + fb[2], fa[3] + fb[3]));
    }
}

// y = silu(x) (bf16 <- f32 or bf16): the activation in front of ResBlock.emb_layers / inside time_embed
__global__ __launch_bounds__(256) void silu_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float t = x[i];
        y[i] = f2bf(t / (1.0f + __expf(-t)));
    }
}

// [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) i / half)   (nn.py:120-140), bf16 rows for the time_embed GEMM
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int ldo, int B, int dim, float max_period) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float f = expf(-logf(max_period) * (float)j / (float)half);
    const float arg = t[b] * f;
    out[(size_t)b * ldo + j] = f2bf(cosf(arg));
    out[(size_t)b * ldo + half + j] = f2bf(sinf(arg));
    if ((dim & 1) && j == 0) out[(size_t)b * ldo + dim - 1] = 0;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// spatial self-attention of AttentionBlock (QKVAttentionLegacy): qkv rows (B * T, H * 3 * ch) with a head's channels as [q | k | v];
// weight = softmax_fp32((q s)(k s)^T), s = ch^-1/4; out rows (B * T, H * ch).  One workgroup per (query, head, sample).
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unet_attn_kernel(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ out, int ldo, int T, int ch, int H) {
    extern __shared__ float sm[];                         // q[ch] | p[T]
    float* qs = sm; float* p = sm + ch;
    __shared__ float red[8];
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const bf16_t* base = qkv + (size_t)b * T * ld + (size_t)h * 3 * ch;
    const float scale = 1.0f / sqrtf((float)ch);          // (q ch^-1/4) . (k ch^-1/4)
    for (int d = threadIdx.x; d < ch; d += 256) qs[d] = bf2f(base[(size_t)q * ld + d]) * scale;
    __syncthreads();
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < T; k += 256) {
        const bf16_t* kr = base + (size_t)k * ld + ch;
        float s = 0.f;
        for (int d = 0; d < ch; d += 8) {
            const uint4 raw = *(const uint4*)(kr + d);
            float f0[4], f1[4];
            unpack_bf4(make_uint2(raw.x, raw.y), f0);
            unpack_bf4(make_uint2(raw.z, raw.w), f1);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(qs[d + j], f0[j], fmaf(qs[d + 4 + j], f1[j], s));
        }
        p[k] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int k = threadIdx.x; k < T; k += 256) { const float e = __expf(p[k] - mx); p[k] = e; sum += e; }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int d = threadIdx.x; d < ch; d += 256) {
        const bf16_t* vr = base + 2 * ch + d;
        float o = 0.f;
        for (int k = 0; k < T; ++k) o = fmaf(p[k], bf2f(vr[(size_t)k * ld]), o);
        out[((size_t)b * T + q) * ldo + (size_t)h * ch + d] = f2bf(o * inv);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Sampling loop, element-wise half (scheduling_ddim.py:226-330, scheduling_ddpm.py:275-345):
//   x0   = c0 * sample + c1 * model_output                       (v / epsilon / sample prediction: the host picks c0, c1)
//   x0  <- clamp(x0, -s_b, s_b) / s_b  (dynamic thresholding, s_b = clamp(quantile_0.995 |x0_b|, 1, max))  or  clamp(x0, -r, r)
//   out  = k0 * x0 + k1 * sample + k2 * model_output + k3 * noise
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diffusion_x0_kernel(const float* __restrict__ sample, const float* __restrict__ mo, float c0, float c1, float* __restrict__ x0,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x0[i] = c0 * sample[i] + c1 * mo[i];
}

// q-quantile of |x| per row with torch.quantile's linear interpolation: v[lo] + (v[lo + 1] - v[lo]) * frac, pos = q (n - 1).  Non-negative
// floats order like their bit patterns: four 8-bit radix passes find the (lo + 1)-th smallest, a fifth pass the smallest value above it
// (or itself when it repeats).  One workgroup per row.
__global__ __launch_bounds__(1024) void quantile_abs_kernel(const float* __restrict__ x, long long n, float q, float* __restrict__ out) {
    const float* xr = x + (size_t)blockIdx.x * n;
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_cnt_le;
    __shared__ float s_next;
    const float pos = q * (float)(n - 1);                 // fp32, like torch.quantile's rank tensor (q takes the input's dtype)
    const long long lo = (long long)floorf(pos);
    const float frac = pos - (float)lo;
    unsigned prefix = 0, rank = (unsigned)lo;             // 0-based rank inside the values that share the prefix found so far
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const unsigned hi_mask = pass == 3 ? 0u : (0xffffffffu << (8 * (pass + 1)));
        for (long long i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned u = __float_as_uint(fabsf(xr[i]));
            if ((u & hi_mask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned acc = 0, d = 0;
            for (; d < 256; ++d) {
                if (acc + hist[d] > rank) break;
                acc += hist[d];
            }
            s_prefix = prefix | (d << (8 * pass));
            s_rank = rank - acc;
        }
        __syncthreads();
        prefix = s_prefix; rank = s_rank;
        __syncthreads();
    }
    const float vlo = __uint_as_float(prefix);
    // v[lo + 1]: vlo again if more copies of it remain above rank lo, else the smallest value above vlo
    if (threadIdx.x == 0) { s_cnt_le = 0; s_next = INFINITY; }
    __syncthreads();
    unsigned cnt = 0;
    float nxt = INFINITY;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const float v = fabsf(xr[i]);
        cnt += v <= vlo;
        if (v > vlo) nxt = fminf(nxt, v);
    }
    atomicAdd(&s_cnt_le, cnt);
    atomicMin((unsigned*)&s_next, __float_as_uint(nxt));   // non-negative floats: unsigned order = float order
    __syncthreads();
    if (threadIdx.x == 0) {
        const float vhi = (long long)s_cnt_le > lo + 1 || lo + 1 >= n ? vlo : s_next;
        out[blockIdx.x] = vlo + (vhi - vlo) * frac;
    }
}

__global__ __launch_bounds__(256) void diffusion_step_kernel(const float* __restrict__ x0, const float* __restrict__ quant, float s_max, float clip_range,
                                                             const float* __restrict__ sample, const float* __restrict__ mo, const float* __restrict__ noise,
                                                             float k0, float k1, float k2, float k3, float* __restrict__ out, float* __restrict__ x0_out,
                                                             long long per_sample, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = x0[i];
        if (quant) {
            const float s = fminf(fmaxf(quant[i / per_sample], 1.0f), s_max);
            v = fminf(fmaxf(v, -s), s) / s;
        } else if (clip_range > 0.f) {
            v = fminf(fmaxf(v, -clip_range), clip_range);
        }
        if (x0_out) x0_out[i] = v;
        float r = k0 * v + k1 * sample[i] + k2 * mo[i];
        if (noise) r += k3 * noise[i];
        out[i] = r;
    }
}

inline unsigned grid_for(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g > 65535 * 16 ? 65535 * 16 : g);
}

}  // namespace

extern "C" int fm_unet_im2col(const void* src1, int ld1, int C1, const void* src2, int ld2, int C2, int H2, int W2, void* out, int ldo, int kpad,
                              int B, int H, int W, int ksize, int stride, int up1, void* stream) {
    FM_CHECK_ARG(src1 && out && B > 0 && H > 0 && W > 0, "fm_unet_im2col: bad argument");
    FM_CHECK_ARG(ksize == 1 || ksize == 3, "fm_unet_im2col: ksize=%d (1 or 3)", ksize);
    FM_CHECK_ARG(stride == 1 || stride == 2, "fm_unet_im2col: stride=%d (1 or 2)", stride);
    FM_CHECK_ARG(C1 > 0 && C1 % 8 == 0 && C2 % 8 == 0 && ld1 % 8 == 0 && (C2 == 0 || (src2 && ld2 % 8 == 0 && H2 > 0 && W2 > 0)), "fm_unet_im2col: channels / leading dims must be multiples of 8");
    FM_CHECK_ARG(kpad % 8 == 0 && kpad >= ksize * ksize * (C1 + C2) && ldo >= kpad && ldo % 8 == 0, "fm_unet_im2col: kpad=%d ldo=%d too small for %d x %d", kpad, ldo, ksize * ksize, C1 + C2);
    FM_CHECK_ARG(up1 == 0 || (up1 == 1 && H % 2 == 0 && W % 2 == 0), "fm_unet_im2col: up1");
    Im2colArgs a{};
    a.src1 = (const bf16_t*)src1; a.src2 = (const bf16_t*)src2; a.out = (bf16_t*)out;
    a.ld1 = ld1; a.ld2 = ld2; a.ldo = ldo; a.B = B; a.H = H; a.W = W; a.C1 = C1; a.C2 = C2; a.H2 = C2 ? H2 : 1; a.W2 = C2 ? W2 : 1;
    a.ksize = ksize; a.stride = stride; a.up1 = up1; a.kpad = kpad;
    const int pad = ksize / 2;
    a.Ho = (H + 2 * pad - ksize) / stride + 1; a.Wo = (W + 2 * pad - ksize) / stride + 1;
    const long long total = (long long)B * a.Ho * a.Wo * (kpad / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, a);
    FM_CHECK_LAUNCH("fm_unet_im2col");
    return 0;
}

extern "C" int fm_groupnorm_nhwc(const void* x, int ldx, const void* add, int ld_add, const void* w, const void* b, void* y, int ldy, void* stats, int B, int HW,
                                 int C, int groups, float eps, int silu, void* stream) {
    FM_CHECK_ARG(x && w && b && y && stats && B > 0 && HW > 0 && C > 0 && groups > 0, "fm_groupnorm_nhwc: bad argument");
    FM_CHECK_ARG(C % groups == 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "fm_groupnorm_nhwc: C=%d groups=%d (C %% groups == 0, C %% 4 == 0)", C, groups);
    FM_CHECK_ARG(C <= 1024 && groups <= 256, "fm_groupnorm_nhwc: C=%d groups=%d (C <= 1024)", C, groups);
    static const int fused_env = [] { const char* e = getenv("FOURM_GN_FUSED"); return e ? atoi(e) : 1; }();
    const int cpg = C / groups;
    if (fused_env && B * groups >= 64 && cpg % 4 == 0 && 256 % (cpg / 4) == 0 && (!add || ld_add % 4 == 0)) {
        hipLaunchKernelGGL(gn_fused_kernel, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const float*)add, ld_add, (const float*)w,
                           (const float*)b, (bf16_t*)y, ldy, HW, C, groups, silu, eps);
        FM_CHECK_LAUNCH("fm_groupnorm_nhwc");
        return 0;
    }
    const int chunks = (HW + GN_ROWS - 1) / GN_ROWS;
    float* partial = (float*)stats + (size_t)B * groups * 2;                   // scratch layout: [B][G][2] sums, then [B][chunks][G][2] partials
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * chunks), dim3(256), (size_t)(256 / (C / 4)) * C * 2 * sizeof(float), (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const float*)add, ld_add, HW, C, groups, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * groups + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)partial, (float*)stats, B, groups, chunks);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for((long long)B * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const float*)add, ld_add,
                       (const float*)stats, (const float*)w, (const float*)b, (bf16_t*)y, ldy, B, HW, C, groups, silu, eps);
    FM_CHECK_LAUNCH("fm_groupnorm_nhwc");
    return 0;
}

extern "C" int fm_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t rows, int C, void* stream) {
    FM_CHECK_ARG(a && b && out && rows > 0 && C > 0 && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, "fm_add_bf16: bad argument");
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo,
                       (long long)rows, C);
    FM_CHECK_LAUNCH("fm_add_bf16");
    return 0;
}

extern "C" int fm_silu_f32_to_bf16(const void* x, void* y, int64_t n, void* stream) {
    FM_CHECK_ARG(x && y && n > 0, "fm_silu_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (bf16_t*)y, (long long)n);
    FM_CHECK_LAUNCH("fm_silu_f32_to_bf16");
    return 0;
}

extern "C" int fm_timestep_embedding(const void* t, void* out, int ldo, int B, int dim, float max_period, void* stream) {
    FM_CHECK_ARG(t && out && B > 0 && dim > 1 && ldo >= dim, "fm_timestep_embedding: bad argument");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * (dim / 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)t, (bf16_t*)out, ldo, B, dim, max_period);
    FM_CHECK_LAUNCH("fm_timestep_embedding");
    return 0;
}

extern "C" int fm_unet_attention(const void* qkv, int ld, void* out, int ldo, int B, int T, int heads, int ch, void* stream) {
    FM_CHECK_ARG(qkv && out && B > 0 && T > 0 && heads > 0 && ch > 0 && ch % 8 == 0 && ld % 8 == 0, "fm_unet_attention: bad argument (ch %% 8 == 0)");
    const size_t lds = (size_t)(ch + T) * 4;
    FM_CHECK_ARG(lds <= 60 * 1024, "fm_unet_attention: ch + T = %d too large", ch + T);
    hipLaunchKernelGGL(unet_attn_kernel, dim3(T, heads, B), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)qkv, ld, (bf16_t*)out, ldo, T, ch, heads);
    FM_CHECK_LAUNCH("fm_unet_attention");
    return 0;
}

extern "C" int fm_diffusion_x0(const void* sample, const void* model_output, float c0, float c1, void* x0, int64_t n, void* stream) {
    FM_CHECK_ARG(sample && model_output && x0 && n > 0, "fm_diffusion_x0: bad argument");
    hipLaunchKernelGGL(diffusion_x0_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)sample, (const float*)model_output, c0, c1, (float*)x0, (long long)n);
    FM_CHECK_LAUNCH("fm_diffusion_x0");
    return 0;
}

extern "C" int fm_quantile_abs(const void* x, int B, int64_t n, float q, void* out, void* stream) {
    FM_CHECK_ARG(x && out && B > 0 && n > 0 && q >= 0.f && q <= 1.f, "fm_quantile_abs: bad argument");
    hipLaunchKernelGGL(quantile_abs_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (const float*)x, (long long)n, q, (float*)out);
    FM_CHECK_LAUNCH("fm_quantile_abs");
    return 0;
}

extern "C" int fm_diffusion_step(const void* x0, const void* quantile, float sample_max_value, float clip_range, const void* sample, const void* model_output,
                                 const void* noise, float k0, float k1, float k2, float k3, void* out, void* x0_out, int B, int64_t per_sample, void* stream) {
    FM_CHECK_ARG(x0 && sample && model_output && out && B > 0 && per_sample > 0, "fm_diffusion_step: bad argument");
    const long long n = (long long)B * per_sample;
    hipLaunchKernelGGL(diffusion_step_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x0, (const float*)quantile, sample_max_value, clip_range,
                       (const float*)sample, (const float*)model_output, (const float*)noise, k0, k1, k2, k3, (float*)out, (float*)x0_out, (long long)per_sample, n);
    FM_CHECK_LAUNCH("fm_diffusion_step");
    return 0;
}
