// Token sampling for the generation step (gfx950): temperature / top-k / top-p filtering + multinomial draw per logits row, and the
// MaskGIT commit (keep the num_select most confident samples, write them into mod_dict).  Replaces, for one decoding step,
//   GenerationSampler.top_k_top_p_filtering / sample_tokens / select_tokens_batched and the scatter updates of
//   maskgit_step_batched   (fourm/models/generate.py:332-371, :373-420, :650-661)
// i.e. two topk + one full sort + softmax + cumsum + argsort + gather + multinomial + three scatters per step, each a (B*N, V) pass.
//
// HBM-bound integer / byte work: ONE read of the logits row (2 or 4 B per entry) into LDS, everything else on chip.
//
// DETERMINISM CONTRACT (what makes "same logits + same uniforms -> same token ids" hold bit for bit against the CPU restatement
// oracle/sample_oracle.py, and run to run):
//   * top-k: the k-th largest logit is found by a radix select on order-preserving integer keys; entries strictly below it are
//     dropped (ties at the threshold all survive, as `logits < kth` upstream);
//   * top-p: probabilities at temperature 1 are quantised to integers q = floor(exp(l - max) * 2^24) and accumulated in 64-bit
//     integers; an entry survives iff the mass of the entries with strictly larger logits is <= floor(top_p * total) (upstream:
//     sorted cumulative softmax shifted by one; equal logits share their fate here);
//   * exp is a fixed polynomial evaluated with separately rounded fp32 multiplies and adds (no fma, no hardware v_exp): the same
//     sequence of IEEE operations in numpy gives the same bits;
//   * the multinomial draw is an inverse CDF over the surviving entries IN INDEX ORDER with a fixed summation tree: 256 contiguous
//     chunks summed left to right, chunk totals scanned left to right, then the chosen chunk walked left to right; the caller
//     supplies one uniform in [0, 1) per row (torch.rand on the device: the RNG is the caller's, not baked into the kernel).
#include "common.h"
#include "fourm_hip.h"

// the determinism contract needs every fp32 multiply and add rounded on its own: no fused multiply-add anywhere in this file
// (__fmul_rn / __fadd_rn are plain * and + in this toolchain and would be contracted under the default -ffp-contract=fast)
#pragma clang fp contract(off)

namespace {

constexpr int NT = 256;                       // threads per row
constexpr int MAX_V = 36864;                  // 144 KB of fp32 logits in LDS (every 4M vocabulary: <= 30000)

__device__ __forceinline__ uint32_t okey(float f) {          // order-preserving: a < b  <=>  okey(a) < okey(b)   (no NaNs)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// exp(x) for x <= 0 with separately rounded fp32 operations (see the contract above); 0 below -87
__device__ __forceinline__ float exp_det(float x) {
    if (x < -80.0f) return 0.0f;
    const float t = __fmul_rn(x, 1.44269504088896341f);
    const float n = rintf(t);
    float r = __fadd_rn(x, -__fmul_rn(n, 0.693359375f));
    r = __fadd_rn(r, -__fmul_rn(n, -2.12194440e-4f));
    float p = 1.3888888888888889e-3f;                                       // 1/720
    p = __fadd_rn(__fmul_rn(p, r), 8.3333333333333333e-3f);                 // 1/120
    p = __fadd_rn(__fmul_rn(p, r), 4.1666666666666667e-2f);                 // 1/24
    p = __fadd_rn(__fmul_rn(p, r), 1.6666666666666667e-1f);                 // 1/6
    p = __fadd_rn(__fmul_rn(p, r), 0.5f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f);
    p = __fadd_rn(__fmul_rn(p, r), 1.0f);
    return ldexpf(p, (int)n);
}

// block-wide helpers on 256 threads ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// k-th largest key among entries with key >= floor_key (k >= 1).  Radix select, 4 passes of 8 bits, integer histograms in LDS.
__device__ uint32_t radix_kth(const float* row, int V, int k, uint32_t* hist, uint32_t* bc) {
    uint32_t prefix = 0, mask = 0;
    int want = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < V; i += NT) {
            const uint32_t key = okey(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int d = 255, acc = 0;
            for (; d > 0; --d) { if (acc + (int)hist[d] >= want) break; acc += hist[d]; }
            bc[0] = (uint32_t)d; bc[1] = (uint32_t)(want - acc);
        }
        __syncthreads();
        prefix |= bc[0] << shift; mask |= 255u << shift; want = (int)bc[1];
        __syncthreads();
    }
    return prefix;
}

// smallest key Kc such that the integer mass of entries with key > Kc is <= thr  (entries with key >= Kc survive top-p)
__device__ __forceinline__ uint32_t mass_q(float logit, float mx, uint32_t floor_key) {
    return okey(logit) >= floor_key ? (uint32_t)(exp_det(__fadd_rn(logit, -mx)) * 16777216.0f) : 0u;
}
__device__ uint32_t radix_mass_cut(const float* row, float mx, uint32_t floor_key, int V, unsigned long long thr, unsigned long long* mh,
                                   uint32_t* bc, unsigned long long* bcm) {
    uint32_t prefix = 0, mask = 0;
    unsigned long long above = 0;                 // mass of keys greater than every key with this prefix
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += NT) mh[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < V; i += NT) {
            const uint32_t key = okey(row[i]);
            if ((key & mask) != prefix) continue;
            const uint32_t qi = mass_q(row[i], mx, floor_key);      // recomputed per pass: the row is the only LDS-resident array
            if (qi) atomicAdd(&mh[(key >> shift) & 255], (unsigned long long)qi);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // lowest digit d whose bin still holds a survivor: above + mass of higher digits <= thr
            unsigned long long a = above;
            int d = 255, cut = 255;
            unsigned long long a_cut = above;
            for (; d >= 0; --d) {
                if (a > thr) break;
                if (mh[d]) { cut = d; a_cut = a; }
                a += mh[d];
            }
            bc[0] = (uint32_t)cut; bcm[0] = a_cut;
        }
        __syncthreads();
        prefix |= bc[0] << shift; mask |= 255u << shift; above = bcm[0];
        __syncthreads();
    }
    return prefix;
}

template <typename T> __device__ __forceinline__ float ldlogit(const T* p, int i);
template <> __device__ __forceinline__ float ldlogit<float>(const float* p, int i) { return p[i]; }
template <> __device__ __forceinline__ float ldlogit<bf16_t>(const bf16_t* p, int i) { return bf2f(p[i]); }

template <typename T>
__global__ __launch_bounds__(NT) void sample_tokens_kernel(const T* __restrict__ logits, int ld, int V, float inv_temp, int greedy, int top_k,
                                                           float top_p, const float* __restrict__ uniforms, long long* __restrict__ out_ids,
                                                           float* __restrict__ out_prob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* row = (float*)smem;                                  // [V] logits, later probabilities
    __shared__ float red[4];
    __shared__ uint32_t hist[256], bc[2];
    __shared__ unsigned long long mh[256], bcm[1];
    __shared__ float csum[NT];
    const int r = blockIdx.x, t = threadIdx.x;
    const T* src = logits + (size_t)r * ld;
    float mx = -INFINITY;
    for (int i = t; i < V; i += NT) { const float v = ldlogit<T>(src, i); row[i] = v; mx = fmaxf(mx, v); }
    mx = block_max(mx, red);
    if (greedy) {                                               // temperature 0: argmax, first index on ties, probability 1
        int best = 0x7fffffff;
        for (int i = t; i < V; i += NT) if (row[i] == mx) best = min(best, i);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
        __syncthreads();
        if ((t & 63) == 0) hist[t >> 6] = (uint32_t)best;
        __syncthreads();
        if (t == 0) { out_ids[r] = (long long)min(min(hist[0], hist[1]), min(hist[2], hist[3])); out_prob[r] = 1.0f; }
        return;
    }
    uint32_t keep_key = 0;                                      // entries with key < keep_key are filtered out
    if (top_k > 0 && top_k < V) keep_key = radix_kth(row, V, top_k, hist, bc);
    if (top_p > 0.0f && top_p < 1.0f) {                          // (top_p = 1 keeps everything)
        // temperature-1 masses of the top-k survivors, as integers
        unsigned long long part = 0;
        for (int i = t; i < V; i += NT) part += mass_q(row[i], mx, keep_key);
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        __syncthreads();
        if ((t & 63) == 0) mh[t >> 6] = part;
        __syncthreads();
        const unsigned long long total = (mh[0] + mh[1]) + (mh[2] + mh[3]);
        __syncthreads();
        const unsigned long long thr = (unsigned long long)((double)top_p * (double)total);
        const uint32_t kp = radix_mass_cut(row, mx, keep_key, V, thr, mh, bc, bcm);
        keep_key = max(keep_key, kp);
    }
    // probabilities at the sampling temperature, zero for filtered entries; fixed summation tree
    const int C = (V + NT - 1) / NT;
    float s = 0.f;
    for (int i = t * C; i < min(V, (t + 1) * C); ++i) {
        const float p = okey(row[i]) >= keep_key ? exp_det(__fmul_rn(__fadd_rn(row[i], -mx), inv_temp)) : 0.f;
        row[i] = p;
        s = __fadd_rn(s, p);
    }
    csum[t] = s;
    __syncthreads();
    if (t == 0) {
        float run = 0.f;
        for (int c = 0; c < NT; ++c) { run = __fadd_rn(run, csum[c]); csum[c] = run; }      // inclusive prefix over chunks
        const float total = run;
        const float target = __fmul_rn(uniforms[r], total);
        int c = 0;
        while (c < NT - 1 && !(csum[c] > target)) ++c;
        float acc = c ? csum[c - 1] : 0.f;
        int pick = -1, last = -1;
        for (int i = c * C; i < min(V, (c + 1) * C); ++i) {
            if (row[i] > 0.f) last = i;
            acc = __fadd_rn(acc, row[i]);
            if (acc > target && row[i] > 0.f) { pick = i; break; }
        }
        if (pick < 0) {                                         // rounding at the very end of the CDF: the last live entry
            pick = last;
            for (int i = V - 1; pick < 0 && i >= 0; --i) if (row[i] > 0.f) pick = i;
        }
        out_ids[r] = pick;
        out_prob[r] = __fdiv_rn(row[pick], total);
    }
}

// MaskGIT commit: per sample, the num_select entries of `prob` (B, N) with the largest value (ties: lower index first) are written
// into the modality's tensors at their positions: tensor[b][pos] = sample, input_mask[b][pos] = 0, target_mask[b][pos] = 1.
__global__ __launch_bounds__(NT) void maskgit_commit_kernel(const float* __restrict__ prob, const long long* __restrict__ samples,
                                                            const int32_t* __restrict__ mod_pos, int N, int num_select, void* tensor,
                                                            int tensor_is_i64, int L, uint8_t* input_mask, uint8_t* target_mask,
                                                            int32_t* __restrict__ top_idx) {
    const int b = blockIdx.x;
    const float* p = prob + (size_t)b * N;
    for (int i = threadIdx.x; i < N; i += NT) {
        const float v = p[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) { const float w = p[j]; rank += (w > v) || (w == v && j < i); }
        if (rank < num_select) {
            top_idx[(size_t)b * num_select + rank] = i;
            const int pos = mod_pos[(size_t)b * N + i];
            if (tensor_is_i64) ((long long*)tensor)[(size_t)b * L + pos] = samples[(size_t)b * N + i];
            else ((int32_t*)tensor)[(size_t)b * L + pos] = (int32_t)samples[(size_t)b * N + i];
            input_mask[(size_t)b * L + pos] = 0;
            target_mask[(size_t)b * L + pos] = 1;
        }
    }
}

}  // namespace

extern "C" int fm_sample_tokens(const void* logits, int ld, int logits_are_f32, int R, int V, float temperature, int top_k, float top_p,
                                const void* uniforms, void* out_ids, void* out_prob, void* stream) {
    FM_CHECK_ARG(logits && out_ids && out_prob && R > 0 && V > 0, "fm_sample_tokens: bad argument");
    FM_CHECK_ARG(V <= MAX_V, "fm_sample_tokens: vocabulary %d exceeds the LDS-resident row (%d)", V, MAX_V);
    FM_CHECK_ARG(top_k >= 0 && top_p >= 0.f && top_p <= 1.f && temperature >= 0.f, "fm_sample_tokens: bad sampling parameters");
    const int greedy = temperature < 1e-10f;
    FM_CHECK_ARG(greedy || uniforms, "fm_sample_tokens: sampling needs one uniform per row");
    const size_t lds = (size_t)V * 4;
    const float inv_temp = greedy ? 1.0f : 1.0f / temperature;
#define FM_LAUNCH_SAMPLE(T)                                                                                                                   \
    {                                                                                                                                         \
        auto k = sample_tokens_kernel<T>;                                                                                                     \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192) == hipSuccess); \
        (void)once;                                                                                                                           \
        hipLaunchKernelGGL(k, dim3(R), dim3(NT), lds, (hipStream_t)stream, (const T*)logits, ld, V, inv_temp, greedy, top_k, top_p,          \
                           (const float*)uniforms, (long long*)out_ids, (float*)out_prob);                                                    \
    }
    if (logits_are_f32) FM_LAUNCH_SAMPLE(float) else FM_LAUNCH_SAMPLE(bf16_t)
#undef FM_LAUNCH_SAMPLE
    FM_CHECK_LAUNCH("fm_sample_tokens");
    return 0;
}

// Classifier-free guidance on logits: out = base + w * (cond - uncond), fp32, one rounding per operation (this file is compiled with
// -ffp-contract=off), base = uncond for the first condition and the running sum for further ones (generate.py:684, :718).
template <typename TU, typename TC>
__global__ __launch_bounds__(256) void guidance_kernel(const TU* __restrict__ uncond, int ldu, const TC* __restrict__ cond, int ldc, float w,
                                                       float* __restrict__ out, int ldo, int R, int V, int accumulate) {
    const size_t total = (size_t)R * V;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / V), c = (int)(e % V);
        const float u = ldlogit<TU>(uncond + (size_t)r * ldu, c);
        const float d = ldlogit<TC>(cond + (size_t)r * ldc, c) - u;
        const float base = accumulate ? out[(size_t)r * ldo + c] : u;
        out[(size_t)r * ldo + c] = base + w * d;
    }
}

extern "C" int fm_maskgit_commit(const void* prob, const void* samples, const int32_t* mod_pos, int B, int N, int num_select, void* tensor,
                                 int tensor_is_i64, int L, void* input_mask, void* target_mask, int32_t* top_idx, void* stream) {
    FM_CHECK_ARG(prob && samples && mod_pos && tensor && input_mask && target_mask && top_idx, "fm_maskgit_commit: null pointer");
    FM_CHECK_ARG(B > 0 && N > 0 && num_select > 0 && num_select <= N && L > 0, "fm_maskgit_commit: bad shape");
    hipLaunchKernelGGL(maskgit_commit_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, (const float*)prob, (const long long*)samples, mod_pos, N,
                       num_select, tensor, tensor_is_i64, L, (uint8_t*)input_mask, (uint8_t*)target_mask, top_idx);
    FM_CHECK_LAUNCH("fm_maskgit_commit");
    return 0;
}

extern "C" int fm_guidance_combine(const void* uncond, int ldu, int uncond_is_f32, const void* cond, int ldc, int cond_is_f32, float weight,
                                   void* out, int ldo, int R, int V, int accumulate, void* stream) {
    FM_CHECK_ARG(uncond && cond && out && R > 0 && V > 0, "fm_guidance_combine: bad argument");
    size_t blocks = ((size_t)R * V + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = (hipStream_t)stream;
#define FM_GC(TU, TC) hipLaunchKernelGGL((guidance_kernel<TU, TC>), dim3((unsigned)blocks), dim3(256), 0, s, (const TU*)uncond, ldu, (const TC*)cond, ldc, \
                                          weight, (float*)out, ldo, R, V, accumulate)
    if (uncond_is_f32) { if (cond_is_f32) FM_GC(float, float); else FM_GC(float, bf16_t); }
    else { if (cond_is_f32) FM_GC(bf16_t, float); else FM_GC(bf16_t, bf16_t); }
#undef FM_GC
    FM_CHECK_LAUNCH("fm_guidance_combine");
    return 0;
}
