// Input / target masks of image-like modalities on the device (first slice).
//
// Replaces, per sample and modality, UnifiedMasking.image_mask (fourm/data/masking.py:237-266): argsort of a uniform noise vector,
// two gathers of budget step functions through that permutation, and the decoder_attention_mask entry that carries the number of
// targets at the first target position.  The loader runs it per sample in worker processes on the host; here one workgroup per
// (sample, modality) ranks the noise in LDS.  The noise is the caller's (torch.rand on the device), so the result is a pure function
// of (noise, budgets): bit-exact against the oracle, which is pinned to upstream's function under the same seed.
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int MASK_MAX_L = 4096;

// ids[r] = index of the r-th smallest noise value (ties: lower index first, i.e. a stable argsort)
__global__ __launch_bounds__(256) void image_mask_kernel(const float* __restrict__ noise, const int* __restrict__ in_budget,
                                                         const int* __restrict__ tgt_budget, int L, uint8_t* __restrict__ input_mask,
                                                         uint8_t* __restrict__ target_mask, int* __restrict__ dam) {
    extern __shared__ float sh[];                     // noise (L) | ids (L, as int)
    float* nz = sh;
    int* ids = (int*)(sh + L);
    __shared__ int first_target, n_target;
    const int b = blockIdx.x;
    const float* src = noise + (size_t)b * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) nz[i] = src[i];
    if (threadIdx.x == 0) { first_target = L; n_target = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const float v = nz[j];
        int r = 0;
        for (int k = 0; k < L; ++k) { const float w = nz[k]; r += (w < v) || (w == v && k < j); }
        ids[r] = j;
    }
    __syncthreads();
    const int kin = in_budget[b], kt = tgt_budget ? tgt_budget[b] : -1;
    int cnt = 0, first = L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int id = ids[i];
        const bool in_vis = id < kin;                                        // an input token
        const bool tgt_vis = kt < 0 ? !in_vis : (id >= kin && id < kin + kt);   // a target token (no target budget: every non-input)
        input_mask[(size_t)b * L + i] = in_vis ? 0 : 1;                      // masks: 1 = masked out
        target_mask[(size_t)b * L + i] = tgt_vis ? 0 : 1;
        dam[(size_t)b * L + i] = 0;
        if (tgt_vis) { ++cnt; first = min(first, i); }
    }
    atomicAdd(&n_target, cnt);
    atomicMin(&first_target, first);
    __syncthreads();
    if (threadIdx.x == 0) {
        // upstream: argmin(target_mask + arange * 1e-6): the first unmasked target position, or position 0 when there is none
        const int pos = first_target < L ? first_target : 0;
        dam[(size_t)b * L + pos] = n_target;
    }
}

// ---- compact host-to-device batch format: what crosses PCIe is uint8 pixels, uint16 ids and bit-packed masks ----------------------

// (B, H, W, C) uint8 -> (B, C, H, W) f32 = ((x / 255) - mean[c]) / std[c]: torchvision's to_tensor + normalize (modality_transforms.py:
// 215-218), IEEE divisions in the same order, so the result is bit-identical to the loader's float pipeline
__global__ __launch_bounds__(256) void unpack_image_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C,
                                                           float m0, float m1, float m2, float m3, float s0, float s1, float s2, float s3) {
    const float mean[4] = {m0, m1, m2, m3}, stdv[4] = {s0, s1, s2, s3};
    const size_t hw = (size_t)H * W, total = (size_t)B * hw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / hw, p = e % hw;
        for (int c = 0; c < C; ++c) {
            const float t = __fdiv_rn((float)src[e * C + c], 255.0f);
            dst[(b * C + c) * hw + p] = __fdiv_rn(t - mean[c], stdv[c]);
        }
    }
}

__global__ __launch_bounds__(256) void unpack_ids_kernel(const uint16_t* __restrict__ src, long long* __restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = (long long)src[e];
}

// bits: row b occupies ceil(L / 8) bytes, bit (i & 7) of byte (i >> 3) = mask[b][i]  (numpy.packbits(..., bitorder="little"))
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint8_t* __restrict__ bits, uint8_t* __restrict__ dst, int B, int L) {
    const int bytes = (L + 7) / 8;
    const size_t total = (size_t)B * L;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / L; const int i = (int)(e % L);
        dst[e] = (bits[b * bytes + (i >> 3)] >> (i & 7)) & 1;
    }
}

// decoder_attention_mask of an image-like modality from its target_mask: the target count at the first target position (masking.py:262-264)
__global__ __launch_bounds__(256) void dam_from_target_kernel(const uint8_t* __restrict__ target_mask, int* __restrict__ dam, int L) {
    __shared__ int first, cnt;
    if (threadIdx.x == 0) { first = L; cnt = 0; }
    __syncthreads();
    const uint8_t* t = target_mask + (size_t)blockIdx.x * L;
    int c = 0, f = L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        dam[(size_t)blockIdx.x * L + i] = 0;
        if (!t[i]) { ++c; f = min(f, i); }
    }
    atomicAdd(&cnt, c); atomicMin(&first, f);
    __syncthreads();
    if (threadIdx.x == 0) dam[(size_t)blockIdx.x * L + (first < L ? first : 0)] = cnt;
}

}  // namespace

extern "C" int fm_image_mask(const void* noise, const int32_t* input_budget, const int32_t* target_budget, int B, int L, void* input_mask,
                             void* target_mask, int32_t* decoder_attention_mask, void* stream) {
    FM_CHECK_ARG(noise && input_budget && input_mask && target_mask && decoder_attention_mask, "fm_image_mask: null pointer");
    FM_CHECK_ARG(B > 0 && L > 0 && L <= MASK_MAX_L, "fm_image_mask: 1 <= L <= %d", MASK_MAX_L);
    hipLaunchKernelGGL(image_mask_kernel, dim3(B), dim3(256), (size_t)L * 8, (hipStream_t)stream, (const float*)noise, input_budget, target_budget, L,
                       (uint8_t*)input_mask, (uint8_t*)target_mask, decoder_attention_mask);
    FM_CHECK_LAUNCH("fm_image_mask");
    return 0;
}

extern "C" int fm_unpack_image_u8(const void* src, void* dst, int B, int H, int W, int C, const float* mean, const float* stdv, void* stream) {
    FM_CHECK_ARG(src && dst && mean && stdv && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, "fm_unpack_image_u8: bad argument (C <= 4; mean / std are HOST arrays)");
    float m[4] = {0, 0, 0, 0}, sd[4] = {1, 1, 1, 1};
    for (int c = 0; c < C; ++c) { m[c] = mean[c]; sd[c] = stdv[c]; }
    size_t blocks = ((size_t)B * H * W + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (float*)dst, B, H, W, C,
                       m[0], m[1], m[2], m[3], sd[0], sd[1], sd[2], sd[3]);
    FM_CHECK_LAUNCH("fm_unpack_image_u8");
    return 0;
}

extern "C" int fm_unpack_ids_u16(const void* src, int64_t* dst, int64_t n, void* stream) {
    FM_CHECK_ARG(src && dst && n > 0, "fm_unpack_ids_u16: bad argument");
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_ids_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (long long*)dst, (size_t)n);
    FM_CHECK_LAUNCH("fm_unpack_ids_u16");
    return 0;
}

extern "C" int fm_unpack_mask_bits(const void* bits, void* dst, int B, int L, void* stream) {
    FM_CHECK_ARG(bits && dst && B > 0 && L > 0, "fm_unpack_mask_bits: bad argument");
    size_t blocks = ((size_t)B * L + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)bits, (uint8_t*)dst, B, L);
    FM_CHECK_LAUNCH("fm_unpack_mask_bits");
    return 0;
}

extern "C" int fm_decoder_attention_from_target(const void* target_mask, int32_t* decoder_attention_mask, int B, int L, void* stream) {
    FM_CHECK_ARG(target_mask && decoder_attention_mask && B > 0 && L > 0, "fm_decoder_attention_from_target: bad argument");
    hipLaunchKernelGGL(dam_from_target_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)target_mask, decoder_attention_mask, L);
    FM_CHECK_LAUNCH("fm_decoder_attention_from_target");
    return 0;
}

// ====================================================================================================================================
// Token budgets and sequence span masking on the device (second slice).  Upstream runs both per sample in the loader's
// worker processes (UnifiedMasking.input_token_budget / target_token_budget, fourm/data/masking.py:181-234; sequence_mask :345-445,
// sequence_token_mask :268-343, sequence_emb_mask_span :448-516 with simple_span_masking :58-91 / chunk_span_masking :94-127).
// Every random number is the caller's (Dirichlet draws, uniform noise, the first keep probability, one integer per sample), in the order
// upstream consumes them, so each kernel is a pure function of its inputs: bit-exact against oracle/masking_oracle.py, which is pinned
// to upstream's functions replaying the same draws (tests/golden/make_golden_masking.py).
// ====================================================================================================================================
namespace {

constexpr int BUDGET_MAX_MODS = 64;

// one thread per sample: the work is a few dozen integer operations per try
__global__ __launch_bounds__(64) void token_budget_kernel(const float* __restrict__ main_draws, const float* __restrict__ extra_draws,
                                                          const int* __restrict__ num_tokens, const int* __restrict__ min_tokens,
                                                          const int* __restrict__ max_tokens, const uint8_t* __restrict__ is_img,
                                                          const int* __restrict__ input_budget, int B, int T, int E, int M,
                                                          int* __restrict__ out, int* __restrict__ tries) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int n = num_tokens[b];
    int bud[BUDGET_MAX_MODS];
    int used = T + 1;                      // no try met every minimum: the last one is kept and reported as T + 1
    for (int t = 0; t < T; ++t) {
        const float* p = main_draws + ((size_t)b * T + t) * M;
        int sum = 0;
        for (int m = 0; m < M; ++m) { bud[m] = (int)floorf(p[m] * (float)n); sum += bud[m]; }      // (sample * n).floor().int()  :187
        int diff = n - sum;
        diff = diff < 0 ? 0 : diff > E ? E : diff;
        for (int e = 0; e < diff; ++e) {                                                         // bincount(sample_n(diff).argmax(-1))  :191
            const float* q = extra_draws + (((size_t)b * T + t) * E + e) * M;
            int arg = 0; float best = q[0];
            for (int m = 1; m < M; ++m) if (q[m] > best) { best = q[m]; arg = m; }
            ++bud[arg];
        }
        bool ok = true;
        for (int m = 0; m < M; ++m) {
            int mx = max_tokens[m];
            if (input_budget) {                                                                   // max_tokens_remaining  :218-219
                if (is_img[m]) mx -= input_budget[(size_t)b * M + m];
                mx = max(min_tokens[m], mx);
            }
            bud[m] = min(bud[m], mx);
            ok = ok && bud[m] >= min_tokens[m];
        }
        if (ok) { used = t + 1; break; }
    }
    for (int m = 0; m < M; ++m) out[(size_t)b * M + m] = bud[m];
    if (tries) tries[b] = used;
}

struct SpanArgs {
    const int* ids; const int* len; const int* unit; const float* noise; const double* keep_prob; const int* r_choice;
    const int* input_budget; const int* target_budget; const int* sentinel_ids;
    int* tensor; uint8_t* input_mask; uint8_t* target_mask; int* dam; int* tries; int* src;
    int B, ld_ids, T, ld_noise, n_sent, max_tokens, vocab_offset, pad_id, emb_mode;
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// one wave per sample
__global__ __launch_bounds__(64) void span_mask_kernel(SpanArgs a) {
    extern __shared__ int sh_i[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int MT = a.max_tokens;
    int* tk = sh_i;                 // tokens (MT)
    int* un = tk + MT;              // mask decision a token follows (MT)
    int* inp = un + MT;             // input sequence (<= MT + 1)
    int* tg = inp + MT + 1;         // target sequence (<= 2 MT + 1)
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;

    // ---- the sequence after truncation (:365 tokens, :376-377 whole chunks) -------------------------------------------------------------
    const int n_raw = min(a.len[b], a.ld_ids);
    int n = min(n_raw, MT);
    if (a.unit) {
        int keep = 0;
        for (int l = lane; l < n; l += 64) {
            const int* u = a.unit + (size_t)b * a.ld_ids;
            if (l + 1 == n_raw || u[l + 1] != u[l]) keep = max(keep, l + 1);
        }
        n = wave_max(keep);
    }
    for (int l = lane; l < n; l += 64) {
        tk[l] = a.emb_mode ? l : a.ids[(size_t)b * a.ld_ids + l] + a.vocab_offset;
        un[l] = a.unit ? a.unit[(size_t)b * a.ld_ids + l] : l;
    }
    __syncthreads();

    // ---- keep probability: lowered by 0.9 per retry until the input fits its budget (:389-408) ------------------------------------------
    const int kin = a.input_budget[b];
    const float* nz = a.noise + (size_t)b * a.T * a.ld_noise;
    double kp = kin == 0 ? 0.0 : a.keep_prob[b];
    int t = 0;
    bool all_masked = false;
    auto is_masked = [&](int l, const float* row, float kpf) { return all_masked || !(row[un[l]] <= kpf); };
    if (kin != 0) {
        for (;;) {
            const float kpf = (float)kp;
            const float* row = nz + (size_t)t * a.ld_noise;
            int cnt = 0;
            for (int l = lane; l < n; l += 64) {
                const bool m = is_masked(l, row, kpf);
                const bool pm = l > 0 && is_masked(l - 1, row, kpf);
                cnt += !m || !pm;                                   // a kept token, or the sentinel that opens a span
            }
            cnt = wave_sum(cnt);
            if (cnt <= kin) break;
            kp = kp * 0.9;
            if (++t >= a.T) { all_masked = true; break; }               // (no draws left: everything masked, the limit kp -> 0; reported as T + 1)
        }
    }
    if (lane == 0 && a.tries) a.tries[b] = t + 1;

    // ---- input and target sequences (simple_span_masking :73-91) ------------------------------------------------------------------------
    const float kpf = (float)kp;
    const float* row = nz + (size_t)(all_masked ? 0 : t) * a.ld_noise;
    int kept_b = 0, masked_b = 0, starts_b = 0;             // counts over the tokens before the current group of 64
    for (int base = 0; base < n; base += 64) {
        const int l = base + lane;
        const bool valid = l < n;
        const bool m = valid && is_masked(l, row, kpf);
        const bool pm = valid && l > 0 && is_masked(l - 1, row, kpf);
        const bool st = m && !pm, kept = valid && !m;
        const unsigned long long bm = __ballot(m), bs = __ballot(st), bk = __ballot(kept);
        const int kb = kept_b + __popcll(bk & below), mb = masked_b + __popcll(bm & below), sb = starts_b + __popcll(bs & below);
        if (kept || st) inp[kb + sb] = st ? (a.emb_mode ? -1 : a.sentinel_ids[min(sb + 1, a.n_sent - 1)]) : tk[l];
        if (m) {
            if (st) tg[mb + sb] = a.sentinel_ids[min(sb + 1, a.n_sent - 1)];
            tg[mb + sb + (st ? 1 : 0)] = tk[l];
        }
        kept_b += __popcll(bk); masked_b += __popcll(bm); starts_b += __popcll(bs);
    }
    if (lane == 0) tg[masked_b + starts_b] = a.sentinel_ids[min(starts_b + 1, a.n_sent - 1)];
    const int in_len = kin == 0 ? 0 : kept_b + starts_b;
    int tgt_len = masked_b + starts_b + 1, tgt_off = 0;
    __syncthreads();

    // more spans than sentinel ids: upstream raises KeyError; reported as tries = -1 (ids are clamped to the last sentinel)
    if (lane == 0 && a.tries && starts_b + 1 > a.n_sent - 1) a.tries[b] = -1;

    if (a.emb_mode) {     // sequence_emb_mask_span (:448-516): inputs only; position i holds embedding row inp[i], a sentinel position (-1) a zero row
        for (int i = lane; i < MT; i += 64) {
            a.input_mask[(size_t)b * MT + i] = i < in_len ? 0 : 1;
            a.target_mask[(size_t)b * MT + i] = 1;
            a.dam[(size_t)b * MT + i] = 0;
            a.src[(size_t)b * MT + i] = i < in_len ? inp[i] : -1;
        }
        return;
    }

    // ---- a target longer than its budget starts at a sentinel (:420-438) ----------------------------------------------------------------
    const int kt = a.target_budget ? a.target_budget[b] : -1;
    if (kt >= 0 && tgt_len > kt) {
        // sentinel positions: "token_id in self.sentinel_ids" - by value, as upstream
        int nS = 0;
        for (int base = 0; base < tgt_len; base += 64) {
            const int i = base + lane;
            bool s = false;
            if (i < tgt_len) for (int k = 0; k < a.n_sent; ++k) s = s || tg[i] == a.sentinel_ids[k];
            nS += __popcll(__ballot(s));
        }
        const int chosen = (a.r_choice ? a.r_choice[b] : 0) % max(1, nS - 1);
        int start = 0x7fffffff, fit = 0x7fffffff, seen = 0;
        for (int base = 0; base < tgt_len; base += 64) {
            const int i = base + lane;
            bool s = false;
            if (i < tgt_len) for (int k = 0; k < a.n_sent; ++k) s = s || tg[i] == a.sentinel_ids[k];
            const unsigned long long bsent = __ballot(s);
            const int ord = seen + __popcll(bsent & below);
            if (s && ord == chosen) start = min(start, i);
            if (s && tgt_len - i <= kt) fit = min(fit, i);             // earliest sentinel from which the rest fits
            seen += __popcll(bsent);
        }
        start = wave_min(start); fit = wave_min(fit);
        if (nS > 0 && tgt_len - start >= kt) { tgt_off = start; tgt_len = kt; }
        else if (fit != 0x7fffffff) { tgt_off = fit; tgt_len -= fit; }
    }

    // ---- the padded row (:411-443) --------------------------------------------------------------------------------------------------------
    const int L = 2 * (MT + 1);
    for (int i = lane; i < L; i += 64) {
        const bool in_i = i < in_len, tg_i = i >= kin && i < kin + tgt_len;
        a.tensor[(size_t)b * L + i] = tg_i ? tg[tgt_off + i - kin] : in_i ? inp[i] : a.pad_id;
        a.input_mask[(size_t)b * L + i] = in_i ? 0 : 1;
        a.target_mask[(size_t)b * L + i] = tg_i ? 0 : 1;
        a.dam[(size_t)b * L + i] = tg_i ? 1 : 0;
    }
}

// rows of a (B, n, D) f32 tensor into (B, max_tokens, D) through a source index per output row (-1: a zero row)
__global__ __launch_bounds__(256) void gather_emb_rows_kernel(const float* __restrict__ emb, const int* __restrict__ src, float* __restrict__ out,
                                                              int n, int MT, int D) {
    const int b = blockIdx.y, i = blockIdx.x;
    const int s = src[(size_t)b * MT + i];
    for (int d = threadIdx.x; d < D; d += 256)
        out[((size_t)b * MT + i) * D + d] = s >= 0 ? emb[((size_t)b * n + s) * D + d] : 0.f;
}

}  // namespace

extern "C" int fm_token_budgets(const void* main_draws, const void* extra_draws, const int32_t* num_tokens, const int32_t* min_tokens,
                                const int32_t* max_tokens, const void* is_img, const int32_t* input_budget, int B, int T, int E, int M,
                                int32_t* budget, int32_t* tries, void* stream) {
    FM_CHECK_ARG(main_draws && extra_draws && num_tokens && min_tokens && max_tokens && budget, "fm_token_budgets: null pointer");
    FM_CHECK_ARG(B > 0 && T > 0 && E >= 0 && M > 0 && M <= BUDGET_MAX_MODS, "fm_token_budgets: 1 <= M <= %d, T >= 1", BUDGET_MAX_MODS);
    FM_CHECK_ARG(!input_budget || is_img, "fm_token_budgets: input_budget (target budgets) needs is_img");
    hipLaunchKernelGGL(token_budget_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)main_draws, (const float*)extra_draws,
                       num_tokens, min_tokens, max_tokens, (const uint8_t*)is_img, input_budget, B, T, E, M, budget, tries);
    FM_CHECK_LAUNCH("fm_token_budgets");
    return 0;
}

extern "C" int fm_span_mask(const fm_span_mask_args* p, void* stream) {
    FM_CHECK_ARG(p && p->len && p->noise && p->keep_prob && p->input_budget && p->sentinel_ids && p->input_mask && p->target_mask &&
                 p->decoder_attention_mask, "fm_span_mask: null pointer");
    FM_CHECK_ARG(p->B > 0 && p->T > 0 && p->max_tokens > 0 && p->max_tokens <= 4096, "fm_span_mask: 1 <= max_tokens <= 4096, T >= 1");
    FM_CHECK_ARG(p->n_sentinels >= 2, "fm_span_mask: at least two sentinel ids");
    const bool emb = p->emb != nullptr;
    if (emb) FM_CHECK_ARG(p->emb_out && p->src && p->emb_rows > 0 && p->emb_dim > 0, "fm_span_mask: embedding mode needs emb_out, src, emb_rows, emb_dim");
    else FM_CHECK_ARG(p->ids && p->tensor && p->ld_ids > 0, "fm_span_mask: ids / tensor missing");
    FM_CHECK_ARG(p->ld_noise >= (emb ? (p->emb_rows < p->max_tokens ? p->emb_rows : p->max_tokens) : 1), "fm_span_mask: noise rows too short");
    SpanArgs a{};
    a.ids = p->ids; a.len = p->len; a.unit = emb ? nullptr : p->unit; a.noise = (const float*)p->noise; a.keep_prob = p->keep_prob; a.r_choice = p->r_choice;
    a.input_budget = p->input_budget; a.target_budget = p->target_budget; a.sentinel_ids = p->sentinel_ids;
    a.tensor = p->tensor; a.input_mask = (uint8_t*)p->input_mask; a.target_mask = (uint8_t*)p->target_mask; a.dam = p->decoder_attention_mask;
    a.tries = p->tries; a.src = p->src;
    a.B = p->B; a.ld_ids = emb ? p->emb_rows : p->ld_ids; a.T = p->T; a.ld_noise = p->ld_noise; a.n_sent = p->n_sentinels; a.max_tokens = p->max_tokens;
    a.vocab_offset = p->vocab_offset; a.pad_id = p->pad_id; a.emb_mode = emb ? 1 : 0;
    const size_t lds = (size_t)(5 * p->max_tokens + 8) * sizeof(int);
    hipLaunchKernelGGL(span_mask_kernel, dim3(p->B), dim3(64), lds, (hipStream_t)stream, a);
    FM_CHECK_LAUNCH("fm_span_mask");
    if (emb) {
        hipLaunchKernelGGL(gather_emb_rows_kernel, dim3(p->max_tokens, p->B), dim3(256), 0, (hipStream_t)stream, (const float*)p->emb, (const int*)p->src,
                           (float*)p->emb_out, p->emb_rows, p->max_tokens, p->emb_dim);
        FM_CHECK_LAUNCH("fm_span_mask (embedding rows)");
    }
    return 0;
}
