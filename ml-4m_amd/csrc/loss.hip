// Per-modality heads: row segmentation by modality + cross-entropy over the discrete-token vocab.
//
// Upstream (fourm/models/fm.py:573-637) boolean-indexes the decoder output per modality (a host
// sync each), runs `to_logits` and F.cross_entropy (fp32, mean) and averages over modalities.
// Here the decoder rows are bucketed on the device into FM_SEG_ROWS-aligned segments (one per
// modality), one grouped GEMM produces bf16 logits for every segment (gemm.hip), and the kernels
// below turn them into per-row losses and in-place d(logits).  No host synchronisation anywhere.
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int SEG_ALIGN = FM_SEG_ROWS;

// grid = n_heads workgroups.  Workgroup g counts every head <= g (cheap: one int per row) to find
// its SEG_ALIGN-aligned start, then writes the stable list of its rows.
__global__ __launch_bounds__(1024) void segment_rows_kernel(const int32_t* __restrict__ head_of_row, int R, int n_heads,
                                                            int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_count,
                                                            int32_t* __restrict__ perm, int32_t* __restrict__ row_to_padded,
                                                            int32_t* __restrict__ tile_group, int Rp) {
    __shared__ int cnt[FM_MAX_MODS];
    __shared__ int wtot[16];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < FM_MAX_MODS) cnt[tid] = 0;
    __syncthreads();
    for (int r = tid; r < R; r += 1024) {
        const int hh = head_of_row[r];
        if (hh >= 0 && hh <= g) atomicAdd(&cnt[hh], 1);
    }
    __syncthreads();
    int start = 0;
    for (int m = 0; m < g; ++m) start += (cnt[m] + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN;
    const int mine = cnt[g];
    const int padded = (mine + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN;
    if (tid == 0) { seg_start[g] = start; seg_count[g] = mine; }
    for (int t = tid; t < padded / SEG_ALIGN; t += 1024) tile_group[start / SEG_ALIGN + t] = g;
    if (g == n_heads - 1) {   // tiles (and rows) past the last segment are unused
        for (int t = (start + padded) / SEG_ALIGN + tid; t < Rp / SEG_ALIGN; t += 1024) tile_group[t] = -1;
        for (int r = start + padded + tid; r < Rp; r += 1024) perm[r] = -1;
    }
    for (int r = start + mine + tid; r < start + padded; r += 1024) perm[r] = -1;
    // stable compaction of this head's rows
    int run = 0;
    for (int base = 0; base < R; base += 1024) {
        const int r = base + tid;
        const int f = (r < R && head_of_row[r] == g) ? 1 : 0;
        const int inc = wave_scan_incl(f);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int off = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) off += wtot[w]; tot += wtot[w]; }
        __syncthreads();
        if (f) {
            const int dst = start + run + off + inc - 1;
            perm[dst] = r;
            row_to_padded[r] = dst;
        }
        run += tot;
    }
}

__global__ void fill_i32_kernel(int32_t* p, int v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// Forward: one workgroup per padded row, ONE streaming pass over the bf16 logits (online max / sum of
// exponentials per thread, merged across the workgroup): row_loss = lse - logits[target], row_lse kept
// for the backward.  Nothing is staged in LDS, so many rows are in flight per CU.
__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const bf16_t* __restrict__ logits, int ldl, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ tile_group, const long long* __restrict__ target_ids,
                                                     const int32_t* __restrict__ vocab, float* __restrict__ row_loss, float* __restrict__ row_lse) {
    __shared__ float red_m[4], red_s[4];
    const int pr = blockIdx.x;
    const int g = tile_group[pr / SEG_ALIGN];
    if (g < 0) return;
    const int src = perm[pr];
    if (src < 0) {
        if (threadIdx.x == 0) { row_loss[pr] = 0.f; row_lse[pr] = 0.f; }
        return;
    }
    const int V = vocab[g];
    const bf16_t* row = logits + (size_t)pr * ldl;
    float m = -INFINITY, ssum = 0.f;
    for (int c = threadIdx.x * 8; c < V; c += 256 * 8) {
        float v[8];
        if (c + 8 <= V) {
            const uint4 p = *(const uint4*)(row + c);
            const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = bf2f((bf16_t)(w[e] & 0xffff)); v[2 * e + 1] = bf2f((bf16_t)(w[e] >> 16)); }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = c + e < V ? bf2f(row[c + e]) : -INFINITY;
        }
        float cm = v[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) cm = fmaxf(cm, v[e]);
        const float mn = fmaxf(m, cm);
        float add = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) add += __expf(v[e] - mn);
        ssum = ssum * __expf(m - mn) + add;
        m = mn;
    }
    // merge across the wave, then across the 4 waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(ssum, o, 64);
        if (m2 > -INFINITY || m > -INFINITY) online_merge(m, ssum, m2, s2);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red_m[wave] = m; red_s[wave] = ssum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = red_m[0], S = red_s[0];
        for (int w = 1; w < 4; ++w)
            if (red_m[w] > -INFINITY) online_merge(M, S, red_m[w], red_s[w]);
        const float lse = M + __logf(S);
        row_lse[pr] = lse;
        row_loss[pr] = lse - bf2f(row[target_ids[src]]);
    }
}

// Backward: logits (bf16) are overwritten in place by d(total)/d(logits) = (exp(x - lse) - onehot) * w_head
// (zero in pad rows and in the pad columns up to roundup64(vocab)): one read + one write per element.
__global__ __launch_bounds__(256) void ce_bwd_kernel(bf16_t* __restrict__ logits, int ldl, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ tile_group, const long long* __restrict__ target_ids,
                                                     const int32_t* __restrict__ vocab, const int32_t* __restrict__ seg_count,
                                                     const float* __restrict__ gscale, int loss_type, int n_heads,
                                                     const float* __restrict__ row_lse) {
    const int pr = blockIdx.x;
    const int g = tile_group[pr / SEG_ALIGN];
    if (g < 0) return;
    const int V = vocab[g];
    const int Vp = (V + 63) / 64 * 64;
    bf16_t* row = logits + (size_t)pr * ldl;
    const int src = perm[pr];
    if (src < 0) {   // pad row of a live segment: gradient must read as zero in the dW / dX GEMMs
        for (int c = threadIdx.x; c < Vp / 8; c += 256) *(uint4*)(row + c * 8) = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    float w;
    if (loss_type == FM_LOSS_MOD) {
        w = 1.0f / ((float)seg_count[g] * (float)n_heads);
    } else {   // FM_LOSS_TOKEN: heads weighted by logits.numel() = rows * vocab (fm.py:633-635)
        float tot = 0.f;
        for (int mm = 0; mm < n_heads; ++mm) tot += (float)seg_count[mm] * (float)vocab[mm];
        w = (float)V / tot;
    }
    w *= gscale ? gscale[0] : 1.0f;
    const float lse = row_lse[pr];
    const int tgt = (int)target_ids[src];
    for (int c = threadIdx.x * 8; c < Vp; c += 256 * 8) {
        const uint4 p = *(const uint4*)(row + c);
        const uint32_t wd[4] = {p.x, p.y, p.z, p.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = bf2f((bf16_t)((e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xffff)));
            o[e] = (c + e < V) ? (__expf(x - lse) - ((c + e) == tgt ? 1.f : 0.f)) * w : 0.f;
        }
        *(uint4*)(row + c) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    }
}

// deterministic per-head means and the total (fm.py:600 / :635)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ row_loss, const int32_t* __restrict__ seg_start,
                                                            const int32_t* __restrict__ seg_count, const int32_t* __restrict__ vocab,
                                                            int n_heads, int loss_type, float* __restrict__ head_loss, float* __restrict__ total) {
    __shared__ float red[4];
    __shared__ float hl[FM_MAX_MODS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int g = 0; g < n_heads; ++g) {
        const int s0 = seg_start[g], n = seg_count[g];
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += row_loss[s0 + i];
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) hl[g] = n > 0 ? ((red[0] + red[1]) + (red[2] + red[3])) / (float)n : 0.f;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float tot = 0.f, wsum = 0.f;
        for (int g = 0; g < n_heads; ++g) {
            head_loss[g] = hl[g];
            if (loss_type == FM_LOSS_MOD) { tot += hl[g]; wsum += 1.f; }
            else { const float w = (float)seg_count[g] * (float)vocab[g]; tot += hl[g] * w; wsum += w; }
        }
        total[0] = tot / wsum;
    }
}

// cross-entropy over fp32 logits in the segmented (per-head) row layout of loss.hip: one workgroup per padded row.
// write_grad == 0: row_loss / row_lse;  write_grad == 1: logits <- d(loss)/d(logits) in place, scaled like ce_bwd_kernel.
__global__ __launch_bounds__(256) void ce_f32_kernel(float* logits, int ld, const int* perm, const int* tile_group, const long long* target_ids,
                                                     const int* vocab, const int* seg_count, const float* grad_scale, int loss_type, int n_heads,
                                                     float* row_loss, float* row_lse, int write_grad, int seg_rows) {
    __shared__ float red[4];
    const int pr = blockIdx.x;
    const int g = tile_group[pr / seg_rows];
    if (g < 0) return;
    const int src = perm[pr];
    const int V = vocab[g];
    float* row = logits + (size_t)pr * ld;
    if (src < 0) {                                          // pad row of a live segment
        if (!write_grad) { if (threadIdx.x == 0) { row_loss[pr] = 0.f; row_lse[pr] = 0.f; } }
        else for (int c = threadIdx.x; c < V; c += 256) row[c] = 0.f;
        return;
    }
    const int tgt = (int)target_ids[src];
    if (!write_grad) {
        float mx = -INFINITY;
        for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, row[c]);
        mx = wave_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 0.f;
        for (int c = threadIdx.x; c < V; c += 256) s += expf(row[c] - mx);
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        s = (red[0] + red[1]) + (red[2] + red[3]);
        const float lse = mx + logf(s);
        if (threadIdx.x == 0) { row_lse[pr] = lse; row_loss[pr] = lse - row[tgt]; }
        return;
    }
    // d(total)/d(logit) = scale_g * (softmax - onehot); 'mod': scale_g = 1 / (n_heads * count_g); 'token': V_g / sum_h count_h * V_h
    float coef;
    if (loss_type == FM_LOSS_MOD) coef = 1.0f / ((float)n_heads * (float)seg_count[g]);
    else {
        double tot = 0.0;
        for (int h = 0; h < n_heads; ++h) tot += (double)seg_count[h] * (double)vocab[h];
        coef = (float)((double)V / tot);
    }
    coef *= grad_scale ? grad_scale[0] : 1.0f;
    const float lse = row_lse[pr];
    for (int c = threadIdx.x; c < V; c += 256) row[c] = coef * (expf(row[c] - lse) - (c == tgt ? 1.0f : 0.f));
}


// padded[pr] = src[perm[pr]] (bf16 rows of width D), zero rows where perm < 0
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, int lds_, const int32_t* __restrict__ perm,
                                                          bf16_t* __restrict__ dst, int ldd, int Rp, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pr = blockIdx.x * 4 + wave; pr < Rp; pr += gridDim.x * 4) {
        const int s = perm[pr];
        for (int c = lane; c < D / 8; c += 64) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (s >= 0) v = *(const uint4*)(src + (size_t)s * lds_ + c * 8);
            *(uint4*)(dst + (size_t)pr * ldd + c * 8) = v;
        }
    }
}

}  // namespace

extern "C" int fm_segment_rows(const int32_t* head_of_row, int R, int n_heads, int32_t* seg_start, int32_t* seg_count,
                               int32_t* perm, int32_t* row_to_padded, int32_t* tile_group, int Rp, void* stream) {
    FM_CHECK_ARG(head_of_row && seg_start && seg_count && perm && row_to_padded && tile_group, "fm_segment_rows: null pointer");
    FM_CHECK_ARG(n_heads > 0 && n_heads <= FM_MAX_MODS && R > 0, "fm_segment_rows: bad shape");
    FM_CHECK_ARG(Rp % SEG_ALIGN == 0 && Rp >= (R + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN + (n_heads - 1) * SEG_ALIGN,
                 "fm_segment_rows: padded capacity Rp=%d too small for R=%d rows in %d segments", Rp, R, n_heads);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, row_to_padded, -1, R);
    hipLaunchKernelGGL(segment_rows_kernel, dim3(n_heads), dim3(1024), 0, (hipStream_t)stream, head_of_row, R, n_heads, seg_start,
                       seg_count, perm, row_to_padded, tile_group, Rp);
    FM_CHECK_LAUNCH("fm_segment_rows");
    return 0;
}

extern "C" int fm_gather_rows(const void* src, int ld_src, const int32_t* perm, void* dst, int ld_dst, int Rp, int D, void* stream) {
    FM_CHECK_ARG(src && perm && dst && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "fm_gather_rows: bad argument");
    int grid = (Rp + 3) / 4;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, perm, (bf16_t*)dst, ld_dst, Rp, D);
    FM_CHECK_LAUNCH("fm_gather_rows");
    return 0;
}

extern "C" int fm_cross_entropy(void* logits, int ldl, const int32_t* perm, const int32_t* tile_group, const int64_t* target_ids,
                                const int32_t* vocab, const int32_t* seg_start, const int32_t* seg_count, const void* grad_scale,
                                int loss_type, int n_heads, int Rp, int max_vocab, void* row_loss, void* row_lse, void* head_loss,
                                void* total_loss, int write_grad, void* stream) {
    FM_CHECK_ARG(logits && perm && tile_group && target_ids && vocab && seg_start && seg_count && row_loss && row_lse && head_loss && total_loss,
                 "fm_cross_entropy: null pointer");
    FM_CHECK_ARG(loss_type == FM_LOSS_MOD || loss_type == FM_LOSS_TOKEN, "fm_cross_entropy: invalid loss type %d", loss_type);
    FM_CHECK_ARG(ldl % 8 == 0 && ldl >= (max_vocab + 63) / 64 * 64, "fm_cross_entropy: ldl=%d too small for vocab %d", ldl, max_vocab);
    FM_CHECK_ARG((((uintptr_t)logits) & 15) == 0, "fm_cross_entropy: logits must be 16-byte aligned");
    if (!write_grad) {
        hipLaunchKernelGGL(ce_fwd_kernel, dim3(Rp), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)logits, ldl, perm, tile_group,
                           (const long long*)target_ids, vocab, (float*)row_loss, (float*)row_lse);
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss, seg_start, seg_count,
                           vocab, n_heads, loss_type, (float*)head_loss, (float*)total_loss);
    } else {
        hipLaunchKernelGGL(ce_bwd_kernel, dim3(Rp), dim3(256), 0, (hipStream_t)stream, (bf16_t*)logits, ldl, perm, tile_group,
                           (const long long*)target_ids, vocab, seg_count, (const float*)grad_scale, loss_type, n_heads, (const float*)row_lse);
    }
    FM_CHECK_LAUNCH("fm_cross_entropy");
    return 0;
}

/* fp32 verification path (csrc/fp32_verify.hip): the same segmented cross-entropy over fp32 logits, exact expf / logf */
extern "C" int fm_cross_entropy_f32(void* logits, int ldl, const int32_t* perm, const int32_t* tile_group, const int64_t* target_ids,
                                    const int32_t* vocab, const int32_t* seg_start, const int32_t* seg_count, const void* grad_scale,
                                    int loss_type, int n_heads, int Rp, int max_vocab, void* row_loss, void* row_lse, void* head_loss,
                                    void* total_loss, int write_grad, void* stream) {
    FM_CHECK_ARG(logits && perm && tile_group && target_ids && vocab && seg_start && seg_count && row_loss && row_lse && head_loss && total_loss,
                 "fm_cross_entropy_f32: null pointer");
    FM_CHECK_ARG(loss_type == FM_LOSS_MOD || loss_type == FM_LOSS_TOKEN, "fm_cross_entropy_f32: invalid loss type %d", loss_type);
    FM_CHECK_ARG(ldl >= max_vocab, "fm_cross_entropy_f32: ldl=%d too small for vocab %d", ldl, max_vocab);
    hipLaunchKernelGGL(ce_f32_kernel, dim3(Rp), dim3(256), 0, (hipStream_t)stream, (float*)logits, ldl, perm, tile_group,
                       (const long long*)target_ids, vocab, seg_count, (const float*)grad_scale, loss_type, n_heads, (float*)row_loss,
                       (float*)row_lse, write_grad, SEG_ALIGN);
    if (!write_grad)
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)row_loss, seg_start, seg_count,
                           vocab, n_heads, loss_type, (float*)head_loss, (float*)total_loss);
    FM_CHECK_LAUNCH("fm_cross_entropy_f32");
    return 0;
}
