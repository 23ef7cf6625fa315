// Fused "concatenate every modality -> keep the first N unmasked positions -> embed only those".
//
// Upstream (fourm/models/encoder_embeddings.py:87-121,184-211,280-309,387-421,
// decoder_embeddings.py:98-139,226-255, fm.py:245-438) embeds *all* O (resp. P) concatenated
// positions into fp32 (B,O,D) tensors, concatenates them, argsorts a float key and gathers N rows.
// Here one workgroup per sample builds the stable partition with integer prefix sums (the float
// argsort is exactly that: tests/test_oracle_golden.py::test_partition_equals_float_argsort) and
// writes only the kept rows.  HBM-bound integer/gather work: no MFMA.
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int MAXPOS = 8192;     // concatenated positions per sample held in LDS
constexpr int MAXKEEP = 1024;    // kept slots per sample

struct SelArgs {
    fm_select_desc d;
};

__device__ __forceinline__ long long load_id(const void* ids, int is64, size_t i) {
    return is64 ? ((const long long*)ids)[i] : (long long)((const int*)ids)[i];
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix and
// adds the block total to *total
__device__ __forceinline__ int block_scan_excl(int v, int* wave_tot /*[4] LDS*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int inc = wave_scan_incl(v);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(256) void select_embed_kernel(SelArgs A) {
    const fm_select_desc& d = A.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: rank_sel[P] (exclusive count of selectable positions before p), rank_pos[P] (exclusive count of
    // unmasked-in-own-mask positions, sequences in the decoder), flags[P], slot_p[keep]
    const int P = d.total_len;
    int* rank_sel = (int*)smem;
    int* rank_pos = rank_sel + P;
    int* slot_p = rank_pos + P;
    uint8_t* flag = (uint8_t*)(slot_p + d.n_keep);
    __shared__ int wave_tot[4];
    __shared__ int mod_off[FM_MAX_MODS + 1];
    __shared__ int cs_scan[MAXKEEP];

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < d.n_mods; ++m) { mod_off[m] = o; o += d.mods[m].L; }
        mod_off[d.n_mods] = o;
    }
    __syncthreads();

    auto mod_of = [&](int p) {
        int m = 0;
        while (m + 1 < d.n_mods && p >= mod_off[m + 1]) ++m;
        return m;
    };

    // ---- pass 1: selectable flags + ranks --------------------------------------------------------
    int run_sel = 0, run_pos = 0;
    for (int base = 0; base < P; base += 256) {
        const int p = base + tid;
        int sel = 0, um = 0;
        if (p < P) {
            const int m = mod_of(p);
            const fm_mod_desc& md = d.mods[m];
            const int j = p - mod_off[m];
            const uint8_t* mk = (const uint8_t*)md.mask + (size_t)b * md.mask_stride;
            if (d.is_decoder && md.shifted) {
                // teacher forcing (fm.py:309-319): position j pairs input j with target j+1
                um = mk[j] == 0;
                sel = (mk[j] == 0) && (mk[j + 1] == 0);
            } else {
                um = sel = mk[j] == 0;
            }
            flag[p] = (uint8_t)(sel | (um << 1));
        }
        int ts, tp;
        const int es = block_scan_excl(sel, wave_tot, &ts);
        const int ep = block_scan_excl(um, wave_tot, &tp);
        if (p < P) { rank_sel[p] = run_sel + es; rank_pos[p] = run_pos + ep; }
        run_sel += ts; run_pos += tp;
    }
    const int n_valid = run_sel;
    __syncthreads();
    // ---- pass 2: stable partition -> slot table ---------------------------------------------------
    for (int p = tid; p < P; p += 256) {
        // raw views (cat_*_tensors / a modality's own forward): every position, in place
        const int s = d.raw ? p : (flag[p] & 1) ? rank_sel[p] : n_valid + (p - rank_sel[p]);
        if (s < d.n_keep) slot_p[s] = p;
    }
    __syncthreads();

    // ---- decoder: cumsum of the compressed attention mask over the kept slots ----------------------
    if (d.is_decoder && !d.raw) {
        // n_keep <= MAXKEEP; serial-by-chunks scan with one wave
        if (wave == 0) {
            int run = 0;
            for (int base = 0; base < d.n_keep; base += 64) {
                const int s = base + lane;
                int v = 0;
                if (s < d.n_keep) {
                    const int p = slot_p[s];
                    const int m = mod_of(p);
                    const fm_mod_desc& md = d.mods[m];
                    v = ((const int*)md.dam)[(size_t)b * md.mask_stride + (p - mod_off[m])];
                }
                const int inc = wave_scan_incl(v);
                if (s < d.n_keep) cs_scan[s] = run + inc;
                run += __shfl(inc, 63, 64);
            }
        }
        __syncthreads();
    }

    // ---- pass 3: one wave per kept slot -------------------------------------------------------------
    const int D = d.dim, nch = D >> 2;
    const int Nt = d.n_reg + d.n_keep;
    // the row movement is split over gridDim.y workgroups per sample (each repeats the cheap index passes above):
    // one workgroup per sample leaves 3/4 of the chip's memory parallelism unused
    for (int s = wave + 4 * blockIdx.y; s < Nt; s += 4 * gridDim.y) {
        const size_t orow = (size_t)b * Nt + s;
        float* tok_o = (float*)d.tokens + orow * D;
        float* emb_o = (float*)d.emb + orow * D;
        float* x0_o = d.x0 ? (float*)d.x0 + orow * D : nullptr;
        if (s < d.n_reg) {   // register tokens: never masked, no embedding (fm.py:375-381)
            for (int c = lane; c < nch; c += 64) {
                const float4 t = *(const float4*)((const float*)d.reg_tokens + (size_t)s * D + c * 4);
                *(float4*)(tok_o + c * 4) = t;
                *(float4*)(emb_o + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x0_o) *(float4*)(x0_o + c * 4) = t;
            }
            if (lane == 0) {
                ((uint8_t*)d.out_mask)[orow] = 0;
                ((int16_t*)d.out_mod)[orow] = -1;
                ((int32_t*)d.slot_mod)[orow] = -2;     // -2 = register token
                ((int32_t*)d.slot_src)[orow] = s;
                ((int32_t*)d.slot_pos)[orow] = 0;
            }
            continue;
        }
        const int sk = s - d.n_reg;
        const int p = slot_p[sk];
        const int m = mod_of(p);
        const fm_mod_desc& md = d.mods[m];
        const int j = p - mod_off[m];
        const bool masked = (flag[p] & 1) == 0;
        const bool own_unmasked = (flag[p] & 2) != 0;     // in the modality's own input / target mask
        const bool raw = d.raw != 0;
        long long id = 0;
        int posrow = 0;
        if (md.kind == FM_KIND_TOK || md.kind == FM_KIND_SEQ) id = load_id(md.ids, md.ids_are_i64, (size_t)b * md.id_stride + j);
        if (md.kind == FM_KIND_SEQ || md.kind == FM_KIND_SEQ_EMB) {
            posrow = rank_pos[p] - rank_pos[mod_off[m]];     // = cumsum(~mask)[j] - 1 for an unmasked j
            if (d.is_decoder && posrow >= md.max_len) posrow = 0;
        } else {
            posrow = j;
        }
        if (lane == 0) {
            ((uint8_t*)d.out_mask)[orow] = masked ? 1 : 0;
            ((int16_t*)d.out_mod)[orow] = (masked && !raw) ? (int16_t)-1 : (int16_t)md.mod_id;
            ((int32_t*)d.slot_mod)[orow] = (masked && !raw) ? -1 : m;
            ((int32_t*)d.slot_src)[orow] = (md.kind == FM_KIND_TOK || md.kind == FM_KIND_SEQ) ? (int)id : j;
            ((int32_t*)d.slot_pos)[orow] = posrow;
            if (d.is_decoder) {
                long long tgt = md.shifted ? load_id(md.ids, md.ids_are_i64, (size_t)b * md.id_stride + j + 1) : id;
                ((long long*)d.target_ids)[orow] = (masked && !raw) ? 0 : tgt;
                ((int32_t*)d.out_cs)[orow] = raw ? 0 : cs_scan[sk];
                ((int16_t*)d.out_mod_pre)[orow] = (int16_t)md.mod_id;     // before pads lose their id (fm.py:431-432)
                ((int32_t*)d.out_mod_index)[orow] = masked ? -1 : md.head_index;
            }
        }
        // token row
        const float* trow = nullptr;
        if (!masked || raw) {
            // raw == 2 (a decoder embedding's own forward_embed): grid tokens embed their ids (decoder_embeddings.py:226-255);
            // in the concatenated decoder views they are queried with the mask token (fm.py:322)
            if (d.is_decoder && md.kind == FM_KIND_TOK && d.raw != 2) trow = (const float*)d.mask_token;
            else if (md.kind == FM_KIND_TOK || md.kind == FM_KIND_SEQ) trow = (const float*)md.table + (size_t)id * D;
            else if (md.kind == FM_KIND_SEQ_EMB) trow = (const float*)md.proj_bias;   // bias of emb_proj; GEMM adds the rest
        }
        const float* prow = (const float*)md.pos + (size_t)posrow * D;
        const float* mrow = (const float*)md.mod_emb;
        // raw views keep what the embedding modules produce before forward_mask_* zeroes anything: grids always carry
        // pos + mod; sequences carry mod alone where their own mask hides the position (encoder_embeddings.py:106-119)
        const bool seq_kind = md.kind == FM_KIND_SEQ || md.kind == FM_KIND_SEQ_EMB;
        const bool with_emb = raw ? true : !masked;
        const bool with_pos = raw ? (!seq_kind || own_unmasked) : true;
        for (int c = lane; c < nch; c += 64) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f), e = make_float4(0.f, 0.f, 0.f, 0.f);
            if (trow) t = *(const float4*)(trow + c * 4);
            if (with_emb) {
                const float4 mm = *(const float4*)(mrow + c * 4);
                float4 pp = make_float4(0.f, 0.f, 0.f, 0.f);
                if (with_pos) pp = *(const float4*)(prow + c * 4);
                e = make_float4(pp.x + mm.x, pp.y + mm.y, pp.z + mm.z, pp.w + mm.w);
            }
            *(float4*)(tok_o + c * 4) = t;
            *(float4*)(emb_o + c * 4) = e;
            if (x0_o) *(float4*)(x0_o + c * 4) = make_float4(t.x + e.x, t.y + e.y, t.z + e.z, t.w + e.w);
        }
        // dense side inputs gathered for the projection GEMM (zero rows elsewhere)
        if (d.patch_rows && d.rows_f32) {                // fp32 verification path: the same rows, unrounded
            float* pr = (float*)d.patch_rows + orow * d.patch_ld;
            const bool here = (!masked || raw) && md.kind == FM_KIND_PATCH;
            const int ps = md.patch, C = md.channels, gw = md.grid_w;
            const float* img = (const float*)md.ids + (size_t)b * md.id_stride;
            for (int f = lane; f < d.patch_ld; f += 64) {
                float v = 0.f;
                if (here && f < ps * ps * C) {
                    const int gy = j / gw, gx = j % gw, Wd = gw * ps, Hd = (md.L / gw) * ps;
                    const int c = f % C, px = (f / C) % ps, py = f / (C * ps);
                    v = img[((size_t)c * Hd + gy * ps + py) * Wd + gx * ps + px];
                }
                pr[f] = v;
            }
        } else if (d.patch_rows) {
            bf16_t* pr = (bf16_t*)d.patch_rows + orow * d.patch_ld;
            if ((!masked || raw) && md.kind == FM_KIND_PATCH) {
                // feature f = (py*ps + px)*C + c  <-  img[b][c][gy*ps + py][gx*ps + px]   (encoder_embeddings.py:301)
                const int ps = md.patch, C = md.channels, gw = md.grid_w;
                const int gy = j / gw, gx = j % gw;
                const float* img = (const float*)md.ids + (size_t)b * md.id_stride;
                const int Wd = gw * ps, Hd = (md.L / gw) * ps;
                for (int f = lane; f < d.patch_ld; f += 64) {
                    float v = 0.f;
                    if (f < ps * ps * C) {
                        const int c = f % C, px = (f / C) % ps, py = f / (C * ps);
                        v = img[((size_t)c * Hd + gy * ps + py) * Wd + gx * ps + px];
                    }
                    pr[f] = f2bf(v);
                }
            } else {
                for (int f = lane; f < d.patch_ld / 4; f += 64) *(uint2*)(pr + f * 4) = make_uint2(0u, 0u);
            }
        }
        if (d.seqemb_rows && d.rows_f32) {
            float* sr = (float*)d.seqemb_rows + orow * d.seqemb_ld;
            const bool here = (!masked || raw) && md.kind == FM_KIND_SEQ_EMB;
            const float* src = (const float*)md.ids + (size_t)b * md.id_stride + (size_t)j * md.orig_dim;
            for (int f = lane; f < d.seqemb_ld; f += 64) sr[f] = (here && f < md.orig_dim) ? src[f] : 0.f;
        } else if (d.seqemb_rows) {
            bf16_t* sr = (bf16_t*)d.seqemb_rows + orow * d.seqemb_ld;
            if ((!masked || raw) && md.kind == FM_KIND_SEQ_EMB) {
                const float* src = (const float*)md.ids + (size_t)b * md.id_stride + (size_t)j * md.orig_dim;
                for (int f = lane; f < d.seqemb_ld; f += 64) sr[f] = f2bf(f < md.orig_dim ? src[f] : 0.f);
            } else {
                for (int f = lane; f < d.seqemb_ld / 4; f += 64) *(uint2*)(sr + f * 4) = make_uint2(0u, 0u);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward of the embedding stage: scatter d(tokens+emb) into the tables.  Column sums per
// modality (mod_emb, mask token) are first reduced in LDS, one global atomic per column per sample.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(fm_embed_bwd_desc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = (float*)smem;                 // [(n_mods + 1)][D] : mod_emb sums, last row = mask token
    const int D = d.dim, nch = D >> 2;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rows = d.n_mods + 1;
    for (int i = threadIdx.x; i < rows * D; i += 256) acc[i] = 0.f;
    __shared__ int used[FM_MAX_MODS + 1];
    if (threadIdx.x <= FM_MAX_MODS) used[threadIdx.x] = 0;
    __syncthreads();
    // Kept slots keep the concatenation order, so a sample's slots are grouped by modality: every wave walks a CONTIGUOUS run of slots and
    // carries the column sums of the current modality in registers (round 4; one LDS atomic per column and slot before: ds_add_f32
    // serialises - 250 us per launch at 0.4 TB/s); they reach LDS once per (wave, modality run).
    constexpr int MAXQ = 8;                                      // float4 chunks per lane: D <= 2048
    const int parts = 4 * gridDim.y, part = wave + 4 * blockIdx.y;
    const int per = (d.Nt + parts - 1) / parts, s_lo = part * per, s_hi = min(d.Nt, s_lo + per);
    float4 racc[MAXQ];
    int cur = -1;
    bool cur_mask = false;
    auto flush = [&]() {
        if (cur < 0) return;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int c = lane + 64 * q;
            if (c >= nch) break;
            float* a0 = acc + (size_t)cur * D + c * 4;
            atomicAdd(a0, racc[q].x); atomicAdd(a0 + 1, racc[q].y); atomicAdd(a0 + 2, racc[q].z); atomicAdd(a0 + 3, racc[q].w);
            if (cur_mask) {
                float* a1 = acc + (size_t)d.n_mods * D + c * 4;
                atomicAdd(a1, racc[q].x); atomicAdd(a1 + 1, racc[q].y); atomicAdd(a1 + 2, racc[q].z); atomicAdd(a1 + 3, racc[q].w);
            }
        }
    };
    for (int s = s_lo; s < s_hi; ++s) {
        const size_t row = (size_t)b * d.Nt + s;
        const int m = ((const int32_t*)d.slot_mod)[row];
        if (m == -1) continue;                                   // masked slot: both inputs were zeroed
        const float* g = (const float*)d.dx + row * d.lddx;
        if (m == -2) {                                           // register token
            if (d.d_reg_tokens)
                for (int c = lane; c < D; c += 64) unsafeAtomicAdd((float*)d.d_reg_tokens + (size_t)s * D + c, g[c]);
            continue;
        }
        const fm_embed_bwd_mod& md = d.mods[m];
        const int src = ((const int32_t*)d.slot_src)[row];
        const int posrow = ((const int32_t*)d.slot_pos)[row];
        const bool to_mask_token = d.is_decoder && md.kind == FM_KIND_TOK;
        if (m != cur) {
            flush();
            cur = m; cur_mask = to_mask_token;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) racc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane == 0) { used[m] = 1; if (to_mask_token) used[d.n_mods] = 1; }
        }
        float* trow = nullptr;
        if (to_mask_token) {}
        else if ((md.kind == FM_KIND_TOK || md.kind == FM_KIND_SEQ) && md.d_table && !(md.has_padding_idx && src == md.padding_idx))
            trow = (float*)md.d_table + (size_t)src * D;
        else if (md.kind == FM_KIND_SEQ_EMB && md.d_proj_bias) trow = (float*)md.d_proj_bias;
        float* prow = md.d_pos ? (float*)md.d_pos + (size_t)posrow * D : nullptr;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int c = lane + 64 * q;
            if (c >= nch) break;
            const float4 v = *(const float4*)(g + c * 4);
            racc[q].x += v.x; racc[q].y += v.y; racc[q].z += v.z; racc[q].w += v.w;
            if (trow) {
                unsafeAtomicAdd(trow + c * 4, v.x); unsafeAtomicAdd(trow + c * 4 + 1, v.y);
                unsafeAtomicAdd(trow + c * 4 + 2, v.z); unsafeAtomicAdd(trow + c * 4 + 3, v.w);
            }
            if (prow) {
                unsafeAtomicAdd(prow + c * 4, v.x); unsafeAtomicAdd(prow + c * 4 + 1, v.y);
                unsafeAtomicAdd(prow + c * 4 + 2, v.z); unsafeAtomicAdd(prow + c * 4 + 3, v.w);
            }
        }
    }
    flush();
    __syncthreads();
    for (int m = 0; m < d.n_mods; ++m) {
        if (!used[m] || !d.mods[m].d_mod_emb) continue;
        for (int c = threadIdx.x; c < D; c += 256) unsafeAtomicAdd((float*)d.mods[m].d_mod_emb + c, acc[(size_t)m * D + c]);
    }
    if (used[d.n_mods] && d.d_mask_token)
        for (int c = threadIdx.x; c < D; c += 256) unsafeAtomicAdd((float*)d.d_mask_token + c, acc[(size_t)d.n_mods * D + c]);
}

// (B, M) compressed decoder mask -> dense (B, M, M) bool, for callers of the upstream sub-API
// (FourM.adapt_decoder_attention_mask, fm.py:440-475)
__global__ void dense_decoder_mask_kernel(const int32_t* cs, const int16_t* mod, uint8_t* out, int B, int M, int causal, int use_cs, int use_sep) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * M * M) return;
    const int k = i % M, q = (i / M) % M, b = i / ((size_t)M * M);
    bool blk = false;
    if (causal) blk = k > q;
    else if (use_cs) blk = k >= cs[(size_t)b * M + q];
    if (use_sep) blk = blk || (mod[(size_t)b * M + q] != mod[(size_t)b * M + k]);
    out[i] = blk ? 1 : 0;
}

}  // namespace

extern "C" int fm_select_embed(const fm_select_desc* d, void* stream) {
    FM_CHECK_ARG(d && d->tokens && d->emb && d->out_mask && d->out_mod && d->slot_mod && d->slot_src && d->slot_pos, "fm_select_embed: null output");
    FM_CHECK_ARG(d->n_mods > 0 && d->n_mods <= FM_MAX_MODS, "fm_select_embed: n_mods=%d out of range (max %d)", d->n_mods, FM_MAX_MODS);
    FM_CHECK_ARG(d->dim > 0 && d->dim % 4 == 0, "fm_select_embed: dim must be a multiple of 4");
    FM_CHECK_ARG(d->batch > 0 && d->n_keep > 0 && d->n_keep <= (d->raw ? MAXPOS : MAXKEEP), "fm_select_embed: n_keep=%d out of range (max %d)", d->n_keep,
                 d->raw ? MAXPOS : MAXKEEP);
    int total = 0;
    for (int m = 0; m < d->n_mods; ++m) {
        const fm_mod_desc& md = d->mods[m];
        FM_CHECK_ARG(md.L > 0 && md.mask && md.pos && md.mod_emb, "fm_select_embed: modality %d incomplete", m);
        FM_CHECK_ARG(md.kind != FM_KIND_PATCH || (d->patch_rows && md.ids), "fm_select_embed: pixel modality needs patch_rows and pixels");
        FM_CHECK_ARG(md.kind != FM_KIND_SEQ_EMB || (d->seqemb_rows && md.ids), "fm_select_embed: seq_emb modality needs seqemb_rows");
        FM_CHECK_ARG(!((md.kind == FM_KIND_TOK || md.kind == FM_KIND_SEQ) && !(d->is_decoder && md.kind == FM_KIND_TOK)) || md.table, "fm_select_embed: modality %d needs a table", m);
        FM_CHECK_ARG(!d->is_decoder || md.dam || d->raw, "fm_select_embed: decoder modality %d needs decoder_attention_mask", m);
        FM_CHECK_ARG(!(d->raw == 2 && md.kind == FM_KIND_TOK) || md.table, "fm_select_embed: modality %d needs a table", m);
        total += md.L;
    }
    FM_CHECK_ARG(total == d->total_len, "fm_select_embed: total_len=%d but the modalities sum to %d", d->total_len, total);
    FM_CHECK_ARG(total <= MAXPOS, "fm_select_embed: %d concatenated positions exceed %d", total, MAXPOS);
    FM_CHECK_ARG(d->n_keep <= total, "fm_select_embed: n_keep=%d exceeds the %d available positions", d->n_keep, total);
    FM_CHECK_ARG(d->raw >= 0 && d->raw <= 2 && (!d->raw || (d->n_keep == total && d->n_reg == 0)),
                 "fm_select_embed: raw views cover every position (n_keep == total_len) and carry no register tokens");
    FM_CHECK_ARG(!d->is_decoder || (d->target_ids && d->out_cs && d->out_mod_pre && d->out_mod_index && d->mask_token), "fm_select_embed: decoder outputs missing");
    FM_CHECK_ARG(d->n_reg == 0 || d->reg_tokens, "fm_select_embed: register tokens missing");
    FM_CHECK_ARG(d->patch_ld % 4 == 0 && d->seqemb_ld % 4 == 0, "fm_select_embed: side buffers need ld %% 4 == 0");
    const size_t lds = (size_t)total * 9 + (size_t)d->n_keep * 4 + 16;
    static bool once = (hipFuncSetAttribute((const void*)select_embed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAXPOS * 9 + MAXPOS * 4 + 16) == hipSuccess);
    (void)once;
    SelArgs A; A.d = *d;
    const int slices = d->batch >= 2048 ? 1 : d->batch >= 1024 ? 2 : 4;
    hipLaunchKernelGGL(select_embed_kernel, dim3(d->batch, slices), dim3(256), lds, (hipStream_t)stream, A);
    FM_CHECK_LAUNCH("fm_select_embed");
    return 0;
}

extern "C" int fm_embed_bwd(const fm_embed_bwd_desc* d, void* stream) {
    FM_CHECK_ARG(d && d->dx && d->slot_mod && d->slot_src && d->slot_pos, "fm_embed_bwd: null pointer");
    FM_CHECK_ARG(d->n_mods > 0 && d->n_mods <= FM_MAX_MODS && d->dim % 4 == 0 && d->dim <= 2048 && d->lddx % 4 == 0, "fm_embed_bwd: bad shape (dim <= 2048)");
    const size_t lds = (size_t)(d->n_mods + 1) * d->dim * sizeof(float);
    FM_CHECK_ARG(lds <= 150 * 1024, "fm_embed_bwd: %zu bytes of LDS needed", lds);
    static bool once = (hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess);
    (void)once;
    const int slices = d->batch >= 2048 ? 1 : d->batch >= 1024 ? 2 : 4;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(d->batch, slices), dim3(256), lds, (hipStream_t)stream, *d);
    FM_CHECK_LAUNCH("fm_embed_bwd");
    return 0;
}

extern "C" int fm_dense_decoder_mask(const int32_t* cs, const int16_t* mod, void* out, int B, int M, int causal, int use_cs, int use_sep, void* stream) {
    FM_CHECK_ARG(out && (causal || !use_cs || cs) && (!use_sep || mod), "fm_dense_decoder_mask: null pointer");
    const size_t n = (size_t)B * M * M;
    hipLaunchKernelGGL(dense_decoder_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, cs, mod, (uint8_t*)out, B, M, causal, use_cs, use_sep);
    FM_CHECK_LAUNCH("fm_dense_decoder_mask");
    return 0;
}
