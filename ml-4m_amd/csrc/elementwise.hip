// HBM-bound element-wise / reduction kernels around the GEMMs: activation backward, bf16 weight
// shadows (plain and transposed, zero padded to the GEMM contracts), bias gradients, AdamW and the
// gradient norm.  All 16-byte vectorised, grid-stride, one pass over memory.
#include "common.h"
#include "fourm_hip.h"

namespace {

__device__ __forceinline__ float4 ld_bf4(const bf16_t* p) {
    const uint2 v = *(const uint2*)p;
    return make_float4(bf2f((bf16_t)(v.x & 0xffff)), bf2f((bf16_t)(v.x >> 16)), bf2f((bf16_t)(v.y & 0xffff)), bf2f((bf16_t)(v.y >> 16)));
}
__device__ __forceinline__ void st_bf4(bf16_t* p, float a, float b, float c, float d) {
    *(uint2*)p = make_uint2(pack2bf(a, b), pack2bf(c, d));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }

// a = silu(g) * u  ->  dg = da * u * silu'(g),  du = da * silu(g)        (GatedMlp, fm_utils.py:142-144)
// gu / dgu: (R, 2*Hp) with g | u halves;  da: (R, Hp).  Pad columns (>= H) are written as zero.
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ da, int ldda, const bf16_t* __restrict__ gu, int ldgu,
                                                         bf16_t* __restrict__ dgu, int lddgu, int R, int H, int Hp, int nt) {
    // 8 features per thread, all six 16-byte loads issued before the arithmetic (a pure HBM stream)
    const int cpr = Hp / 8;
    const size_t total = (size_t)R * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = i / cpr, c = (i % cpr) * 8;
        typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
        uint4 dp, gp, up;
        if (nt & 2) { const u4_t t = __builtin_nontemporal_load((const u4_t*)(da + (size_t)r * ldda + c)); dp = make_uint4(t[0], t[1], t[2], t[3]); }
        else dp = *(const uint4*)(da + (size_t)r * ldda + c);
        if (nt & 1) {          // the saved (g | u) are read exactly once, here
            const u4_t t = __builtin_nontemporal_load((const u4_t*)(gu + (size_t)r * ldgu + c)), v = __builtin_nontemporal_load((const u4_t*)(gu + (size_t)r * ldgu + Hp + c));
            gp = make_uint4(t[0], t[1], t[2], t[3]); up = make_uint4(v[0], v[1], v[2], v[3]);
        } else { gp = *(const uint4*)(gu + (size_t)r * ldgu + c); up = *(const uint4*)(gu + (size_t)r * ldgu + Hp + c); }
        const uint32_t dw[4] = {dp.x, dp.y, dp.z, dp.w}, gw[4] = {gp.x, gp.y, gp.z, gp.w}, uw[4] = {up.x, up.y, up.z, up.w};
        uint32_t og[4], ou[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dg[2], du[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float dv = bf2f((bf16_t)(e ? dw[q] >> 16 : dw[q] & 0xffff));
                const float gv = bf2f((bf16_t)(e ? gw[q] >> 16 : gw[q] & 0xffff));
                const float uv = bf2f((bf16_t)(e ? uw[q] >> 16 : uw[q] & 0xffff));
                const float sg = sigmoid_f(gv);
                const float sl = bfround(gv * sg);
                const float ds = bfround(dv * uv);
                const bool live = c + 2 * q + e < H;
                du[e] = live ? dv * sl : 0.f;
                dg[e] = live ? ds * (sg * (1.0f + gv * (1.0f - sg))) : 0.f;
            }
            og[q] = pack2bf(dg[0], dg[1]); ou[q] = pack2bf(du[0], du[1]);
        }
        *(uint4*)(dgu + (size_t)r * lddgu + c) = make_uint4(og[0], og[1], og[2], og[3]);
        *(uint4*)(dgu + (size_t)r * lddgu + Hp + c) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
    }
}

// h = gelu(pre)  ->  dpre = dh * gelu'(pre)        (Mlp, fm_utils.py:121-126; exact erf GELU)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dh, int lddh, const bf16_t* __restrict__ pre, int ldp,
                                                       bf16_t* __restrict__ dpre, int lddp, int R, int H, int Hp) {
    const int cpr = Hp / 4;
    const size_t total = (size_t)R * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = i / cpr, c = (i % cpr) * 4;
        const float4 d = ld_bf4(dh + (size_t)r * lddh + c), x = ld_bf4(pre + (size_t)r * ldp + c);
        const float dv[4] = {d.x, d.y, d.z, d.w}, xv[4] = {x.x, x.y, x.z, x.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float cdf = 0.5f * (1.0f + erf_fast(xv[e] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * xv[e] * xv[e]);
            o[e] = (c + e < H) ? dv[e] * (cdf + xv[e] * pdf) : 0.f;
        }
        st_bf4(dpre + (size_t)r * lddp + c, o[0], o[1], o[2], o[3]);
    }
}

// dst (rows, ldd) bf16 <- src (rows, cols) f32, columns [cols, ldd) zero
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ dst, int ldd, int rows, int cols) {
    const int cpr = ldd / 4;
    const size_t total = (size_t)rows * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = i / cpr, c = (i % cpr) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c + e < cols) ? src[(size_t)r * lds_ + c + e] : 0.f;
        st_bf4(dst + (size_t)r * ldd + c, v[0], v[1], v[2], v[3]);
    }
}

// dst (cols, dst_cols >= rows; row stride ldd) bf16 <- transpose of src (rows, cols) f32; dst columns [rows, dst_cols) zero.
// 64x64 tiles through LDS: coalesced on both sides.
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, int lds_, bf16_t* __restrict__ dst, int ldd, int dst_cols, int rows, int cols) {
    __shared__ float tile[64][65];
    const int tr0 = blockIdx.y * 64, tc0 = blockIdx.x * 64;     // tile origin in src (row, col)
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i / 64, c = i % 64;
        const int gr = tr0 + r, gc = tc0 + c;
        tile[r][c] = (gr < rows && gc < cols) ? src[(size_t)gr * lds_ + gc] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i / 64, r = i % 64;                     // dst row = src col
        const int gc = tc0 + c, gr = tr0 + r;
        if (gc < cols && gr < dst_cols) dst[(size_t)gc * ldd + gr] = f2bf(tile[r][c]);
    }
}

// Multi-tensor refresh of the bf16 weight shadows: one launch walks a device table of (fp32 master -> bf16
// shadow) jobs in 64x64 tiles; a job either casts in place or casts + transposes through LDS.  Pad rows /
// columns of the destinations are never written (they are zero from allocation).
constexpr int SHADOW_TILES_PER_BLOCK = 8;   // consecutive tiles per workgroup: one table search amortised over 128 KB of work

__global__ __launch_bounds__(256) void shadow_refresh_kernel(const fm_shadow_desc* __restrict__ descs, int n, int total_tiles) {
    __shared__ float tile[64][65];
    const int t_first = blockIdx.x * SHADOW_TILES_PER_BLOCK;
    int lo = 0, hi = n - 1;                                   // last job with tile_start <= t_first
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_start <= t_first) lo = mid; else hi = mid - 1;
    }
    fm_shadow_desc d = descs[lo];
    int next_start = lo + 1 < n ? descs[lo + 1].tile_start : 0x7fffffff;
    const int q = (threadIdx.x & 15) * 4, p = threadIdx.x >> 4;
    for (int tt = t_first; tt < min(total_tiles, t_first + SHADOW_TILES_PER_BLOCK); ++tt) {
        if (tt >= next_start) {                               // walked into the next job
            ++lo;
            d = descs[lo];
            next_start = lo + 1 < n ? descs[lo + 1].tile_start : 0x7fffffff;
        }
        const int t = tt - d.tile_start;
        const int tiles_c = (d.cols + 63) / 64;
        const int tr0 = (t / tiles_c) * 64, tc0 = (t % tiles_c) * 64;
        const float* src = (const float*)d.src;
        bf16_t* dst = (bf16_t*)d.dst;
        const bool vec_src = ((((uintptr_t)src) & 15) == 0) && (d.ld_src % 4 == 0);
        const bool vec_dst = ((((uintptr_t)dst) & 7) == 0) && (d.ld_dst % 4 == 0);
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                         // all 4 loads of this tile in flight together
            const int gr = tr0 + p + 16 * i, gc = tc0 + q;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
            if (gr < d.rows) {
                const float* sp = src + (size_t)gr * d.ld_src + gc;
                if (vec_src && gc + 3 < d.cols) {
                    const float4 f = *(const float4*)sp;
                    v[i][0] = f.x; v[i][1] = f.y; v[i][2] = f.z; v[i][3] = f.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gc + e < d.cols) v[i][e] = sp[e];
                }
            }
        }
        if (d.col_scale) {                                    // folded LayerNorm weight: src[r][c] * scale[c] (context-norm hoist)
            const float* cs = (const float*)d.col_scale;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sc = tc0 + q + e < d.cols ? cs[tc0 + q + e] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i][e] *= sc;
            }
        }
        if (d.dst_f32) {                                      // fp32 verification mode: same walk, fp32 destination, scalar stores
            float* df = (float*)d.dst;
            if (!d.transpose) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gr = tr0 + p + 16 * i, gc = tc0 + q;
                    if (gr >= d.rows) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gc + e < d.cols) df[(size_t)gr * d.ld_dst + gc + e] = v[i][e];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gr = tr0 + p + 16 * i, gc = tc0 + q;
                    if (gr >= d.rows) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gc + e < d.cols) df[(size_t)(gc + e) * d.ld_dst + gr] = v[i][e];
                }
            }
            continue;
        }
        if (!d.transpose) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gr = tr0 + p + 16 * i, gc = tc0 + q;
                if (gr >= d.rows) continue;
                bf16_t* dp = dst + (size_t)gr * d.ld_dst + gc;
                if (vec_dst && gc + 3 < d.cols) *(uint2*)dp = make_uint2(pack2bf(v[i][0], v[i][1]), pack2bf(v[i][2], v[i][3]));
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (gc + e < d.cols) dp[e] = f2bf(v[i][e]);
            }
            continue;
        }
        __syncthreads();                                       // previous tile's readers are done with `tile`
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[p + 16 * i][q + e] = v[i][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = p + 16 * i, gc = tc0 + c, gr = tr0 + q;          // dst row = src column
            if (gc >= d.cols) continue;
            bf16_t* dp = dst + (size_t)gc * d.ld_dst + gr;
            if (vec_dst && gr + 3 < d.rows) *(uint2*)dp = make_uint2(pack2bf(tile[q][c], tile[q + 1][c]), pack2bf(tile[q + 2][c], tile[q + 3][c]));
            else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gr + e < d.rows) dp[e] = f2bf(tile[q + e][c]);
        }
    }
}

// db[n] += sum_r dY[r][n].  A block covers 256 columns x rows_per_block rows as 32 column groups (8 columns = one 16-byte load) x 8 row
// lanes: a wave reads two 512-byte row pieces per load; the 8 row lanes are summed through LDS, one atomic per column and block.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dy, int ldy, float* __restrict__ db, int R, int N, int rows_per_block) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int n0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n0 < N) {
        const bool wide = n0 + 8 <= N && (ldy & 7) == 0 && (((uintptr_t)dy) & 15) == 0;
        for (int r = r0 + rl; r < r1; r += 8) {
            const bf16_t* row = dy + (size_t)r * ldy + n0;
            if (wide) {
                const uint4 v = *(const uint4*)row;
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[2 * e] += bf2f((bf16_t)(w[e] & 0xffff)); s[2 * e + 1] += bf2f((bf16_t)(w[e] >> 16)); }
            } else {
                for (int e = 0; e < 8; ++e) if (n0 + e < N) s[e] += bf2f(row[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][cg * 8 + e] = s[e];
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        unsafeAtomicAdd(db + n, t);
    }
}

// torch.optim.AdamW (decoupled weight decay), one launch per contiguous run of one parameter group.
// grad_mult (device scalar, optional) = gradient clipping coefficient.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    size_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ grad_mult, const float* __restrict__ hyper, float* __restrict__ sumsq) {
    if (hyper) { lr = hyper[0]; wd = hyper[1]; bc1 = hyper[2]; bc2_sqrt = hyper[3]; }      // captured launches: values live on the device
    const float gm = grad_mult ? grad_mult[0] : 1.0f;
    const float step = lr / bc1;
    float ss = 0.f;                                   // sum of the RAW gradients' squares (sumsq != NULL: the norm rides on this pass)
    auto update = [&](float& pe, float& me, float& ve, float ge) {
        ss += ge * ge;
        ge *= gm;
        pe *= 1.0f - lr * wd;
        me = beta1 * me + (1.0f - beta1) * ge;
        ve = beta2 * ve + (1.0f - beta2) * ge * ge;
        pe -= step * me / (sqrtf(ve) / bc2_sqrt + eps);
    };
    auto quad = [&](size_t o, float4 P, float4 M, float4 V, float4 G) {
        update(P.x, M.x, V.x, G.x); update(P.y, M.y, V.y, G.y); update(P.z, M.z, V.z, G.z); update(P.w, M.w, V.w, G.w);
        *(float4*)(p + o) = P; *(float4*)(m + o) = M; *(float4*)(v + o) = V;
    };
    auto slow = [&](size_t o) {                       // a position near the end: one full quad, or the last < 4 elements
        if (o + 4 <= n) quad(o, *(float4*)(p + o), *(float4*)(m + o), *(float4*)(v + o), *(const float4*)(g + o));
        else for (size_t j = o; j < n; ++j) update(p[j], m[j], v[j], g[j]);
    };
    // two 16-byte quads per thread and iteration, all eight loads issued before the arithmetic (a 28 B/param HBM stream)
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += 2 * stride) {
        const size_t i2 = i + stride;
        if (i2 + 4 <= n) {
            const float4 P0 = *(float4*)(p + i), P1 = *(float4*)(p + i2), M0 = *(float4*)(m + i), M1 = *(float4*)(m + i2);
            const float4 V0 = *(float4*)(v + i), V1 = *(float4*)(v + i2), G0 = *(const float4*)(g + i), G1 = *(const float4*)(g + i2);
            quad(i, P0, M0, V0, G0);
            quad(i2, P1, M1, V1, G1);
        } else {
            slow(i);
            if (i2 < n) slow(i2);
        }
    }
    if (sumsq) {
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(sumsq, ss);
    }
}

// AdamW on weight matrices + their plain bf16 shadow in one STREAMING pass: see fm_adamw_shadow in the header.  A "tile" is a run of
// ADAMW_CHUNK consecutive elements of one matrix (linear in memory: full-width HBM bursts; the 64x64 tile walk of the transposing
// refresh reaches 0.8 TB/s, this 5 TB/s), the bf16 copy goes to dst_plain[r * ld_plain + c].  Transposed shadows stay with
// fm_shadow_refresh.  The update is adamw_kernel's, term by term.
constexpr int ADAMW_CHUNK = 8192;           // elements per tile: 256 threads x 8 quads

__global__ __launch_bounds__(256) void adamw_shadow_kernel(const fm_adamw_job* __restrict__ jobs, int n, int total_tiles, float lr, float beta1,
                                                           float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                           const float* __restrict__ grad_mult, const float* __restrict__ hyper, float* __restrict__ sumsq) {
    if (hyper) { lr = hyper[0]; wd = hyper[1]; bc1 = hyper[2]; bc2_sqrt = hyper[3]; }
    const float gm = grad_mult ? grad_mult[0] : 1.0f;
    const float step = lr / bc1;
    float ss = 0.f;
    auto update = [&](float& pe, float& me, float& ve, float ge) {
        ss += ge * ge;
        ge *= gm;
        pe *= 1.0f - lr * wd;
        me = beta1 * me + (1.0f - beta1) * ge;
        ve = beta2 * ve + (1.0f - beta2) * ge * ge;
        pe -= step * me / (sqrtf(ve) / bc2_sqrt + eps);
    };
    for (int tt = blockIdx.x; tt < total_tiles; tt += gridDim.x) {
        int lo = 0, hi = n - 1;                               // last job with tile_start <= tt
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].tile_start <= tt) lo = mid; else hi = mid - 1;
        }
        const fm_adamw_job d = jobs[lo];
        const size_t total = (size_t)d.rows * d.cols;
        const size_t base = (size_t)(tt - d.tile_start) * ADAMW_CHUNK;
        float* P = (float*)d.p; const float* G = (const float*)d.g; float* M = (float*)d.m; float* V = (float*)d.v;
        bf16_t* dst = (bf16_t*)d.dst_plain;
        const bool vec = (d.cols % 4 == 0) && ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
        const bool vec_dst = dst && ((((uintptr_t)dst) & 7) == 0) && (d.ld_plain % 4 == 0);
        if (vec) {
            // two sweeps of 4 quads per thread: all 16 loads of a sweep in flight before the arithmetic
#pragma unroll
            for (int sweep = 0; sweep < 2; ++sweep) {
                float4 Pq[4], Gq[4], Mq[4], Vq[4];
                size_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = base + ((size_t)(sweep * 4 + i) * 256 + threadIdx.x) * 4;
                    if (o[i] < total) { Pq[i] = *(const float4*)(P + o[i]); Gq[i] = *(const float4*)(G + o[i]); Mq[i] = *(const float4*)(M + o[i]); Vq[i] = *(const float4*)(V + o[i]); }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (o[i] >= total) continue;
                    update(Pq[i].x, Mq[i].x, Vq[i].x, Gq[i].x); update(Pq[i].y, Mq[i].y, Vq[i].y, Gq[i].y);
                    update(Pq[i].z, Mq[i].z, Vq[i].z, Gq[i].z); update(Pq[i].w, Mq[i].w, Vq[i].w, Gq[i].w);
                    *(float4*)(P + o[i]) = Pq[i]; *(float4*)(M + o[i]) = Mq[i]; *(float4*)(V + o[i]) = Vq[i];
                    if (dst) {
                        const size_t r = o[i] / d.cols, c = o[i] % d.cols;       // a quad never straddles rows (cols % 4 == 0)
                        bf16_t* dp = dst + r * d.ld_plain + c;
                        if (vec_dst) *(uint2*)dp = make_uint2(pack2bf(Pq[i].x, Pq[i].y), pack2bf(Pq[i].z, Pq[i].w));
                        else { dp[0] = f2bf(Pq[i].x); dp[1] = f2bf(Pq[i].y); dp[2] = f2bf(Pq[i].z); dp[3] = f2bf(Pq[i].w); }
                    }
                }
            }
        } else {
            for (size_t o = base + threadIdx.x; o < min(total, base + ADAMW_CHUNK); o += 256) {
                update(P[o], M[o], V[o], G[o]);
                if (dst) dst[(o / d.cols) * d.ld_plain + o % d.cols] = f2bf(P[o]);
            }
        }
    }
    if (sumsq) {
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(sumsq, ss);
    }
}

// out[0] += sum x^2   (fp32 partials per workgroup, one atomic each)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
        if (i + 4 <= n) {
            const float4 v = *(const float4*)(x + i);
            s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
            for (size_t j = i; j < n; ++j) s += x[j] * x[j];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
}

// norm = sqrt(sumsq); coef = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float* norm_out, float* coef_out) {
    const float nrm = sqrtf(sumsq[0]);
    norm_out[0] = nrm;
    if (coef_out) coef_out[0] = max_norm > 0.f ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
        if (i + 4 <= n) {
            const float4 v = *(const float4*)(src + i);
            st_bf4(dst + i, v.x, v.y, v.z, v.w);
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
        }
    }
}

__global__ __launch_bounds__(256) void bf16_to_f32_scaled_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n, float scale) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
        if (i + 4 <= n) {
            const uint2 p = *(const uint2*)(src + i);
            *(float4*)(dst + i) = make_float4(scale * bf2f((bf16_t)(p.x & 0xffff)), scale * bf2f((bf16_t)(p.x >> 16)),
                                              scale * bf2f((bf16_t)(p.y & 0xffff)), scale * bf2f((bf16_t)(p.y >> 16)));
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = scale * bf2f(src[j]);
        }
    }
}

// out = x + float(delta): the residual add of a bf16 Linear output onto the fp32 stream (what fm_layernorm_fwd_res does on the way)
__global__ __launch_bounds__(256) void add_bf16_f32_kernel(const float* __restrict__ x, const bf16_t* __restrict__ d, float* __restrict__ out, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
        const float4 a = *(const float4*)(x + i);
        const uint2 p = *(const uint2*)(d + i);
        *(float4*)(out + i) = make_float4(a.x + bf2f((bf16_t)(p.x & 0xffff)), a.y + bf2f((bf16_t)(p.x >> 16)),
                                          a.z + bf2f((bf16_t)(p.y & 0xffff)), a.w + bf2f((bf16_t)(p.y >> 16)));
    }
}

inline int grid_for(size_t work_items) {
    size_t g = (work_items + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int fm_swiglu_bwd(const void* da, int ldda, const void* gu, int ldgu, void* dgu, int lddgu, int R, int H, int Hp, void* stream) {
    FM_CHECK_ARG(da && gu && dgu && R > 0 && H > 0 && Hp >= H && Hp % 8 == 0, "fm_swiglu_bwd: bad argument (Hp must be a multiple of 8)");
    FM_CHECK_ARG(ldda % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0 && ((((uintptr_t)da | (uintptr_t)gu | (uintptr_t)dgu) & 15) == 0),
                 "fm_swiglu_bwd: 16-byte aligned buffers with leading dims that are multiples of 8");
    static const int ew_nt = [] { const char* e = getenv("FOURM_EW_NT"); return e ? atoi(e) : 3; }();      // bit 0 = the saved (g | u), bit 1 = d(act) as NON-TEMPORAL loads: both are read exactly once, here - they need not displace the (dg | du) this kernel writes for the GEMMs that run next (56.58 -> 56.37 ms per 4M-B step same-box; FOURM_EW_NT=0: plain loads)
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((size_t)R * Hp / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)da, ldda,
                       (const bf16_t*)gu, ldgu, (bf16_t*)dgu, lddgu, R, H, Hp, ew_nt);
    FM_CHECK_LAUNCH("fm_swiglu_bwd");
    return 0;
}

extern "C" int fm_gelu_bwd(const void* dh, int lddh, const void* pre, int ldp, void* dpre, int lddp, int R, int H, int Hp, void* stream) {
    FM_CHECK_ARG(dh && pre && dpre && R > 0 && H > 0 && Hp >= H && Hp % 4 == 0, "fm_gelu_bwd: bad argument");
    FM_CHECK_ARG(lddh % 4 == 0 && ldp % 4 == 0 && lddp % 4 == 0, "fm_gelu_bwd: leading dims must be multiples of 4");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for((size_t)R * Hp / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, lddh,
                       (const bf16_t*)pre, ldp, (bf16_t*)dpre, lddp, R, H, Hp);
    FM_CHECK_LAUNCH("fm_gelu_bwd");
    return 0;
}

extern "C" int fm_cast_pad(const void* src, int ld_src, void* dst, int ld_dst, int rows, int cols, void* stream) {
    FM_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_dst >= cols && ld_dst % 4 == 0, "fm_cast_pad: bad argument");
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((size_t)rows * ld_dst / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src,
                       (bf16_t*)dst, ld_dst, rows, cols);
    FM_CHECK_LAUNCH("fm_cast_pad");
    return 0;
}

extern "C" int fm_transpose_cast_pad(const void* src, int ld_src, void* dst, int ld_dst, int dst_cols, int rows, int cols, void* stream) {
    FM_CHECK_ARG(src && dst && rows > 0 && cols > 0 && dst_cols >= rows && ld_dst >= dst_cols, "fm_transpose_cast_pad: bad argument");
    dim3 grid((cols + 63) / 64, (dst_cols + 63) / 64);
    hipLaunchKernelGGL(transpose_cast_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, (bf16_t*)dst, ld_dst, dst_cols, rows, cols);
    FM_CHECK_LAUNCH("fm_transpose_cast_pad");
    return 0;
}

extern "C" int fm_shadow_refresh(const fm_shadow_desc* descs, int n_descs, int total_tiles, void* stream) {
    FM_CHECK_ARG(descs && n_descs > 0 && total_tiles > 0, "fm_shadow_refresh: bad argument");
    hipLaunchKernelGGL(shadow_refresh_kernel, dim3((total_tiles + SHADOW_TILES_PER_BLOCK - 1) / SHADOW_TILES_PER_BLOCK), dim3(256), 0,
                       (hipStream_t)stream, descs, n_descs, total_tiles);
    FM_CHECK_LAUNCH("fm_shadow_refresh");
    return 0;
}

// Gradient of a LayerNorm weight folded into the following Linear's weight image (fm_fold_colscale_grad in the header).
// Block = 64 columns x 64 rows of one job: thread (cl, rl) walks rows rl, rl + 4, ...; the column sums meet in LDS, one atomic per column.
struct FoldArgs { fm_fold_grad_job job[FM_FOLD_MAX_JOBS]; int n; };
__global__ __launch_bounds__(256) void fold_colscale_grad_kernel(FoldArgs a) {
    __shared__ float red[4][64];
    const fm_fold_grad_job j = a.job[blockIdx.z];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, r0 = blockIdx.y * 64;
    float s = 0.f;
    if (c < j.cols && r0 < j.rows) {
        const float g = ((const float*)j.gamma)[c];
        const float* dwp = (const float*)j.dWp;
        const float* w = (const float*)j.W;
        float* gw = (float*)j.gW;
        const int r1 = min(j.rows, r0 + 64);
        for (int r = r0 + rl; r < r1; r += 4) {
            const float d = dwp[(size_t)r * j.ld_dwp + c];
            if (gw) gw[(size_t)r * j.cols + c] += d * g;
            s += d * w[(size_t)r * j.cols + c];
        }
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < j.cols && j.ggamma && r0 < j.rows) atomicAdd((float*)j.ggamma + c, red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}

extern "C" int fm_fold_colscale_grad(const fm_fold_grad_job* jobs, int n_jobs, void* stream) {
    FM_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= FM_FOLD_MAX_JOBS, "fm_fold_colscale_grad: 1..FM_FOLD_MAX_JOBS jobs");
    FoldArgs a{};
    int max_r = 0, max_c = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const fm_fold_grad_job& j = jobs[i];
        FM_CHECK_ARG(j.dWp && j.W && j.gamma && j.rows > 0 && j.cols > 0 && j.ld_dwp >= j.cols, "fm_fold_colscale_grad: bad job");
        a.job[i] = j;
        max_r = j.rows > max_r ? j.rows : max_r;
        max_c = j.cols > max_c ? j.cols : max_c;
    }
    a.n = n_jobs;
    hipLaunchKernelGGL(fold_colscale_grad_kernel, dim3((max_c + 63) / 64, (max_r + 63) / 64, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    FM_CHECK_LAUNCH("fm_fold_colscale_grad");
    return 0;
}

extern "C" int fm_colsum(const void* dy, int ldy, void* db, int R, int N, void* stream) {
    FM_CHECK_ARG(dy && db && R > 0 && N > 0, "fm_colsum: bad argument");
    const int rpb = 128;
    dim3 grid((N + 255) / 256, (R + rpb - 1) / rpb);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, ldy, (float*)db, R, N, rpb);
    FM_CHECK_LAUNCH("fm_colsum");
    return 0;
}

extern "C" int fm_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int64_t step, const void* grad_mult, const void* hyper, void* sumsq, void* stream) {
    FM_CHECK_ARG(p && g && m && v && n > 0 && step > 0, "fm_adamw: bad argument");
    FM_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "fm_adamw: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)(n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (float*)p, (const float*)g, (float*)m,
                       (float*)v, (size_t)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), (const float*)grad_mult,
                       (const float*)hyper, (float*)sumsq);
    FM_CHECK_LAUNCH("fm_adamw");
    return 0;
}

extern "C" int fm_adamw_shadow(const fm_adamw_job* jobs, int n_jobs, int total_tiles, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, const void* grad_mult, const void* hyper, void* sumsq, void* stream) {
    FM_CHECK_ARG(jobs && n_jobs > 0 && total_tiles > 0 && step > 0, "fm_adamw_shadow: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const int blocks = total_tiles < 8192 ? total_tiles : 8192;
    hipLaunchKernelGGL(adamw_shadow_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs, total_tiles, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), (const float*)grad_mult, (const float*)hyper, (float*)sumsq);
    FM_CHECK_LAUNCH("fm_adamw_shadow");
    return 0;
}

extern "C" int fm_sumsq(const void* x, int64_t n, void* out, void* stream) {
    FM_CHECK_ARG(x && out && n > 0 && (((uintptr_t)x) & 15) == 0, "fm_sumsq: bad argument");
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((size_t)(n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (size_t)n, (float*)out);
    FM_CHECK_LAUNCH("fm_sumsq");
    return 0;
}

extern "C" int fm_clip_coef(const void* sumsq, float max_norm, void* norm_out, void* coef_out, void* stream) {
    FM_CHECK_ARG(sumsq && norm_out, "fm_clip_coef: null pointer");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const float*)sumsq, max_norm, (float*)norm_out, (float*)coef_out);
    FM_CHECK_LAUNCH("fm_clip_coef");
    return 0;
}

extern "C" int fm_bf16_to_f32_scaled(const void* src, void* dst, int64_t n, float scale, void* stream) {
    FM_CHECK_ARG(src && dst && n > 0 && (((uintptr_t)dst) & 15) == 0 && (((uintptr_t)src) & 7) == 0, "fm_bf16_to_f32_scaled: bad argument");
    hipLaunchKernelGGL(bf16_to_f32_scaled_kernel, dim3(grid_for((size_t)(n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (float*)dst, (size_t)n, scale);
    FM_CHECK_LAUNCH("fm_bf16_to_f32_scaled");
    return 0;
}

extern "C" int fm_add_bf16_f32(const void* x, const void* delta, void* out, int64_t n, void* stream) {
    FM_CHECK_ARG(x && delta && out && n > 0 && n % 4 == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0) && (((uintptr_t)delta) & 7) == 0,
                 "fm_add_bf16_f32: n must be a multiple of 4, x / out 16-byte and delta 8-byte aligned");
    hipLaunchKernelGGL(add_bf16_f32_kernel, dim3(grid_for((size_t)n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const bf16_t*)delta,
                       (float*)out, (size_t)n);
    FM_CHECK_LAUNCH("fm_add_bf16_f32");
    return 0;
}

extern "C" int fm_f32_to_bf16(const void* src, void* dst, int64_t n, void* stream) {
    FM_CHECK_ARG(src && dst && n > 0 && (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0, "fm_f32_to_bf16: bad argument");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for((size_t)(n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (bf16_t*)dst, (size_t)n);
    FM_CHECK_LAUNCH("fm_f32_to_bf16");
    return 0;
}

// ---- stochastic depth (DropPath, fm_utils.py:64-87): x[r][:] *= scale[r / rows_per_sample] on a bf16 (R, N) tile in place --------------
namespace {
__global__ __launch_bounds__(256) void scale_rows_bf16_kernel(bf16_t* __restrict__ x, int ld, const float* __restrict__ scale, int rows_per_sample, int R, int N) {
    const int nv = N / 8;                                   // 16-byte pieces per row (N % 8 == 0)
    const size_t total = (size_t)R * nv;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int r = (int)(e / nv), c = (int)(e % nv);
        const float s = scale[r / rows_per_sample];
        uint4* p = (uint4*)(x + (size_t)r * ld + c * 8);
        uint4 v = *p;
        uint32_t* w = (uint32_t*)&v;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)f2bf(bf2f((bf16_t)(w[i] & 0xffff)) * s) | ((uint32_t)f2bf(bf2f((bf16_t)(w[i] >> 16)) * s) << 16);
        *p = v;
    }
}
}  // namespace

extern "C" int fm_scale_rows_bf16(void* x, int ld, const void* scale, int rows_per_sample, int R, int N, void* stream) {
    FM_CHECK_ARG(x && scale && R > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0 && rows_per_sample > 0 && (((uintptr_t)x) & 15) == 0,
                 "fm_scale_rows_bf16: bad argument (N, ld multiples of 8; 16-byte aligned)");
    size_t blocks = ((size_t)R * (N / 8) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(scale_rows_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, (const float*)scale, rows_per_sample, R, N);
    FM_CHECK_LAUNCH("fm_scale_rows_bf16");
    return 0;
}
