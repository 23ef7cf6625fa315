// LayerNorm over the last dimension, eps inside the sqrt, fp32 statistics (F.layer_norm semantics,
// fourm/models/fm_utils.py:93-108; under autocast upstream runs it in fp32 and the consumer Linear
// casts to bf16 — here the cast is fused into the store).
//
// HBM-bound: one wavefront per row, the row lives in registers (<= 2048 columns), 16-byte accesses.
#include <algorithm>
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int MAXC_LIMIT = 8;   // float4 chunks per lane  ->  D <= 64 * 4 * 8 = 2048
// The kernels are instantiated for 2 / 3 / 4 / 8 chunks per lane (D <= 512 / 768 / 1024 / 2048): the row
// lives in registers, so a tight bound keeps the VGPR count low and the occupancy (= memory-level
// parallelism of this HBM-bound kernel) high.

// 16-byte / 8-byte loads with the non-temporal hint (lab: a stream that is read once need not displace what the next kernel will read)
__device__ __forceinline__ float4 ld4(const float* p, bool nt) {
    if (nt) { const f32x4_t t = __builtin_nontemporal_load((const f32x4_t*)p); return make_float4(t[0], t[1], t[2], t[3]); }
    return *(const float4*)p;
}
__device__ __forceinline__ uint2 ld2u(const bf16_t* p, bool nt) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2_t;
    if (nt) { const u2_t t = __builtin_nontemporal_load((const u2_t*)p); return make_uint2(t[0], t[1]); }
    return *(const uint2*)p;
}
template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *(uint2*)p = make_uint2(pack2bf(a, b), pack2bf(c, d));
}

// RES: the row is x + delta (delta = the bf16 output of the Linear that precedes the norm in the residual stream, fm_utils.py:332-333,
// 363-365); the sum is written to `xo` (the new residual stream, fp32) on the way.  This moves the residual add out of the GEMM
// epilogue - where every workgroup of the chip reads and rewrites its fp32 tile at the same moment, at ~2.8 TB/s - into this
// streaming kernel (5.5 TB/s): same bytes, same arithmetic (x + float(bf16)), bit-identical stream.
template <typename OutT, int MAXC, bool RES = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ b, OutT* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     const int* __restrict__ row_map, int R, int D, float eps,
                                                     const bf16_t* __restrict__ delta = nullptr, int ldd = 0, float* __restrict__ xo = nullptr, int ldxo = 0,
                                                     int nt_res = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    // weights once per wave; two rows per iteration with both rows' loads issued first (HBM stream: more bytes in flight)
    float4 ww[MAXC], bb[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        ww[c] = ch < nch ? *(const float4*)(w + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bb[c] = (b && ch < nch) ? *(const float4*)(b + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    constexpr int NR = 2;
    for (int r0 = blockIdx.x * 4 + wave; r0 < R; r0 += gridDim.x * 4 * NR) {
        float4 v[NR][MAXC];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + q * gridDim.x * 4;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                v[q][c] = (ch < nch && r < R) ? ld4(x + (size_t)r * ldx + ch * 4, nt_res & 2) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (RES) {
            uint2 dl[NR][MAXC];
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int r = r0 + q * gridDim.x * 4;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int ch = lane + 64 * c;
                    dl[q][c] = (ch < nch && r < R) ? ld2u(delta + (size_t)r * ldd + ch * 4, nt_res & 2) : make_uint2(0u, 0u);
                }
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int r = r0 + q * gridDim.x * 4;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int ch = lane + 64 * c;
                    float d4[4];
                    unpack_bf4(dl[q][c], d4);
                    v[q][c].x += d4[0]; v[q][c].y += d4[1]; v[q][c].z += d4[2]; v[q][c].w += d4[3];
                    if (ch < nch && r < R) {
                        if (nt_res & 1) __builtin_nontemporal_store(f32x4_t{v[q][c].x, v[q][c].y, v[q][c].z, v[q][c].w}, (f32x4_t*)(xo + (size_t)r * ldxo + ch * 4));
                        else *(float4*)(xo + (size_t)r * ldxo + ch * 4) = v[q][c];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + q * gridDim.x * 4;
            if (r >= R) continue;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) s += (v[q][c].x + v[q][c].y) + (v[q][c].z + v[q][c].w);
            const float mu = wave_sum(s) / (float)D;
            float sq = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) {
                    const float a0 = v[q][c].x - mu, a1 = v[q][c].y - mu, a2 = v[q][c].z - mu, a3 = v[q][c].w - mu;
                    sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            const float rs = rsqrtf(wave_sum(sq) / (float)D + eps);
            if (lane == 0) {
                if (mean) mean[r] = mu;
                if (rstd) rstd[r] = rs;
            }
            const int dst = row_map ? row_map[r] : r;
            if (dst < 0) continue;
            OutT* yr = y + (size_t)dst * ldy;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch)
                    store4<OutT>(yr + ch * 4, (v[q][c].x - mu) * rs * ww[c].x + bb[c].x, (v[q][c].y - mu) * rs * ww[c].y + bb[c].y,
                                 (v[q][c].z - mu) * rs * ww[c].z + bb[c].z, (v[q][c].w - mu) * rs * ww[c].w + bb[c].w);
            }
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w ;  dw += sum_r dy * xhat ;  db += sum_r dy
// rows per workgroup (4 waves x 8 ... 32 rows): every workgroup ends with one fp32 atomic per column and gradient, all workgroups on the same
// D addresses - the more rows a workgroup covers, the fewer of those (FOURM_LN_BWD_ROWS: lab)
// lab (FOURM_LN_NT): bit 0 = the forward's new fp32 residual stream, bit 1 = the backward's fp32 residual gradient leave as non-temporal stores;
// bit 2 = the forward's x / delta, bit 3 = the backward's dy / residual gradient, bit 4 = the backward's saved h arrive as non-temporal loads
static int ln_nt_flags() {
    static const int f = [] { const char* e = getenv("FOURM_LN_NT"); return e ? atoi(e) : 24; }();      // default: the backward's read-once streams (dy, the residual gradient, the saved h) as non-temporal loads: 56.73 -> 56.40 ms per 4M-B step same-box; the forward's (bits 0, 2) and the gradient store (bit 1) measured equal or worse
    return f;
}
static int bwd_rows(int R) {
    static const int env = [] { const char* e = getenv("FOURM_LN_BWD_ROWS"); return e ? atoi(e) : 0; }();
    if (env >= 4) return env;
    return R >= 512 * 64 ? 64 : 32;        // (32768 x 768 from h: 78.6 us at 32 rows, 73.2 at 64, 74.3 at 128; without the atomics 67)
}

// FROM_H (bias-free norms whose bf16 output h = bf16(xhat * w) was saved for the weight-gradient GEMM anyway): xhat = h / w is rebuilt
// from the 2-byte h instead of the 4-byte x (16 -> 14 bytes per element of this HBM-bound kernel; relative error of xhat 2^-9, the
// rounding h already carries into the GEMMs).  Chunks holding a weight of magnitude < 1e-20 (h = 0 there: xhat is not recoverable)
// read x as before.
template <int MAXC, bool FROM_H = false, bool HAS_DB = true>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const int* __restrict__ dy_row_map,
                                                     const bf16_t* __restrict__ h, int ldh,
                                                     const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* dres, float* dx, int lddx, bf16_t* __restrict__ dx_bf, int lddxbf,
                                                     float* __restrict__ dw, float* __restrict__ db, int R, int D, int BWD_ROWS, int nt_dx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;   // [2][4 waves][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    float4 aw[MAXC], ab[MAXC], wv[MAXC], iw[FROM_H ? MAXC : 1];
    bool from_x[FROM_H ? MAXC : 1];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        aw[c] = ab[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ch = lane + 64 * c;
        wv[c] = ch < nch ? *(const float4*)(w + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FROM_H) {
            from_x[c] = fminf(fminf(fabsf(wv[c].x), fabsf(wv[c].y)), fminf(fabsf(wv[c].z), fabsf(wv[c].w))) < 1e-20f;
            iw[c] = from_x[c] ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(1.f / wv[c].x, 1.f / wv[c].y, 1.f / wv[c].z, 1.f / wv[c].w);
        }
    }
    const int r_end = min(R, (int)(blockIdx.x + 1) * BWD_ROWS);
    // Two rows per iteration, every load of both rows (dy, x, residual gradient) issued before any arithmetic:
    // the kernel is a pure HBM stream and a wave otherwise waits out one memory round trip per row.
    constexpr int NR = 2;
    for (int r0 = blockIdx.x * BWD_ROWS + wave; r0 < r_end; r0 += 4 * NR) {
        uint2 dyp[NR][MAXC], hp[NR][FROM_H ? MAXC : 1];
        float4 xv[NR][MAXC], rv[NR][MAXC];
        float mu[NR], rs[NR];
        bool live[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = r0 + 4 * q;
            live[q] = r < r_end;
            const int rr = live[q] ? r : r0;
            const int src = dy_row_map ? dy_row_map[rr] : rr;
            mu[q] = mean[rr]; rs[q] = rstd[rr];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                dyp[q][c] = make_uint2(0u, 0u);
                xv[q][c] = rv[q][c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ch < nch && live[q]) {
                    if (src >= 0) dyp[q][c] = ld2u(dy + (size_t)src * lddy + ch * 4, nt_dx & 2);
                    if constexpr (FROM_H) {
                        hp[q][c] = ld2u(h + (size_t)rr * ldh + ch * 4, nt_dx & 4);
                        if (from_x[c]) xv[q][c] = *(const float4*)(x + (size_t)rr * ldx + ch * 4);
                    } else xv[q][c] = *(const float4*)(x + (size_t)rr * ldx + ch * 4);
                    if (dres) rv[q][c] = ld4(dres + (size_t)rr * lddx + ch * 4, nt_dx & 2);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            if (!live[q]) continue;
            const int r = r0 + 4 * q;
            float4 g[MAXC], xh[MAXC];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                g[c] = xh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ch < nch) {
                    const uint2 p = dyp[q][c];
                    const float4 d = make_float4(bf2f((bf16_t)(p.x & 0xffff)), bf2f((bf16_t)(p.x >> 16)), bf2f((bf16_t)(p.y & 0xffff)), bf2f((bf16_t)(p.y >> 16)));
                    const float4 v = xv[q][c];
                    xh[c] = make_float4((v.x - mu[q]) * rs[q], (v.y - mu[q]) * rs[q], (v.z - mu[q]) * rs[q], (v.w - mu[q]) * rs[q]);
                    if constexpr (FROM_H) {
                        const uint2 hh = hp[q][c];
                        if (!from_x[c]) xh[c] = make_float4(bf2f((bf16_t)(hh.x & 0xffff)) * iw[c].x, bf2f((bf16_t)(hh.x >> 16)) * iw[c].y,
                                                            bf2f((bf16_t)(hh.y & 0xffff)) * iw[c].z, bf2f((bf16_t)(hh.y >> 16)) * iw[c].w);
                    }
                    aw[c].x += d.x * xh[c].x; aw[c].y += d.y * xh[c].y; aw[c].z += d.z * xh[c].z; aw[c].w += d.w * xh[c].w;
                    if constexpr (HAS_DB) { ab[c].x += d.x; ab[c].y += d.y; ab[c].z += d.z; ab[c].w += d.w; }
                    g[c] = make_float4(d.x * wv[c].x, d.y * wv[c].y, d.z * wv[c].z, d.w * wv[c].w);
                    s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
                    s2 += (g[c].x * xh[c].x + g[c].y * xh[c].y) + (g[c].z * xh[c].z + g[c].w * xh[c].w);
                }
            }
            const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nch) {
                    const float4 t = rv[q][c];
                    const float4 o = make_float4(rs[q] * (g[c].x - m1 - xh[c].x * m2) + t.x, rs[q] * (g[c].y - m1 - xh[c].y * m2) + t.y,
                                                 rs[q] * (g[c].z - m1 - xh[c].z * m2) + t.z, rs[q] * (g[c].w - m1 - xh[c].w * m2) + t.w);
                    if (nt_dx & 1) __builtin_nontemporal_store(f32x4_t{o.x, o.y, o.z, o.w}, (f32x4_t*)(dx + (size_t)r * lddx + ch * 4));
                    else *(float4*)(dx + (size_t)r * lddx + ch * 4) = o;
                    if (dx_bf) *(uint2*)(dx_bf + (size_t)r * lddxbf + ch * 4) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
                }
            }
        }
    }
    if (!dw && !db) return;
    // cross-wave reduction of the column sums, then one atomic per column per workgroup
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
            *(float4*)(red + (size_t)wave * D + ch * 4) = aw[c];
            if constexpr (HAS_DB) *(float4*)(red + (size_t)(4 + wave) * D + ch * 4) = ab[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
        if (dw) unsafeAtomicAdd(dw + i, (red[i] + red[D + i]) + (red[2 * D + i] + red[3 * D + i]));
        if constexpr (HAS_DB) if (db) unsafeAtomicAdd(db + i, (red[4 * D + i] + red[5 * D + i]) + (red[6 * D + i] + red[7 * D + i]));
    }
}


// ------------------------------------------------------------------------------------------------
// Per-head LayerNorm of q / k (qk_norm models: NormAttention / NormCrossAttention, fm_utils.py:222-308).
// A head vector has 64 elements: 16 lanes x 4 elements, so one wave normalises 4 heads; under autocast the
// norm runs in fp32 on the bf16 q / k and its output is rounded to bf16 by the following matmul.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}

__global__ __launch_bounds__(256) void headnorm_fwd_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w,
                                                           const float* __restrict__ b, bf16_t* __restrict__ y, int ldy,
                                                           float2* __restrict__ stats, int R, int H, float eps) {
    const int sub = threadIdx.x & 15;
    const float4 wv = *(const float4*)(w + sub * 4);
    const float4 bv = b ? *(const float4*)(b + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t total = (size_t)R * H;
    for (size_t v = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); v < total; v += (size_t)gridDim.x * 16) {
        const int r = v / H, h = v % H;
        const uint2 p = *(const uint2*)(x + (size_t)r * ldx + h * 64 + sub * 4);
        const float e[4] = {bf2f((bf16_t)(p.x & 0xffff)), bf2f((bf16_t)(p.x >> 16)), bf2f((bf16_t)(p.y & 0xffff)), bf2f((bf16_t)(p.y >> 16))};
        const float mean = group16_sum(e[0] + e[1] + e[2] + e[3]) * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) q += (e[i] - mean) * (e[i] - mean);
        const float rs = rsqrtf(group16_sum(q) * (1.0f / 64.0f) + eps);
        const float o0 = (e[0] - mean) * rs * wv.x + bv.x, o1 = (e[1] - mean) * rs * wv.y + bv.y;
        const float o2 = (e[2] - mean) * rs * wv.z + bv.z, o3 = (e[3] - mean) * rs * wv.w + bv.w;
        *(uint2*)(y + (size_t)r * ldy + h * 64 + sub * 4) = make_uint2(pack2bf(o0, o1), pack2bf(o2, o3));
        if (sub == 0 && stats) stats[v] = make_float2(mean, rs);
    }
}

// dx = rs * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat));  dw += sum dy*xhat;  db += sum dy
__global__ __launch_bounds__(256) void headnorm_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx,
                                                           const float* __restrict__ w, const float2* __restrict__ stats,
                                                           bf16_t* __restrict__ dx, int lddx, float* __restrict__ dw, float* __restrict__ db,
                                                           int R, int H) {
    __shared__ float red[2][16][64];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const float4 wv = *(const float4*)(w + sub * 4);
    const float wj[4] = {wv.x, wv.y, wv.z, wv.w};
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t total = (size_t)R * H;
    for (size_t v = (size_t)blockIdx.x * 16 + grp; v < total; v += (size_t)gridDim.x * 16) {
        const int r = v / H, h = v % H;
        const uint2 px = *(const uint2*)(x + (size_t)r * ldx + h * 64 + sub * 4);
        const uint2 pd = *(const uint2*)(dy + (size_t)r * lddy + h * 64 + sub * 4);
        const float2 st = stats[v];
        const float e[4] = {bf2f((bf16_t)(px.x & 0xffff)), bf2f((bf16_t)(px.x >> 16)), bf2f((bf16_t)(px.y & 0xffff)), bf2f((bf16_t)(px.y >> 16))};
        const float d[4] = {bf2f((bf16_t)(pd.x & 0xffff)), bf2f((bf16_t)(pd.x >> 16)), bf2f((bf16_t)(pd.y & 0xffff)), bf2f((bf16_t)(pd.y >> 16))};
        float xh[4], g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xh[i] = (e[i] - st.x) * st.y;
            g[i] = d[i] * wj[i];
            s1 += g[i]; s2 += g[i] * xh[i];
            aw[i] += d[i] * xh[i]; ab[i] += d[i];
        }
        s1 = group16_sum(s1) * (1.0f / 64.0f); s2 = group16_sum(s2) * (1.0f / 64.0f);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = st.y * (g[i] - s1 - xh[i] * s2);
        *(uint2*)(dx + (size_t)r * lddx + h * 64 + sub * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
    // column sums over the workgroup's 16 vector slots, then one atomic per column
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[0][grp][sub * 4 + i] = aw[i]; red[1][grp][sub * 4 + i] = ab[i]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) s += red[which][gq][c];
        float* dst = which ? db : dw;
        if (dst) unsafeAtomicAdd(dst + c, s);
    }
}

}  // namespace

static int chunks_for(int D) { const int c = (D / 4 + 63) / 64; return c <= 2 ? 2 : c <= 3 ? 3 : c <= 4 ? 4 : 8; }

extern "C" int fm_layernorm_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, int y_is_f32,
                                void* mean, void* rstd, const int32_t* row_map, int R, int D, float eps, void* stream) {
    FM_CHECK_ARG(x && w && y, "fm_layernorm_fwd: null pointer");
    FM_CHECK_ARG(R > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXC_LIMIT, "fm_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * MAXC_LIMIT);
    FM_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, "fm_layernorm_fwd: leading dims must be multiples of 4");
    int grid = (R + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
#define LN_FWD(T, C)                                                                                                        \
    hipLaunchKernelGGL((ln_fwd_kernel<T, C>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (const float*)w, \
                       (const float*)b, (T*)y, ldy, (float*)mean, (float*)rstd, row_map, R, D, eps)
#define LN_FWD_C(T)                                    \
    switch (chunks_for(D)) {                           \
        case 2: LN_FWD(T, 2); break;                   \
        case 3: LN_FWD(T, 3); break;                   \
        case 4: LN_FWD(T, 4); break;                   \
        default: LN_FWD(T, 8); break;                  \
    }
    if (y_is_f32) { LN_FWD_C(float) } else { LN_FWD_C(bf16_t) }
#undef LN_FWD_C
#undef LN_FWD
    FM_CHECK_LAUNCH("fm_layernorm_fwd");
    return 0;
}

extern "C" int fm_layernorm_fwd_res(const void* x, int ldx, const void* delta, int ldd, void* x_out, int ldxo, const void* w, const void* b,
                                    void* y, int ldy, int y_is_f32, void* mean, void* rstd, const int32_t* row_map, int R, int D, float eps,
                                    void* stream) {
    FM_CHECK_ARG(x && delta && x_out && w && y, "fm_layernorm_fwd_res: null pointer");
    FM_CHECK_ARG(R > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXC_LIMIT, "fm_layernorm_fwd_res: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * MAXC_LIMIT);
    FM_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldd % 4 == 0 && ldxo % 4 == 0, "fm_layernorm_fwd_res: leading dims must be multiples of 4");
    FM_CHECK_ARG((((uintptr_t)delta) & 7) == 0 && (((uintptr_t)x_out | (uintptr_t)x) & 15) == 0, "fm_layernorm_fwd_res: alignment (delta 8 B, x / x_out 16 B)");
    int grid = (R + 3) / 4;
    if (grid > 256 * 16) grid = 256 * 16;
#define LN_FWD(T, C)                                                                                                        \
    hipLaunchKernelGGL((ln_fwd_kernel<T, C, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (const float*)w, \
                       (const float*)b, (T*)y, ldy, (float*)mean, (float*)rstd, row_map, R, D, eps, (const bf16_t*)delta, ldd, (float*)x_out, ldxo, (ln_nt_flags() & 1) | ((ln_nt_flags() >> 1) & 2))
#define LN_FWD_C(T)                                    \
    switch (chunks_for(D)) {                           \
        case 2: LN_FWD(T, 2); break;                   \
        case 3: LN_FWD(T, 3); break;                   \
        case 4: LN_FWD(T, 4); break;                   \
        default: LN_FWD(T, 8); break;                  \
    }
    if (y_is_f32) { LN_FWD_C(float) } else { LN_FWD_C(bf16_t) }
#undef LN_FWD_C
#undef LN_FWD
    FM_CHECK_LAUNCH("fm_layernorm_fwd_res");
    return 0;
}

static int layernorm_bwd_launch(const void* dy, int lddy, const int32_t* dy_row_map, const void* h, int ldh, const void* x, int ldx, const void* w,
                                const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                                int lddxbf, void* dw, void* db, int R, int D, void* stream) {
    FM_CHECK_ARG(dy && x && w && mean && rstd && dx, "fm_layernorm_bwd: null pointer");
    FM_CHECK_ARG(!h || ldh % 4 == 0, "fm_layernorm_bwd_h: ldh must be a multiple of 4");
    FM_CHECK_ARG(R > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXC_LIMIT, "fm_layernorm_bwd: D=%d unsupported", D);
    FM_CHECK_ARG(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddxbf % 4 == 0, "fm_layernorm_bwd: leading dims must be multiples of 4");
    const int BWD_ROWS = bwd_rows(R);
    const int grid = (R + BWD_ROWS - 1) / BWD_ROWS;
    const size_t lds = (size_t)8 * D * sizeof(float);
#define LN_BWD_(C, FH)                                                                                                      \
    {                                                                                                                       \
        auto k = ln_bwd_kernel<C, FH, !FH>;                                                                                     \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 4) == hipSuccess); \
        (void)once;                                                                                                         \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy, lddy, dy_row_map, (const bf16_t*)h, ldh, \
                           (const float*)x, ldx, (const float*)w, (const float*)mean, (const float*)rstd, (const float*)dres, (float*)dx, lddx, \
                           (bf16_t*)dx_bf16, lddxbf, (float*)dw, (float*)db, R, D, BWD_ROWS, ((ln_nt_flags() >> 1) & 1) | ((ln_nt_flags() >> 2) & 2) | ((ln_nt_flags() >> 2) & 4));                               \
    }
#define LN_BWD(C) { if (h) LN_BWD_(C, true) else LN_BWD_(C, false) }
    switch (chunks_for(D)) {
        case 2: LN_BWD(2) break;
        case 3: LN_BWD(3) break;
        case 4: LN_BWD(4) break;
        default: LN_BWD(8) break;
    }
#undef LN_BWD
#undef LN_BWD_
    FM_CHECK_LAUNCH("fm_layernorm_bwd");
    return 0;
}

extern "C" int fm_layernorm_bwd(const void* dy, int lddy, const int32_t* dy_row_map, const void* x, int ldx, const void* w,
                                const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                                int lddxbf, void* dw, void* db, int R, int D, void* stream) {
    return layernorm_bwd_launch(dy, lddy, dy_row_map, nullptr, 0, x, ldx, w, mean, rstd, dres, dx, lddx, dx_bf16, lddxbf, dw, db, R, D, stream);
}

extern "C" int fm_layernorm_bwd_h(const void* dy, int lddy, const void* h, int ldh, const void* x, int ldx, const void* w,
                                  const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                                  int lddxbf, void* dw, int R, int D, void* stream) {
    FM_CHECK_ARG(h, "fm_layernorm_bwd_h: null pointer");
    return layernorm_bwd_launch(dy, lddy, nullptr, h, ldh, x, ldx, w, mean, rstd, dres, dx, lddx, dx_bf16, lddxbf, dw, nullptr, R, D, stream);
}

extern "C" int fm_headnorm_fwd(const void* x, int ldx, const void* w, const void* b, void* y, int ldy, void* stats, int R, int H, float eps,
                               void* stream) {
    FM_CHECK_ARG(x && w && y && R > 0 && H > 0, "fm_headnorm_fwd: bad argument");
    FM_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ((((uintptr_t)x | (uintptr_t)y) & 7) == 0) && ((((uintptr_t)w | (uintptr_t)b) & 15) == 0),
                 "fm_headnorm_fwd: alignment (x/y 8 B, w/b 16 B, leading dims %% 4)");
    const size_t total = (size_t)R * H;
    const int grid = (int)std::min<size_t>((total + 15) / 16, 256 * 32);
    hipLaunchKernelGGL(headnorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (const float*)w, (const float*)b,
                       (bf16_t*)y, ldy, (float2*)stats, R, H, eps);
    FM_CHECK_LAUNCH("fm_headnorm_fwd");
    return 0;
}

extern "C" int fm_headnorm_bwd(const void* dy, int lddy, const void* x, int ldx, const void* w, const void* stats, void* dx, int lddx, void* dw,
                               void* db, int R, int H, void* stream) {
    FM_CHECK_ARG(dy && x && w && stats && dx && R > 0 && H > 0, "fm_headnorm_bwd: bad argument");
    FM_CHECK_ARG(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 7) == 0) &&
                     ((((uintptr_t)w) & 15) == 0), "fm_headnorm_bwd: alignment (x/dy/dx 8 B, w 16 B, leading dims %% 4)");
    const size_t total = (size_t)R * H;
    const int grid = (int)std::min<size_t>((total + 15) / 16, 256 * 8);
    hipLaunchKernelGGL(headnorm_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx,
                       (const float*)w, (const float2*)stats, (bf16_t*)dx, lddx, (float*)dw, (float*)db, R, H);
    FM_CHECK_LAUNCH("fm_headnorm_bwd");
    return 0;
}
