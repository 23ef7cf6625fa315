// LayerNorm over the last dimension, eps inside the sqrt, fp32 statistics (F.layer_norm semantics,
// fourm/models/fm_utils.py:93-108; under autocast upstream runs it in fp32 and the consumer Linear
// casts to bf16 — here the cast is fused into the store).
//
// HBM-bound: one wavefront per row, the row lives in registers (<= 2048 columns), 16-byte accesses.
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int MAXC_LIMIT = 8;   // float4 chunks per lane  ->  D <= 64 * 4 * 8 = 2048
// The kernels are instantiated for 2 / 3 / 4 / 8 chunks per lane (D <= 512 / 768 / 1024 / 2048): the row
// lives in registers, so a tight bound keeps the VGPR count low and the occupancy (= memory-level
// parallelism of this HBM-bound kernel) high.

template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    *(uint2*)p = make_uint2(pack2bf(a, b), pack2bf(c, d));
}

template <typename OutT, int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ b, OutT* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     const int* __restrict__ row_map, int R, int D, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* xr = x + (size_t)r * ldx;
        float4 v[MAXC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            v[c] = ch < nch ? *(const float4*)(xr + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const float a0 = v[c].x - mu, a1 = v[c].y - mu, a2 = v[c].z - mu, a3 = v[c].w - mu;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
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
            if (ch < nch) {
                const float4 ww = *(const float4*)(w + ch * 4);
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b) bb = *(const float4*)(b + ch * 4);
                store4<OutT>(yr + ch * 4, (v[c].x - mu) * rs * ww.x + bb.x, (v[c].y - mu) * rs * ww.y + bb.y,
                             (v[c].z - mu) * rs * ww.z + bb.z, (v[c].w - mu) * rs * ww.w + bb.w);
            }
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w ;  dw += sum_r dy * xhat ;  db += sum_r dy
constexpr int BWD_ROWS = 32;   // rows per workgroup (4 waves x 8)

template <int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, int lddy, const int* __restrict__ dy_row_map,
                                                     const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* dres, float* dx, int lddx, bf16_t* __restrict__ dx_bf, int lddxbf,
                                                     float* __restrict__ dw, float* __restrict__ db, int R, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;   // [2][4 waves][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = D >> 2;
    float4 aw[MAXC], ab[MAXC], wv[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        aw[c] = ab[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ch = lane + 64 * c;
        wv[c] = ch < nch ? *(const float4*)(w + ch * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int r_end = min(R, (int)(blockIdx.x + 1) * BWD_ROWS);
    for (int r = blockIdx.x * BWD_ROWS + wave; r < r_end; r += 4) {
        const int src = dy_row_map ? dy_row_map[r] : r;
        const float mu = mean[r], rs = rstd[r];
        const float* xr = x + (size_t)r * ldx;
        float4 g[MAXC], xh[MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            g[c] = xh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < nch) {
                float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                if (src >= 0) {
                    const uint2 p = *(const uint2*)(dy + (size_t)src * lddy + ch * 4);
                    d = make_float4(bf2f((bf16_t)(p.x & 0xffff)), bf2f((bf16_t)(p.x >> 16)), bf2f((bf16_t)(p.y & 0xffff)), bf2f((bf16_t)(p.y >> 16)));
                }
                const float4 xv = *(const float4*)(xr + ch * 4);
                xh[c] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                aw[c].x += d.x * xh[c].x; aw[c].y += d.y * xh[c].y; aw[c].z += d.z * xh[c].z; aw[c].w += d.w * xh[c].w;
                ab[c].x += d.x; ab[c].y += d.y; ab[c].z += d.z; ab[c].w += d.w;
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
                float4 o = make_float4(rs * (g[c].x - m1 - xh[c].x * m2), rs * (g[c].y - m1 - xh[c].y * m2),
                                       rs * (g[c].z - m1 - xh[c].z * m2), rs * (g[c].w - m1 - xh[c].w * m2));
                if (dres) {
                    const float4 t = *(const float4*)(dres + (size_t)r * lddx + ch * 4);
                    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                }
                *(float4*)(dx + (size_t)r * lddx + ch * 4) = o;
                if (dx_bf) *(uint2*)(dx_bf + (size_t)r * lddxbf + ch * 4) = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
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
            *(float4*)(red + (size_t)(4 + wave) * D + ch * 4) = ab[c];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 256) {
        if (dw) unsafeAtomicAdd(dw + i, (red[i] + red[D + i]) + (red[2 * D + i] + red[3 * D + i]));
        if (db) unsafeAtomicAdd(db + i, (red[4 * D + i] + red[5 * D + i]) + (red[6 * D + i] + red[7 * D + i]));
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

extern "C" int fm_layernorm_bwd(const void* dy, int lddy, const int32_t* dy_row_map, const void* x, int ldx, const void* w,
                                const void* mean, const void* rstd, const void* dres, void* dx, int lddx, void* dx_bf16,
                                int lddxbf, void* dw, void* db, int R, int D, void* stream) {
    FM_CHECK_ARG(dy && x && w && mean && rstd && dx, "fm_layernorm_bwd: null pointer");
    FM_CHECK_ARG(R > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXC_LIMIT, "fm_layernorm_bwd: D=%d unsupported", D);
    FM_CHECK_ARG(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && lddxbf % 4 == 0, "fm_layernorm_bwd: leading dims must be multiples of 4");
    const int grid = (R + BWD_ROWS - 1) / BWD_ROWS;
    const size_t lds = (size_t)8 * D * sizeof(float);
#define LN_BWD(C)                                                                                                           \
    {                                                                                                                       \
        auto k = ln_bwd_kernel<C>;                                                                                          \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2048 * 4) == hipSuccess); \
        (void)once;                                                                                                         \
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy, lddy, dy_row_map, (const float*)x, ldx, \
                           (const float*)w, (const float*)mean, (const float*)rstd, (const float*)dres, (float*)dx, lddx,   \
                           (bf16_t*)dx_bf16, lddxbf, (float*)dw, (float*)db, R, D);                                         \
    }
    switch (chunks_for(D)) {
        case 2: LN_BWD(2) break;
        case 3: LN_BWD(3) break;
        case 4: LN_BWD(4) break;
        default: LN_BWD(8) break;
    }
#undef LN_BWD
    FM_CHECK_LAUNCH("fm_layernorm_bwd");
    return 0;
}
