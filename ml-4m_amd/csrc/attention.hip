// Masked multi-head attention, forward and backward, head_dim = 64 (every named 4M config), gfx950.
//
// Replaces  q@k^T * scale -> masked_fill(-finfo.max) -> softmax -> @v  of
// fourm/models/fm_utils.py:160-180 (self) and :197-219 (cross) and the autograd graph behind them.
//
// Mask semantics are upstream's: a blocked score is *replaced by* -finfo(bf16).max (not -inf), so a
// fully blocked query row attends uniformly to all keys.  Mask kinds:
//   FM_MASK_NONE     nothing blocked
//   FM_MASK_KEYPAD   blocked[b][q][k] = kpad[b][k]                              (fm.py:388, (B,1,N) mask)
//   FM_MASK_DECODER  blocked[b][q][k] = k >= cs[b][q]  ||  mod[b][q] != mod[b][k]   (fm.py:440-475,
//                    cs = cumsum of the compressed decoder_attention_mask; either term can be disabled)
//   FM_MASK_DENSE    blocked[b][q][k] = dense[b][q][k]                           (arbitrary (B,n1,n2) bool)
//
// Scores are computed transposed (S^T = K Q^T) so that a lane owns one query: the online-softmax
// statistics are lane-local and P^T feeds the PV MFMA straight from registers.  Softmax runs in the
// base-2 domain (t = s * scale * log2 e, p = exp2(t - max)): one FMA + one v_exp per score; stat_m holds
// the row maximum of t, stat_l the row sum of p (a private fwd -> bwd convention).  V (and, in the
// backward, K / Q / dO) are consumed "column-wise" from row-major LDS tiles through lds_col_frag.
#include "common.h"
#include "fourm_hip.h"

#ifndef ATTN_V2_SCHED
#define ATTN_V2_SCHED 0        // lab knob (tools/r05_attn_variants.sh builds the alternatives into separate libraries)
#endif
#ifndef ATTN_DOT2
#define ATTN_DOT2 1            // lab knob: 0 = delta without v_dot2c_f32_bf16
#endif
#ifndef ATTN_KV_DMA
#define ATTN_KV_DMA 1          // lab knob: 0 = K / V fragments of attn_bwd128_kernel loaded straight from global memory (32-byte requests)
#endif
#ifndef ATTN_DELTA_LDS
#define ATTN_DELTA_LDS 1       // lab knob: 0 = delta of attn_bwd128_kernel from two-threads-per-row global loads of O and dO (16-byte requests)
#endif
#ifndef ATTN_STAGED_STORES
#define ATTN_STAGED_STORES 1   // lab knob: 0 = dQ / dK / dV of attn_bwd128_kernel through store_bf16_groups (32-byte write requests)
#endif

namespace {

constexpr int HD = 64;
constexpr int ROWB = 128;                       // bytes per LDS row (64 bf16)
constexpr float NEG_FILL = -3.3895313892515355e38f;   // -finfo(bfloat16).max

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O;
    float* stat_m; float* stat_l;
    int ldq, ldk, ldv, ldo;
    int B, H, Nq, Nk;
    float scale;
    int mask_kind;
    const uint8_t* kpad;          // (B, Nk)
    const int32_t* cs;            // (B, Nq)   or null (no cumsum term)
    const int16_t* modq;          // (B, Nq)   or null (no modality term)
    const int16_t* modk;          // (B, Nk)
    const uint8_t* dense;         // (B, Nq, Nk)
    int causal;                   // FM_MASK_DECODER: use k > q instead of the cumsum rule
    // backward only
    const bf16_t* dO; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    int lddo, lddq, lddk, lddv;
    int kvr;                      // forward: rows between two samples in K / V (>= Nk: a K/V cache filled up to Nk)
    int zero_attn;                // softmax1: one extra zero logit in the denominator (allow_zero_attn)
    int o16;                      // forward (attn_fwd128_kernel): the rows of O are 16-byte aligned - whole-line stores staged through LDS
    int chunk;                    // rows of the two LDS tiles of the backward (a multiple of 32; >= max(Nq, Nk) padded when one chunk does)
};

// 16-byte chunk c of row r sits at chunk c ^ sw3(r).  sw3 = the bit-REVERSED row-pair index: rows r and r + 2 (the same 128-byte half of
// the 256-byte bank window) differ in bit 2 of the key, so the four rows a ds_read_b64_tr_b16 half-wave touches (4 rows x 64 bytes) spread
// over all 64 banks (round 4's key (r >> 1) & 7 put rows r and r + 2 on the same 16 banks: every transpose read was a 2-way conflict, 23 %
// of the backward's LDS cycles); the 16 rows of a ds_read_b128 lane group still see 8 distinct keys per row parity (a bijection of 3 bits).
__device__ __forceinline__ int sw3(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ __forceinline__ const char* tile_addr(const char* tile, int row, int col) {
    return tile + row * ROWB + ((((col >> 3) ^ sw3(row))) << 4) + (col & 7) * 2;
}

// copy `rows` x 64 bf16 (row stride ld elements) into a swizzled LDS tile; rows >= limit are clamped
// AUX: cache-policy bits of the LDS-DMA (2 = non-temporal: the backward's saved q / k / v are read exactly once)
#ifndef ATTN_BWD_SAVED_NT
#define ATTN_BWD_SAVED_NT 2          // (-DATTN_BWD_SAVED_NT=0: plain loads; 56.58 -> 56.36 ms per 4M-B step same-box, profiles/r06_ab_nontemporal.txt)
#endif
#ifndef ATTN_FWD_KV_NT
#define ATTN_FWD_KV_NT 2             // k / v of the 128 x 128 forward as non-temporal LDS-DMA: they are not read again before the backward (56.85 -> 56.60 ms per step same-box; -DATTN_FWD_KV_NT=0: plain)
#endif
#ifndef ATTN_BWD_DO_NT
#define ATTN_BWD_DO_NT 0             // lab build (-DATTN_BWD_DO_NT=2): dO of the 128 x 128 backward (just written by the proj dX GEMM, read once here) as non-temporal LDS-DMA
#endif
template <int NWAVES, int AUX = 0>
__device__ __forceinline__ void stage_rows(const bf16_t* src, int ld, int row0, int limit, int rows, char* tile, int wave, int lane) {
    for (int p = wave; p < rows / 8; p += NWAVES) {
        const int t = p * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ sw3(t);
        int r = row0 + t;
        r = r < limit ? r : limit - 1;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + (size_t)r * ld + lc * 8), LDS_PTR(tile + p * 8 * ROWB), 16, 0, AUX);
    }
}

__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int kk, int fhi) {
    return *(const bf16x8_t*)(tile + row * ROWB + ((((kk * 2 + fhi) ^ sw3(row))) << 4));
}

__device__ __forceinline__ bf16x8_t pack8(const float* p) {
    union { bf16x8_t v; uint32_t u[4]; } x;
    x.u[0] = pack2bf(p[0], p[1]); x.u[1] = pack2bf(p[2], p[3]); x.u[2] = pack2bf(p[4], p[5]); x.u[3] = pack2bf(p[6], p[7]);
    return x.v;
}

// blocked(q, k) for lane-owned q and key index k = k0 + c (c = position inside a 64-key window):
//   KEYPAD : bit c of `bits` (bit mask of the window's padded keys, pre-shifted by 4*fhi)
//   DECODER: (c >= cs_rel) | (mk != mq)   with cs_rel = cs[q] - k0 (or the causal bound q + 1 - k0)
//   DENSE  : byte load
template <int MASK>
__device__ __forceinline__ bool blocked_at(const AttnArgs& a, int b, int q, int k, int c, unsigned long long bits, int cs_rel, int mq, int mk) {
    if constexpr (MASK == FM_MASK_KEYPAD) return (bits >> c) & 1ull;
    else if constexpr (MASK == FM_MASK_DECODER) {
        bool blk = c >= cs_rel;
        if (a.modq) blk = blk || (mq != mk);
        return blk;
    } else if constexpr (MASK == FM_MASK_DENSE) return a.dense[((size_t)b * a.Nq + q) * a.Nk + k] != 0;
    else return false;
}
constexpr float LOG2E = 1.4426950408889634f;
// A wave's 32 rows x 64 bf16 (two 32 x 32 accumulators: register 4 g + j of t[df] = feature 32 df + 8 g + 4 fhi + j of row lane & 31) leave
// as WHOLE 128-byte lines: through a wave-private 4 KB LDS area (16-byte chunks XOR-swizzled by the row) and back row-contiguous, 8 lanes
// per row.  store_bf16_groups hands the L2 four 32-byte write requests per line; this form one - the request count, not the bytes, is what the
// 150 MB of dQ / dK / dV of an attention backward cost (4.7 M of its 6.7 M L2 requests per launch at the bench shape).
// row_scale multiplies the lane's row.  The rows must be 16-byte aligned (checked by the dispatch).
__device__ __forceinline__ void store_rows_staged(const f32x16_t (&t)[2], float row_scale, char* stage, bf16_t* g_row0, int ld, int lane) {
    const int r = lane & 31, fhi = lane >> 5;
#pragma unroll
    for (int df = 0; df < 2; ++df)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(uint2*)(stage + r * 128 + ((((df * 4 + g) ^ (r & 7))) << 4) + fhi * 8) =
                make_uint2(pack2bf(t[df][4 * g] * row_scale, t[df][4 * g + 1] * row_scale), pack2bf(t[df][4 * g + 2] * row_scale, t[df][4 * g + 3] * row_scale));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = q * 8 + (lane >> 3), c = lane & 7;
        const uint4 v = *(const uint4*)(stage + row * 128 + ((c ^ (row & 7)) << 4));
        *(uint4*)((char*)(g_row0 + (size_t)row * ld) + c * 16) = v;
    }
}

// acc + a.lo * b.lo + a.hi * b.hi on two packed bf16 pairs: ONE v_dot2c_f32_bf16 (products of bf16 are exact in fp32) instead of four unpacks
// and two FMAs - delta = rowsum(dO o O) in the prologue of the backward kernels (~80 of a wave's ~960 VALU instructions before round 5's end)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16pair_t;
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
#if ATTN_DOT2
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair_t, a), __builtin_bit_cast(bf16pair_t, b), acc, false);
#else       // lab (tools/r05_attn_variants.sh): the unpack + FMA form
    return acc + (bf2f((bf16_t)(a & 0xffff)) * bf2f((bf16_t)(b & 0xffff)) + bf2f((bf16_t)(a >> 16)) * bf2f((bf16_t)(b >> 16)));
#endif
}

// ------------------------------------------------------------------------------------------------
// forward: grid (ceil(Nq/128), H, B), 4 waves x 32 queries, keys in tiles of 64 with online softmax
// ------------------------------------------------------------------------------------------------
template <bool TR, int MASK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    constexpr int KT = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2 buffers][K tile 8 KB | V tile 8 KB] then the key metadata of the whole sequence (one global
    // round trip in the prologue instead of one per key tile)
    char* tiles = smem;
    const int NkP = (a.Nk + KT - 1) / KT * KT;
    int16_t* kmod_l = (int16_t*)(smem + 2 * 2 * KT * ROWB);      // [NkP]
    uint8_t* kpad_l = (uint8_t*)(kmod_l + NkP);                  // [NkP]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int q = q0 + (lane & 31);
    const int qc = q < a.Nq ? q : a.Nq - 1;

    const bf16_t* Qb = a.Q + (size_t)b * a.Nq * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * a.kvr * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * a.kvr * a.ldv + h * HD;

    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(Qb + (size_t)qc * a.ldq + (kk * 2 + fhi) * 8);
    int csq = 0, mq = 0;
    if constexpr (MASK == FM_MASK_DECODER) {
        if (a.cs) csq = a.cs[(size_t)b * a.Nq + qc];
        if (a.modq) mq = a.modq[(size_t)b * a.Nq + qc];
    }
    if constexpr (MASK == FM_MASK_DECODER || MASK == FM_MASK_KEYPAD) {
        for (int k = threadIdx.x; k < NkP; k += 256) {
            const int kc = k < a.Nk ? k : a.Nk - 1;
            if constexpr (MASK == FM_MASK_DECODER) kmod_l[k] = a.modk ? a.modk[(size_t)b * a.Nk + kc] : (int16_t)0;
            else kpad_l[k] = a.kpad[(size_t)b * a.Nk + kc];
        }
    }

    auto stage = [&](int t, int buf) {
        char* kt = tiles + buf * 2 * KT * ROWB;
        stage_rows<4>(Kb, a.ldk, t * KT, a.Nk, KT, kt, wave, lane);
        stage_rows<4>(Vb, a.ldv, t * KT, a.Nk, KT, kt + KT * ROWB, wave, lane);
    };

    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = a.scale * LOG2E;

    const int NT = (a.Nk + KT - 1) / KT;
    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < NT; ++t) {
        const int buf = t & 1;
        if (t + 1 < NT) stage(t + 1, buf ^ 1);
        const char* kt = tiles + buf * 2 * KT * ROWB;
        const char* vt = kt + KT * ROWB;

        // ---- S^T = K Q^T for the 64 keys of this tile ----------------------------------------
        f32x16_t st[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(kt, kb * 32 + (lane & 31), kk, fhi), qf[kk], st[kb], 0, 0, 0);
        }
        // ---- mask + running max (base-2 domain) ------------------------------------------------
        float p[2][16];
        float tmax = -INFINITY;
        unsigned long long bits = 0;
        int cs_rel = 0x7fffffff;
        if constexpr (MASK == FM_MASK_KEYPAD) bits = __ballot(kpad_l[t * KT + lane] != 0) >> (4 * fhi);
        if constexpr (MASK == FM_MASK_DECODER) {
            if (a.causal) cs_rel = qc + 1 - t * KT - 4 * fhi;
            else if (a.cs) cs_rel = csq - t * KT - 4 * fhi;
        }
        const bool full_tile = (t + 1) * KT <= a.Nk;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = kb * 32 + (r & 3) + 8 * (r >> 2);          // compile-time; key = t*KT + c + 4*fhi
                float sv = st[kb][r] * c2;
                if constexpr (MASK != FM_MASK_NONE) {
                    int mk = 0;
                    if constexpr (MASK == FM_MASK_DECODER) mk = kmod_l[t * KT + c + 4 * fhi];
                    const int k = t * KT + c + 4 * fhi;
                    sv = blocked_at<MASK>(a, b, qc, k < a.Nk ? k : a.Nk - 1, c, bits, cs_rel, mq, mk) ? NEG_FILL : sv;
                }
                if (!full_tile) sv = (t * KT + c + 4 * fhi) < a.Nk ? sv : -INFINITY;
                p[kb][r] = sv;
                tmax = fmaxf(tmax, sv);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[kb][r] = __builtin_amdgcn_exp2f(p[kb][r] - m_new);
                psum += p[kb][r];
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        // ---- O^T += V^T P^T ---------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8_t pb = pack8(&p[kb][8 * s]);
                const int rA = kb * 32 + s * 16 + 4 * fhi;
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const bf16x8_t vf = lds_col_frag<TR>([&](int r, int c) { return tile_addr(vt, r, c); }, rA, rA + 8, df * 32);
                    o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[df], 0, 0, 0);
                }
            }
        __syncthreads();
    }
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    float m_fin = m_run, oscale = 1.0f;
    if (a.zero_attn) {      // softmax1 (fm_utils.py:28-30): the padded zero logit joins the maximum and the sum, its probability is dropped
        m_fin = fmaxf(m_run, 0.f);
        oscale = __builtin_amdgcn_exp2f(m_run - m_fin);
        l_tot = l_tot * oscale + __builtin_amdgcn_exp2f(-m_fin);
    }
    const float inv = oscale / l_tot;
    {   // lanes l and l+32 own the same query row: 16-byte stores (store_bf16_groups)
        bf16_t* orow = a.O + ((size_t)b * a.Nq + qc) * a.ldo + h * HD;
        const bool wide_o = (a.ldo & 7) == 0 && (((uintptr_t)a.O) & 15) == 0;
        const int lim = q < a.Nq ? HD : 0;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint2 po[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    po[u] = make_uint2(pack2bf(o[df][4 * (g + u)] * inv, o[df][4 * (g + u) + 1] * inv),
                                       pack2bf(o[df][4 * (g + u) + 2] * inv, o[df][4 * (g + u) + 3] * inv));
                store_bf16_groups(orow, df * 32 + 8 * g, po[0], po[1], fhi, lim, wide_o);
            }
    }
    if (q < a.Nq) {
        if (fhi == 0 && a.stat_m) {
            const size_t si = ((size_t)b * a.H + h) * a.Nq + q;
            a.stat_m[si] = m_fin;
            a.stat_l[si] = l_tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward, 128 x 128 tokens exactly (the bench shape; same dispatch rule as attn_bwd128_kernel) - round 5.
// The general kernel above is VALU-bound at this shape (profiles/r05_final_pmc_attn.txt: 1 400 VALU instructions per wave = 26 % of a
// wave's cycles with THREE waves per SIMD, MFMA pipe 4.5 % busy): 12 VALU per score for the online softmax over two 64-key tiles (scale,
// two selects, max, subtract, exp, sum, the rescale of O through AGPR moves).  With all 128 keys of a (b, h) in LDS at once there is no
// running maximum and no rescale: 4 - 7 VALU per score
//   * the mask is applied to the raw MFMA result (blocked -> -inf: one v_cndmask; key padding: the lane mask comes from two s_bfe of the
//     128-bit padding bitmap through inverse_ballot - no VALU compare; decoder rule: the packed subtraction of attn_bwd128_kernel),
//   * one v_max3 per two scores for the TRUE row maximum m, p = exp2(fma(s, c2, -m c2)) (the scale is inside the FMA), one add, one
//     v_cvt_pk_bf16_f32 per two scores;
//   * fully blocked rows (m = -inf) leave the loop with p = 0 and are set to p = 1, l = 128, m = NEG_FILL behind a wave-uniform branch
//     (upstream: every score replaced by -finfo.max -> uniform attention);
//   * ONE memory round trip (K, V by LDS-DMA, Q into registers, the mask metadata) and ONE barrier per workgroup.
// stat_m / stat_l keep the general kernel's convention (row maximum of s * scale * log2 e, row sum of exp2).
// ------------------------------------------------------------------------------------------------
template <int MASK>
__global__ __launch_bounds__(256, 3) void attn_fwd128_kernel(AttnArgs a) {
    constexpr int N = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kt = smem;
    char* Vt = smem + N * ROWB;
    int* uk_l = (int*)(smem + 2 * N * ROWB);             // decoder: (mod_k << 9) + k per key; key padding: the 128-bit bitmap of padded keys

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5;
    const int b = blockIdx.y, h = blockIdx.x;
    const int q = wave * 32 + (lane & 31);
    const bf16_t* Qb = a.Q + (size_t)b * N * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * a.kvr * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * a.kvr * a.ldv + h * HD;

    stage_rows<4, ATTN_FWD_KV_NT>(Kb, a.ldk, 0, N, N, Kt, wave, lane);
    stage_rows<4, ATTN_FWD_KV_NT>(Vb, a.ldv, 0, N, N, Vt, wave, lane);
    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(Qb + (size_t)q * a.ldq + (kk * 2 + fhi) * 8);
    int wq = 0;                                          // decoder: (mod_q << 9) + cs_q - 1: blocked <=> (unsigned)(wq - uk) >= 255
    if constexpr (MASK == FM_MASK_DECODER) {
        int csv = 255, lov = 0;
        if (a.causal) csv = q + 1;
        else if (a.cs) csv = min(max(a.cs[(size_t)b * N + q], 0), 255);
        if (a.modq) lov = (int)a.modq[(size_t)b * N + q] << 9;
        wq = lov + csv - 1;
        if (threadIdx.x < N) uk_l[threadIdx.x] = (((a.modq && a.modk) ? (int)a.modk[(size_t)b * N + threadIdx.x] : 0) << 9) + (int)threadIdx.x;
    }
    if constexpr (MASK == FM_MASK_KEYPAD) {
        if (wave < 2) {                                  // waves 0 / 1: keys 0..63 / 64..127
            const unsigned long long bits = __ballot(a.kpad ? a.kpad[(size_t)b * N + threadIdx.x] != 0 : false);
            if (lane == 0) *(unsigned long long*)(uk_l + 2 * wave) = bits;
        }
    }
    __syncthreads();

    // ---- S^T = K Q^T: key block kb, accumulator row r <-> key 32 kb + (r & 3) + 8 (r >> 2) + 4 fhi; the lane's query is the column ----
    f32x16_t st[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, kb * 32 + (lane & 31), kk, fhi), qf[kk], st[kb], 0, 0, 0);
    }
    // ---- mask (blocked -> -inf) and the row maximum of the raw scores ------------------------------------
    float mraw = -INFINITY;
    unsigned bm[4] = {0u, 0u, 0u, 0u};
    if constexpr (MASK == FM_MASK_KEYPAD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bm[i] = (unsigned)__builtin_amdgcn_readfirstlane(uk_l[i]);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            i32x4_t uk4 = {0, 0, 0, 0};
            if constexpr (MASK == FM_MASK_DECODER) uk4 = *(const i32x4_t*)(uk_l + kb * 32 + 8 * g + 4 * fhi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j, cc = j + 8 * g;                  // key = 32 kb + cc + 4 fhi
                float s = st[kb][r];
                if constexpr (MASK == FM_MASK_KEYPAD) {
                    const unsigned long long mk = (unsigned long long)(0u - ((bm[kb] >> cc) & 1u)) | ((unsigned long long)(0u - ((bm[kb] >> (cc + 4)) & 1u)) << 32);
                    s = __builtin_amdgcn_inverse_ballot_w64(mk) ? -INFINITY : s;
                }
                if constexpr (MASK == FM_MASK_DECODER) s = (unsigned)(wq - uk4[j]) >= 255u ? -INFINITY : s;
                st[kb][r] = s;
                mraw = fmaxf(mraw, s);
            }
        }
    mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
    const float c2 = a.scale * LOG2E;                    // > 0 (checked by the dispatch): max(s c2) = c2 max(s)
    bool full = false;
    if constexpr (MASK != FM_MASK_NONE) full = mraw == -INFINITY;
    float m_run = full ? 0.f : mraw * c2;
    const float nm = -m_run;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c2, nm));
            st[kb][r] = p;
            psum += p;
        }
    if constexpr (MASK != FM_MASK_NONE) {
        if (__ballot(full) != 0ull) {                    // rare: empty samples, decoder rows in front of the first visible token
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = full ? 1.0f : st[kb][r];
            psum = full ? 64.f : psum;
            m_run = full ? NEG_FILL : m_run;
        }
    }
    // ---- O^T = V^T P^T ----------------------------------------------------------------------------------
    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            union { bf16x8_t v; uint32_t u[4]; } pb;
#pragma unroll
            for (int i = 0; i < 4; ++i) pb.u[i] = pack2bf(st[kb][8 * s + 2 * i], st[kb][8 * s + 2 * i + 1]);
            const int rA = kb * 32 + s * 16 + 4 * fhi;
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                const bf16x8_t vf = lds_col_frag<true>([&](int r, int c) { return tile_addr(Vt, r, c); }, rA, rA + 8, df * 32);
                o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb.v, o[df], 0, 0, 0);
            }
        }
    float l_tot = psum + __shfl_xor(psum, 32, 64);
    float m_fin = m_run, oscale = 1.0f;
    if (a.zero_attn) {      // softmax1 (fm_utils.py:28-30), as in attn_fwd_kernel
        m_fin = fmaxf(m_run, 0.f);
        oscale = __builtin_amdgcn_exp2f(m_run - m_fin);
        l_tot = l_tot * oscale + __builtin_amdgcn_exp2f(-m_fin);
    }
    const float inv = oscale / l_tot;
#if ATTN_STAGED_STORES
    if (a.o16) {        // (set by the dispatch: O rows 16-byte aligned) whole-line stores through the K tile: every wave is done with it behind the barrier
        __syncthreads();
        store_rows_staged(o, inv, Kt + wave * 4096, a.O + ((size_t)b * N + wave * 32) * a.ldo + h * HD, a.ldo, lane);
    } else
#endif
    {
        bf16_t* orow = a.O + ((size_t)b * N + q) * a.ldo + h * HD;
        const bool wide_o = (a.ldo & 7) == 0 && (((uintptr_t)a.O) & 15) == 0;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint2 po[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    po[u] = make_uint2(pack2bf(o[df][4 * (g + u)] * inv, o[df][4 * (g + u) + 1] * inv),
                                       pack2bf(o[df][4 * (g + u) + 2] * inv, o[df][4 * (g + u) + 3] * inv));
                store_bf16_groups(orow, df * 32 + 8 * g, po[0], po[1], fhi, HD, wide_o);
            }
    }
    if (fhi == 0 && a.stat_m) {
        const size_t si = ((size_t)b * a.H + h) * N + q;
        a.stat_m[si] = m_fin;
        a.stat_l[si] = l_tot;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, Nq and Nk multiples of 128, Nk <= 512 (256 x 256: the 4M-L / mod21 training shapes): attn_fwd128_kernel's per-score arithmetic
// inside an online softmax over key tiles of 128 (two LDS buffers, one barrier per tile).  A blocked score is replaced by NEGR = -2^120 on
// the RAW score: NEGR * c2 is exact for every scale, so in a fully blocked row fma(NEGR, c2, -max) = 0 exactly and p = 1 on every key
// without a fix-up (upstream's uniform row), while next to any real score exp2 underflows to exactly 0; a row maximum below -1e30 is
// reported as NEG_FILL (the backward's "fully blocked" marker).  Decoder rule packed with an 11-bit key field (attn_bwd_kernel's DIET form).
// ------------------------------------------------------------------------------------------------
template <int MASK, bool DB>
__global__ __launch_bounds__(256, DB ? 2 : 3) void attn_fwdt_kernel(AttnArgs a) {
    constexpr int N = 128;
    constexpr float NEGR = -1.329227995784916e36f;       // -2^120
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NT = a.Nk / N;
    char* tiles = smem;                                  // [buffer][K tile 16 KB | V tile 16 KB]
    int* uk_l = (int*)(smem + ((DB && NT > 1) ? 2 : 1) * 2 * N * ROWB);      // decoder: (mod_k << 11) + k per key; key padding: the bitmap of padded keys

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * N + wave * 32 + (lane & 31);
    const bf16_t* Qb = a.Q + (size_t)b * a.Nq * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * a.kvr * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * a.kvr * a.ldv + h * HD;
    auto stage = [&](int t, int buf) {
        char* kt = tiles + buf * 2 * N * ROWB;
        stage_rows<4>(Kb, a.ldk, t * N, a.Nk, N, kt, wave, lane);
        stage_rows<4>(Vb, a.ldv, t * N, a.Nk, N, kt + N * ROWB, wave, lane);
    };
    stage(0, 0);
    bf16x8_t qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8_t*)(Qb + (size_t)q * a.ldq + (kk * 2 + fhi) * 8);
    int wq = 0;                                          // decoder: (mod_q << 11) + cs_q - 1: blocked <=> (unsigned)(wq - uk) >= 1023
    if constexpr (MASK == FM_MASK_DECODER) {
        int csv = 1023, lov = 0;
        if (a.causal) csv = min(q + 1, 1023);                 // keys are < 512: any bound >= 512 shows them all
        else if (a.cs) csv = min(max(a.cs[(size_t)b * a.Nq + q], 0), 1023);
        if (a.modq) lov = (int)a.modq[(size_t)b * a.Nq + q] << 11;
        wq = lov + csv - 1;
        for (int k = threadIdx.x; k < a.Nk; k += 256) uk_l[k] = (((a.modq && a.modk) ? (int)a.modk[(size_t)b * a.Nk + k] : 0) << 11) + k;
    }
    if constexpr (MASK == FM_MASK_KEYPAD) {
        for (int k0 = wave * 64; k0 < a.Nk; k0 += 256) {      // wave-uniform: 64 keys per wave and round
            const unsigned long long bits = __ballot(a.kpad ? a.kpad[(size_t)b * a.Nk + k0 + lane] != 0 : false);
            if (lane == 0) *(unsigned long long*)(uk_l + k0 / 32) = bits;
        }
    }
    __syncthreads();

    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = a.scale * LOG2E;                    // > 0 (checked by the dispatch): max(s c2) = c2 max(s)
    for (int t = 0; t < NT; ++t) {
        const int buf = DB ? (t & 1) : 0;
        if (DB && t + 1 < NT) stage(t + 1, buf ^ 1);
        const char* Kt = tiles + buf * 2 * N * ROWB;
        const char* Vt = Kt + N * ROWB;
        f32x16_t st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kt, kb * 32 + (lane & 31), kk, fhi), qf[kk], st[kb], 0, 0, 0);
        }
        float mraw = NEGR;
        unsigned bm[4] = {0u, 0u, 0u, 0u};
        if constexpr (MASK == FM_MASK_KEYPAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) bm[i] = (unsigned)__builtin_amdgcn_readfirstlane(uk_l[t * 4 + i]);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((ext_vector_type(4))) int i32x4_t;
                i32x4_t uk4 = {0, 0, 0, 0};
                if constexpr (MASK == FM_MASK_DECODER) uk4 = *(const i32x4_t*)(uk_l + t * N + kb * 32 + 8 * g + 4 * fhi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j, cc = j + 8 * g;              // key = 128 t + 32 kb + cc + 4 fhi
                    float s = st[kb][r];
                    if constexpr (MASK == FM_MASK_KEYPAD) {
                        const unsigned long long mk = (unsigned long long)(0u - ((bm[kb] >> cc) & 1u)) | ((unsigned long long)(0u - ((bm[kb] >> (cc + 4)) & 1u)) << 32);
                        s = __builtin_amdgcn_inverse_ballot_w64(mk) ? NEGR : s;
                    }
                    if constexpr (MASK == FM_MASK_DECODER) s = (unsigned)(wq - uk4[j]) >= 1023u ? NEGR : s;
                    st[kb][r] = s;
                    mraw = fmaxf(mraw, s);
                }
            }
        mraw = fmaxf(mraw, __shfl_xor(mraw, 32, 64));
        const float m_new = fmaxf(m_run, mraw * c2);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // first tile: exp2(-inf) = 0
        const float nm = -m_new;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], c2, nm));
                st[kb][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (t > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                union { bf16x8_t v; uint32_t u[4]; } pb;
#pragma unroll
                for (int i = 0; i < 4; ++i) pb.u[i] = pack2bf(st[kb][8 * s + 2 * i], st[kb][8 * s + 2 * i + 1]);
                const int rA = kb * 32 + s * 16 + 4 * fhi;
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const bf16x8_t vf = lds_col_frag<true>([&](int r, int c) { return tile_addr(Vt, r, c); }, rA, rA + 8, df * 32);
                    o[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb.v, o[df], 0, 0, 0);
                }
            }
        if (t + 1 < NT) {
            __syncthreads();
            if (!DB) {      // one buffer (34 KB, three workgroups per CU): the next tile is requested when every wave is done with this one
                stage(t + 1, 0);
                __syncthreads();
            }
        }
    }
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (m_run < -1e30f) m_run = NEG_FILL;                // every key blocked (p = 1 on each of them): the general kernel's marker
    float m_fin = m_run, oscale = 1.0f;
    if (a.zero_attn) {      // softmax1 (fm_utils.py:28-30), as in attn_fwd_kernel
        m_fin = fmaxf(m_run, 0.f);
        oscale = __builtin_amdgcn_exp2f(m_run - m_fin);
        l_tot = l_tot * oscale + __builtin_amdgcn_exp2f(-m_fin);
    }
    const float inv = oscale / l_tot;
#if ATTN_STAGED_STORES
    if (a.o16) {        // whole-line stores through the (dead) K / V buffer, as in attn_fwd128_kernel
        __syncthreads();
        store_rows_staged(o, inv, tiles + wave * 4096, a.O + ((size_t)b * a.Nq + blockIdx.x * N + wave * 32) * a.ldo + h * HD, a.ldo, lane);
    } else
#endif
    {
        bf16_t* orow = a.O + ((size_t)b * a.Nq + q) * a.ldo + h * HD;
        const bool wide_o = (a.ldo & 7) == 0 && (((uintptr_t)a.O) & 15) == 0;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint2 po[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    po[u] = make_uint2(pack2bf(o[df][4 * (g + u)] * inv, o[df][4 * (g + u) + 1] * inv),
                                       pack2bf(o[df][4 * (g + u) + 2] * inv, o[df][4 * (g + u) + 3] * inv));
                store_bf16_groups(orow, df * 32 + 8 * g, po[0], po[1], fhi, HD, wide_o);
            }
    }
    if (fhi == 0 && a.stat_m) {
        const size_t si = ((size_t)b * a.H + h) * a.Nq + q;
        a.stat_m[si] = m_fin;
        a.stat_l[si] = l_tot;
    }
}

// ------------------------------------------------------------------------------------------------
// backward: one workgroup per (b, h).
//   pass A  (a wave owns 32 keys, loops over queries)   -> dK, dV      LDS: (Q, dO)
//   pass B  (a wave owns 32 queries, loops over keys)   -> dQ          LDS: (K, V)
// S and dP are recomputed in both passes, so no atomics and no register-tile transposes.
// Sequences of up to `chunk` (512) rows sit in LDS whole; longer ones (upstream trains 1024 + 1024 tokens,
// cfgs/default/4m/models/main/*1024*) are walked in chunks that are re-staged once per round of 4 x 32 owned rows.
// ------------------------------------------------------------------------------------------------
// DS (sequences of up to 128 x 128: every 128-token 4M configuration): pass A also leaves dS^T (bf16, keys x queries) in LDS as
// 64-query-wide swizzled sub-tiles, and pass B is then dQ = dS K alone - no second QK^T / dO V^T, no second softmax (32 KB more
// LDS: 68 KB per workgroup, still two per CU).
template <bool TR, int MASK, bool CHUNKED, bool DS = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NqP = (a.Nq + 31) & ~31, NkP = (a.Nk + 31) & ~31;
    const int NP = a.chunk;                              // rows per LDS tile
    // !CHUNKED (every 4M training configuration up to 512 tokens): one chunk, known at compile time - the loops below fold to the
    // single-pass kernel, dead waves leave instead of idling through barriers
    const int nQC = CHUNKED ? (NqP + NP - 1) / NP : 1, nKC = CHUNKED ? (NkP + NP - 1) / NP : 1;
    // Two LDS tiles, used twice: (Q, dO) during pass A, then (K, V) during pass B.  The operand a wave keeps in
    // registers for a whole pass (its K/V block in A, its Q/dO block in B) comes straight from global memory.
    // Half the LDS of holding all four tiles -> twice the resident workgroups: this kernel streams 400 MB per
    // launch and needs the memory-level parallelism more than anything else.
    char* T0 = smem;
    char* T1 = T0 + NP * ROWB;
    char* Ql = T0; char* dOl = T1; char* Kl = T0; char* Vl = T1;
    // DIET (round 5; every non-chunked instantiation = sequences up to 512 tokens): the per-score arithmetic of attn_bwd128_kernel - the
    // forward's statistics as one FMA addend, a blocked score selects a per-query exponent, the softmax scale leaves dS (dK / dQ are scaled
    // at their stores), the decoder rule is one unsigned compare, fully blocked rows are handled outside the loops (Q row zeroed in LDS,
    // dQ row zeroed at the store).  The chunked form (up to 32 k tokens, tiles re-staged per round) keeps the round-2 arithmetic.
    constexpr bool DIET = !CHUNKED;
    float4* qs_l = (float4*)(T1 + NP * ROWB);           // per query: {row max (base 2), 1/row sum, delta, cs | mod}; DIET: {nm, pb, delta, wq}
    int32_t* modk_l = (int32_t*)(qs_l + NqP);           // per key: modality id; DIET + decoder mask: (mod_k << 11) + k
    uint8_t* kpad_l = (uint8_t*)(modk_l + NkP);
    uint8_t* full_l = kpad_l + NkP;                      // DIET: the query's row is fully blocked
    char* dSl = smem + (((size_t)((char*)(full_l + NqP) - smem) + 15) & ~(size_t)15);      // DS: sub-tile t = queries [64 t, 64 t + 64)
    static_assert(!DS || !CHUNKED, "dS^T is kept for single-chunk sequences only");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5;
    const int b = blockIdx.y, h = blockIdx.x;

    const bf16_t* Qb = a.Q + (size_t)b * a.Nq * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * a.Nk * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * a.Nk * a.ldv + h * HD;
    const bf16_t* Ob = a.O + (size_t)b * a.Nq * a.ldo + h * HD;
    const bf16_t* dOb = a.dO + (size_t)b * a.Nq * a.lddo + h * HD;

    if (nQC == 1) {
        stage_rows<4>(Qb, a.ldq, 0, a.Nq, NqP, Ql, wave, lane);
        stage_rows<4>(dOb, a.lddo, 0, a.Nq, NqP, dOl, wave, lane);
    }
    // delta[q] = sum_d dO[q][d] * O[q][d]: two threads per query row, 4 x 16-byte loads each from O and dO,
    // all independent (one memory round trip for the whole prologue)
    for (int q0 = 0; q0 < NqP; q0 += 128) {
        const int q = q0 + (threadIdx.x >> 1), half = threadIdx.x & 1;
        float dl = 0.f;
        if (q < a.Nq) {
            const uint4* op = (const uint4*)(Ob + (size_t)q * a.ldo + half * 32);
            const uint4* gp = (const uint4*)(dOb + (size_t)q * a.lddo + half * 32);
            uint4 ov[4], gv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { ov[i] = op[i]; gv[i] = gp[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t* o32 = (const uint32_t*)&ov[i];
                const uint32_t* g32 = (const uint32_t*)&gv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dl = dot2_bf16(o32[e], g32[e], dl);
            }
        }
        dl += __shfl_xor(dl, 1, 64);
        if (half == 0 && q < NqP) {
            const size_t si = ((size_t)b * a.H + h) * a.Nq + (q < a.Nq ? q : 0);
            int csv = 0x7fff, modv = 0;
            if constexpr (MASK == FM_MASK_DECODER) {
                if (q < a.Nq) {
                    if (a.causal) csv = q + 1;
                    else if (a.cs) csv = min(a.cs[(size_t)b * a.Nq + q], 0x7fff);
                    if (a.modq) modv = (uint16_t)a.modq[(size_t)b * a.Nq + q];
                }
            }
            if constexpr (DIET) {
                // nm = -(m + log2 l) (the exponent's addend), pb = log2 of a blocked key's probability (-inf, or -log2 l in a fully blocked
                // row), wq = (mod_q << 11) + cs_q - 1 with cs_q clamped to [0, 1023]; out-of-range queries: nm = pb = -inf (p = 0)
                const float m = q < a.Nq ? a.stat_m[si] : 0.f, l = q < a.Nq ? a.stat_l[si] : 1.f;
                const bool full = MASK != FM_MASK_NONE && q < a.Nq && m < -1e38f;
                const float nm = q >= a.Nq ? -INFINITY : full ? 0.f : -(m + __log2f(l));
                const float pb = full ? -__log2f(l) : -INFINITY;
                const int wq = (int)((unsigned)(int16_t)modv << 11) + min(max(csv == 0x7fff ? 1023 : csv, 0), 1023) - 1;
                qs_l[q] = make_float4(nm, pb, dl, __int_as_float(wq));
                full_l[q] = full;
            } else {
            // out-of-range queries get linv = 0: their probabilities vanish
            qs_l[q] = make_float4(q < a.Nq ? a.stat_m[si] : 0.f, q < a.Nq ? 1.0f / a.stat_l[si] : 0.f, dl,
                                  __int_as_float((csv << 16) | modv));
            }
        }
    }
    for (int k = threadIdx.x; k < NkP; k += 256) {
        const int kc = k < a.Nk ? k : a.Nk - 1;
        const int mk_ = (MASK == FM_MASK_DECODER && a.modk) ? (int)a.modk[(size_t)b * a.Nk + kc] : 0;
        modk_l[k] = DIET ? (int)((unsigned)mk_ << 11) + kc : mk_;
        kpad_l[k] = (MASK == FM_MASK_KEYPAD) ? a.kpad[(size_t)b * a.Nk + kc] : (uint8_t)0;
    }
    __syncthreads();
    if constexpr (DIET && MASK != FM_MASK_NONE) {
        // fully blocked rows: zero their Q row in LDS (no dK contribution; their scores all select pb); the dQ row is zeroed at the store
        bool anyfull = false;
        for (int q0 = 0; q0 < NqP; q0 += 64) anyfull = anyfull || __ballot(q0 + lane < NqP && full_l[q0 + lane]) != 0ull;
        if (anyfull) {
            for (int q = threadIdx.x >> 1; q < NqP; q += 128) {
                if (full_l[q]) {
                    const int half = threadIdx.x & 1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) *(uint4*)(Ql + q * ROWB + half * 64 + i * 16) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            __syncthreads();
        }
    }
    auto sel = [](float if0, float if1, unsigned long long mask) {      // v_cndmask on the EXPONENT (never behind a v_exp: see attn_bwd128_kernel)
        float r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
        return r;
    };
    (void)sel;

    const int nQB = NqP / 32, nKB = NkP / 32;
    const float c2 = a.scale * LOG2E;
    const bool wide_q = (a.lddq & 7) == 0 && (((uintptr_t)a.dQ) & 15) == 0;
    const bool wide_k = (a.lddk & 7) == 0 && (((uintptr_t)a.dK) & 15) == 0;
    const bool wide_v = (a.lddv & 7) == 0 && (((uintptr_t)a.dV) & 15) == 0;

    // ---- pass A: dK, dV -------------------------------------------------------------------------
    for (int kb0 = 0; kb0 < nKB; kb0 += 4) {
        if (!CHUNKED && kb0 + wave >= nKB) break;
        const bool live = kb0 + wave < nKB;              // (a dead wave of the last round still joins the chunk barriers)
        const int kb = live ? kb0 + wave : nKB - 1;
        const int k = kb * 32 + (lane & 31);            // this lane's key
        const int kc = k < a.Nk ? k : a.Nk - 1;
        const int mk = modk_l[kc];
        const bool kp = kpad_l[kc] != 0;
        const unsigned long long kpmask = __ballot(kp), kvalid = __ballot(k < a.Nk);
        (void)kpmask; (void)kvalid;
        bf16x8_t kf[4], vf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                 // rows past Nk repeat the last key (their p is forced to 0 below)
            kf[kk] = *(const bf16x8_t*)(Kb + (size_t)kc * a.ldk + (kk * 2 + fhi) * 8);
            vf[kk] = *(const bf16x8_t*)(Vb + (size_t)kc * a.ldv + (kk * 2 + fhi) * 8);
        }
        f32x16_t dKt[2], dVt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dKt[i][r] = dVt[i][r] = 0.f;
        for (int qc_ = 0; qc_ < nQC; ++qc_) {
        const int qrow0 = qc_ * NP, qrows = min(NP, NqP - qrow0);
        if (nQC > 1) {
            __syncthreads();                             // everyone is done with the previous chunk
            stage_rows<4>(Qb, a.ldq, qrow0, a.Nq, qrows, Ql, wave, lane);
            stage_rows<4>(dOb, a.lddo, qrow0, a.Nq, qrows, dOl, wave, lane);
            __syncthreads();
        }
        for (int qb = qrow0 / 32; qb < (qrow0 + qrows) / 32; ++qb) {
            const int lq = qb * 32 - qrow0;              // LDS row of the block's first query
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ql, lq + (lane & 31), kk, fhi), kf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(dOl, lq + (lane & 31), kk, fhi), vf[kk], dp, 0, 0, 0);
            }
            float pv[16], dsv[16];
            if constexpr (DIET) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                    const float4 qs = qs_l[q];                               // {nm, pb, delta, wq}: one 16-byte broadcast read
                    float t = __builtin_fmaf(s[r], c2, qs.x);
                    if constexpr (MASK == FM_MASK_KEYPAD) t = sel(t, qs.y, kpmask);
                    if constexpr (MASK == FM_MASK_DECODER) t = sel(t, qs.y, __ballot((unsigned)(__float_as_int(qs.w) - mk) >= 1023u));
                    if constexpr (MASK == FM_MASK_DENSE) t = a.dense[((size_t)b * a.Nq + (q < a.Nq ? q : a.Nq - 1)) * a.Nk + kc] != 0 ? qs.y : t;
                    if constexpr (DS) t = sel(-INFINITY, t, kvalid);          // keys past Nk: p = 0 (their dS^T rows feed pass B)
                    const float pr = __builtin_amdgcn_exp2f(t);
                    pv[r] = pr;
                    dsv[r] = pr * (dp[r] - qs.z);                            // (the softmax scale is applied at the dK / dQ stores)
                }
            } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                const float4 qs = qs_l[q];                                   // one 16-byte broadcast read
                float sc = s[r] * c2;
                bool blk = false;
                if constexpr (MASK == FM_MASK_KEYPAD) blk = kp;
                if constexpr (MASK == FM_MASK_DECODER) {
                    const int w = __float_as_int(qs.w);
                    blk = kc >= (w >> 16);
                    if (a.modq) blk = blk || ((w & 0xffff) != (mk & 0xffff));
                }
                if constexpr (MASK == FM_MASK_DENSE) blk = a.dense[((size_t)b * a.Nq + (q < a.Nq ? q : a.Nq - 1)) * a.Nk + kc] != 0;
                sc = blk ? NEG_FILL : sc;
                float pr = __builtin_amdgcn_exp2f(sc - qs.x) * qs.y;
                pr = k < a.Nk ? pr : 0.f;
                pv[r] = pr;
                // masked_fill stops the gradient at blocked scores (they matter only in fully blocked rows)
                dsv[r] = blk ? 0.f : pr * (dp[r] - qs.z) * a.scale;
            }
            }
            if constexpr (DS) {     // this lane's key row, 4 x 4 consecutive queries: 8-byte stores into the swizzled sub-tile
                char* sub = dSl + (qb >> 1) * (NkP * ROWB);
                const int krow = kb * 32 + (lane & 31);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *(uint2*)tile_addr(sub, krow, (qb & 1) * 32 + 8 * g4 + 4 * fhi) =
                        make_uint2(pack2bf(dsv[4 * g4], dsv[4 * g4 + 1]), pack2bf(dsv[4 * g4 + 2], dsv[4 * g4 + 3]));
            }
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const bf16x8_t pb = pack8(&pv[8 * sblk]), db = pack8(&dsv[8 * sblk]);
                const int rA = lq + sblk * 16 + 4 * fhi;
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const bf16x8_t dof = lds_col_frag<TR>([&](int r, int c) { return tile_addr(dOl, r, c); }, rA, rA + 8, df * 32);
                    dVt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pb, dVt[df], 0, 0, 0);
                    const bf16x8_t qcf = lds_col_frag<TR>([&](int r, int c) { return tile_addr(Ql, r, c); }, rA, rA + 8, df * 32);
                    dKt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qcf, db, dKt[df], 0, 0, 0);
                }
            }
        }
        }
        if (live) {   // lanes l and l+32 own the same key row: 16-byte stores through v_permlane32_swap (store_bf16_groups)
            bf16_t* dkrow = a.dK + ((size_t)b * a.Nk + kc) * a.lddk + h * HD;
            bf16_t* dvrow = a.dV + ((size_t)b * a.Nk + kc) * a.lddv + h * HD;
            const int lim = k < a.Nk ? HD : 0;           // rows past Nk write nothing
            // one output after the other: the four 32-byte pieces of a row's 128-byte line leave back to back
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int df = 0; df < 2; ++df)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const f32x16_t& t = which ? dVt[df] : dKt[df];
                        const float ks = (DIET && !which) ? a.scale : 1.0f;      // DIET: dS left pass A without the softmax scale
                        uint2 pk[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            pk[u] = make_uint2(pack2bf(t[4 * (g + u)] * ks, t[4 * (g + u) + 1] * ks), pack2bf(t[4 * (g + u) + 2] * ks, t[4 * (g + u) + 3] * ks));
                        store_bf16_groups(which ? dvrow : dkrow, df * 32 + 8 * g, pk[0], pk[1], fhi, lim, which ? wide_v : wide_k);
                    }
        }
    }

    // ---- (K, V) take the place of (Q, dO) in LDS -------------------------------------------------
    __syncthreads();
    if (nKC == 1) {
        stage_rows<4>(Kb, a.ldk, 0, a.Nk, NkP, Kl, wave, lane);
        if constexpr (!DS) stage_rows<4>(Vb, a.ldv, 0, a.Nk, NkP, Vl, wave, lane);
        __syncthreads();
    }

    // ---- pass B: dQ -----------------------------------------------------------------------------
    for (int qb0 = 0; qb0 < nQB; qb0 += 4) {
        if (!CHUNKED && qb0 + wave >= nQB) break;
        const bool live = qb0 + wave < nQB;
        const int qb = live ? qb0 + wave : nQB - 1;
        const int q = qb * 32 + (lane & 31);
        const int qc = q < a.Nq ? q : a.Nq - 1;
        const float4 qs = qs_l[q];
        const float mq_ = qs.x, li = qs.y, dl = qs.z;                          // DIET: nm, pb, delta
        const int csq = __float_as_int(qs.w) >> 16, mq = __float_as_int(qs.w) & 0xffff;
        const int wq = __float_as_int(qs.w);                                   // DIET: the packed decoder constant
        f32x16_t dQt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        if constexpr (DS) {     // dQ[q][:] = sum_k dS[q][k] K[k][:] with dS^T read column-wise (queries = the MFMA's n index)
            const char* sub = dSl + (qb >> 1) * (NkP * ROWB);
            for (int kb = 0; kb < nKB; ++kb)
#pragma unroll
                for (int sblk = 0; sblk < 2; ++sblk) {
                    const int rA = kb * 32 + sblk * 16 + 4 * fhi;
                    const bf16x8_t db = lds_col_frag<TR>([&](int r, int c) { return tile_addr(sub, r, c); }, rA, rA + 8, (qb & 1) * 32);
#pragma unroll
                    for (int df = 0; df < 2; ++df) {
                        const bf16x8_t kcf = lds_col_frag<TR>([&](int r, int c) { return tile_addr(Kl, r, c); }, rA, rA + 8, df * 32);
                        dQt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcf, db, dQt[df], 0, 0, 0);
                    }
                }
        } else {
        bf16x8_t qf[4], dof[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = *(const bf16x8_t*)(Qb + (size_t)qc * a.ldq + (kk * 2 + fhi) * 8);
            dof[kk] = *(const bf16x8_t*)(dOb + (size_t)qc * a.lddo + (kk * 2 + fhi) * 8);
        }
        for (int kc_ = 0; kc_ < nKC; ++kc_) {
        const int krow0 = kc_ * NP, krows = min(NP, NkP - krow0);
        if (nKC > 1) {
            __syncthreads();
            stage_rows<4>(Kb, a.ldk, krow0, a.Nk, krows, Kl, wave, lane);
            stage_rows<4>(Vb, a.ldv, krow0, a.Nk, krows, Vl, wave, lane);
            __syncthreads();
        }
        for (int kb = krow0 / 32; kb < (krow0 + krows) / 32; ++kb) {
            const int lk = kb * 32 - krow0;              // LDS row of the block's first key
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Kl, lk + (lane & 31), kk, fhi), qf[kk], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vl, lk + (lane & 31), kk, fhi), dof[kk], dp, 0, 0, 0);
            }
            float dsv[16];
            unsigned long long bits = 0;
            if constexpr (MASK == FM_MASK_KEYPAD) bits = (kb * 32 < NkP ? __ballot(kpad_l[min(kb * 32 + (lane & 31), NkP - 1)] != 0) : 0ull) >> (4 * fhi);
            if constexpr (DIET) {
                const bool ragged = (kb + 1) * 32 > a.Nk;                    // (only the last key block can hold keys past Nk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2);
                    const int k = kb * 32 + c + 4 * fhi;
                    const int kc = k < a.Nk ? k : a.Nk - 1;
                    float t = __builtin_fmaf(s[r], c2, mq_);
                    if constexpr (MASK == FM_MASK_KEYPAD) t = sel(t, li, __ballot((bits >> c) & 1ull));
                    if constexpr (MASK == FM_MASK_DECODER) t = sel(t, li, __ballot((unsigned)(wq - modk_l[kc]) >= 1023u));
                    if constexpr (MASK == FM_MASK_DENSE) t = a.dense[((size_t)b * a.Nq + qc) * a.Nk + kc] != 0 ? li : t;
                    if (ragged) t = k < a.Nk ? t : -INFINITY;
                    dsv[r] = __builtin_amdgcn_exp2f(t) * (dp[r] - dl);
                }
            } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2);
                const int k = kb * 32 + c + 4 * fhi;
                const int kc = k < a.Nk ? k : a.Nk - 1;
                float sc = s[r] * c2;
                bool blk = false;
                if constexpr (MASK == FM_MASK_KEYPAD) blk = (bits >> c) & 1ull;
                if constexpr (MASK == FM_MASK_DECODER) {
                    blk = kc >= csq;
                    if (a.modq) blk = blk || (mq != (modk_l[kc] & 0xffff));
                }
                if constexpr (MASK == FM_MASK_DENSE) blk = a.dense[((size_t)b * a.Nq + qc) * a.Nk + kc] != 0;
                sc = blk ? NEG_FILL : sc;
                float pr = __builtin_amdgcn_exp2f(sc - mq_) * li;
                pr = k < a.Nk ? pr : 0.f;
                dsv[r] = blk ? 0.f : pr * (dp[r] - dl) * a.scale;
            }
            }
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const bf16x8_t db = pack8(&dsv[8 * sblk]);
                const int rA = lk + sblk * 16 + 4 * fhi;
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const bf16x8_t kcf = lds_col_frag<TR>([&](int r, int c) { return tile_addr(Kl, r, c); }, rA, rA + 8, df * 32);
                    dQt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcf, db, dQt[df], 0, 0, 0);
                }
            }
        }
        }
        }
        (void)mq_; (void)li; (void)dl; (void)csq; (void)mq; (void)wq;
        if (live) {
            bf16_t* dqrow = a.dQ + ((size_t)b * a.Nq + qc) * a.lddq + h * HD;
            const int lim = q < a.Nq ? HD : 0;
            float qsc = 1.0f;                                                  // DIET: the softmax scale; 0 for a fully blocked row
            if constexpr (DIET) qsc = (MASK != FM_MASK_NONE && full_l[q]) ? 0.f : a.scale;
#pragma unroll
            for (int df = 0; df < 2; ++df)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    uint2 pq[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        pq[u] = make_uint2(pack2bf(dQt[df][4 * (g + u)] * qsc, dQt[df][4 * (g + u) + 1] * qsc), pack2bf(dQt[df][4 * (g + u) + 2] * qsc, dQt[df][4 * (g + u) + 3] * qsc));
                    store_bf16_groups(dqrow, df * 32 + 8 * g, pq[0], pq[1], fhi, lim, wide_q);
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// backward, 128 x 128 tokens exactly (every 128-token 4M configuration: the bench shape) - round 5.
// Same two passes and the same dS^T hand-off as attn_bwd_kernel<.., DS = true>, re-cut around what the round-4 counters showed
// (profiles/r04_pmc_attn.txt: 1 800 VALU instructions per wave = 43 % of a SIMD's issue time with two waves on it, 23 % LDS conflict cycles):
//   * 5 - 8 VALU per score instead of 15: the forward's (max, sum) enter as ONE addend nm = -(m + log2 l) of the exponent's FMA, the
//     softmax scale leaves dS (dK and dQ are scaled once at their stores), a blocked score selects a per-QUERY constant pb
//     (0, or 1 / l in a fully blocked row: exactly what exp2(NEG_FILL - m) / l evaluates to), the decoder rule is one unsigned compare of
//     (mod_k << 8 | k) - (mod_q << 8) against cs_q, and fully blocked rows are handled OUTSIDE the loop (their Q row is zeroed in LDS, their
//     dQ row at the store: masked_fill passes no gradient) so dS needs no select;
//   * per-query constants as separate arrays read with one ds_read_b128 per 4 queries (16 ds_read_b96 per q-block before: 8 LDS cycles each);
//   * the q-block loop unrolled: every LDS address is a lane constant + an immediate;
//   * K / V fragments requested in the prologue with everything else (ONE memory round trip), and K reaches pass B from the registers that
//     hold it (4 ds_write_b128 per lane) instead of through a third dependent global round trip.
// ------------------------------------------------------------------------------------------------
template <int MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd128_kernel(AttnArgs a) {
    constexpr int N = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* T0 = smem;                                     // Q, then K
    char* T1 = smem + N * ROWB;                          // dO
    char* dSl = smem + 2 * N * ROWB;                     // dS^T: 2 sub-tiles (64 queries each) x 128 key rows x 128 bytes
    float* nm_l = (float*)(smem + 4 * N * ROWB);         // -(row max + log2 row sum)
    float* dl_l = nm_l + N;                              // delta = sum_d dO O
    float* pb_l = dl_l + N;                              // log2 of the probability of a blocked key: -inf, or -log2 l in a fully blocked row
    int* wq_l = (int*)(pb_l + N);                        // decoder: (mod_q << 9) + cs_q - 1, cs_q = the query's visible-key bound clamped to [0, 255]
    int* full_l = wq_l + N;                              // the row is fully blocked
    int* uk_l = full_l + N;                              // per key - decoder: (mod_k << 9) + k, key padding: blocked
    // decoder rule in one subtraction: (unsigned)(wq - uk) < 255  <=>  same modality and k < cs_q   (|cs - k - 1| < 256 < 512 = one modality step)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5;
    const int b = blockIdx.y, h = blockIdx.x;
    const bf16_t* Qb = a.Q + (size_t)b * N * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * N * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * N * a.ldv + h * HD;
    const bf16_t* Ob = a.O + (size_t)b * N * a.ldo + h * HD;
    const bf16_t* dOb = a.dO + (size_t)b * N * a.lddo + h * HD;

    stage_rows<4, ATTN_BWD_SAVED_NT>(Qb, a.ldq, 0, N, N, T0, wave, lane);
    stage_rows<4, ATTN_BWD_DO_NT>(dOb, a.lddo, 0, N, N, T1, wave, lane);
    const int key = wave * 32 + (lane & 31);             // pass A: this lane's key
    bf16x8_t kf[4], vf[4];
#if ATTN_KV_DMA
    // K and V reach their registers through the (still unused) dS^T area: whole 128-byte lines by LDS-DMA instead of 32-byte pieces per row and
    // instruction (the fragment loads were 1 024 of a workgroup's ~1 900 L2 requests)
    stage_rows<4, ATTN_BWD_SAVED_NT>(Kb, a.ldk, 0, N, N, dSl, wave, lane);
    stage_rows<4, ATTN_BWD_SAVED_NT>(Vb, a.ldv, 0, N, N, dSl + N * ROWB, wave, lane);
#else
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *(const bf16x8_t*)(Kb + (size_t)key * a.ldk + (kk * 2 + fhi) * 8);
        vf[kk] = *(const bf16x8_t*)(Vb + (size_t)key * a.ldv + (kk * 2 + fhi) * 8);
    }
#endif
#if ATTN_DELTA_LDS && ATTN_KV_DMA
    // delta = rowsum(dO o O): O as whole lines (8 lanes per row, rows 32 i + (t >> 3)), dO from its LDS tile behind the barrier.  The two-threads-
    // per-row form below asked the L2 for 16 bytes per lane and instruction, 64 bytes apart: 2 048 sixteen-byte requests per workgroup, more than
    // everything else of the kernel together.
    uint4 ovl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ovl[i] = *(const uint4*)(Ob + (size_t)(32 * i + (threadIdx.x >> 3)) * a.ldo + (threadIdx.x & 7) * 8);
#endif
    {   // per-query constants: two threads per query row (4 x 16-byte loads each from O and dO)
        const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
#if !(ATTN_DELTA_LDS && ATTN_KV_DMA)
        const uint4* op = (const uint4*)(Ob + (size_t)q * a.ldo + half * 32);
        const uint4* gp = (const uint4*)(dOb + (size_t)q * a.lddo + half * 32);
        uint4 ov[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ov[i] = op[i]; gv[i] = gp[i]; }
#endif
        const size_t si = ((size_t)b * a.H + h) * N + q;
        const float m = a.stat_m[si], l = a.stat_l[si];
        int csv = 255, lov = 0;
        if constexpr (MASK == FM_MASK_DECODER) {
            if (a.causal) csv = q + 1;
            else if (a.cs) csv = min(max(a.cs[(size_t)b * N + q], 0), 255);
            if (a.modq) lov = (int)a.modq[(size_t)b * N + q] << 9;
        }
        if (threadIdx.x < N) {
            const int k = threadIdx.x;
            if constexpr (MASK == FM_MASK_DECODER) uk_l[k] = ((a.modk ? (int)a.modk[(size_t)b * N + k] : 0) << 9) + k;
            else if constexpr (MASK == FM_MASK_KEYPAD) uk_l[k] = a.kpad ? a.kpad[(size_t)b * N + k] != 0 : 0;
        }
#if !(ATTN_DELTA_LDS && ATTN_KV_DMA)
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t* o32 = (const uint32_t*)&ov[i];
            const uint32_t* g32 = (const uint32_t*)&gv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl = dot2_bf16(o32[e], g32[e], dl);
        }
        dl += __shfl_xor(dl, 1, 64);
#endif
        if (half == 0) {
            const bool full = MASK != FM_MASK_NONE && m < -1e38f;      // the forward kept NEG_FILL as the maximum: every key was blocked
            nm_l[q] = full ? 0.f : -(m + __log2f(l));
#if !(ATTN_DELTA_LDS && ATTN_KV_DMA)
            dl_l[q] = dl;
#endif
            pb_l[q] = full ? -__log2f(l) : -INFINITY;       // log2 of a blocked key's probability: exp2 -> 1 / l, or exactly 0
            wq_l[q] = lov + csv - 1; full_l[q] = full;
        }
    }
    __syncthreads();
#if ATTN_KV_DMA
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = row_frag(dSl, key, kk, fhi);
        vf[kk] = row_frag(dSl + N * ROWB, key, kk, fhi);
    }
#if ATTN_DELTA_LDS
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 32 * i + (threadIdx.x >> 3), c = threadIdx.x & 7;
        const uint4 gv = *(const uint4*)(T1 + row * ROWB + ((c ^ sw3(row)) << 4));
        float d = dot2_bf16(ovl[i].x, gv.x, 0.f);
        d = dot2_bf16(ovl[i].y, gv.y, d); d = dot2_bf16(ovl[i].z, gv.z, d); d = dot2_bf16(ovl[i].w, gv.w, d);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (c == 0) dl_l[row] = d;
    }
#endif
    __syncthreads();                                     // every wave holds its K / V fragments: the area may take dS^T
#endif
    if constexpr (MASK != FM_MASK_NONE) {
        // fully blocked rows (empty samples, decoder rows in front of the first visible token): P = 1 / l on every key, no gradient through
        // the scores.  Their Q row is zeroed here (no dK contribution; S is irrelevant, every score selects pb), their dQ row at the store.
        if (__ballot(full_l[lane] | full_l[lane + 64]) != 0ull) {
            const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
            if (full_l[q]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint4*)(T0 + q * ROWB + half * 64 + i * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
    }

    // LDS addresses = lane constant + immediate from here on (offsets relative to smem; the q-block / key-block loops are unrolled)
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    const lds_ptr L = (lds_ptr)smem;
    constexpr uint32_t oT0 = 0, oT1 = N * ROWB, oDS = 2 * N * ROWB, oNM = 4 * N * ROWB, oDL = oNM + 4 * N, oPB = oDL + 4 * N, oWQ = oPB + 4 * N,
                       oUK = oWQ + 8 * N;
    const int r31 = lane & 31;
    uint32_t rowo[4];                                    // row fragment (MFMA A operand) of row r31, k-step kk
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rowo[kk] = r31 * ROWB + (((kk * 2 + fhi) ^ sw3(r31)) << 4);
    uint32_t colo[2][2];                                 // transpose-read base of rows 4 fhi + 8 e + (i >> 2), columns 32 df + ...
    {
        const int i = lane & 15;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                const int rl = 4 * fhi + 8 * e + (i >> 2), c = df * 32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
                colo[e][df] = rl * ROWB + (((c >> 3) ^ sw3(rl)) << 4) + (c & 7) * 2;
            }
    }
    auto col_frag = [&](uint32_t tile, int e0df, int rows) -> bf16x8_t {   // rows rows + 4 fhi + {0..3, 8..11}, columns of block df
        union { bf16x8_t v; s16x4_t h[2]; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + colo[0][e0df] + tile + rows * ROWB));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + colo[1][e0df] + tile + rows * ROWB));
        return u.v;
    };
    const float c2 = a.scale * LOG2E;
    // lane bit set = this lane's key is blocked for ... KEYPAD: every query (a loop constant in two SGPRs); DECODER: per score (v_cmp result)
    unsigned long long kpmask = 0;
    int ukey = 0;
    if constexpr (MASK == FM_MASK_DECODER) ukey = *(const __attribute__((address_space(3))) int*)(L + oUK + 4 * key);
    if constexpr (MASK == FM_MASK_KEYPAD) kpmask = __ballot(*(const __attribute__((address_space(3))) int*)(L + oUK + 4 * key) != 0);
    // v_cndmask, never a branch (hipcc turns the ternary on a loop constant into control flow).  Applied to the EXPONENT, in front of
    // v_exp_f32: hipcc's hazard recogniser does not see inside an asm statement, and a VALU reading a transcendental's result in the next
    // issue slot gets the stale register on gfx950 (the first form of this kernel selected behind the v_exp and failed exactly so).
    auto sel = [](float if0, float if1, unsigned long long mask) {
        float r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
        return r;
    };
    const uint32_t dswo = key * ROWB + (sw3(key) << 4) + 8 * fhi;        // dS^T store of key row `key`: chunk c at dswo ^ (c << 4)

    // ---- pass A: this wave's 32 keys against every query -> dK, dV, dS^T ---------------------------
    f32x16_t dKt[2], dVt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dKt[i][r] = dVt[i][r] = 0.f;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const __attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + oT0 + qb * 32 * ROWB), kf[kk], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const __attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + oT1 + qb * 32 * ROWB), vf[kk], dp, 0, 0, 0);
        }
        float pv[16], dsv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t qo = 4 * (qb * 32 + 8 * g) + 16 * fhi;        // accumulator rows 4 g .. 4 g + 3 = queries 32 qb + 8 g + 4 fhi + 0..3
            const f32x4_t nm4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oNM + qo);
            const f32x4_t dl4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oDL + qo);
            f32x4_t pb4 = {0.f, 0.f, 0.f, 0.f};
            typedef __attribute__((ext_vector_type(4))) int i32x4_t;
            i32x4_t wq4 = {0, 0, 0, 0};
            if constexpr (MASK != FM_MASK_NONE) pb4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oPB + qo);
            if constexpr (MASK == FM_MASK_DECODER) wq4 = *(const __attribute__((address_space(3))) i32x4_t*)(L + oWQ + qo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float t = __builtin_fmaf(s[r], c2, nm4[j]);
                if constexpr (MASK == FM_MASK_KEYPAD) t = sel(t, pb4[j], kpmask);
                if constexpr (MASK == FM_MASK_DECODER) t = sel(t, pb4[j], __ballot((unsigned)(wq4[j] - ukey) >= 255u));
                const float p = __builtin_amdgcn_exp2f(t);
                pv[r] = p;
                dsv[r] = p * (dp[r] - dl4[j]);
            }
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)      // this lane's key row, 4 x 4 consecutive queries: 8-byte stores into the swizzled sub-tile
            *(__attribute__((address_space(3))) u32x2_t*)(L + ((dswo ^ ((((qb & 1) * 4 + g4)) << 4)) + oDS + (qb >> 1) * (N * ROWB))) =
                u32x2_t{pack2bf(dsv[4 * g4], dsv[4 * g4 + 1]), pack2bf(dsv[4 * g4 + 2], dsv[4 * g4 + 3])};
#pragma unroll
        for (int sblk = 0; sblk < 2; ++sblk) {
            const bf16x8_t pb = pack8(&pv[8 * sblk]), db = pack8(&dsv[8 * sblk]);
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                dVt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(oT1, df, qb * 32 + sblk * 16), pb, dVt[df], 0, 0, 0);
                dKt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(oT0, df, qb * 32 + sblk * 16), db, dKt[df], 0, 0, 0);
            }
        }
#if ATTN_V2_SCHED == 1
        __builtin_amdgcn_sched_barrier(0);      // lab variant: no instruction crosses a q-block boundary
#endif
    }
#if !ATTN_STAGED_STORES
    {   // lanes l and l+32 own the same key row: 16-byte stores (store_bf16_groups), one output after the other
        const bool wide_k = (a.lddk & 7) == 0 && (((uintptr_t)a.dK) & 15) == 0;
        const bool wide_v = (a.lddv & 7) == 0 && (((uintptr_t)a.dV) & 15) == 0;
        bf16_t* dkrow = a.dK + ((size_t)b * N + key) * a.lddk + h * HD;
        bf16_t* dvrow = a.dV + ((size_t)b * N + key) * a.lddv + h * HD;
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int df = 0; df < 2; ++df)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const f32x16_t& t = which ? dVt[df] : dKt[df];
                    const float sc = which ? 1.0f : a.scale;            // dS left pass A without the softmax scale
                    uint2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        pk[u] = make_uint2(pack2bf(t[4 * (g + u)] * sc, t[4 * (g + u) + 1] * sc), pack2bf(t[4 * (g + u) + 2] * sc, t[4 * (g + u) + 3] * sc));
                    store_bf16_groups(which ? dvrow : dkrow, df * 32 + 8 * g, pk[0], pk[1], fhi, HD, which ? wide_v : wide_k);
                }
    }

#endif
    // ---- K takes the place of Q in LDS, from the registers that hold it ---------------------------
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) *(__attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + oT0 + wave * 32 * ROWB) = kf[kk];
#if ATTN_STAGED_STORES
    // dK, dV: behind the barrier every wave is done with the dO tile - 4 KB of it per wave stage the outputs into whole-line stores
    store_rows_staged(dKt, a.scale, T1 + wave * 4096, a.dK + ((size_t)b * N + wave * 32) * a.lddk + h * HD, a.lddk, lane);      // (dS left pass A without the softmax scale)
    store_rows_staged(dVt, 1.0f, T1 + wave * 4096, a.dV + ((size_t)b * N + wave * 32) * a.lddv + h * HD, a.lddv, lane);
#endif
    __syncthreads();

    // ---- pass B: dQ = dS K for this wave's 32 queries (dS^T read column-wise: queries = the MFMA's n index) ---
    {
        const int qb = wave;
        const int q = qb * 32 + r31;
        f32x16_t dQt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const uint32_t sub = oDS + (qb >> 1) * (N * ROWB);
        // the wave's 32-query half of the sub-tile: colo[.][qb & 1] without a dynamically indexed register array
        const uint32_t dsc0 = (qb & 1) ? colo[0][1] : colo[0][0], dsc1 = (qb & 1) ? colo[1][1] : colo[1][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const int rows = kb * 32 + sblk * 16;
                union { bf16x8_t v; s16x4_t h[2]; } db;
                db.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + dsc0 + sub + rows * ROWB));
                db.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + dsc1 + sub + rows * ROWB));
#pragma unroll
                for (int df = 0; df < 2; ++df) dQt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(oT0, df, rows), db.v, dQt[df], 0, 0, 0);
            }
        float sc = a.scale;
        if constexpr (MASK != FM_MASK_NONE) sc = full_l[q] ? 0.f : sc;
#if ATTN_STAGED_STORES
        store_rows_staged(dQt, sc, T1 + wave * 4096, a.dQ + ((size_t)b * N + qb * 32) * a.lddq + h * HD, a.lddq, lane);
        return;
#endif
        const bool wide_q = (a.lddq & 7) == 0 && (((uintptr_t)a.dQ) & 15) == 0;
        bf16_t* dqrow = a.dQ + ((size_t)b * N + q) * a.lddq + h * HD;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint2 pq[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    pq[u] = make_uint2(pack2bf(dQt[df][4 * (g + u)] * sc, dQt[df][4 * (g + u) + 1] * sc),
                                       pack2bf(dQt[df][4 * (g + u) + 2] * sc, dQt[df][4 * (g + u) + 3] * sc));
                store_bf16_groups(dqrow, df * 32 + 8 * g, pq[0], pq[1], fhi, HD, wide_q);
            }
    }
}


// ------------------------------------------------------------------------------------------------
// The same backward with 51 KB of LDS and <= 168 registers: THREE workgroups per CU instead of two (the r05 counters of attn_bwd128_kernel:
// 52 % of the wave cycles are issue stalls on dependent MFMA / VALU chains, 24 % waits - more resident waves is what covers them).
//   * dS^T overlays the rows it was computed from: region qb (8 KB) holds Q[32 qb ..] and dO[32 qb ..] (32 rows x 128 bytes each) until
//     every wave has multiplied q-block qb (one barrier per q-block), then the 128 keys x 32 queries of dS^T (64-byte rows);
//   * K has its own 16 KB tile from the prologue on (pass B's operand; pass A re-reads its 32 key rows per q-block instead of holding
//     them in 16 registers); V stays in registers.
// ------------------------------------------------------------------------------------------------
template <int MASK>
__global__ __launch_bounds__(256, 3) void attn_bwd128o_kernel(AttnArgs a) {
    constexpr int N = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_ptr;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    const lds_ptr L = (lds_ptr)smem;
    constexpr uint32_t RG = 8192, oK = 4 * RG, oNM = oK + N * ROWB, oDL = oNM + 4 * N, oPB = oDL + 4 * N, oWQ = oPB + 4 * N, oFULL = oWQ + 4 * N,
                       oUK = oFULL + 4 * N;
    float* nm_l = (float*)(smem + oNM); float* dl_l = (float*)(smem + oDL); float* pb_l = (float*)(smem + oPB);
    int* wq_l = (int*)(smem + oWQ); int* full_l = (int*)(smem + oFULL); int* uk_l = (int*)(smem + oUK);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fhi = lane >> 5, r31 = lane & 31;
    const int b = blockIdx.y, h = blockIdx.x;
    const bf16_t* Qb = a.Q + (size_t)b * N * a.ldq + h * HD;
    const bf16_t* Kb = a.K + (size_t)b * N * a.ldk + h * HD;
    const bf16_t* Vb = a.V + (size_t)b * N * a.ldv + h * HD;
    const bf16_t* Ob = a.O + (size_t)b * N * a.ldo + h * HD;
    const bf16_t* dOb = a.dO + (size_t)b * N * a.lddo + h * HD;

    // LDS-DMA: piece p = rows 8 p .. 8 p + 7 of a tile; Q / dO pieces land in their q-block's region, K in its own tile
    for (int p = wave; p < N / 8; p += 4) {
        const int t = p * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ sw3(t);
        const uint32_t dst = (p >> 2) * RG + (p & 3) * 1024;
        __builtin_amdgcn_global_load_lds(GLB_PTR(Qb + (size_t)t * a.ldq + lc * 8), LDS_PTR(smem + dst), 16, 0, ATTN_BWD_SAVED_NT);      // (saved q / k: read once)
        __builtin_amdgcn_global_load_lds(GLB_PTR(dOb + (size_t)t * a.lddo + lc * 8), LDS_PTR(smem + dst + 4096), 16, 0, ATTN_BWD_DO_NT);
        __builtin_amdgcn_global_load_lds(GLB_PTR(Kb + (size_t)t * a.ldk + lc * 8), LDS_PTR(smem + oK + p * 1024), 16, 0, ATTN_BWD_SAVED_NT);
    }
    const int key = wave * 32 + r31;                     // pass A: this lane's key
    bf16x8_t vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) vf[kk] = *(const bf16x8_t*)(Vb + (size_t)key * a.ldv + (kk * 2 + fhi) * 8);
    {   // per-query constants: two threads per query row (4 x 16-byte loads each from O and dO)
        const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
        const uint4* op = (const uint4*)(Ob + (size_t)q * a.ldo + half * 32);
        const uint4* gp = (const uint4*)(dOb + (size_t)q * a.lddo + half * 32);
        uint4 ov[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ov[i] = op[i]; gv[i] = gp[i]; }
        const size_t si = ((size_t)b * a.H + h) * N + q;
        const float m = a.stat_m[si], l = a.stat_l[si];
        int csv = 255, lov = 0;
        if constexpr (MASK == FM_MASK_DECODER) {
            if (a.causal) csv = q + 1;
            else if (a.cs) csv = min(max(a.cs[(size_t)b * N + q], 0), 255);
            if (a.modq) lov = (int)a.modq[(size_t)b * N + q] << 9;
        }
        if (threadIdx.x < N) {
            const int k = threadIdx.x;
            if constexpr (MASK == FM_MASK_DECODER) uk_l[k] = ((a.modk ? (int)a.modk[(size_t)b * N + k] : 0) << 9) + k;
            else if constexpr (MASK == FM_MASK_KEYPAD) uk_l[k] = a.kpad ? a.kpad[(size_t)b * N + k] != 0 : 0;
        }
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t* o32 = (const uint32_t*)&ov[i];
            const uint32_t* g32 = (const uint32_t*)&gv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl = dot2_bf16(o32[e], g32[e], dl);
        }
        dl += __shfl_xor(dl, 1, 64);
        if (half == 0) {
            const bool full = MASK != FM_MASK_NONE && m < -1e38f;
            nm_l[q] = full ? 0.f : -(m + __log2f(l));
            dl_l[q] = dl;
            pb_l[q] = full ? -__log2f(l) : -INFINITY;
            wq_l[q] = lov + csv - 1; full_l[q] = full;
        }
    }
    __syncthreads();
    if constexpr (MASK != FM_MASK_NONE) {
        if (__ballot(full_l[lane] | full_l[lane + 64]) != 0ull) {      // fully blocked rows: zero their Q row (see attn_bwd128_kernel)
            const int q = threadIdx.x >> 1, half = threadIdx.x & 1;
            if (full_l[q]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint4*)(smem + (q >> 5) * RG + (q & 31) * ROWB + half * 64 + i * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
        }
    }

    uint32_t rowo[4];                                    // row fragment (MFMA A operand) of row r31 of a 32-row block, k-step kk
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) rowo[kk] = r31 * ROWB + (((kk * 2 + fhi) ^ sw3(r31)) << 4);
    uint32_t colo[2][2], dsco[2];                        // transpose-read bases: 128-byte rows (Q, dO, K) and 64-byte rows (dS^T)
    {
        const int i = lane & 15;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rl = 4 * fhi + 8 * e + (i >> 2);
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                const int c = df * 32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
                colo[e][df] = rl * ROWB + (((c >> 3) ^ sw3(rl)) << 4) + (c & 7) * 2;
            }
            const int c = ((lane >> 4) & 1) * 16 + (i & 3) * 4;
            dsco[e] = rl * 64 + (((c >> 3) ^ ((rl >> 1) & 3)) << 4) + (c & 7) * 2;
        }
    }
    auto col_frag = [&](uint32_t base, int df) -> bf16x8_t {       // rows base + 4 fhi + {0..3, 8..11} of a 128-byte-row tile, columns of block df
        union { bf16x8_t v; s16x4_t h[2]; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + colo[0][df] + base));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + colo[1][df] + base));
        return u.v;
    };
    const float c2 = a.scale * LOG2E;
    unsigned long long kpmask = 0;
    int ukey = 0;
    if constexpr (MASK == FM_MASK_DECODER) ukey = *(const __attribute__((address_space(3))) int*)(L + oUK + 4 * key);
    if constexpr (MASK == FM_MASK_KEYPAD) kpmask = __ballot(*(const __attribute__((address_space(3))) int*)(L + oUK + 4 * key) != 0);
    auto sel = [](float if0, float if1, unsigned long long mask) {      // (see attn_bwd128_kernel: never behind a v_exp)
        float r;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
        return r;
    };
    const uint32_t dswo = key * 64 + (((key >> 1) & 3) << 4) + 8 * fhi;       // dS^T store of key row `key`: chunk c at dswo ^ (c << 4)

    // ---- pass A ----------------------------------------------------------------------------------
    f32x16_t dKt[2], dVt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dKt[i][r] = dVt[i][r] = 0.f;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t kf = *(const __attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + oK + wave * 32 * ROWB);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const __attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + qb * RG), kf, s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const __attribute__((address_space(3))) bf16x8_t*)(L + rowo[kk] + qb * RG + 4096), vf[kk], dp, 0, 0, 0);
        }
        float pv[16], dsv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t qo = 4 * (qb * 32 + 8 * g) + 16 * fhi;
            const f32x4_t nm4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oNM + qo);
            const f32x4_t dl4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oDL + qo);
            f32x4_t pb4 = {0.f, 0.f, 0.f, 0.f};
            i32x4_t wq4 = {0, 0, 0, 0};
            if constexpr (MASK != FM_MASK_NONE) pb4 = *(const __attribute__((address_space(3))) f32x4_t*)(L + oPB + qo);
            if constexpr (MASK == FM_MASK_DECODER) wq4 = *(const __attribute__((address_space(3))) i32x4_t*)(L + oWQ + qo);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j;
                float t = __builtin_fmaf(s[r], c2, nm4[j]);
                if constexpr (MASK == FM_MASK_KEYPAD) t = sel(t, pb4[j], kpmask);
                if constexpr (MASK == FM_MASK_DECODER) t = sel(t, pb4[j], __ballot((unsigned)(wq4[j] - ukey) >= 255u));
                const float p = __builtin_amdgcn_exp2f(t);
                pv[r] = p;
                dsv[r] = p * (dp[r] - dl4[j]);
            }
        }
        u32x2_t dsp[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dsp[g4] = u32x2_t{pack2bf(dsv[4 * g4], dsv[4 * g4 + 1]), pack2bf(dsv[4 * g4 + 2], dsv[4 * g4 + 3])};
#pragma unroll
        for (int sblk = 0; sblk < 2; ++sblk) {
            const bf16x8_t pb = pack8(&pv[8 * sblk]);
            union { bf16x8_t v; u32x2_t h[2]; } db;
            db.h[0] = dsp[2 * sblk]; db.h[1] = dsp[2 * sblk + 1];
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                dVt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(qb * RG + 4096 + sblk * 16 * ROWB, df), pb, dVt[df], 0, 0, 0);
                dKt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(qb * RG + sblk * 16 * ROWB, df), db.v, dKt[df], 0, 0, 0);
            }
        }
        __syncthreads();                                 // every wave has multiplied q-block qb: its region takes dS^T[:, 32 qb ..]
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(__attribute__((address_space(3))) u32x2_t*)(L + ((dswo ^ (g4 << 4)) + qb * RG)) = dsp[g4];
    }
#if !ATTN_STAGED_STORES
    {
        const bool wide_k = (a.lddk & 7) == 0 && (((uintptr_t)a.dK) & 15) == 0;
        const bool wide_v = (a.lddv & 7) == 0 && (((uintptr_t)a.dV) & 15) == 0;
        bf16_t* dkrow = a.dK + ((size_t)b * N + key) * a.lddk + h * HD;
        bf16_t* dvrow = a.dV + ((size_t)b * N + key) * a.lddv + h * HD;
#pragma unroll
        for (int which = 0; which < 2; ++which)
#pragma unroll
            for (int df = 0; df < 2; ++df)
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const f32x16_t& t = which ? dVt[df] : dKt[df];
                    const float sc = which ? 1.0f : a.scale;
                    uint2 pk[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        pk[u] = make_uint2(pack2bf(t[4 * (g + u)] * sc, t[4 * (g + u) + 1] * sc), pack2bf(t[4 * (g + u) + 2] * sc, t[4 * (g + u) + 3] * sc));
                    store_bf16_groups(which ? dvrow : dkrow, df * 32 + 8 * g, pk[0], pk[1], fhi, HD, which ? wide_v : wide_k);
                }
    }
#endif
    __syncthreads();                                     // the dS^T of the last q-block is in place

    // ---- pass B: dQ = dS K for this wave's 32 queries -----------------------------------------------
    {
        const int qb = wave;
        const int q = qb * 32 + r31;
        f32x16_t dQt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const int rows = kb * 32 + sblk * 16;
                union { bf16x8_t v; s16x4_t h[2]; } db;
                db.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + dsco[0] + wave * RG + rows * 64));
                db.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(L + dsco[1] + wave * RG + rows * 64));
#pragma unroll
                for (int df = 0; df < 2; ++df) dQt[df] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(oK + rows * ROWB, df), db.v, dQt[df], 0, 0, 0);
            }
        float sc = a.scale;
        if constexpr (MASK != FM_MASK_NONE) sc = full_l[q] ? 0.f : sc;
#if ATTN_STAGED_STORES
        // dK and dV stayed in registers through pass B (<= 168 registers still: three workgroups per CU): behind this barrier every wave is done
        // with dS^T and K, and 4 KB of the K tile per wave stage all three outputs into whole-line stores
        __syncthreads();
        char* stg = smem + oK + wave * 4096;
        store_rows_staged(dKt, a.scale, stg, a.dK + ((size_t)b * N + wave * 32) * a.lddk + h * HD, a.lddk, lane);
        store_rows_staged(dVt, 1.0f, stg, a.dV + ((size_t)b * N + wave * 32) * a.lddv + h * HD, a.lddv, lane);
        store_rows_staged(dQt, sc, stg, a.dQ + ((size_t)b * N + qb * 32) * a.lddq + h * HD, a.lddq, lane);
        return;
#endif
        const bool wide_q = (a.lddq & 7) == 0 && (((uintptr_t)a.dQ) & 15) == 0;
        bf16_t* dqrow = a.dQ + ((size_t)b * N + q) * a.lddq + h * HD;
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                uint2 pq[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    pq[u] = make_uint2(pack2bf(dQt[df][4 * (g + u)] * sc, dQt[df][4 * (g + u) + 1] * sc),
                                       pack2bf(dQt[df][4 * (g + u) + 2] * sc, dQt[df][4 * (g + u) + 3] * sc));
                store_bf16_groups(dqrow, df * 32 + 8 * g, pq[0], pq[1], fhi, HD, wide_q);
            }
    }
}

int fill(AttnArgs& a, const fm_attn_args* p, const char* who) {
    FM_CHECK_ARG(p && p->Q && p->K && p->V && p->O, "%s: null pointer", who);
    FM_CHECK_ARG(p->head_dim == HD, "%s: head_dim=%d unsupported (this build handles 64)", who, p->head_dim);
    FM_CHECK_ARG(p->B > 0 && p->H > 0 && p->Nq > 0 && p->Nk > 0, "%s: bad shape", who);
    FM_CHECK_ARG(p->ldq % 8 == 0 && p->ldk % 8 == 0 && p->ldv % 8 == 0 && p->ldo % 4 == 0, "%s: leading dims must be multiples of 8", who);
    FM_CHECK_ARG(p->mask_kind != FM_MASK_KEYPAD || p->kpad, "%s: FM_MASK_KEYPAD needs kpad", who);
    FM_CHECK_ARG(p->mask_kind != FM_MASK_DENSE || p->dense, "%s: FM_MASK_DENSE needs dense", who);
    FM_CHECK_ARG(p->mask_kind != FM_MASK_DECODER || ((p->modq == nullptr) == (p->modk == nullptr)), "%s: modq and modk go together", who);
    a.Q = (const bf16_t*)p->Q; a.K = (const bf16_t*)p->K; a.V = (const bf16_t*)p->V; a.O = (bf16_t*)p->O;
    a.stat_m = (float*)p->stat_m; a.stat_l = (float*)p->stat_l;
    a.ldq = p->ldq; a.ldk = p->ldk; a.ldv = p->ldv; a.ldo = p->ldo;
    a.B = p->B; a.H = p->H; a.Nq = p->Nq; a.Nk = p->Nk; a.scale = p->scale; a.mask_kind = p->mask_kind;
    a.kpad = (const uint8_t*)p->kpad; a.cs = p->cs; a.modq = p->modq; a.modk = p->modk; a.dense = (const uint8_t*)p->dense;
    a.causal = p->causal;
    a.kvr = p->kv_batch_rows > 0 ? p->kv_batch_rows : p->Nk;
    a.zero_attn = p->zero_attn;
    FM_CHECK_ARG(a.kvr >= p->Nk, "%s: kv_batch_rows=%d < Nk=%d", who, p->kv_batch_rows, p->Nk);
    a.dO = (const bf16_t*)p->dO; a.dQ = (bf16_t*)p->dQ; a.dK = (bf16_t*)p->dK; a.dV = (bf16_t*)p->dV;
    a.lddo = p->lddo; a.lddq = p->lddq; a.lddk = p->lddk; a.lddv = p->lddv;
    return 0;
}

int g_attn_tr = 1;   // transpose reads verified on hardware (tools/probe_gfx950.hip)

}  // namespace

extern "C" void fm_set_attn_transpose_read(int on) { g_attn_tr = on; }
extern "C" int fm_get_attn_transpose_read(void) { return g_attn_tr; }

extern "C" int fm_attn_fwd(const fm_attn_args* p, void* stream) {
    AttnArgs a{};
    if (int rc = fill(a, p, "fm_attn_fwd")) return rc;
    FM_CHECK_ARG((a.stat_m == nullptr) == (a.stat_l == nullptr), "fm_attn_fwd: stat_m and stat_l go together");
    const int NkP = (a.Nk + 63) / 64 * 64;
    const size_t lds = 2 * 2 * 64 * ROWB + (size_t)NkP * 3 + 16;
    FM_CHECK_ARG(lds <= 150 * 1024, "fm_attn_fwd: Nk=%d too long for the key metadata buffer", a.Nk);
    dim3 grid((a.Nq + 127) / 128, a.H, a.B);
    const int tr = p->force_tr >= 0 ? p->force_tr : g_attn_tr;
    // 128 x 128 tokens exactly: the round-5 kernel (all keys in LDS at once, 4 - 7 VALU per score); FOURM_ATTN_FWD_V2=0 keeps the general one
    static const bool v2_on = [] { const char* e = getenv("FOURM_ATTN_FWD_V2"); return !e || atoi(e) != 0; }();
    static const bool t_for_128 = [] { const char* e = getenv("FOURM_ATTN_FWD_T"); return e && atoi(e) == 2; }();
    if (v2_on && !t_for_128 && tr && a.Nq == 128 && a.Nk == 128 && a.mask_kind != FM_MASK_DENSE && a.scale > 0.f) {
        const size_t lds128 = (size_t)2 * 128 * ROWB + 128 * 4;
        const dim3 grid128(a.H, a.B);
        a.o16 = ((a.ldo & 7) == 0 && (((uintptr_t)a.O) & 15) == 0) ? 1 : 0;
        switch (a.mask_kind) {
            case FM_MASK_NONE: hipLaunchKernelGGL((attn_fwd128_kernel<FM_MASK_NONE>), grid128, dim3(256), lds128, (hipStream_t)stream, a); break;
            case FM_MASK_KEYPAD: hipLaunchKernelGGL((attn_fwd128_kernel<FM_MASK_KEYPAD>), grid128, dim3(256), lds128, (hipStream_t)stream, a); break;
            case FM_MASK_DECODER: hipLaunchKernelGGL((attn_fwd128_kernel<FM_MASK_DECODER>), grid128, dim3(256), lds128, (hipStream_t)stream, a); break;
            default: fm_set_error("fm_attn_fwd: unknown mask kind %d", a.mask_kind); return -1;
        }
        FM_CHECK_LAUNCH("fm_attn_fwd");
        return 0;
    }
    // Nq, Nk multiples of 128, Nk <= 512 (256 x 256: 4M-L / mod21): the same arithmetic in an online softmax over 128-key tiles.
    // FOURM_ATTN_FWD_T=0 keeps the general kernel; =2 also routes the 128 x 128 shape here (lab comparison with attn_fwd128_kernel).
    static const int t_mode = [] { const char* e = getenv("FOURM_ATTN_FWD_T"); return e ? atoi(e) : 1; }();
    if (t_mode && tr && a.Nq % 128 == 0 && a.Nk % 128 == 0 && a.Nk <= 512 && a.mask_kind != FM_MASK_DENSE && a.scale > 0.f) {
        static const bool t_db = [] { const char* e = getenv("FOURM_ATTN_FWD_DB"); return e && atoi(e) != 0; }();      // lab: two K/V buffers, two workgroups per CU
        const size_t ldst = (size_t)((t_db && a.Nk > 128) ? 2 : 1) * 2 * 128 * ROWB + (size_t)a.Nk * 4;
        const dim3 gridt(a.Nq / 128, a.H, a.B);
        a.o16 = ((a.ldo & 7) == 0 && (((uintptr_t)a.O) & 15) == 0) ? 1 : 0;
#define FWDT(MK)                                                                                                              \
    if (t_db) hipLaunchKernelGGL((attn_fwdt_kernel<MK, true>), gridt, dim3(256), ldst, (hipStream_t)stream, a);                \
    else hipLaunchKernelGGL((attn_fwdt_kernel<MK, false>), gridt, dim3(256), ldst, (hipStream_t)stream, a);
        switch (a.mask_kind) {
            case FM_MASK_NONE: FWDT(FM_MASK_NONE) break;
            case FM_MASK_KEYPAD: FWDT(FM_MASK_KEYPAD) break;
            case FM_MASK_DECODER: FWDT(FM_MASK_DECODER) break;
            default: fm_set_error("fm_attn_fwd: unknown mask kind %d", a.mask_kind); return -1;
        }
#undef FWDT
        FM_CHECK_LAUNCH("fm_attn_fwd");
        return 0;
    }
#define FWD(TR, MK) hipLaunchKernelGGL((attn_fwd_kernel<TR, MK>), grid, dim3(256), lds, (hipStream_t)stream, a)
#define FWD_MASK(TR)                                                        \
    switch (a.mask_kind) {                                                  \
        case FM_MASK_NONE: FWD(TR, FM_MASK_NONE); break;                    \
        case FM_MASK_KEYPAD: FWD(TR, FM_MASK_KEYPAD); break;                \
        case FM_MASK_DECODER: FWD(TR, FM_MASK_DECODER); break;              \
        case FM_MASK_DENSE: FWD(TR, FM_MASK_DENSE); break;                  \
        default: fm_set_error("fm_attn_fwd: unknown mask kind %d", a.mask_kind); return -1; \
    }
    if (tr) { FWD_MASK(true) } else { FWD_MASK(false) }
#undef FWD_MASK
#undef FWD
    FM_CHECK_LAUNCH("fm_attn_fwd");
    return 0;
}

extern "C" int fm_attn_bwd(const fm_attn_args* p, void* stream) {
    AttnArgs a{};
    if (int rc = fill(a, p, "fm_attn_bwd")) return rc;
    FM_CHECK_ARG(a.dO && a.dQ && a.dK && a.dV && a.stat_m && a.stat_l, "fm_attn_bwd: null pointer");
    FM_CHECK_ARG(a.kvr == a.Nk, "fm_attn_bwd: kv_batch_rows is a forward-only (K/V cache) option");
    FM_CHECK_ARG(p->lddo % 8 == 0 && p->lddq % 4 == 0 && p->lddk % 4 == 0 && p->lddv % 4 == 0, "fm_attn_bwd: leading dims");
    const int NqP = (a.Nq + 31) & ~31, NkP = (a.Nk + 31) & ~31;
    // up to 512 rows per tile in one piece (128 KB of tiles); longer sequences in chunks of 256 rows (two workgroups per CU)
    const int NPmax = NqP > NkP ? NqP : NkP;
    a.chunk = NPmax <= 512 ? NPmax : 256;
    // dS^T through LDS (one softmax recompute instead of two) when it fits beside the tiles with two workgroups per CU
    static const bool ds_on = [] { const char* e = getenv("FOURM_ATTN_BWD_DS"); return !e || atoi(e) != 0; }();
    const size_t ds_bytes = (size_t)((NqP + 63) / 64) * NkP * ROWB;
    const bool ds = ds_on && a.chunk == NPmax && ds_bytes <= 32 * 1024;
    const size_t lds = (size_t)2 * a.chunk * ROWB + (size_t)NqP * (16 + 1) + (size_t)NkP * (4 + 1) + 64 + (ds ? ds_bytes + 16 : 0);
    FM_CHECK_ARG(lds <= 160 * 1024, "fm_attn_bwd: Nq=%d Nk=%d need %zu bytes of LDS for the per-row statistics", a.Nq, a.Nk, lds);
    FM_CHECK_ARG(a.Nq < 0x7fff && a.Nk < 0x7fff, "fm_attn_bwd: sequence too long for the packed 15-bit mask bounds");
    dim3 grid(a.H, a.B);
    const int tr = p->force_tr >= 0 ? p->force_tr : g_attn_tr;
    // 128 x 128 tokens exactly (the 128-token 4M configurations): the round-5 kernel; FOURM_ATTN_BWD_V2=0 keeps the general one
    static const bool v2_on = [] { const char* e = getenv("FOURM_ATTN_BWD_V2"); return !e || atoi(e) != 0; }();
    const bool rows16 = ((a.lddq | a.lddk | a.lddv) & 7) == 0 && ((((uintptr_t)a.dQ) | ((uintptr_t)a.dK) | ((uintptr_t)a.dV)) & 15) == 0;      // whole-line staged stores
    if (v2_on && tr && rows16 && a.Nq == 128 && a.Nk == 128 && a.mask_kind != FM_MASK_DENSE) {
        const size_t lds128 = (size_t)4 * 128 * ROWB + 6 * 128 * 4;
        // Which of the two 128 x 128 kernels (profiles/r05_attn_ov.txt, B = 256, H = 12): the 3-workgroups-per-CU overlay form wins where the
        // per-score VALU work is largest (decoder rule: 91.6 vs 106.6 us) and loses slightly elsewhere (key padding 95.4 vs 93.2, none 98.3 vs
        // 95.9): default = overlay for the decoder mask only.  FOURM_ATTN_BWD_OV=0 / 1 forces one form for every mask.
        static const int ov_mode = [] { const char* e = getenv("FOURM_ATTN_BWD_OV"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
        const bool ov_on = ov_mode < 0 ? a.mask_kind == FM_MASK_DECODER : ov_mode == 1;
        if (ov_on) {      // 51 KB / <= 168 registers: three workgroups per CU
            const size_t ldso = (size_t)4 * 8192 + 128 * ROWB + 6 * 128 * 4;
#define BWD128O(MK)                                                                                                                       \
    {                                                                                                                                     \
        auto k = attn_bwd128o_kernel<MK>;                                                                                                 \
        hipLaunchKernelGGL(k, grid, dim3(256), ldso, (hipStream_t)stream, a);                                                             \
    }
            if (a.mask_kind == FM_MASK_NONE) { a.kpad = nullptr; BWD128O(FM_MASK_KEYPAD) }
            else if (a.mask_kind == FM_MASK_KEYPAD) BWD128O(FM_MASK_KEYPAD)
            else BWD128O(FM_MASK_DECODER)
#undef BWD128O
            FM_CHECK_LAUNCH("fm_attn_bwd");
            return 0;
        }
#define BWD128(MK)                                                                                                                        \
    {                                                                                                                                     \
        auto k = attn_bwd128_kernel<MK>;                                                                                                  \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess);   \
        (void)once;                                                                                                                       \
        hipLaunchKernelGGL(k, grid, dim3(256), lds128, (hipStream_t)stream, a);                                                           \
    }
        // The unmasked case runs on the key-padding instantiation with no padded key (kpad = NULL): hipcc schedules that body better than
        // the mask-free one (98 vs 108 us at the bench shape, profiles/r05_attn_variants.txt); FOURM_ATTN_NONE_AS_KEYPAD=0 keeps the latter.
        static const bool none_as_keypad = [] { const char* e = getenv("FOURM_ATTN_NONE_AS_KEYPAD"); return !e || atoi(e) != 0; }();
        if (a.mask_kind == FM_MASK_NONE && none_as_keypad) {
            a.kpad = nullptr;
            BWD128(FM_MASK_KEYPAD);
            FM_CHECK_LAUNCH("fm_attn_bwd");
            return 0;
        }
        switch (a.mask_kind) {
            case FM_MASK_NONE: BWD128(FM_MASK_NONE); break;
            case FM_MASK_KEYPAD: BWD128(FM_MASK_KEYPAD); break;
            case FM_MASK_DECODER: BWD128(FM_MASK_DECODER); break;
            default: fm_set_error("fm_attn_bwd: unknown mask kind %d", a.mask_kind); return -1;
        }
#undef BWD128
        FM_CHECK_LAUNCH("fm_attn_bwd");
        return 0;
    }
#define BWD(TR, MK)                                                                                                   \
    if (a.chunk < NPmax) BWD2(TR, MK, true, false) else if (ds) BWD2(TR, MK, false, true) else BWD2(TR, MK, false, false)
#define BWD2(TR, MK, CH, DSV)                                                                                         \
    {                                                                                                                 \
        auto k = attn_bwd_kernel<TR, MK, CH, DSV>;                                                                       \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); \
        (void)once;                                                                                                   \
        hipLaunchKernelGGL(k, grid, dim3(256), lds, (hipStream_t)stream, a);                                          \
    }
#define BWD_MASK(TR)                                                        \
    switch (a.mask_kind) {                                                  \
        case FM_MASK_NONE: BWD(TR, FM_MASK_NONE); break;                    \
        case FM_MASK_KEYPAD: BWD(TR, FM_MASK_KEYPAD); break;                \
        case FM_MASK_DECODER: BWD(TR, FM_MASK_DECODER); break;              \
        case FM_MASK_DENSE: BWD(TR, FM_MASK_DENSE); break;                  \
        default: fm_set_error("fm_attn_bwd: unknown mask kind %d", a.mask_kind); return -1; \
    }
    if (tr) { BWD_MASK(true) } else { BWD_MASK(false) }
#undef BWD_MASK
#undef BWD
#undef BWD2
    FM_CHECK_LAUNCH("fm_attn_bwd");
    return 0;
}
