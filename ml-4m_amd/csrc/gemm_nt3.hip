// Lock-step, large-tile NT GEMM (gfx950):  out[m][n] = sum_k X[m][k] * W[n][k]   for the dense launches of the trunk.
//
// What the round-3 micro-benchmarks (tools/ubench.hip, profiles/r03_ubench.txt) say about this chip, and what this kernel does
// with it:
//   * v_mfma_f32_32x32x16_bf16 on random bf16 data sustains 1.79-1.89 PFLOP/s (32.0 cycles per MFMA per SIMD at the 1.72 GHz the
//     power limit leaves): that is the ceiling, not 2.5.
//   * fragment reads (ds_read_b128) cost the matrix pipe nothing for any wave tile as long as they are SPREAD between the MFMAs;
//     a burst of reads right behind a workgroup barrier (every wave at once) costs ~200 cycles per barrier.
//   * s_barrier itself is free under MFMAs (48 cycles, hidden), so ONE barrier per K-tile is fine; the two barriers per K-tile
//     of the ping-pong schedule in gemm.hip, each followed by a read burst, are what held its main loop at ~1.05 PFLOP/s.
//   * global_load_lds delivers <= 86 B/ns per CU from L2 and ~30 B/ns per CU from the Infinity Cache or HBM (~8 TB/s chip wide):
//     a 128 x 256 tile needs 80 B/ns per CU at the MFMA ceiling, a 256 x 256 tile 54, so the tile must be large.
// Hence: 256 x 256 (or 192 x 256) output tiles, 8 waves as 2 x 4 (wave tile 128 x 64 / 96 x 64), K-step 64, TWO LDS stages,
// all waves in lock step with one barrier per K-tile.  The barrier sits before the LAST k-step of a K-tile: the fragments of that
// k-step are already in registers, so behind the barrier every wave has 8 MFMAs to issue while it reads the first fragments of
// the next K-tile and issues the LDS-DMA of the K-tile after it (into the buffer the barrier just freed), one piece and one
// fragment read per MFMA.  The K-tiles of ALL output tiles of a persistent workgroup form one stream (as in gemm_nt_flat.hip):
// the ring never drains at a tile boundary; a finished tile is converted and stored between two K-tiles.
//
// vmcnt is one in-order queue of loads and stores: the wait before the barrier allows exactly the stores of a tile that ended
// one K-tile ago to be outstanding (buffer stores with hardware bounds checks: always the same number of instructions).
#include <type_traits>
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"

namespace {
using namespace fmk;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
constexpr uint32_t OOB = 0x80000000u;       // an offset with bit 31 set is outside num_records: the access is dropped / reads 0

// STG: the bf16 outputs of a tile leave through a wave-private LDS staging area (the 32 / 48 KB the two operand stages leave of the 160 KB):
// a wave holds 32 rows x 32 bytes per store instruction in its accumulator layout - 32-byte write requests at the L2, 4.7 M of the 13 M
// requests of a qkv launch (profiles/r04_pmc_l2_fullline.txt) - and re-reads the staged rows so that consecutive lanes cover whole
// 128-byte lines: the same store instructions, half the write requests (64 bytes each), bit-identical outputs.
// BIAS (staged dense epilogues only): out = bf16(acc + bf16(bias[n])); EPI_GELU: pre = that value (out2, optional), out = bf16(gelu(pre)) -
// the biased Linears of the tokenizers' ViTs and of the *_gelu 4M variants (gemm.hip's epilogue arithmetic, bit for bit).
// DEVM (the per-modality logits GEMMs, round 5): the row range of the launch is read from device memory - rows [row0_dev[0], + m_dev[0] rounded
// up to whole 256-row tiles) of X / out (a head's FM_SEG_ROWS-aligned segment; the caller's pad rows are zero) - so that one dense launch per
// head replaces the grouped kernel of gemm.hip without the host knowing the row counts (hipGraph capture).  The grid is every CU in use;
// workgroups without a tile leave at once.
template <int TW, int EPI, bool SPLIT = false, bool STG = false, bool BIAS = false, bool DEVM = false>
__global__ __launch_bounds__(512) void gemm_nt3_kernel(NTArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (DEVM) {
        static_assert(EPI == EPI_BF16, "device-side row ranges: bf16 outputs only");
        const int m = (a.m_dev[0] + 255) & ~255;
        const size_t r0 = (size_t)a.row0_dev[0];
        a.M = m; a.n_tiles_x = m >> 8;
        a.X += r0 * a.ldx;
        a.out = (char*)a.out + r0 * a.ldo * 2;
        if (a.n_tiles_w * a.n_tiles_x <= (int)blockIdx.x) return;
    }
    constexpr int TX = 256, KB = 64, WW = 2, WX = 4, NWAVES = 8, KS = 4;
    constexpr int RB = 128, CPR = 8, RPP = 8;
    constexpr int FW = TW / WW / 32, FX = TX / WX / 32;             // 4 (3) x 2 fragments of 32 x 32 per wave
    constexpr int PW = TW / (RPP * NWAVES), PX = TX / (RPP * NWAVES), LOADS = PW + PX;
    constexpr int STAGE = (TW + TX) * RB;
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;
    constexpr int NMF = FW * FX;                                    // MFMAs per k-step
    // stores per wave and tile (16 bytes per lane each)
    constexpr int NST = (EPI == EPI_BF16 || EPI == EPI_GELU) ? FW * FX * 2 : EPI == EPI_SWIGLU ? FX * (FW / 2) * 2 * 3 : EPI == EPI_SWIGLU_BWD ? 60 : FW * FX * 4;
    static_assert(TW % (RPP * NWAVES) == 0, "tile rows must split evenly over the DMA pieces");
    static_assert(EPI != EPI_SWIGLU || FW % 2 == 0, "SwiGLU needs (g,u) fragment pairs per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;
    const int frow = lane & 31, fhi = lane >> 5;
    const int fswz = (frow >> 1) & (CPR - 1);

    const int total = a.n_tiles_w * a.n_tiles_x;
    if (a.dephase_groups > 1) {      // experiment (fm_lab_set 0 / 1): groups of CUs start out of phase, so that store epilogues do not coincide
        const int ph = ((int)blockIdx.x / 8) % a.dephase_groups;
        for (int i = 0; i < ph * a.dephase_step; ++i) __builtin_amdgcn_s_sleep(32);
    }
    const int n_my = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int KT = a.K / KB;
    const int G = n_my * KT;
    const int N = a.N;

    // ---- the DMA stream -------------------------------------------------------------------------------------------------------
    uint32_t woff[PW], xoff[PX];
    __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.W), rs_w2 = rsrc_of(a.W2 ? a.W2 : a.W), rs_x = rsrc_of(a.X);
    (void)rs_w2;
    // Tile order.  Consecutive ids run on one XCD (xcd_remap) with the W tiles fastest: the X tile is shared in that XCD's L2.  When all
    // of W does not fit the 4 MB L2 next to the streaming X tiles (SwiGLU: 6.3 MB of fc1 | fc3; qkv: 3.5 MB) every round re-fetches it
    // from the Infinity Cache (FETCH_SIZE 3-14 x the operands, profiles/r02_lab_nt_fetch.txt).  group_w > 0: an XCD walks its X range
    // once per GROUP of group_w W tiles - the group stays resident, the X range is streamed n_tiles_w / group_w times instead.
    const int gw = a.group_w;
    const bool grouped_order = gw > 0 && total % 8 == 0 && a.n_tiles_x % 8 == 0 && a.n_tiles_w % gw == 0;
    auto tile_origin = [&](int j, int& n0, int& m0) {
        int tile = xcd_remap((int)blockIdx.x + j * (int)gridDim.x, total);
        if (a.reverse) tile = total - 1 - tile;      // newest rows of the producer first (see NTArgs::reverse)
        int tw = tile % a.n_tiles_w, tx = tile / a.n_tiles_w;
        if (grouped_order) {
            const int per_xcd = total / 8, xr = a.n_tiles_x / 8;
            const int xcd = tile / per_xcd, l = tile % per_xcd;
            const int grp = l / (xr * gw), rem = l % (xr * gw);
            tx = xcd * xr + rem / gw; tw = grp * gw + rem % gw;
        }
        n0 = tw * NPT; m0 = tx * TX;
    };
    auto set_sources = [&](int j) __attribute__((always_inline)) {
        int n0, m0;
        tile_origin(j, n0, m0);
        rs_w = rsrc_of(a.W + (size_t)n0 * a.ldw);
        if constexpr (EPI == EPI_SWIGLU) rs_w2 = rsrc_of(a.W2 + (size_t)n0 * a.ldw);
        rs_x = rsrc_of(a.X + (size_t)m0 * a.ldx);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> 1) & (CPR - 1));
            int r;
            if constexpr (EPI == EPI_SWIGLU) r = (t >> 6) * 32 + (t & 31);     // rows [0,32) of a 64-row group: g (W), [32,64): u (W2)
            else r = t;
            r = n0 + r < N ? r : N - 1 - n0;
            woff[p] = (uint32_t)r * (uint32_t)a.ldw * 2u + (uint32_t)lc * 16u;
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> 1) & (CPR - 1));
            const int r = m0 + t < a.M ? t : a.M - 1 - m0;
            xoff[p] = (uint32_t)r * (uint32_t)a.ldx * 2u + (uint32_t)lc * 16u;
        }
    };
    int s_kt = 0, s_j = 0, s_buf = 0;
    auto stage_piece = [&](int q) __attribute__((always_inline)) {            // piece q of the stream's current stage (q < PW: W rows, else X rows)
        const int soff = s_kt * (KB * 2);
        if (q < PW) {
            const int pi = q < PW ? q : 0;
            const int t0 = (pi * NWAVES + wave) * RPP;
            if (EPI == EPI_SWIGLU && ((t0 >> 5) & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, LDS_PTR(smem + s_buf * STAGE + (pi * NWAVES + wave) * 1024), 16, woff[pi], soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(smem + s_buf * STAGE + (pi * NWAVES + wave) * 1024), 16, woff[pi], soff, 0, 0);
        } else {
            const int pi = q >= PW ? q - PW : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(smem + s_buf * STAGE + TW * RB + (pi * NWAVES + wave) * 1024), 16, xoff[pi], soff, 0, 0);
        }
    };
    auto stage_advance = [&]() __attribute__((always_inline)) {
        ++s_kt; s_buf ^= 1;
        if (s_kt == KT) { s_kt = 0; ++s_j; if (s_j < n_my) set_sources(s_j); }
    };

    f32x16_t acc[FW][FX];
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- epilogue of the tile whose last K-tile was just multiplied --------------------------------------------------------------
    auto finish_tile = [&](int j_done) __attribute__((always_inline)) {
        int n0, m0;
        tile_origin(j_done, n0, m0);
        if constexpr ((EPI == EPI_BF16 || EPI == EPI_GELU) && STG) {
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
            const __amdgpu_buffer_rsrc_t rs_pre = rsrc_of((const char*)(a.out2 ? a.out2 : a.out) + (size_t)m0 * (a.out2 ? a.ldo2 : a.ldo) * 2);
            (void)rs_pre;
            constexpr int CW = (TW / WW) / 8;                                   // 16-byte chunks per staged row: 16 (two rounds of 8) or 12
            constexpr int RCH = CW == 16 ? 8 : 12, ROWB = RCH * 16, NRND = CW / RCH, SPR = 32 * RCH / 64;     // stores per round
            char* sc = smem + 2 * STAGE + wave * (32 * ROWB);
            constexpr int NWHICH = EPI == EPI_GELU ? 2 : 1;                     // GELU: the pre-activation (when asked for), then the activation
#pragma unroll
            for (int j = 0; j < FX; ++j)
#pragma unroll
                for (int rnd = 0; rnd < NRND; ++rnd)
#pragma unroll
                for (int which = 0; which < NWHICH; ++which) {
                    if (EPI == EPI_GELU && which == 0 && !a.out2) continue;
#pragma unroll
                    for (int ii = 0; ii < RCH / 4; ++ii)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int i = rnd * (RCH / 4) + ii, ch = ii * 4 + g;
                            float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            if constexpr (BIAS) {
                                const int cb = n0 + ww * (TW / WW) + i * 32 + 8 * g + 4 * fhi;
                                if (cb < N) {
                                    const float4 t = *(const float4*)(a.bias + cb);
                                    v[0] += bfround(t.x); v[1] += bfround(t.y); v[2] += bfround(t.z); v[3] += bfround(t.w);
                                }
                            }
                            if (EPI == EPI_GELU && which == 1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = gelu_f(bfround(v[e]));
                            }
                            const uint2 pv = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                            int pos;
                            if constexpr (RCH == 8) pos = ch ^ (frow & 7);
                            else { pos = ch + (frow & 7); pos = pos >= 12 ? pos - 12 : pos; }
                            *(uint2*)(sc + frow * ROWB + pos * 16 + fhi * 8) = pv;
                        }
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4_t rv[SPR];
                    int rr[SPR], cc[SPR];
#pragma unroll
                    for (int q = 0; q < SPR; ++q) {
                        const int t = q * 64 + lane;
                        rr[q] = t / RCH; cc[q] = t % RCH;
                        int pos;
                        if constexpr (RCH == 8) pos = cc[q] ^ (rr[q] & 7);
                        else { pos = cc[q] + (rr[q] & 7); pos = pos >= 12 ? pos - 12 : pos; }
                        rv[q] = *(const u32x4_t*)(sc + rr[q] * ROWB + pos * 16);
                    }
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < SPR; ++q) {
                        const int rl = wx * (TX / WX) + j * 32 + rr[q];
                        const int c = n0 + ww * (TW / WW) + rnd * (RCH * 8) + cc[q] * 8;
                        if (EPI == EPI_GELU && which == 0) {
                            const uint32_t so = ((m0 + rl < a.M && c < N) ? 0u : OOB) | ((uint32_t)(rl * a.ldo2) * 2u + (uint32_t)c * 2u);
                            __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_pre, so, 0, 0);
                        } else {
                            const uint32_t so = ((m0 + rl < a.M && c < N) ? 0u : OOB) | ((uint32_t)(rl * a.ldo) * 2u + (uint32_t)c * 2u);
                            __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out, so, 0, 0);
                        }
                    }
                }
        } else if constexpr (EPI == EPI_BF16) {
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
            const int c0 = n0 + ww * (TW / WW) + 8 * fhi;                       // this lane's first column (chunk 0)
#pragma unroll
            for (int j = 0; j < FX; ++j) {
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const uint32_t rowoff = (m0 + rl < a.M ? 0u : OOB) | ((uint32_t)(rl * a.ldo) * 2u);
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int g = 2 * gp;
                        const uint2 p0 = make_uint2(pack2bf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                        const uint2 p1 = make_uint2(pack2bf(acc[i][j][4 * g + 4], acc[i][j][4 * g + 5]), pack2bf(acc[i][j][4 * g + 6], acc[i][j][4 * g + 7]));
                        const auto x = __builtin_amdgcn_permlane32_swap(p0.x, p1.x, false, false);
                        const auto y = __builtin_amdgcn_permlane32_swap(p0.y, p1.y, false, false);
                        const u32x4_t v = {x[0], y[0], x[1], y[1]};
                        const int c = c0 + i * 32 + 16 * gp;
                        const uint32_t so = (c < N ? rowoff : OOB) + (uint32_t)c * 2u;
                        if (a.lab & 32) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, so, 0, 2);            // nt
                        else if (a.lab & 64) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, so, 0, 17);      // sc0 sc1 (write-through)
                        else if (a.lab & 128) __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, so, 0, 3);      // sc0 nt
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, so, 0, 0);
                    }
            }
        } else if constexpr (EPI == EPI_RES) {
            // out = res + bf16(acc)  (fp32, may alias): the residual values of one fragment are requested while the previous one is
            // added and stored
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 4);
            const __amdgpu_buffer_rsrc_t rs_res = rsrc_of((const char*)a.res + (size_t)m0 * a.ldr * 4);
            const int c0 = n0 + ww * (TW / WW) + 4 * fhi;                       // this lane's first column (group 0)
            f32x4v_t rv[2][4];
            auto req = [&](int f, int par) __attribute__((always_inline)) {
                const int j = f / FW, i = f % FW;
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const uint32_t rowoff = (m0 + rl < a.M ? 0u : OOB) | ((uint32_t)(rl * a.ldr) * 4u);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = c0 + i * 32 + 8 * g;
                    rv[par][g] = __builtin_bit_cast(f32x4v_t, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (c < N ? rowoff : OOB) + (uint32_t)c * 4u, 0, 0));
                }
            };
            req(0, 0);
#pragma unroll
            for (int f = 0; f < FW * FX; ++f) {
                if (f + 1 < FW * FX) req(f + 1, (f + 1) & 1);
                const int j = f / FW, i = f % FW;
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const uint32_t rowoff = (m0 + rl < a.M ? 0u : OOB) | ((uint32_t)(rl * a.ldo) * 4u);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = c0 + i * 32 + 8 * g;
                    f32x4v_t o = rv[f & 1][g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += bfround(acc[i][j][4 * g + e]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), rs_out, (c < N ? rowoff : OOB) + (uint32_t)c * 4u, 0, 0);
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU_BWD) {
            // acc = d(act); res = saved (g | u) (bf16, width 2 * Hp); out = (dg | du) (bf16): the activation backward of GatedMlp
            // (fm_utils.py:142-144) on registers - d(act) never reaches HBM.  Same arithmetic as fm_swiglu_bwd (bit-identical).
            // The saved values of one fragment are requested while the previous one is computed and stored.
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
            const __amdgpu_buffer_rsrc_t rs_gu = rsrc_of((const char*)a.res + (size_t)m0 * a.ldr * 2);
            const int c0 = n0 + ww * (TW / WW) + 8 * fhi;                       // this lane's first column (chunk 0 of a 16-column pair)
            u32x4_t sv[2][2];                                                   // [parity][g, u] of one 16-column pair of one fragment
            constexpr int NU = FW * FX * 2;                                     // units: (fragment, 16-column pair)
            auto src_off = [&](int unit, int ld) __attribute__((always_inline)) {
                const int f = unit / 2, gp = unit % 2, j = f / FW, i = f % FW;
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const int c = c0 + i * 32 + 16 * gp;
                return ((m0 + rl < a.M && c < N) ? 0u : OOB) | ((uint32_t)(rl * ld) * 2u + (uint32_t)c * 2u);
            };
            auto req = [&](int unit, int par) __attribute__((always_inline)) {
                const uint32_t off = src_off(unit, a.ldr);
                sv[par][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_gu, off, 0, 0);
                sv[par][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_gu, off + (uint32_t)a.Hp * 2u, 0, 0);
            };
            req(0, 0);
#pragma unroll
            for (int unit = 0; unit < NU; ++unit) {
                if (unit + 1 < NU) req(unit + 1, (unit + 1) & 1);
                const int f = unit / 2, gp = unit % 2, j = f / FW, i = f % FW;
                uint2 g2[2], u2[2], dg2[2], du2[2];
                {   // each half-wave loaded 16 bytes = groups (2 gp, 2 gp + 1) of its rows: two swaps give every lane its 4 + 4 values
                    const u32x4_t tg = sv[unit & 1][0], tu = sv[unit & 1][1];
                    const auto gx = __builtin_amdgcn_permlane32_swap(tg[0], tg[2], false, false);
                    const auto gy = __builtin_amdgcn_permlane32_swap(tg[1], tg[3], false, false);
                    const auto ux = __builtin_amdgcn_permlane32_swap(tu[0], tu[2], false, false);
                    const auto uy = __builtin_amdgcn_permlane32_swap(tu[1], tu[3], false, false);
                    g2[0] = make_uint2(gx[0], gy[0]); g2[1] = make_uint2(gx[1], gy[1]);
                    u2[0] = make_uint2(ux[0], uy[0]); u2[1] = make_uint2(ux[1], uy[1]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int g = 2 * gp + u;
                    float xv[4], uv[4], r1[4], r2[4];
                    unpack_bf4(g2[u], xv); unpack_bf4(u2[u], uv);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = bfround(acc[i][j][4 * g + e]);
                        const float sg = 1.0f / (1.0f + __expf(-xv[e]));
                        const float sl = bfround(xv[e] * sg);
                        const float ds = bfround(d * uv[e]);
                        r2[e] = d * sl;
                        r1[e] = ds * (sg * (1.0f + xv[e] * (1.0f - sg)));
                    }
                    dg2[u] = make_uint2(pack2bf(r1[0], r1[1]), pack2bf(r1[2], r1[3]));
                    du2[u] = make_uint2(pack2bf(r2[0], r2[1]), pack2bf(r2[2], r2[3]));
                }
                const auto ax = __builtin_amdgcn_permlane32_swap(dg2[0].x, dg2[1].x, false, false);
                const auto ay = __builtin_amdgcn_permlane32_swap(dg2[0].y, dg2[1].y, false, false);
                const auto bx = __builtin_amdgcn_permlane32_swap(du2[0].x, du2[1].x, false, false);
                const auto by = __builtin_amdgcn_permlane32_swap(du2[0].y, du2[1].y, false, false);
                const u32x4_t vg = {ax[0], ay[0], ax[1], ay[1]}, vu = {bx[0], by[0], bx[1], by[1]};
                const uint32_t off = src_off(unit, a.ldo);
                __builtin_amdgcn_raw_buffer_store_b128(vg, rs_out, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(vu, rs_out, off + (uint32_t)a.Hp * 2u, 0, 0);
            }
        } else if constexpr (EPI == EPI_SWIGLU && STG) {
            // (g, u) -> gu buffer [g | u], act -> out; per wave and 32-row block: three staged rounds of 32 rows x 64 hidden units (128 bytes per row)
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
            const __amdgpu_buffer_rsrc_t rs_out2 = rsrc_of((const char*)(a.out2 ? a.out2 : a.out) + (size_t)m0 * (a.out2 ? a.ldo2 : a.ldo) * 2);
            char* sc = smem + 2 * STAGE + wave * 4096;
            const int hw0 = n0 + (ww * (TW / WW) / 64) * 32;                    // this wave's first hidden unit
#pragma unroll
            for (int j = 0; j < FX; ++j) {
                // g and u are kept PACKED (32 registers) across the three rounds; act is formed from the packed values (bf16 -> f32 is exact)
                uint2 qg[FW / 2][4], qu[FW / 2][4];
#pragma unroll
                for (int ip = 0; ip < FW / 2; ++ip)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        qg[ip][g] = make_uint2(pack2bf(acc[2 * ip][j][4 * g], acc[2 * ip][j][4 * g + 1]), pack2bf(acc[2 * ip][j][4 * g + 2], acc[2 * ip][j][4 * g + 3]));
                        qu[ip][g] = make_uint2(pack2bf(acc[2 * ip + 1][j][4 * g], acc[2 * ip + 1][j][4 * g + 1]), pack2bf(acc[2 * ip + 1][j][4 * g + 2], acc[2 * ip + 1][j][4 * g + 3]));
                        asm volatile("" : "+v"(qg[ip][g].x), "+v"(qg[ip][g].y), "+v"(qu[ip][g].x), "+v"(qu[ip][g].y));      // opaque: no float copies kept alive
                    }
#pragma unroll
                for (int which = 0; which < 3; ++which) {
                    if (which < 2 && !a.out2) continue;
#pragma unroll
                    for (int ip = 0; ip < FW / 2; ++ip)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            uint2 pv = which == 0 ? qg[ip][g] : qu[ip][g];
                            if (which == 2) {
                                float gv[4], uv[4];
                                unpack_bf4(qg[ip][g], gv); unpack_bf4(qu[ip][g], uv);
                                pv = make_uint2(pack2bf(bfround(silu_f(gv[0])) * uv[0], bfround(silu_f(gv[1])) * uv[1]),
                                                pack2bf(bfround(silu_f(gv[2])) * uv[2], bfround(silu_f(gv[3])) * uv[3]));
                            }
                            const int ch = ip * 4 + g;
                            *(uint2*)(sc + frow * 128 + ((ch ^ (frow & 7)) * 16) + fhi * 8) = pv;
                        }
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4_t rv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = q * 8 + (lane >> 3), c = lane & 7;
                        rv[q] = *(const u32x4_t*)(sc + r * 128 + ((c ^ (r & 7)) * 16));
                    }
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rl = wx * (TX / WX) + j * 32 + q * 8 + (lane >> 3);
                        const int h = hw0 + (lane & 7) * 8;
                        const uint32_t ok = (m0 + rl < a.M && h < N) ? 0u : OOB;
                        if (which == 2) __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out, ok | ((uint32_t)(rl * a.ldo) * 2u + (uint32_t)h * 2u), 0, 0);
                        // g and u are saved for the backward only: NON-TEMPORAL stores, so that their 268 MB do not push the activation (the operand of the fc2
                        // GEMM that runs next) out of the 256 MB Infinity Cache - fc2 forward 101 -> 86 us in situ, 0.4 - 0.7 ms per 4M-B step same-box
                        // (FOURM_NT3_LAB bit 32: plain stores, bit 64 / 128: sc0 sc1 / sc0 nt - measured equal to plain / to nt)
                        else if (a.lab & 32) __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out2, ok | ((uint32_t)(rl * a.ldo2) * 2u + (uint32_t)(h + (which == 1 ? a.Hp : 0)) * 2u), 0, 0);
                        else if (a.lab & 64) __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out2, ok | ((uint32_t)(rl * a.ldo2) * 2u + (uint32_t)(h + (which == 1 ? a.Hp : 0)) * 2u), 0, 17);
                        else if (a.lab & 128) __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out2, ok | ((uint32_t)(rl * a.ldo2) * 2u + (uint32_t)(h + (which == 1 ? a.Hp : 0)) * 2u), 0, 3);
                        else __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out2, ok | ((uint32_t)(rl * a.ldo2) * 2u + (uint32_t)(h + (which == 1 ? a.Hp : 0)) * 2u), 0, 2);
                    }
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU) {
            const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
            const __amdgpu_buffer_rsrc_t rs_out2 = rsrc_of((const char*)(a.out2 ? a.out2 : a.out) + (size_t)m0 * (a.out2 ? a.ldo2 : a.ldo) * 2);
            const int h0 = n0 + (ww * (TW / WW) / 64) * 32 + 8 * fhi;           // this lane's first hidden unit (chunk 0)
#pragma unroll
            for (int j = 0; j < FX; ++j) {
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const uint32_t ok = m0 + rl < a.M ? 0u : OOB;
                const uint32_t ro = ok | ((uint32_t)(rl * a.ldo) * 2u), ro2 = ok | ((uint32_t)(rl * a.ldo2) * 2u);
                u32x4_t pg[FW / 2][2], pu[FW / 2][2], pa[FW / 2][2];
#pragma unroll
                for (int ip = 0; ip < FW / 2; ++ip)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        uint2 qg[2], qu[2], qa[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int g = 2 * gp + u;
                            float gv[4], uv[4], av[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                gv[e] = bfround(acc[2 * ip][j][4 * g + e]);
                                uv[e] = bfround(acc[2 * ip + 1][j][4 * g + e]);
                                av[e] = bfround(silu_f(gv[e])) * uv[e];
                            }
                            qg[u] = make_uint2(pack2bf(gv[0], gv[1]), pack2bf(gv[2], gv[3]));
                            qu[u] = make_uint2(pack2bf(uv[0], uv[1]), pack2bf(uv[2], uv[3]));
                            qa[u] = make_uint2(pack2bf(av[0], av[1]), pack2bf(av[2], av[3]));
                        }
                        auto swz = [&](const uint2 (&p)[2]) {
                            const auto x = __builtin_amdgcn_permlane32_swap(p[0].x, p[1].x, false, false);
                            const auto y = __builtin_amdgcn_permlane32_swap(p[0].y, p[1].y, false, false);
                            const u32x4_t v = {x[0], y[0], x[1], y[1]};
                            return v;
                        };
                        pg[ip][gp] = swz(qg); pu[ip][gp] = swz(qu); pa[ip][gp] = swz(qa);
                    }
                // output by output: the 16-byte pieces of one 128-byte line leave back to back
#pragma unroll
                for (int which = 0; which < 3; ++which)
#pragma unroll
                    for (int ip = 0; ip < FW / 2; ++ip)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            const int h = h0 + ip * 32 + 16 * gp;
                            const uint32_t col = (uint32_t)h * 2u;
                            if (which == 2) __builtin_amdgcn_raw_buffer_store_b128(pa[ip][gp], rs_out, (h < N ? ro : OOB) + col, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b128(which == 0 ? pg[ip][gp] : pu[ip][gp], rs_out2,
                                                                        ((h < N && a.out2) ? ro2 : OOB) + col + (which == 1 ? (uint32_t)a.Hp * 2u : 0u), 0, 0);
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < FW; ++i)
#pragma unroll
            for (int j = 0; j < FX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- prologue: stages 0 and 1 --------------------------------------------------------------------------------------------------
    set_sources(0);
#pragma unroll
    for (int q = 0; q < LOADS; ++q) stage_piece(q);
    stage_advance();
    if (G > 1) {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) stage_piece(q);
        stage_advance();
        wait_vmcnt<LOADS>();
    } else wait_vmcnt<0>();
    block_barrier();                                     // stage 0 is in LDS for everyone

    bf16x8_t wf[2][FW], xf[2][FX];
    auto read_frags = [&](int buf, int kk, int par) __attribute__((always_inline)) {
        const char* wt = smem + buf * STAGE + (ww * (TW / WW) + frow) * RB;
        const char* xt = smem + buf * STAGE + TW * RB + (wx * (TX / WX) + frow) * RB;
        const int off = ((kk * 2 + fhi) ^ fswz) * 16;
#pragma unroll
        for (int i = 0; i < FW; ++i) wf[par][i] = *(const bf16x8_t*)(wt + i * 32 * RB + off);
#pragma unroll
        for (int j = 0; j < FX; ++j) xf[par][j] = *(const bf16x8_t*)(xt + j * 32 * RB + off);
    };
    auto mfmas = [&](int par) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FW; ++i)
#pragma unroll
            for (int j = 0; j < FX; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[par][i], xf[par][j], acc[i][j], 0, 0, 0);
    };
    read_frags(0, 0, 0);

    int buf = 0, c_kt = 0, c_j = 0;
    bool stores_behind = false;                          // did the previous K-tile end an output tile (its stores are the youngest in the queue)?
    // one K-tile.  NEXT: a stage g + 1 exists; MORE: a stage g + 2 exists; FIRST: g = 0 (compile time: the steady state has no branch
    // between the barrier and the end of the k-step, so the reads, DMA pieces and MFMAs behind the barrier can be interleaved).
    // SPLIT: the X pieces of stage g + 2 (streamed from HBM: longest latency) are issued behind the barrier in k-step 3, its W
    // pieces (L2 hits) in k-step 0 of the next K-tile: half the DMA instructions per k-step in the vector-memory queue.
    auto k_tile = [&](auto first_c, auto next_c, auto more_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, NEXT = decltype(next_c)::value, MORE = decltype(more_c)::value;
        constexpr bool HALF = SPLIT && NEXT && !FIRST;       // the W pieces of stage g + 1 are still to be issued
        // ---- k-steps 0 .. 2: MFMAs of step kk, fragment reads of step kk + 1 between them ---------------------------------------
#pragma unroll
        for (int kk = 0; kk < KS - 1; ++kk) {
            read_frags(buf, kk + 1, (kk + 1) & 1);
            if (HALF && kk == 0) {
#pragma unroll
                for (int q = 0; q < PW; ++q) stage_piece(q);
            }
            mfmas(kk & 1);
#pragma unroll
            for (int q = 0; q < FW + FX; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (HALF && kk == 0 && q < PW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HALF && kk == 0) stage_advance();
        }
        // ---- the barrier: stage g + 1 has landed for everyone, stage g has been read by everyone -------------------------------
        if constexpr (NEXT) {
            if (!(a.lab & 1)) { if (stores_behind && !SPLIT) wait_vmcnt<NST>(); else wait_vmcnt<0>(); }
        }
        wait_lgkmcnt<0>();
        block_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 3: its MFMAs; first the fragments of stage g + 1 (two reads per MFMA), then the LDS-DMA of stage g + 2 into
        //      the freed buffer (the compiler orders a DMA piece behind every LDS read in flight: reads first, pieces after)
        if constexpr (NEXT) read_frags(buf ^ 1, 0, 0);
        constexpr int NOW = SPLIT ? PX : LOADS;              // pieces issued in this k-step
        if constexpr (MORE) {
            if (!(a.lab & 2)) {
#pragma unroll
                for (int q = SPLIT ? PW : 0; q < LOADS; ++q) stage_piece(q);
            }
        }
        mfmas(1);
        constexpr int RD = NEXT ? FW + FX : 0, RSTEPS = (RD + 1) / 2;           // MFMAs that carry two reads each
#pragma unroll
        for (int q = 0; q < RSTEPS; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        if constexpr (MORE) {
            constexpr int LEFT = NMF - RSTEPS - 1;                               // MFMAs that carry DMA pieces (the last one carries none)
            constexpr int PER = (NOW + LEFT - 1) / LEFT;
#pragma unroll
            for (int q = 0; q < LEFT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, PER, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MORE && !SPLIT) stage_advance();
        buf ^= 1;
        stores_behind = false;
        if (++c_kt == KT) {
            if (!(a.lab & 4)) finish_tile(c_j);
            c_kt = 0; ++c_j; stores_behind = true;
        }
    };
    using T = std::true_type; using F = std::false_type;
    if (G > 2) k_tile(T{}, T{}, T{}); else k_tile(T{}, T{}, F{});        // (G >= 2: K >= 128)
    for (int g = 1; g + 2 < G; ++g) k_tile(F{}, T{}, T{});
    if (G > 2) k_tile(F{}, T{}, F{});
    k_tile(F{}, F{}, F{});
#endif
}

template <int TW, int EPI, bool SPLIT = false, bool STG = false, bool BIAS = false, bool DEVM = false>
int launch_nt3(NTArgs a, hipStream_t s) {
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;
    a.n_tiles_w = (a.N + NPT - 1) / NPT;
    a.n_tiles_x = (a.M + 255) / 256;
    // W tiles per resident group (tile_origin): with the staged epilogue the SwiGLU launch gains 7 % from groups of 4 (its 6.3 MB of
    // fc1 | fc3 are fetched 8 x per XCD otherwise; 227 -> 212 us, profiles/r04_lab_wgroup_staged.txt) - the other shapes do not move.
    // fm_lab_set 3, bits 12-15 override (15 = off).
    a.group_w = (a.lab >> 12) & 15;
    if (a.group_w == 0 && EPI == EPI_SWIGLU && STG) a.group_w = 4;
    if (a.group_w == 15) a.group_w = 0;
    int grid = a.n_tiles_w * a.n_tiles_x;
    const int cus = fm_grid_cus();
    if (grid > cus || DEVM) grid = cus;                  // (DEVM: a.M is the caller's upper bound; the kernel reads the row count)
    // Tiles are walked from the LAST row block to the first: the rows the producer kernel wrote last are the ones still in the Infinity
    // Cache (a forward walk over freshly written data larger than the cache meets the oldest, evicted rows first), and this launch's own
    // outputs are then written last-rows-first for the ascending streaming kernel behind it.  62.0 -> 61.85 ms per 4M-B step, same box
    // (profiles/r04_ab_reverse_walk.txt); FOURM_NT3_LAB bit 2048: forward walk.
    a.reverse = (a.lab & 2048) ? 0 : 1;
    const size_t lds = (size_t)2 * (TW + 256) * 128 + (STG ? (TW == 256 ? 32768 : 49152) : 0);
    auto k = gemm_nt3_kernel<TW, EPI, SPLIT, STG, BIAS, DEVM>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
    if (hipGetLastError() != hipSuccess) return -2;
    return 1;
}

}  // namespace

// mode: 1 = 256-wide W tiles only, 2 = 192-wide only, 3 = by shape.  Returns 1 when it took the launch, 0 when the arguments are
// outside what it handles, < 0 on a launch error.
int fm_launch_nt3(const fmk::NTArgs& a, int epilogue, int mode, hipStream_t s) {
    using namespace fmk;
    if (!mode || a.groups || a.bias2) return 0;
    if (a.m_dev) {      // row range in device memory: the staged bf16 form only; tile width by N alone (the row count is not known here)
        if (!a.row0_dev || epilogue != FM_EPI_BF16 || a.bias || a.K % 64 != 0 || a.K < 128 || a.N % 8 != 0 || a.ldo % 64 != 0 ||
            (((uintptr_t)a.out) & 127) != 0 || (a.lab & (16 | 1024)))
            return 0;
        if ((size_t)256 * (size_t)a.ldo * 4 >= 0x7fffffffull || (size_t)256 * (size_t)a.ldx * 2 >= 0x7fffffffull || (size_t)256 * (size_t)a.ldw * 2 >= 0x7fffffffull) return 0;
        return (a.N % 192 == 0 && a.N % 256 != 0) ? launch_nt3<192, EPI_BF16, true, true, false, true>(a, s) : launch_nt3<256, EPI_BF16, true, true, false, true>(a, s);
    }
    if (a.bias && !(epilogue == FM_EPI_BF16 || epilogue == FM_EPI_GELU)) return 0;
    if (a.bias && ((((uintptr_t)a.bias) & 15) != 0 || (a.lab & (16 | 1024)))) return 0;       // (biased launches exist in the staged form only)
    if (a.M < 2048 || a.K % 64 != 0 || a.K < 128 || a.N % 8 != 0) return 0;
    if ((size_t)256 * (size_t)a.ldo * 4 >= 0x7fffffffull || (size_t)256 * (size_t)a.ldx * 2 >= 0x7fffffffull || (size_t)256 * (size_t)a.ldw * 2 >= 0x7fffffffull) return 0;
    const bool al16 = (((uintptr_t)a.out | (uintptr_t)a.out2 | (uintptr_t)a.res) & 15) == 0;
    if (!al16) return 0;
    // Tile width by cost: a persistent workgroup per CU walks ceil(tiles / CUs) tiles, each costing ~ its width, so the launch costs
    // rounds x width; 192-wide tiles only where they tile N exactly.  On all 256 CUs this picks 192 for N = 768 / 2304 (512 / 1536 tiles =
    // 2 / 6 whole rounds; 256-wide: 384 / 1152 tiles = 1.5 / 4.5) and 256 elsewhere, as the fixed rule of round 3 did; with CUs reserved for
    // RCCL (fm_set_reserved_cus: 248 CUs) it moves N = 768 / 2304 to 256-wide tiles (2 / 5 rounds instead of 3 / 7 of the 192-wide ones).
    const int cus = fm_grid_cus();
    const long xt = (a.M + 255) / 256;
    const long t256 = (long)((a.N + 255) / 256) * xt, t192 = (long)((a.N + 191) / 192) * xt;
    const bool fits192 = a.N % 192 == 0;
    const long c256 = (t256 + cus - 1) / cus * 256, c192 = (t192 + cus - 1) / cus * 192;
    const bool use192 = mode == 2 ? fits192 : mode == 3 ? (fits192 && c192 < c256) : false;
    if (epilogue == FM_EPI_GELU) {      // biased (or not) Linear + exact GELU, optional pre-activation copy: staged form only
        // Off unless asked for (FOURM_NT3_LAB bit 131072): the erf polynomial on top of 128 accumulator registers spills (8 - 104 bytes per lane)
        // and the launch is slower than gemm.hip's GELU epilogue (tokenizer fc1, M = 12544, N = 3072: 98 vs 86 us).
        if (!(a.lab & 131072)) return 0;
        if (a.ldo % 64 != 0 || (((uintptr_t)a.out) & 127) != 0 || (a.out2 && (a.ldo2 % 64 != 0 || (((uintptr_t)a.out2) & 127) != 0)) || (a.lab & (16 | 1024))) return 0;
        if (a.bias) return use192 ? launch_nt3<192, EPI_GELU, true, true, true>(a, s) : launch_nt3<256, EPI_GELU, true, true, true>(a, s);
        return use192 ? launch_nt3<192, EPI_GELU, true, true, false>(a, s) : launch_nt3<256, EPI_GELU, true, true, false>(a, s);
    }
    if (epilogue == FM_EPI_BF16) {
        if (a.ldo % 8 != 0) return 0;
        if (a.bias) {
            if (a.ldo % 64 != 0 || (((uintptr_t)a.out) & 127) != 0) return 0;
            return use192 ? launch_nt3<192, EPI_BF16, true, true, true>(a, s) : launch_nt3<256, EPI_BF16, true, true, true>(a, s);
        }
        if (!(a.lab & 1024) && !(a.lab & 16) && a.ldo % 64 == 0 && (((uintptr_t)a.out) & 127) == 0)      // staged epilogue (whole-line stores); lab bit 1024: legacy
            return use192 ? launch_nt3<192, EPI_BF16, true, true>(a, s) : launch_nt3<256, EPI_BF16, true, true>(a, s);
        if (!(a.lab & 16)) return use192 ? launch_nt3<192, EPI_BF16, true>(a, s) : launch_nt3<256, EPI_BF16, true>(a, s);
        return use192 ? launch_nt3<192, EPI_BF16>(a, s) : launch_nt3<256, EPI_BF16>(a, s);
    }
    if (epilogue == FM_EPI_RESIDUAL) {
        // in situ the in-epilogue residual reads are slower than gemm.hip's tile-start prefetch (88 vs 83 us at N = K = 768): off unless asked for
        if (!(a.lab & 256)) return 0;
        if (a.ldo % 4 != 0 || a.ldr % 4 != 0 || !a.res) return 0;
        if (!(a.lab & 16)) return use192 ? launch_nt3<192, EPI_RES, true>(a, s) : launch_nt3<256, EPI_RES, true>(a, s);
        return use192 ? launch_nt3<192, EPI_RES>(a, s) : launch_nt3<256, EPI_RES>(a, s);
    }
    if (epilogue == FM_EPI_SWIGLU_BWD) {
        if (a.N % 16 != 0 || a.Hp % 8 != 0 || a.ldo % 8 != 0 || a.ldr % 8 != 0 || !a.res || a.Hp < a.N) return 0;
        if ((size_t)256 * (size_t)a.ldr * 2 >= 0x7fffffffull) return 0;
        return launch_nt3<256, EPI_SWIGLU_BWD, true>(a, s);
    }
    if (epilogue == FM_EPI_SWIGLU) {
        if (a.N % 64 != 0 || a.Hp % 8 != 0 || a.ldo % 8 != 0 || (a.out2 && a.ldo2 % 8 != 0) || !a.W2) return 0;
        if (!(a.lab & 1024) && !(a.lab & 16) && a.ldo % 64 == 0 && (!a.out2 || a.ldo2 % 64 == 0) && a.Hp % 64 == 0 && a.N % 64 == 0)
            return launch_nt3<256, EPI_SWIGLU, true, true>(a, s);
        if (!(a.lab & 16)) return launch_nt3<256, EPI_SWIGLU, true>(a, s);
        return launch_nt3<256, EPI_SWIGLU>(a, s);
    }
    return 0;
}
