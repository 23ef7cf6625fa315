// Four-wave, 512-register NT GEMM (gfx950):  out[m][n] = sum_k X[m][k] * W[n][k]  on 256 (rows) x 384 (features) output tiles.
//
// Why another kernel.  gemm_nt3.hip (8 waves x 256 registers, 256 x 256 / 192 x 256 tiles) is bound by the CU <- L2 path: its launch time
// follows the number of 128-byte LDS-DMA requests (profiles/r04_pmc_l2_fullline.txt), and the requests per flop are fixed by the tile:
// (TW + TX) rows of 128 bytes per TW x TX x 64 multiply-adds.  A 256 x 384 tile needs 640 rows where two-and-a-bit 192 x 256 tiles need
// 896 (N = 768, 2304: -29 %) and 1.5 tiles of 256 x 256 need 768 (N = 1536: -17 %).  Its 384 accumulator registers per lane only fit
// ONE wave per SIMD: 4 waves as 2 x 2, wave tile 192 (features) x 128 (rows) = 6 x 4 MFMA fragments, 24 MFMAs per k-step against 10
// fragment reads (nt3: 8 against 6) - the single wave covers its own LDS latency with the MFMAs of the running k-step (32 cycles each).
// M = 32768 rows tile N = 768 in exactly ONE round of 256 workgroups, N = 1536 / 2304 in 2 / 3.
//
// Schedule = nt3's: K-step 64, TWO LDS stages of 80 KB (all 160 KB), one barrier per K-tile placed before the last k-step, the K-tiles
// of all output tiles of a persistent workgroup as one stream, SPLIT issue of the DMA pieces (X rows of stage g + 2 behind the barrier,
// W rows in k-step 0 of the next K-tile).  With no LDS left over, the staged whole-line epilogue borrows the W region of the stage buffer
// the last barrier of a tile has just freed: SPLIT leaves it untouched until k-step 0 of the next K-tile (one extra barrier per tile).
// Shapes: M % 256 == 0, N % 384 == 0, K % 64 == 0, K >= 192 (every row offset is then a lane constant + a scalar: no per-piece
// offset registers).  Accumulation order per output = nt3's: bit-identical results.
#include <type_traits>
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"

namespace {
using namespace fmk;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

// compile-time loop: the body sees its index as a constant expression (register-class constraints of the MFMA statements depend on it)
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}
// Accumulators by hand.  With the MFMA builtin, hipcc (ROCm 7.2) mixes accumulators, fragments and copies over both register files once
// more than 256 registers are live and spills by the thousand (2 657 spilled registers for the 256 x 384 tile; still 241 with asm
// statements whose accumulator operand is constrained to the AGPR file: 16 tuples of 16 fill that file with no slack for its copies).
// So the first 16 accumulator fragments are NOT C++ values: fragment f IS a[16 f : 16 f + 15], named in the asm text; the compiler never
// sees them (it has no reason to touch the AGPR file: everything it allocates fits the VGPR file - checked on the generated code by
// tools/check_nt4_asm.py: no AGPR reference outside these statements, no scratch).  Fragments 16 .. 23 of the 256 x 384 tile are ordinary
// "+v" operands (128 VGPRs, leaving 128 for fragments and addresses).  A tile's first MFMA per fragment takes the constant 0 as its
// addend (no zero fill).  Hazards the compiler no longer sees: an accumulator is reused 16 - 24 MFMAs after its last write (none);
// the epilogue's first read is padded with s_nop by hand.
template <int F, bool ZERO> __device__ __forceinline__ void mfma_agpr(const bf16x8_t& w, const bf16x8_t& x) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, 0" ::"v"(w), "v"(x), "n"(16 * F), "n"(16 * F + 15));
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(w), "v"(x), "n"(16 * F), "n"(16 * F + 15));
}
template <bool ZERO> __device__ __forceinline__ void mfma_vgpr(f32x16_t& acc, const bf16x8_t& w, const bf16x8_t& x) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(w), "v"(x));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
// four consecutive values of AGPR fragment F (registers 16 F + 4 G .. + 3), packed to bf16
template <int F, int G> __device__ __forceinline__ uint2 read_agpr_pack4() {
    float v0, v1, v2, v3;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "n"(16 * F + 4 * G), "n"(16 * F + 4 * G + 1), "n"(16 * F + 4 * G + 2), "n"(16 * F + 4 * G + 3));
    return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
}

template <int TW, int TX, int WW, int WX, bool STG>
__global__ __launch_bounds__(WW * WX * 64) void gemm_nt4_kernel(NTArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KB = 64, NWAVES = WW * WX, KS = 4;
    constexpr int RB = 128, CPR = 8, RPP = 8;
    constexpr int FW = TW / WW / 32, FX = TX / WX / 32;
    constexpr int PW = TW / (RPP * NWAVES), PX = TX / (RPP * NWAVES), LOADS = PW + PX;
    constexpr int STAGE = (TW + TX) * RB;
    constexpr int NMF = FW * FX;
    static_assert(TW % (RPP * NWAVES) == 0 && TX % (RPP * NWAVES) == 0, "tile rows must split evenly over the DMA pieces");
    static_assert((RPP * NWAVES) % 16 == 0, "the swizzle key of a piece row must not depend on the piece index");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63, lane_k = lane;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;
    const int frow = lane & 31, fhi = lane >> 5;
    const int fswz = (frow >> 1) & (CPR - 1);

    const int total = a.n_tiles_w * a.n_tiles_x;
    const int n_my = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int KT = a.K / KB;
    const int G = n_my * KT;

    // ---- the DMA stream -------------------------------------------------------------------------------------------------------
    // piece p of a wave covers tile rows (p * NWAVES + wave) * 8 + lane / 8; the chunk swizzle key ((row >> 1) & 7) does not depend on p,
    // so one lane offset serves every piece and the piece's row advance rides in the scalar offset with the k advance.
    __amdgpu_buffer_rsrc_t rs_w = rsrc_of(a.W), rs_x = rsrc_of(a.X);
    const int t0 = wave * RPP + lane / CPR;
    const int lc0 = (lane % CPR) ^ ((t0 >> 1) & (CPR - 1));
    const uint32_t woff = (uint32_t)t0 * (uint32_t)a.ldw * 2u + (uint32_t)lc0 * 16u;
    const uint32_t xoff = (uint32_t)t0 * (uint32_t)a.ldx * 2u + (uint32_t)lc0 * 16u;
    const int w_pstride = RPP * NWAVES * a.ldw * 2, x_pstride = RPP * NWAVES * a.ldx * 2;
    auto tile_origin = [&](int j, int& n0, int& m0) {
        int tile = xcd_remap((int)blockIdx.x + j * (int)gridDim.x, total);
        if (a.reverse) tile = total - 1 - tile;
        n0 = (tile % a.n_tiles_w) * TW; m0 = (tile / a.n_tiles_w) * TX;
    };
    auto set_sources = [&](int j) __attribute__((always_inline)) {
        int n0, m0;
        tile_origin(j, n0, m0);
        rs_w = rsrc_of(a.W + (size_t)n0 * a.ldw);
        rs_x = rsrc_of(a.X + (size_t)m0 * a.ldx);
    };
    int s_kt = 0, s_j = 0, s_buf = 0;
    auto stage_piece = [&](int q) __attribute__((always_inline)) {            // piece q of the stream's current stage (q < PW: W rows, else X rows)
        if (q < PW) {
            const int pi = q < PW ? q : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(smem + s_buf * STAGE + (pi * NWAVES + wave) * 1024), 16, woff, s_kt * (KB * 2) + pi * w_pstride, 0, 0);
        } else {
            const int pi = q >= PW ? q - PW : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(smem + s_buf * STAGE + TW * RB + (pi * NWAVES + wave) * 1024), 16, xoff, s_kt * (KB * 2) + pi * x_pstride, 0, 0);
        }
    };
    auto stage_advance = [&]() __attribute__((always_inline)) {
        ++s_kt; s_buf ^= 1;
        if (s_kt == KT) { s_kt = 0; ++s_j; if (s_j < n_my) set_sources(s_j); }
    };

    constexpr int NACC_A = NMF < 16 ? NMF : 16, NACC_V = NMF - NACC_A;          // accumulator fragments in the AGPR file (by hand) / in VGPRs
    f32x16_t acc_v[NACC_V > 0 ? NACC_V : 1];
    asm volatile("" ::: "a255");                         // the kernel owns the whole AGPR file (resource usage: NumAgprs = 256)
    // bf16 of values 4 g .. 4 g + 3 of accumulator fragment (i, j)
    auto acc_pack4 = [&](auto fc, auto gc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, g = decltype(gc)::value;
        if constexpr (f < NACC_A) return read_agpr_pack4<f, g>();
        else return make_uint2(pack2bf(acc_v[f - NACC_A][4 * g], acc_v[f - NACC_A][4 * g + 1]), pack2bf(acc_v[f - NACC_A][4 * g + 2], acc_v[f - NACC_A][4 * g + 3]));
    };

    // ---- epilogue of the tile whose last K-tile was just multiplied; `free_buf`: the stage buffer whose W region nobody reads or fills now ----
    auto finish_tile = [&](int j_done, int free_buf) __attribute__((always_inline)) {
        int n0, m0;
        tile_origin(j_done, n0, m0);
        // the epilogue's addresses are functions of the lane id only: computed HERE from an opaque copy, or the compiler hoists two dozen of them
        // out of the K loop and keeps them in registers the main loop needs (the 256 x 384 tile leaves 128 VGPRs for everything else)
        int lane = lane_k, frow = lane & 31, fhi = lane >> 5;
        asm volatile("" : "+v"(lane), "+v"(frow), "+v"(fhi));
        const uint32_t oob = (a.lab & 8) ? 0x80000000u : 0u;             // lab: every store dropped by the bounds check (its instruction is still issued)
        const __amdgpu_buffer_rsrc_t rs_out = rsrc_of((const char*)a.out + (size_t)m0 * a.ldo * 2);
        if constexpr (STG) {
            // a wave's 32 rows x (TW / WW) features go through a wave-private 4 KB of the free W region (8-byte swizzled writes in the
            // accumulator layout, 16-byte reads back row-contiguous) and leave as whole 128-byte lines: nt3's staged epilogue
            constexpr int CW = (TW / WW) / 8;                                   // 16-byte chunks per staged row: 24 (384-wide tiles) or 16
            constexpr int RCH = 8, ROWB = RCH * 16, NRND = CW / RCH;           // rounds of 64 features
            static_assert(CW % RCH == 0, "whole rounds of 64 features");
            char* sc = smem + free_buf * STAGE + wave * (32 * ROWB);
            static_for<0, FX>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                static_for<0, NRND>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int rnd = decltype(rc)::value;
                    static_for<0, 8>([&](auto cc) __attribute__((always_inline)) {
                        constexpr int ch = decltype(cc)::value, ii = ch / 4, g = ch % 4, i = rnd * 2 + ii;
                        const uint2 pv = acc_pack4(IC<i * FX + j>{}, IC<g>{});
                        *(uint2*)(sc + frow * ROWB + ((ch ^ (frow & 7)) * 16) + fhi * 8) = pv;
                    });
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
                    u32x4_t rv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = q * 8 + (lane >> 3), c = lane & 7;
                        rv[q] = *(const u32x4_t*)(sc + r * ROWB + ((c ^ (r & 7)) * 16));
                    }
                    wait_lgkmcnt<0>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rl = wx * (TX / WX) + j * 32 + q * 8 + (lane >> 3);
                        const int c = n0 + ww * (TW / WW) + rnd * 64 + (lane & 7) * 8;
                        __builtin_amdgcn_raw_buffer_store_b128(rv[q], rs_out, oob | ((uint32_t)(rl * a.ldo) * 2u + (uint32_t)c * 2u), 0, 0);
                    }
                });
            });
        } else {
            const int c0 = n0 + ww * (TW / WW) + 8 * fhi;                       // this lane's first column (chunk 0)
            static_for<0, FX>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                const int rl = wx * (TX / WX) + j * 32 + frow;
                const uint32_t rowoff = (uint32_t)(rl * a.ldo) * 2u;
                static_for<0, FW * 2>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value / 2, gp = decltype(ic)::value % 2, g = 2 * gp;
                    const uint2 p0 = acc_pack4(IC<i * FX + j>{}, IC<g>{}), p1 = acc_pack4(IC<i * FX + j>{}, IC<g + 1>{});
                    const auto x = __builtin_amdgcn_permlane32_swap(p0.x, p1.x, false, false);
                    const auto y = __builtin_amdgcn_permlane32_swap(p0.y, p1.y, false, false);
                    const u32x4_t v = {x[0], y[0], x[1], y[1]};
                    const int c = c0 + i * 32 + 16 * gp;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, oob | (rowoff + (uint32_t)c * 2u), 0, 0);
                });
            });
        }
    };

    // ---- prologue: stages 0 and 1 --------------------------------------------------------------------------------------------------
    set_sources(0);
#pragma unroll
    for (int q = 0; q < LOADS; ++q) stage_piece(q);
    stage_advance();
#pragma unroll
    for (int q = 0; q < LOADS; ++q) stage_piece(q);
    stage_advance();
    wait_vmcnt<LOADS>();
    block_barrier();                                     // stage 0 is in LDS for everyone

    // Fragments: the X side double-buffered (a k-step's four X fragments are used by every MFMA row), the W side single-buffered - W fragment i
    // is re-read for the next k-step right behind its last MFMA (i-major order), 20 MFMAs before its next use.
    bf16x8_t wf[FW], xf[2][FX];
    const char* wt_base = smem + (ww * (TW / WW) + frow) * RB;
    const char* xt_base = smem + TW * RB + (wx * (TX / WX) + frow) * RB;
    auto read_w = [&](int buf, int kk, int i) __attribute__((always_inline)) {
        wf[i] = *(const bf16x8_t*)(wt_base + buf * STAGE + i * 32 * RB + (((kk * 2 + fhi) ^ fswz) * 16));
    };
    auto read_x = [&](int buf, int kk, int par, int j) __attribute__((always_inline)) {
        xf[par][j] = *(const bf16x8_t*)(xt_base + buf * STAGE + j * 32 * RB + (((kk * 2 + fhi) ^ fswz) * 16));
    };
#pragma unroll
    for (int i = 0; i < FW; ++i) read_w(0, 0, i);
#pragma unroll
    for (int j = 0; j < FX; ++j) read_x(0, 0, 0, j);

    using T = std::true_type; using F = std::false_type;
    int buf = 0, c_kt = 0, c_j = 0;
    // One k-step: MFMA q = (i, j) = (q / FX, q % FX) in source order, each followed by what rides on it: the reads of the NEXT k-step's
    // fragments (X fragment q behind MFMA q < FX into the other X set; W fragment i behind its last MFMA) and at most one DMA piece;
    // a sched_barrier pins every group.  rbuf / rkk: where the next k-step's fragments live (rbuf < 0: there is none).
    // pieces: first, count of the DMA pieces of this k-step (of the stream's current stage).
    auto k_step = [&](auto zero_c, auto par_c, int rbuf, int rkk, auto p0_c, auto np_c) __attribute__((always_inline)) {
        constexpr bool ZERO = decltype(zero_c)::value;
        constexpr int PAR = decltype(par_c)::value, P0 = decltype(p0_c)::value, NP = decltype(np_c)::value;
        static_for<0, NMF>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value, i = q / FX, j = q % FX;
            if constexpr (q < NACC_A) mfma_agpr<q, ZERO>(wf[i], xf[PAR][j]);
            else mfma_vgpr<ZERO>(acc_v[q - NACC_A], wf[i], xf[PAR][j]);
            if (rbuf >= 0) {
                if constexpr (q < FX) read_x(rbuf, rkk, PAR ^ 1, q);
                if constexpr (j == FX - 1) read_w(rbuf, rkk, i);
            }
            // the DMA pieces ride on the MFMAs that carry no read, spread over the k-step
            constexpr int slot = q - FX;                       // MFMAs FX .. NMF - 2 minus the W-read ones
            if constexpr (NP > 0 && q >= FX && j != FX - 1) {
                constexpr int s_idx = (q - FX) - (q - FX) / FX;          // index among the free slots
                constexpr int NFREE = (NMF - FX) - (NMF - FX) / FX;
                constexpr int NPS = NP > 0 ? NP : 1, STEP = NFREE / NPS > 0 ? NFREE / NPS : 1;
                if constexpr (s_idx % STEP == 0 && s_idx / STEP < NP) stage_piece(P0 + s_idx / STEP);
            }
            (void)slot;
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto k_tile = [&](auto first_c, auto next_c, auto more_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, NEXT = decltype(next_c)::value, MORE = decltype(more_c)::value;
        constexpr bool HALF = NEXT && !FIRST;                // the W pieces of stage g + 1 are still to be issued
        // ---- k-steps 0 .. 2 (k-step 0 of a tile's first K-tile starts the accumulators from the constant 0) ------------------------
        if (c_kt == 0) {
            if constexpr (HALF) k_step(T{}, IC<0>{}, buf, 1, IC<0>{}, IC<PW>{}); else k_step(T{}, IC<0>{}, buf, 1, IC<0>{}, IC<0>{});
        } else {
            if constexpr (HALF) k_step(F{}, IC<0>{}, buf, 1, IC<0>{}, IC<PW>{}); else k_step(F{}, IC<0>{}, buf, 1, IC<0>{}, IC<0>{});
        }
        if constexpr (HALF) stage_advance();
        k_step(F{}, IC<1>{}, buf, 2, IC<0>{}, IC<0>{});
        k_step(F{}, IC<0>{}, buf, 3, IC<0>{}, IC<0>{});
        // ---- the barrier: stage g + 1 has landed for everyone, stage g has been read by everyone -------------------------------
        if constexpr (NEXT) {
            if (!(a.lab & 1)) wait_vmcnt<0>();
        }
        wait_lgkmcnt<0>();
        block_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-step 3: the fragments of stage g + 1 ride on it, and the X pieces of stage g + 2 into the buffer the barrier freed ----
        if constexpr (MORE) k_step(F{}, IC<1>{}, buf ^ 1, 0, IC<PW>{}, IC<PX>{});
        else k_step(F{}, IC<1>{}, NEXT ? (buf ^ 1) : -1, 0, IC<0>{}, IC<0>{});
        if (++c_kt == KT) {
            if (!(a.lab & 4)) {
                asm volatile("s_nop 15\n\ts_nop 15");      // the last MFMAs' results (asm: no hazard handling by the compiler)
                finish_tile(c_j, buf);                   // `buf` (stage g) is the buffer the barrier freed; its W region is nobody's until k-step 0
                asm volatile("s_nop 7");
                if constexpr (STG && NEXT) block_barrier();      // everyone has read its staged rows back: the next W pieces may overwrite them
            }
            c_kt = 0; ++c_j;
        }
        buf ^= 1;
    };
    k_tile(T{}, T{}, T{});                                                // (K >= 192: G >= 3)
    for (int g = 1; g + 2 < G; ++g) k_tile(F{}, T{}, T{});
    k_tile(F{}, T{}, F{});
    k_tile(F{}, F{}, F{});
#endif
}

template <int TW, int TX, int WW, int WX, bool STG>
int launch_nt4(NTArgs a, hipStream_t s) {
    a.n_tiles_w = a.N / TW;
    a.n_tiles_x = a.M / TX;
    int grid = a.n_tiles_w * a.n_tiles_x;
    const int cus = fm_grid_cus();
    if (grid > cus) grid = cus;
    a.reverse = (a.lab & 2048) ? 0 : 1;
    const size_t lds = (size_t)2 * (TW + TX) * 128;
    auto k = gemm_nt4_kernel<TW, TX, WW, WX, STG>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(grid), dim3(WW * WX * 64), lds, s, a);
    if (hipGetLastError() != hipSuccess) return -2;
    return 1;
}

}  // namespace

// mode: bit 0 = take N % 384 == 0 on 256 x 384 tiles (when that costs no more rounds x width than gemm_nt3's tiles), bit 1 = take N % 256 == 0 on 4-wave
// 256 x 256 tiles, bit 2 = legacy (unstaged) stores, bit 3 = lab: bit 0 whatever the rounds.
// Returns 1 when it took the launch, 0 when the arguments are outside what it handles, < 0 on a launch error.
int fm_launch_nt4(const fmk::NTArgs& a, int epilogue, int mode, hipStream_t s) {
    using namespace fmk;
    if (!mode || a.groups || a.bias || a.bias2 || a.m_dev || epilogue != FM_EPI_BF16) return 0;
    if (a.M % 256 != 0 || a.M < 2048 || a.K % 64 != 0 || a.K < 192 || a.ldo % 64 != 0 || (((uintptr_t)a.out) & 127) != 0) return 0;
    if ((size_t)384 * (size_t)a.ldo * 2 >= 0x7fffffffull || (size_t)384 * (size_t)a.ldx * 2 >= 0x7fffffffull || (size_t)384 * (size_t)a.ldw * 2 >= 0x7fffffffull) return 0;
    const bool stg = !(mode & 4);
    if ((mode & 1) && a.N % 384 == 0) {
        // Rounds x tile width, like gemm_nt3's choice between its 256- and 192-wide tiles: with CUs reserved for RCCL (fm_set_reserved_cus: 240 of 256)
        // N = 768 is 256 tiles of 256 x 384 = TWO rounds (cost 768) where gemm_nt3's 256-wide tiles need two rounds of 256 (cost 512) - the launch
        // then goes to gemm_nt3.  On all 256 CUs the costs tie (384 = 2 x 192, 1152 = 6 x 192) and this kernel wins on requests per multiply-add.
        const long cus = fm_grid_cus(), xt = a.M / 256;
        const long c4 = ((long)(a.N / 384) * xt + cus - 1) / cus * 384;
        const long c256 = ((long)((a.N + 255) / 256) * xt + cus - 1) / cus * 256, c192 = a.N % 192 == 0 ? ((long)(a.N / 192) * xt + cus - 1) / cus * 192 : c256;
        if (c4 > (c256 < c192 ? c256 : c192) && !(mode & 8)) return 0;          // (mode bit 3: lab - take the launch whatever the rounds)
        return stg ? launch_nt4<384, 256, 2, 2, true>(a, s) : launch_nt4<384, 256, 2, 2, false>(a, s);
    }
    if ((mode & 2) && a.N % 256 == 0) return stg ? launch_nt4<256, 256, 2, 2, true>(a, s) : launch_nt4<256, 256, 2, 2, false>(a, s);
    return 0;
}
