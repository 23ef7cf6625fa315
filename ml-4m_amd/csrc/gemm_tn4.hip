// Four-wave, 512-register form of the TN job list (fm_gemm_tn_multi):  out[n][k] += sum_r A[r][n] * B[r][k]  on 256 (A columns) x 384
// (B columns) output tiles - the weight-gradient GEMMs of one transformer layer in one launch, scheduled exactly like gemm.hip's
// gemm_tn_multi_kernel (same tile list, same cut of the last partial round: tn_multi_walk in gemm_args.h).
//
// Why.  Like the NT kernels the list kernel is bound by the CU <- L2 path: its time follows the LDS-DMA requests, (TA + TB) rows of a
// K-tile per TA x TB x 64 multiply-adds.  256 x 384 needs 640 where one-and-a-half 256 x 256 tiles need 768 (-17 %); gemm_nt4.hip has the
// same step on the NT side.  384 accumulator registers per lane leave ONE wave per SIMD: 4 waves as 2 x 2, wave tile 128 (A) x 192 (B)
// = 4 x 6 MFMA fragments, 24 MFMAs per k-step against 20 transpose reads (10 fragments of two ds_read_b64_tr_b16 each).
//
// LDS: a stage is five [64 rows][128 columns] bf16 regions of 16 KB (A0 A1 B0 B1 B2), two stages = all 160 KB.  With 256-byte region rows
// a 1 KB LDS-DMA piece is 4 whole rows and the 16-byte chunk swizzle ((row & 3) << 2) depends on the lane only: ONE lane offset per operand
// serves all 20 pieces of a wave, the piece's rows / the region's columns / the K-tile ride in the scalar offset.
// Schedule = gemm_nt4.hip's: one barrier per K-tile before its last k-step, the A pieces of stage g + 2 behind the barrier, the B pieces in
// k-step 0 of the next K-tile; the A fragments double-buffered, B fragment i re-read right behind its last MFMA of the k-step.
// Accumulator fragments 0 .. 15 live in hand-named AGPRs (the compiler cannot allocate 384 accumulator registers without spilling:
// gemm_nt4.hip), 16 .. 23 in VGPRs.
// Shapes: R % 64 == 0 per job, N % 128 == 0, K % 128 == 0 (a region is wholly inside or wholly outside the operand; outside regions are
// clamped to the last one inside and their results dropped), operands 16-byte aligned.  fm_gemm_tn_multi picks this kernel when the
// 256 x 384 tiling wastes < 8 % of its MFMAs on such regions (4M-B: fc2's K = 2048 -> 6 x 384: 2.7 % of a layer).
#include <type_traits>
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"
#include "agpr_mfma.h"

namespace {
using namespace fmk;

template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (B < E) { fn(IC<B>{}); static_for<B + 1, E>(fn); }
}
__device__ __forceinline__ void mfma_vgpr(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int F, int R> __device__ __forceinline__ float read_agpr() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(16 * F + R));
    return v;
}
// one half (4 reduction rows x 16 columns per 16-lane group) of a transpose-read fragment; the caller owns the waits
template <int OFF> __device__ __forceinline__ s16x4_t tr_read4(uint32_t lds_addr) {
    s16x4_t h;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h) : "v"(lds_addr), "n"(OFF));
    return h;
}

constexpr int TA = 256, TB = 384, KB = 64, RG = 128;                 // tile, K-step, columns per LDS region
constexpr int RB = RG * 2, REGION = KB * RB, NREG = (TA + TB) / RG, STAGE = NREG * REGION;     // 256-byte rows, 16 KB regions, 80 KB stages
constexpr int NWAVES = 4, FA = 4, FB = 6, NMF = FA * FB;
constexpr int PPR = REGION / 1024 / NWAVES;                          // DMA pieces per region and wave: 4
constexpr int NPA = (TA / RG) * PPR, NPB = (TB / RG) * PPR;          // 8 + 12 pieces per wave and stage

__global__ __launch_bounds__(256) void gemm_tn4_multi_kernel(TNMultiArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const int fhi = lane >> 5, li = lane & 15;
    const int G = gridDim.x;
    const int w = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    asm volatile("" ::: "a255");                         // the kernel owns the whole AGPR file

    // ---- fragment addresses: lane (fhi, li) addresses rows fhi * 8 + li / 4 (+ 4 for the second half), 4 columns at ((lane >> 4) & 1) * 16 +
    // (li & 3) * 4 of its 32-column fragment; k-steps and halves are immediate offsets.  Fragment f of a region (32 columns = chunks 4 f .. 4 f + 3)
    // sits at chunk group f ^ (row & 3): byte (f << 6) ^ swz - ONE lane base per operand and the swizzle key instead of ten addresses held in
    // registers through the main loop (made opaque at every use: the compiler would hoist the ten sums out of the loop again) ----
    const int frow0 = fhi * 8 + (li >> 2), fcol = ((lane >> 4) & 1) * 16 + (li & 3) * 4;
    const uint32_t lane_base = smem_lds + (uint32_t)(frow0 * RB + (fcol >> 3) * 16 + (fcol & 7) * 2);
    const uint32_t swz = (uint32_t)((frow0 & 3) << 6);
    const int tb0 = wb * FB;                                          // B fragment i of this wave = fragment (tb0 + i) % 4 of region 2 + (tb0 + i) / 4
    // ---- DMA lane pattern: lane l of a piece covers row l / 16 of its 4 rows, chunk (l % 16) ^ ((l / 16) << 2) ----
    const int prow = lane >> 4, pch = (lane & 15) ^ (prow << 2);

    union Frag { bf16x8_t v; s16x4_t h[2]; };

    auto run = [&](int tile, int t_begin, int t_end) __attribute__((always_inline)) {
        if (t_begin >= t_end) return;
        int jx = 0;
        while (jx + 1 < a.n_jobs && tile >= a.job[jx + 1].tile_start) ++jx;
        const TNJob& jb = a.job[jx];
        const int local = tile - jb.tile_start;
        const int n0 = (local / jb.n_tiles_b) * TA, k0 = (local % jb.n_tiles_b) * TB;
        const int lda = jb.lda, ldb = jb.ldb;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(jb.A), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(jb.B), 0, 0x7fffffff, 0x00020000);
        const uint32_t voff_a = (uint32_t)(prow * lda * 2 + pch * 16), voff_b = (uint32_t)(prow * ldb * 2 + pch * 16);
        // first column of each region, clamped into the operand (regions past its last column: results dropped by the epilogue)
        int colr[NREG];
#pragma unroll
        for (int g = 0; g < NREG; ++g) {
            const bool isa = g < TA / RG;
            const int c = isa ? n0 + g * RG : k0 + (g - TA / RG) * RG, cols = isa ? jb.a_cols : jb.b_cols;
            colr[g] = c <= cols - RG ? c : cols - RG;
        }
        // piece q (< NPA: A) of K-tile t into stage buffer `buf`
        auto dma = [&](int t, int buf, auto q_c) __attribute__((always_inline)) {
            constexpr int q = decltype(q_c)::value, g = q / PPR, p = q % PPR;
            const int rows = t * KB + (p * NWAVES + wave) * 4;
            char* dst = smem + buf * STAGE + g * REGION + (p * NWAVES + wave) * 1024;
            if constexpr (q < NPA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(dst), 16, voff_a, (rows * lda + colr[g]) * 2, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(dst), 16, voff_b, (rows * ldb + colr[g]) * 2, 0, 0);
        };
        const int NT_ = t_end - t_begin, t_last = t_end - 1;
        // prologue: stage 0 and the A pieces of stage 1 (its B pieces ride on k-step 0 of the first K-tile: the steady-state pattern).  K-tile
        // indices past the segment are clamped to its last K-tile: those pieces land in a buffer nobody reads any more (the main loop has ONE
        // straight-line body - variants of it for the segment's ends cost more registers than the allocator has: see the note at k_step).
        static_for<0, NPA + NPB>([&](auto qc) __attribute__((always_inline)) { dma(t_begin, 0, qc); });
        static_for<0, NPA>([&](auto qc) __attribute__((always_inline)) { dma(min(t_begin + 1, t_last), 1, qc); });
        wait_vmcnt<NPA>();
        block_barrier();

        Frag af[2][FA], bfr[FB];
        f32x16_t acc_v[NMF - 16];
#pragma unroll
        for (int x = 0; x < NMF - 16; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_v[x][r] = 0.f;
        agpr_zero_all();
        // address of A fragment j / B fragment i in the stage at byte offset st.  Opaque when it leaves: the sum is then computed BEFORE the asm
        // statement that follows in source order (the MFMA the read rides on) - computed behind it, the compiler builds it in the registers the read
        // is about to fill, i.e. writes a source register of the MFMA issued one instruction earlier.
        auto addr_a = [&](uint32_t st, int j) __attribute__((always_inline)) {
            uint32_t key = swz;
            asm volatile("" : "+v"(key));
            uint32_t ad = lane_base + (st + (uint32_t)(wa * REGION)) + (key ^ (uint32_t)(j << 6));
            asm volatile("" : "+v"(ad));
            return ad;
        };
        auto addr_b = [&](uint32_t st, int i) __attribute__((always_inline)) {
            uint32_t key = swz;
            asm volatile("" : "+v"(key));
            const int t = tb0 + i;
            uint32_t ad = lane_base + (st + (uint32_t)((TA / RG + (t >> 2)) * REGION)) + (key ^ (uint32_t)((t & 3) << 6));
            asm volatile("" : "+v"(ad));
            return ad;
        };
        auto read_a = [&](uint32_t ad, auto kk_c, auto par_c, auto j_c) __attribute__((always_inline)) {
            constexpr int kk = decltype(kk_c)::value, PAR = decltype(par_c)::value, j = decltype(j_c)::value;
            af[PAR][j].h[0] = tr_read4<kk * 16 * RB>(ad);
            af[PAR][j].h[1] = tr_read4<kk * 16 * RB + 4 * RB>(ad);
        };
        auto read_b = [&](uint32_t ad, auto kk_c, auto i_c) __attribute__((always_inline)) {
            constexpr int kk = decltype(kk_c)::value, i = decltype(i_c)::value;
            bfr[i].h[0] = tr_read4<kk * 16 * RB>(ad);
            bfr[i].h[1] = tr_read4<kk * 16 * RB + 4 * RB>(ad);
        };
        static_for<0, FA>([&](auto jc) __attribute__((always_inline)) { read_a(addr_a(0u, decltype(jc)::value), IC<0>{}, IC<0>{}, jc); });
        static_for<0, FB>([&](auto ic) __attribute__((always_inline)) { read_b(addr_b(0u, decltype(ic)::value), IC<0>{}, ic); });
        wait_lgkmcnt<0>();
        __builtin_amdgcn_sched_barrier(0);

        // One k-step on the fragments in registers (A set PAR): MFMA q = (i, j) = (q / 4, q % 4) -> accumulator fragment q, each followed by
        // what rides on it: the transpose reads of the NEXT k-step (st_next: byte offset of its stage; KN: its k-step) - A fragment
        // q behind MFMA q < 4 into the other A set, B fragment i behind its last MFMA - and at most one DMA piece (P0 .. P0 + NP - 1 of K-tile
        // t_dma into buf_dma) on the MFMAs that carry no read.  LDS reads return in order: "at most 4 outstanding" before each group of four
        // MFMAs leaves only the two fragments read last in flight (the fragments a group needs were read at least 16 MFMAs earlier).
        auto k_step = [&](auto par_c, int st_next, auto kn_c, int t_dma, int buf_dma, auto p0_c, auto np_c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_c)::value, P0 = decltype(p0_c)::value, NP = decltype(np_c)::value;
            static_for<0, NMF>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, i = q / FA, j = q % FA;
                uint32_t ad_a = 0, ad_b = 0;
                if constexpr (q < FA) ad_a = addr_a((uint32_t)st_next, q);
                if constexpr (j == FA - 1) ad_b = addr_b((uint32_t)st_next, i);
                if constexpr (j == 0) wait_lgkmcnt<4>();
                if constexpr (q < 16) mfma_agpr_clob<q, false>(af[PAR][j].v, bfr[i].v);
                else mfma_vgpr(acc_v[q - 16], af[PAR][j].v, bfr[i].v);
                if constexpr (q < FA) read_a(ad_a, kn_c, IC<PAR ^ 1>{}, IC<(q < FA ? q : 0)>{});
                if constexpr (j == FA - 1) read_b(ad_b, kn_c, IC<i>{});
                if constexpr (NP > 0 && q >= FA && j != FA - 1) {
                    constexpr int s_idx = (q - FA) - (q - FA) / FA;              // index among the 15 MFMAs that carry no read
                    constexpr int NFREE = (NMF - FA) - (NMF - FA) / FA;
                    constexpr int NPS = NP > 0 ? NP : 1, STEP = NFREE / NPS > 0 ? NFREE / NPS : 1;
                    if constexpr (s_idx % STEP == 0 && s_idx / STEP < NP) dma(t_dma, buf_dma, IC<P0 + s_idx / STEP>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using I0 = IC<0>; using I1 = IC<1>; using I2 = IC<2>; using I3 = IC<3>;
        // ONE loop body for every K-tile (no first / last variants: with tied 16-register accumulator operands flowing through a loop whose body
        // branches into variants, the register allocator gives each variant its own tuples and copies between them - 2 x 128 registers).
        int buf = 0;
        for (int it = 0; it < NT_; ++it) {
            const int st = buf * STAGE, stn = (buf ^ 1) * STAGE;
            k_step(I0{}, st, I1{}, min(t_begin + it + 1, t_last), buf ^ 1, IC<NPA>{}, IC<NPB>{});      // + the B pieces of stage it + 1
            k_step(I1{}, st, I2{}, 0, 0, I0{}, I0{});
            k_step(I0{}, st, I3{}, 0, 0, I0{}, I0{});
            // stage it + 1 has landed for everyone, stage `it` has been read by everyone (k-step 3's fragments are in registers)
            wait_vmcnt<0>();
            wait_lgkmcnt<0>();
            block_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // k-step 3: the first fragments of stage it + 1 ride on it (stale bytes, unused, behind the last K-tile), and the A pieces of
            // stage it + 2 into the buffer the barrier freed
            k_step(I1{}, stn, I0{}, min(t_begin + it + 2, t_last), buf, I0{}, IC<NPA>{});
            buf ^= 1;
        }
        // ---- epilogue: fp32 atomics; accumulator fragment (i, j): registers <-> n (A columns), lanes <-> k (B columns) ----
        asm volatile("s_nop 15\n\ts_nop 15");          // the last MFMAs' results (asm: no hazard handling by the compiler)
        float* out = jb.out;
        const int N = jb.N, K = jb.K, ldo = jb.ldo;
        static_for<0, FB>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            const int k = k0 + wb * (TB / 2) + i * 32 + (lane & 31);
            if (k < K && !(a.lab & 1)) {
                static_for<0, FA>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value, f = i * FA + j;
                    const int nb = n0 + wa * (TA / 2) + j * 32 + 4 * fhi;
                    if (nb < N) {              // (N % 128 == 0: a 32-column fragment is wholly inside or outside)
                        float* o = out + (size_t)nb * ldo + k;
                        static_for<0, 16>([&](auto rc) __attribute__((always_inline)) {
                            constexpr int r = decltype(rc)::value;
                            float v;
                            if constexpr (f < 16) v = read_agpr<f, r>(); else v = acc_v[f - 16][r];
                            unsafeAtomicAdd(o + (size_t)((r & 3) + 8 * (r >> 2)) * ldo, v);
                        });
                    }
                });
            }
        });
        // (no barrier: every wave's LDS reads of this segment were complete before its last barrier - the next prologue may refill both stages)
    };
    tn_multi_walk(a, w, G, run);
#endif
}

}  // namespace


int fm_launch_tn4_multi(const fmk::TNMultiArgs& a, int grid, hipStream_t s) {
    auto k = gemm_tn4_multi_kernel;
    const size_t lds = (size_t)2 * STAGE;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, a);
    return 0;
}
