// Flattened persistent NT GEMM (gfx950):  out[m][n] = sum_k X[m][k] * W[n][k]  for the big dense launches of the trunk.
//
// Same arithmetic, fragment maps, LDS image and ping-pong schedule as gemm_nt_kernel<.., PP, PERSIST> in gemm.hip (results are
// bit-identical: one accumulation order), different shape in TIME.  What the tile-at-a-time kernel pays per output tile —
// measured with tools/gemm_lab ablations, profiles/r02_lab_nt_ablation_*.txt, 128 x 256 tiles at N = 2304, K = 768:
//   * ~20 % : the LDS ring drains and refills at every tile boundary (the first DMA of a tile is waited for with nothing to do);
//   * ~18 % : the store tail — 8 waves issue all their stores at once (issue bound) and, because the stores sit behind the next
//             tile's prologue DMA in one in-order vmcnt queue, the next tile starts with a full drain;
// is removed here by treating the K-tiles of ALL tiles of a workgroup as ONE stream:
//   * iteration g = (tile j, k-tile kt) runs j-major; the DMA stream simply stays 2 (3 for the trailing wave row) iterations
//     ahead of the MFMA stream across tile boundaries, so the ring never drains;
//   * at a tile's last k-tile the accumulators are converted to their output format INTO REGISTERS and cleared; the stores leave
//     one per MFMA half during the next tile (buffer stores with hardware bounds checking: always exactly one VMEM instruction,
//     so the counted vmcnt waits stay exact: vmcnt is one in-order queue of loads and stores on gfx950);
//   * both wave rows issue their LDS-DMA pieces between the MFMAs of their MFMA half (the trailing row one slot earlier than in
//     gemm.hip: it targets the buffer both rows finished reading at the previous barrier), so the two halves of a slot are
//     symmetric: 16 fragment reads on one row against 16 MFMAs + 6 DMA pieces + 1 store on the other.
//
// Slots (between consecutive workgroup barriers), iteration g, stage = LDS ring buffer g % 3:
//     even slot g : lead  MEM_g  (reads stage g)                  | trail MFMA_{g-1}, issues its pieces of stage g+2, 1 store
//     odd  slot g : lead  MFMA_g, issues its pieces of stage g+2, | trail MEM_g (reads stage g), waits for its pieces of g+1
//                   1 store, waits for its pieces of stage g+1    |
// RAW: every wave waits (counted) for its own pieces of stage g+1 in odd slot g; the barrier closing that slot orders them for the
//      lead's reads (even slot g+1) and the trail's (odd slot g+1).
// WAR: stage g+2 reuses the buffer of stage g-1, last read by the trail in odd slot g-1, whose closing barrier precedes even slot g.
//
// Handles: dense, no bias, 16-byte aligned outputs with N % 8 == 0, K % 64 == 0, K / 64 >= the stores per tile (else returns 0 and
// gemm.hip's kernels run).  Epilogues: FM_EPI_BF16, FM_EPI_SWIGLU, FM_EPI_RESIDUAL.
#include <type_traits>
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"

namespace {
using namespace fmk;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// one bounds-checked 16-byte store: lanes whose offset has bit 31 set (row offset OOB + in-range column bytes) fall outside
// num_records and are dropped by the hardware
__device__ __forceinline__ void bstore16(u32x4_t v, __amdgpu_buffer_rsrc_t r, uint32_t voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
constexpr uint32_t OOB = 0x80000000u;

template <int TW, int TX, int KB, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_flat_kernel(NTArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the stub: the body is all device builtins)
    constexpr int WW = 2, WX = 4, NWAVES = 8, STAGES = 3;
    constexpr int RB = KB * 2, CPR = RB / 16, RPP = 1024 / RB, SWSH = (RB == 128) ? 1 : 2;
    constexpr int PW = TW / (RPP * NWAVES), PX = TX / (RPP * NWAVES), LOADS = PW + PX;
    constexpr int FW = TW / WW / 32, FX = TX / WX / 32;
    constexpr int STAGE = (TW + TX) * RB;
    constexpr int KS = KB / 16;
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;        // output features per W tile
    // deferred stores per wave and tile (16 bytes per lane each)
    static_assert(EPI == EPI_BF16 || EPI == EPI_SWIGLU, "epilogues of the flattened kernel");
    constexpr int NST = EPI == EPI_BF16 ? FX * FW * 2 : FX * (FW / 2) * 2 * 3;
    constexpr int NPK = NST;                                      // packed 16-byte values kept per tile
    static_assert(TW % (RPP * NWAVES) == 0 && TX % (RPP * NWAVES) == 0, "tile rows must split evenly over the DMA pieces");
    static_assert(EPI != EPI_SWIGLU || FW % 2 == 0, "SwiGLU needs (g,u) fragment pairs per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;
    const bool lead = ww == 0;
    const int frow = lane & 31, fhi = lane >> 5;
    const int fswz = (frow >> SWSH) & (CPR - 1);

    const int total = a.n_tiles_w * a.n_tiles_x;
    const int n_my = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int KT = a.K / KB;
    const int G = n_my * KT;
    const int N = a.N;

    // ---- the DMA stream: (tile s_j, k-tile s_kt) of stage s_g ---------------------------------------------------------------------
    // Sources are addressed through buffer descriptors (one for the W rows, one for the X rows of the current tile) with a
    // per-lane 32-bit byte offset computed once per tile and the k advance in the scalar offset operand: a piece is
    // `s_mov m0; buffer_load_dwordx4 voff, rsrc, soff offen lds` - no vector address arithmetic in the MFMA half, and half the
    // address payload of a flat 64-bit load.  (Offsets stay below 2^31: a tile's rows span at most TX * ld * 2 bytes.)
    uint32_t woff[PW], xoff[PX];
    __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.W), rs_w2 = make_rsrc(a.W2 ? a.W2 : a.W), rs_x = make_rsrc(a.X);
    (void)rs_w2;
    auto tile_origin = [&](int j, int& n0, int& m0) {
        const int tile = xcd_remap((int)blockIdx.x + j * (int)gridDim.x, total);
        n0 = (tile % a.n_tiles_w) * NPT; m0 = (tile / a.n_tiles_w) * TX;      // W tiles fastest: the X tile is shared in L2
    };
    auto set_sources = [&](int j) {
        int n0, m0;
        tile_origin(j, n0, m0);
        rs_w = make_rsrc(a.W + (size_t)n0 * a.ldw);
        if constexpr (EPI == EPI_SWIGLU) rs_w2 = make_rsrc(a.W2 + (size_t)n0 * a.ldw);
        rs_x = make_rsrc(a.X + (size_t)m0 * a.ldx);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> SWSH) & (CPR - 1));
            int r;                                               // row of this tile's W block
            if constexpr (EPI == EPI_SWIGLU) r = (t >> 6) * 32 + (t & 31);     // rows [0,32) of a 64-row group: g (W), [32,64): u (W2)
            else r = t;
            r = n0 + r < N ? r : N - 1 - n0;
            woff[p] = (uint32_t)r * (uint32_t)a.ldw * 2u + (uint32_t)lc * 16u;
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> SWSH) & (CPR - 1));
            const int r = m0 + t < a.M ? t : a.M - 1 - m0;
            xoff[p] = (uint32_t)r * (uint32_t)a.ldx * 2u + (uint32_t)lc * 16u;
        }
    };
    int s_g = 0, s_kt = 0, s_j = 0, s_buf = 0;
    auto stage_piece = [&](int q) {            // piece q of stage s_g (q < PW: W rows, else X rows)
        const int soff = s_kt * (KB * 2);
        if (q < PW) {
            const int pi = q < PW ? q : 0;
            const int t0 = (pi * NWAVES + wave) * RPP;           // first tile row of the piece (wave uniform): picks W or W2
            if (EPI == EPI_SWIGLU && ((t0 >> 5) & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, LDS_PTR(smem + s_buf * STAGE + (pi * NWAVES + wave) * 1024), 16, woff[pi], soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, LDS_PTR(smem + s_buf * STAGE + (pi * NWAVES + wave) * 1024), 16, woff[pi], soff, 0, 0);
        } else {
            const int pi = q >= PW ? q - PW : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, LDS_PTR(smem + s_buf * STAGE + TW * RB + (pi * NWAVES + wave) * 1024), 16, xoff[pi], soff, 0, 0);
        }
    };
    auto stage_advance = [&]() {               // after the last piece of stage s_g
        ++s_g; ++s_kt;
        s_buf = s_buf + 1 == STAGES ? 0 : s_buf + 1;
        if (s_kt == KT) { s_kt = 0; ++s_j; if (s_j < n_my) set_sources(s_j); }
    };
    auto stage_all = [&]() {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) stage_piece(q);
        stage_advance();
    };

    // ---- deferred epilogue state ---------------------------------------------------------------------------------------------
    u32x4_t pk[NPK];                            // the finished tile's outputs (bf16 packed), 16 bytes per lane each
    uint32_t rowoff[EPI == EPI_SWIGLU ? 2 * FX : FX];   // byte offset of this lane's rows inside the finished tile's row block (OOB if none)
    uint32_t coloff = 0;                        // byte offset of this lane's first column chunk
    __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out), rs_out2 = make_rsrc(a.out2 ? a.out2 : a.out);
    int pend = NST;                             // next deferred store (NST = nothing pending)
    (void)rs_out2;

    f32x16_t acc[FW][FX];
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // store number `s` of the finished tile (compile-time s: the values live in registers)
    auto store_one = [&](auto s_c) {
        constexpr int s = decltype(s_c)::value;
        if constexpr (EPI == EPI_BF16) {
            // s = (j * FW + i) * 2 + gp : 16 columns [i*32 + 16*gp, +16) of row block j; this lane's 8 of them start at + 8*fhi
            constexpr int j = s / (FW * 2), i = (s / 2) % FW, gp = s % 2;
            bstore16(pk[s], rs_out, rowoff[j] + coloff + (uint32_t)((i * 32 + 16 * gp) * 2));
        } else if constexpr (EPI == EPI_SWIGLU) {
            // s = which * (FX * FW/2 * 2) + (j * (FW/2) + ip) * 2 + gp ; which = 0: g, 1: u (out2), 2: act (out)
            constexpr int per = FX * (FW / 2) * 2;
            constexpr int which = s / per, r = s % per, j = r / ((FW / 2) * 2), ip = (r / 2) % (FW / 2), gp = r % 2;
            const uint32_t col = coloff + (uint32_t)((ip * 32 + 16 * gp) * 2);
            if constexpr (which == 2) bstore16(pk[s], rs_out, rowoff[j] + col);
            else bstore16(pk[s], rs_out2, rowoff[FX + j] + col + (which == 1 ? (uint32_t)a.Hp * 2u : 0u));
        }
    };
    auto store_switch = [&](int s) {
        // uniform dispatch on the run-time store index
#define FM_ST(k) case k: if constexpr (k < NST) store_one(std::integral_constant<int, (k < NST ? k : 0)>{}); break;
        switch (s) {
            FM_ST(0) FM_ST(1) FM_ST(2) FM_ST(3) FM_ST(4) FM_ST(5) FM_ST(6) FM_ST(7) FM_ST(8) FM_ST(9) FM_ST(10) FM_ST(11)
            FM_ST(12) FM_ST(13) FM_ST(14) FM_ST(15) FM_ST(16) FM_ST(17) FM_ST(18) FM_ST(19) FM_ST(20) FM_ST(21) FM_ST(22) FM_ST(23)
            default: break;
        }
#undef FM_ST
    };
    static_assert(NST <= 24, "store dispatch table");

    // the tile whose last k-tile was just multiplied: accumulators -> packed outputs, accumulators cleared
    auto finish_tile = [&](int j_done) {
        int n0, m0;
        tile_origin(j_done, n0, m0);
        // row offsets inside the output (bytes), per row block; rows past M are dropped
        const size_t esz = 2;
#pragma unroll
        for (int j = 0; j < FX; ++j) {
            const int m = m0 + wx * (TX / WX) + j * 32 + frow;
            rowoff[j] = m < a.M ? 0u : OOB;
        }
        if constexpr (EPI == EPI_BF16) {
            // per-tile resource: base = first row of the tile, offsets stay far below 2 GB whatever the matrix size
            rs_out = make_rsrc((const char*)a.out + (size_t)m0 * a.ldo * esz);
#pragma unroll
            for (int j = 0; j < FX; ++j) rowoff[j] |= (uint32_t)((wx * (TX / WX) + j * 32 + frow) * a.ldo) * 2u;
            const int c0 = n0 + ww * (TW / WW) + 8 * fhi;                       // this lane's first column (chunk 0)
            coloff = (uint32_t)c0 * 2u;
            // columns past N: N % 8 == 0 and chunks are 8 wide, so a chunk is all in or all out
#pragma unroll
            for (int j = 0; j < FX; ++j)
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const int g = 2 * gp;
                        uint2 p0 = make_uint2(pack2bf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                        uint2 p1 = make_uint2(pack2bf(acc[i][j][4 * g + 4], acc[i][j][4 * g + 5]), pack2bf(acc[i][j][4 * g + 6], acc[i][j][4 * g + 7]));
                        const auto x = __builtin_amdgcn_permlane32_swap(p0.x, p1.x, false, false);
                        const auto y = __builtin_amdgcn_permlane32_swap(p0.y, p1.y, false, false);
                        u32x4_t v = {x[0], y[0], x[1], y[1]};
                        pk[(j * FW + i) * 2 + gp] = v;
                    }
            // (no ragged right edge: the launcher requires N % TW == 0)
        } else if constexpr (EPI == EPI_SWIGLU) {
            rs_out = make_rsrc((const char*)a.out + (size_t)m0 * a.ldo * esz);
            rs_out2 = make_rsrc((const char*)a.out2 + (size_t)m0 * a.ldo2 * esz);
#pragma unroll
            for (int j = 0; j < FX; ++j) {
                const uint32_t r = (uint32_t)(wx * (TX / WX) + j * 32 + frow);
                rowoff[FX + j] = rowoff[j] | (r * (uint32_t)a.ldo2 * 2u);
                rowoff[j] |= r * (uint32_t)a.ldo * 2u;
            }
            const int h0 = n0 + (ww * (TW / WW) / 64) * 32 + 8 * fhi;           // this lane's first hidden unit (chunk 0)
            coloff = (uint32_t)h0 * 2u;
            constexpr int per = FX * (FW / 2) * 2;
#pragma unroll
            for (int j = 0; j < FX; ++j)
#pragma unroll
                for (int ip = 0; ip < FW / 2; ++ip)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        uint2 pg_[2], pu_[2], pa_[2];
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
                            pg_[u] = make_uint2(pack2bf(gv[0], gv[1]), pack2bf(gv[2], gv[3]));
                            pu_[u] = make_uint2(pack2bf(uv[0], uv[1]), pack2bf(uv[2], uv[3]));
                            pa_[u] = make_uint2(pack2bf(av[0], av[1]), pack2bf(av[2], av[3]));
                        }
                        const int r = (j * (FW / 2) + ip) * 2 + gp;
                        auto swz = [&](const uint2 (&p)[2]) {
                            const auto x = __builtin_amdgcn_permlane32_swap(p[0].x, p[1].x, false, false);
                            const auto y = __builtin_amdgcn_permlane32_swap(p[0].y, p[1].y, false, false);
                            u32x4_t v = {x[0], y[0], x[1], y[1]};
                            return v;
                        };
                        pk[0 * per + r] = swz(pg_); pk[1 * per + r] = swz(pu_); pk[2 * per + r] = swz(pa_);
                    }
        }
#pragma unroll
        for (int i = 0; i < FW; ++i)
#pragma unroll
            for (int j = 0; j < FX; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        pend = (EPI == EPI_SWIGLU && !a.out2) ? 2 * (NST / 3) : 0;             // inference: only the activation is written
    };

    // ---- prologue: stages 0, 1 (and 2 for the trailing row, which runs its DMA stream one iteration further ahead) --------------
    set_sources(0);
    stage_all();
    if (G > 1) stage_all();
    if (G > 1) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
    block_barrier();                                     // stage 0 is in LDS for everyone
    bool dma_prev = false;                               // trailing row: did its previous MFMA half issue DMA pieces?
    int st_prev = 0;                                     //               ... and a deferred store?
    if (!lead) {
        if (G > 2) { stage_all(); dma_prev = true; }
        block_barrier();                                 // the trailing row runs one barrier behind
    }

    int buf = 0, c_kt = 0, c_j = 0;
    for (int g = 0; g < G; ++g) {
        // ---- MEM half: every fragment of stage g -----------------------------------------------------------------------
        const char* wt = smem + buf * STAGE + (ww * (TW / WW) + frow) * RB;
        const char* xt = smem + buf * STAGE + TW * RB + (wx * (TX / WX) + frow) * RB;
        bf16x8_t wf[KS][FW], xf[KS][FX];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int off = ((kk * 2 + fhi) ^ fswz) * 16;
#pragma unroll
            for (int i = 0; i < FW; ++i) wf[kk][i] = *(const bf16x8_t*)(wt + i * 32 * RB + off);
#pragma unroll
            for (int j = 0; j < FX; ++j) xf[kk][j] = *(const bf16x8_t*)(xt + j * 32 * RB + off);
        }
        if (!lead) {
            // the trailing row's pieces of stage g+1 (issued two MFMA halves ago) must have landed before the closing barrier;
            // younger in its queue: the pieces of stage g+2 and the store of its previous MFMA half, when there were any
            if (g + 1 < G) {
                if (dma_prev) { if (st_prev) wait_vmcnt<LOADS + 1>(); else wait_vmcnt<LOADS>(); }
                else { if (st_prev) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
            }
        }
        wait_lgkmcnt<0>();
        block_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA half: 16 MFMAs, this wave's DMA pieces of stage g+2 (lead) / g+3 (trail) between them, one deferred store -------
        const bool more = lead ? (g + 2 < G) : (g + 3 < G);
        if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
                for (int j = 0; j < FX; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
            if (more) {
#pragma unroll
                for (int q = kk * LOADS / KS; q < (kk + 1) * LOADS / KS; ++q) stage_piece(q);
            }
        }
        if (a.prio) __builtin_amdgcn_s_setprio(0);
        int st_now = 0;
        if (pend < NST) { store_switch(pend); ++pend; st_now = 1; }
        if (more) stage_advance();
        if (lead) {
            // the lead's pieces of stage g+1 (previous MFMA half); younger: this half's pieces of stage g+2 and its store
            if (g + 1 < G) {
                if (more) { if (st_now) wait_vmcnt<LOADS + 1>(); else wait_vmcnt<LOADS>(); }
                else { if (st_now) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
            }
        } else { dma_prev = more; st_prev = st_now; }
        __builtin_amdgcn_sched_barrier(0);
        if (lead || g + 1 < G) block_barrier();          // barrier counts: lead 1 + 2G, trail 2 + 2G - 1
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        // ---- tile boundary of the MFMA stream ------------------------------------------------------------------------------
        if (++c_kt == KT) {
            // (stores of the previous tile all left: NST <= KT halves have passed)
            finish_tile(c_j);
            c_kt = 0; ++c_j;
        }
    }
    // the last tile's outputs
    for (; pend < NST; ++pend) store_switch(pend);
#endif
}

template <int TW, int TX, int KB, int EPI>
int launch_flat(NTArgs a, hipStream_t s) {
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;
    a.n_tiles_w = (a.N + NPT - 1) / NPT;
    a.n_tiles_x = (a.M + TX - 1) / TX;
    int grid = a.n_tiles_w * a.n_tiles_x;
    static int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n / 8 * 8;
    }();
    if (grid > cus) grid = cus;
    const size_t lds = (size_t)3 * (TW + TX) * KB * 2;
    auto k = gemm_nt_flat_kernel<TW, TX, KB, EPI>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);
    if (hipGetLastError() != hipSuccess) return -2;
    return 1;
}

}  // namespace

int g_nt_flat = 0;      // experimental (measured equal to the tile-at-a-time kernels, profiles/r02_lab_nt_flat.txt): fm_set_gemm_nt_config bit 29 turns it on

int fm_launch_nt_flat(const fmk::NTArgs& a, int epilogue, hipStream_t s) {
    using namespace fmk;
    if (!g_nt_flat || a.groups || a.bias || a.bias2) return 0;
    if (a.M < 2048 || a.K % 64 != 0) return 0;
    const bool al16 = (((uintptr_t)a.out | (uintptr_t)a.out2) & 15) == 0;
    if (epilogue == FM_EPI_BF16) {
        constexpr int TW = 128, NST = 8;
        if (a.N % TW != 0 || a.ldo % 8 != 0 || !al16 || a.K / 64 < NST) return 0;
        if ((size_t)256 * a.ldo * 2 >= 0x7fffffffull) return 0;
        return launch_flat<128, 256, 64, EPI_BF16>(a, s);
    }
    if (epilogue == FM_EPI_SWIGLU) {
        constexpr int NST = 12;
        if (a.N % 64 != 0 || a.Hp % 8 != 0 || a.ldo % 8 != 0 || (a.out2 && a.ldo2 % 8 != 0) || !al16 || a.K / 64 < NST || !a.W2) return 0;
        if ((size_t)256 * (a.out2 ? a.ldo2 : a.ldo) * 2 >= 0x7fffffffull) return 0;
        return launch_flat<128, 256, 64, EPI_SWIGLU>(a, s);
    }
    return 0;
}
