// bf16 MFMA GEMMs for the 4M trunk (gfx950).
//
//   NT  out[m][n] = sum_k X[m][k] * W[n][k]          forward Linear and dX (with a W^T shadow)
//   TN  out[n][k] += sum_r A[r][n] * B[r][k]         dW = dY^T X   (fp32 accumulate into the grad)
//
// Both use v_mfma_f32_32x32x16_bf16 with the *weight-like* operand on the MFMA row side, so that in
// the accumulator a lane holds 4 consecutive output features of one token row: epilogues (bias, GELU,
// SwiGLU, residual add) run on registers and stores are 8/16-byte row-contiguous chunks.
//
// LDS tiles are filled with global_load_lds (16 B per lane, LDS image is lane-linear) and read with
// ds_read_b128; bank conflicts are removed by XOR-swizzling the 16-B chunk index on the *source*
// address and again on the read (guide rule 21).
//
// Contracts (the Python layer allocates every bf16 buffer and guarantees them):
//   * NT: the reduction dimension K is a multiple of 64 and zero padded (TN masks its reduction rows itself);
//   * leading dimensions are multiples of 8 elements (16-byte aligned rows).
#include <type_traits>
#include <cstdlib>
#include <algorithm>
#include "common.h"
#include "fourm_hip.h"
#include "gemm_args.h"

namespace {
using namespace fmk;

constexpr int BK = 64;           // reduction elements per LDS stage

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
// STAGES-deep LDS ring: the DMA of K-tile t+STAGES-1 is issued while tile t is being multiplied; waits are
// counted (never a full drain inside the loop) and there is one workgroup barrier per K-tile.
// KB = reduction elements per stage (64 or 32): LDS rows are KB*2 bytes, 16-byte chunks XOR-swizzled so that
// the 16 rows a ds_read_b128 lane group touches fall on 16 different 16-byte bank slots.
//
// PP ("ping-pong", 8 waves, 3 stages): the two wave rows (waves 0-3 / 4-7; wave i and i+4 share a SIMD) run
// the same loop one barrier apart, so that on every SIMD one wave is in its MFMA half while the other is in
// its LDS-read half.  Every K-tile iteration is  [MEM: all fragment reads of the tile] barrier [MFMA: 16 or 8
// MFMAs] barrier.  Wall-clock slots (between consecutive workgroup barriers) alternate
//     even: row 0 MEM_k   | row 1 MFMA_{k-1}        odd: row 0 MFMA_k | row 1 MEM_k
// All LDS-DMA for tile k+2 is issued, and each wave's counted wait for its pieces of tile k+1 is done, in the
// ODD slot k by both rows, so the barrier closing that slot orders tile k+1 for row 0 (read in the next slot)
// and for row 1 (one slot later).  Tile k+2 overwrites the buffer of tile k-1, whose last reads (row 1, odd
// slot k-1) were retired by lgkmcnt(0) before that slot's closing barrier.
//
// PERSIST (with PP, dense only): the workgroup walks tiles blockIdx.x, + gridDim.x, ... of the same XCD-aware order.
// The LDS-DMA of the next tile's first stages is issued BEFORE the store epilogue of the current one (every LDS read
// of a tile is retired before the last barrier both wave rows pass), so that latency and the workgroup relaunch
// disappear under the stores.  The first wait of such a tile is a full vmcnt(0): the wave's own epilogue stores are
// younger than those loads and vmcnt only promises order among loads.
// (a 16-byte block of zeros: the source of LDS-DMA pieces that must contribute nothing - reduction rows past the live count of the TN kernels,
// taps outside the image of the implicit convolution)
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0u, 0u, 0u, 0u};

// CONV (dense, not persistent): the X operand is gathered - see NTArgs::conv_*: row t of the tile is an output pixel, K-tile kt lies inside ONE
// tap (conv_C % KB == 0) and reads KB channels of the input pixel that tap selects, or zeros.  Everything else (LDS layout, main loop, epilogues,
// the K-slices of a split launch) is the plain kernel's: the result is bit-identical to fm_unet_im2col + the plain launch.
template <int TW, int TX, int WW, int WX, int KB, int STAGES, int EPI, bool GROUPED, bool PP = false, bool PERSIST = false, bool CONV = false>
__global__ __launch_bounds__(WW * WX * 64) void gemm_nt_kernel(NTArgs a) {
    constexpr int NWAVES = WW * WX;
    constexpr int RB = KB * 2;                                 // bytes per LDS row
    constexpr int CPR = RB / 16;                               // 16-byte chunks per row (8 or 4)
    constexpr int RPP = 1024 / RB;                             // rows per 1-KiB DMA piece (8 or 16)
    constexpr int SWSH = (RB == 128) ? 1 : 2;                  // rows per 256-byte bank row = 1 << SWSH
    constexpr int LOADS = (TW + TX) / (RPP * NWAVES);          // LDS-DMA instructions per wave per stage
    constexpr int FW = TW / WW / 32, FX = TX / WX / 32;        // 32x32 fragments per wave
    constexpr int STAGE = (TW + TX) * RB;
    static_assert(TW % (RPP * NWAVES) == 0 && TX % (RPP * NWAVES) == 0, "tile rows must split evenly over the DMA pieces");
    static_assert(EPI != EPI_SWIGLU || FW % 2 == 0, "SwiGLU needs (g,u) fragment pairs per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ww = wave / WX, wx = wave % WX;

    // ---- which tile ------------------------------------------------------------------------
    int tx, tw;
    if constexpr (GROUPED) {
        // Head GEMMs: a vocabulary matrix (up to 46 MB) does not fit an L2 and the groups have very different
        // widths (4096 ... 30000 of max_N columns), so (1) tiles are ordered in column blocks of `gw` W-tiles: all X
        // tiles sweep one block, then the next (W is fetched about once, the much smaller X tiles once per block);
        // (2) the order is cut into chunks of gw x 4 tiles (one L2 working set) dealt ROUND-ROBIN to the 8 XCDs
        // (dispatch puts workgroup b on XCD b % 8), so the tiles a narrow group leaves empty thin out every XCD's
        // share evenly instead of emptying some XCDs (contiguous shares: 465 TF on a 4096 | 30000 pair, dense: 730).
        constexpr int GX = 4;
        const int gw = a.group_w, chunk_sz = gw * GX;
        const int nblk = (a.n_tiles_w + gw - 1) / gw, nxg = (a.n_tiles_x + GX - 1) / GX;
        const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
        const int chunk = (j / chunk_sz) * 8 + xcd, within = j % chunk_sz;
        const int blk = chunk / nxg, txg = chunk % nxg;
        tx = txg * GX + within / gw; tw = blk * gw + within % gw;
        if (blk >= nblk || tx >= a.n_tiles_x || tw >= a.n_tiles_w) return;
    } else {
        const int tile = xcd_remap(blockIdx.x, a.n_tiles_w * a.n_tiles_x);
        tx = tile / a.n_tiles_w; tw = tile % a.n_tiles_w;         // W tiles fastest: X tile shared in L2
    }
    int conv_k0 = 0;                                          // CONV: first reduction index of this launch / K-slice
    if constexpr (EPI == EPI_F32 && !GROUPED && !PERSIST) {
        if (a.split_k > 1) {          // K-slice blockIdx.y of a split launch: its own operand columns, its own fp32 partial output
            const int z = blockIdx.y, k0 = z * a.k_slice;
            a.W += k0;
            if constexpr (CONV) conv_k0 = k0; else a.X += k0;
            a.K = min(a.k_slice, a.K - k0);
            a.out = (float*)a.out + (size_t)z * (size_t)a.split_stride;
        }
    }
    const bf16_t* Wp = a.W;
    int N = a.N, K = a.K, ldw = a.ldw;
    if constexpr (GROUPED) {
        const int g = a.tile_group[tx];
        if (g < 0) return;
        Wp = (const bf16_t*)a.groups[g].W; N = a.groups[g].N; K = a.groups[g].K; ldw = a.groups[g].ldw;
    }
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;        // output features per W tile
    int n0 = tw * NPT, m0 = tx * TX;
    if (n0 >= N) return;

    // ---- staging: each wave-instruction moves RPP rows x RB bytes; the per-lane source addresses only
    // advance by KB elements per K-tile, so they are computed once per tile -----------------------
    constexpr int PW = TW / (RPP * NWAVES), PX = TX / (RPP * NWAVES);
    const bf16_t* wsrc[PW];
    const bf16_t* xsrc[PX];
    int cv_y[CONV ? PX : 1], cv_x[CONV ? PX : 1];         // CONV: input coordinates of tap (0, 0) for the piece's output pixel
    int cs_c0 = 0, cs_ky = 0, cs_kx = 0;                   // CONV: first channel and tap of the K-tile staged next
    if constexpr (CONV) { const int tap = conv_k0 / a.conv_C; cs_c0 = conv_k0 - tap * a.conv_C; cs_ky = tap / 3; cs_kx = tap - cs_ky * 3; }
    auto set_sources = [&]() {
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> SWSH) & (CPR - 1));
            if constexpr (EPI == EPI_SWIGLU) {
                // rows [0,32) of every 64-row group come from W (g), rows [32,64) from W2 (u), same hidden units
                int n = n0 + (t >> 6) * 32 + (t & 31);
                n = n < N ? n : N - 1;
                wsrc[p] = (((t >> 5) & 1) ? a.W2 : Wp) + (size_t)n * ldw + lc * 8;
            } else {
                int n = n0 + t;
                n = n < N ? n : N - 1;
                wsrc[p] = Wp + (size_t)n * ldw + lc * 8;
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int t = (p * NWAVES + wave) * RPP + lane / CPR;
            const int lc = (lane % CPR) ^ ((t >> SWSH) & (CPR - 1));
            int m = m0 + t;
            if constexpr (CONV) {
                const bool live = m < a.M;
                m = live ? m : a.M - 1;
                const int hw = a.conv_Ho * a.conv_Wo, b = m / hw, r = m - b * hw, oy = r / a.conv_Wo, ox = r - oy * a.conv_Wo;
                cv_y[p] = live ? oy * a.conv_stride - 1 : -(1 << 20);          // (a dead row never passes the bounds test)
                cv_x[p] = ox * a.conv_stride - 1;
                xsrc[p] = a.X + (size_t)b * (a.conv_H >> a.conv_up) * (a.conv_W >> a.conv_up) * a.ldx + lc * 8;
                // no up-sampling: the pixel of tap (ky, kx) is (ky W + kx) rows behind the pixel of tap (0, 0) - a uniform offset per K-tile
                if (!a.conv_up) xsrc[p] += ((long long)cv_y[p] * a.conv_W + cv_x[p]) * a.ldx;
            } else {
                m = m < a.M ? m : a.M - 1;
                xsrc[p] = a.X + (size_t)m * a.ldx + lc * 8;
            }
        }
    };
    set_sources();
    auto stage_w = [&](int kt, int buf, int p) {
        __builtin_amdgcn_global_load_lds(GLB_PTR(wsrc[p] + kt * KB), LDS_PTR(smem + buf * STAGE + (p * NWAVES + wave) * 1024), 16, 0, 0);
    };
    auto stage_x = [&](int kt, int buf, int p) {
        if constexpr (CONV) {
            const int c0 = cs_c0, ky = cs_ky, kx = cs_kx;     // (the K-tiles of a launch are staged in order: the tap advances with them, no division per K-tile)
            const int y = cv_y[p] + ky, x = cv_x[p] + kx;
            const bool in = (unsigned)y < (unsigned)a.conv_H && (unsigned)x < (unsigned)a.conv_W;
            const bf16_t* src;
            if (!a.conv_up) src = xsrc[p] + ((long long)(ky * a.conv_W + kx) * a.ldx + c0);          // (uniform offset: scalar arithmetic)
            else src = xsrc[p] + (size_t)((y >> 1) * (a.conv_W >> 1) + (x >> 1)) * a.ldx + c0;
            src = in ? src : (const bf16_t*)g_zero16;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(smem + buf * STAGE + TW * RB + (p * NWAVES + wave) * 1024), 16, 0, 0);
        } else
        __builtin_amdgcn_global_load_lds(GLB_PTR(xsrc[p] + kt * KB), LDS_PTR(smem + buf * STAGE + TW * RB + (p * NWAVES + wave) * 1024), 16, 0, 0);
    };
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int p = 0; p < PW; ++p) stage_w(kt, buf, p);
#pragma unroll
        for (int p = 0; p < PX; ++p) stage_x(kt, buf, p);
        if constexpr (CONV) {                               // next K-tile: KB channels further, or the next tap
            cs_c0 += KB;
            if (cs_c0 == a.conv_C) { cs_c0 = 0; if (++cs_kx == 3) { cs_kx = 0; ++cs_ky; } }
        }
    };

    if (a.dephase_groups > 1) {       // experiment: start the workgroups of a CU / of the chip out of phase (epilogue of one under the main loop of another)
        const int ph = (gridDim.x > 256 ? blockIdx.x / 256 : blockIdx.x / 8) % a.dephase_groups;
        for (int i = 0; i < ph * a.dephase_step; ++i) __builtin_amdgcn_s_sleep(32);
    }
    f32x16_t acc[FW][FX];
    const int KT = K / KB;
    auto prologue = [&]() {
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p)
            if (p < KT) stage(p, p);
    };
    prologue();

    // per-lane constants of the fragment reads
    const int frow = lane & 31;                          // row inside a 32-row fragment
    const int fswz = (frow >> SWSH) & (CPR - 1);         // swizzle key (fragment bases are multiples of 32)
    const int fhi = lane >> 5;

    // Epilogues that READ (fp32 residual; saved activations of the activation-backward forms) on the ping-pong
    // schedules: the values of the tile are requested at tile START and ride in registers through the main loop (one
    // workgroup per CU: registers are free), so the epilogue is arithmetic + stores only.  They are the youngest loads
    // when the loop is entered, hence the + epi_loads in its first wait.
    constexpr bool RES_PREFETCH = (EPI == EPI_RES) && PP && FW * FX <= 4;      // 64 registers at most
    constexpr bool ACT_PREFETCH = (EPI == EPI_SWIGLU_BWD || EPI == EPI_GELU_BWD) && PP && FW * FX <= 4;
    constexpr int NSRC = (EPI == EPI_SWIGLU_BWD) ? 2 : 1;                       // (g | u) or the pre-activation
    constexpr int EPI_LOADS = RES_PREFETCH ? FW * FX * 4 : ACT_PREFETCH ? FW * FX * 2 * NSRC : 0;
    float4 rpre[RES_PREFETCH ? FW : 1][RES_PREFETCH ? FX : 1][4];
    uint4 apre[ACT_PREFETCH ? FW : 1][ACT_PREFETCH ? FX : 1][2][NSRC];
    // 16-byte accesses need the 8-element granularity (else the epilogue falls back to its own narrow loads)
    const bool act_wide = ACT_PREFETCH && ((N | a.Hp | a.ldo | a.ldr) & 7) == 0 && (((uintptr_t)a.out | (uintptr_t)a.res) & 15) == 0;
    const bool epi_prefetch = RES_PREFETCH || act_wide;
    auto load_res = [&]() {
        if constexpr (RES_PREFETCH) {
#pragma unroll
            for (int j = 0; j < FX; ++j) {
                int m = m0 + wx * (TX / WX) + j * 32 + frow;
                m = m < a.M ? m : a.M - 1;
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        int n = n0 + ww * (TW / WW) + i * 32 + 8 * g + 4 * fhi;
                        n = n + 4 <= N ? n : (N >= 4 ? N - 4 : 0);           // stay inside the row; the value is unused there
                        rpre[i][j][g] = *(const float4*)(a.res + (size_t)m * a.ldr + n);
                    }
            }
        }
        if constexpr (ACT_PREFETCH) {
            if (act_wide) {
#pragma unroll
                for (int j = 0; j < FX; ++j) {
                    int m = m0 + wx * (TX / WX) + j * 32 + frow;
                    m = m < a.M ? m : a.M - 1;
                    const bf16_t* srow = (const bf16_t*)a.res + (size_t)m * a.ldr;
#pragma unroll
                    for (int i = 0; i < FW; ++i)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            int c = n0 + ww * (TW / WW) + i * 32 + 16 * gp + 8 * fhi;
                            c = c + 8 <= N ? c : N - 8;                      // N % 8 == 0 here
#pragma unroll
                            for (int q = 0; q < NSRC; ++q) apre[i][j][gp][q] = *(const uint4*)(srow + q * a.Hp + c);
                        }
                }
            }
        }
    };
    load_res();

    int t_lin = blockIdx.x;                              // PERSIST: position in the tile order
    bool first_tile = true;
    for (;;) {                                           // one pass unless PERSIST
#pragma unroll
    for (int i = 0; i < FW; ++i)
#pragma unroll
        for (int j = 0; j < FX; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (PP) {
        static_assert(WW == 2 && NWAVES == 8 && STAGES == 3, "ping-pong schedule: 2 wave rows of 4 waves, 3-deep ring");
        constexpr int KS = KB / 16;                          // MFMA k-steps per K-tile
        constexpr int PIECES = PW + PX;
        const bool lead = ww == 0;
        // Only loads are ordered by vmcnt, and the youngest LOADS + EPI_LOADS of them are stage 1 and the epilogue's values:
        // at most that many operations outstanding => stage 0 has landed (epilogue stores of the previous tile still in
        // flight only make the condition stricter).
        if (epi_prefetch) { if (KT > 1) wait_vmcnt<LOADS + EPI_LOADS>(); else wait_vmcnt<EPI_LOADS>(); }
        else { if (KT > 1) wait_vmcnt<LOADS>(); else wait_vmcnt<0>(); }
        block_barrier();                                     // tile 0 is in LDS for everyone
        if (!lead) block_barrier();                          // the trailing row runs one barrier behind
        int buf = 0;
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = kt + 2 < KT;
            const int nbuf = buf >= 1 ? buf - 1 : 2;         // (buf + 2) % 3: the buffer tile kt-1 lived in
            // ---- MEM half: every fragment of tile kt ------------------------------------------------
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
            if (!lead) {                                     // odd slot: the trailing row stages from its MEM half
                if (more) { stage(kt + 2, nbuf); wait_vmcnt<LOADS>(); } else wait_vmcnt<0>();
            }
            wait_lgkmcnt<0>();
            block_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA half (the leading row also stages here: same odd slot) -------------------------
            if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int j = 0; j < FX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
                if (lead && more) {                          // DMA pieces spread between the MFMA groups
#pragma unroll
                    for (int q = kk * PIECES / KS; q < (kk + 1) * PIECES / KS; ++q) {
                        if (q < PW) stage_w(kt + 2, nbuf, q); else stage_x(kt + 2, nbuf, q - PW);
                    }
                }
            }
            if (a.prio) __builtin_amdgcn_s_setprio(0);
            if (lead) { if (more) wait_vmcnt<LOADS>(); else wait_vmcnt<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            if (lead || kt + 1 < KT) block_barrier();        // barrier counts: lead 1+2*KT, trailing 2+2*KT-1
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
    } else {
        int buf = 0;
        for (int kt = 0; kt < KT; ++kt) {
            // tile kt has landed once at most min(STAGES-2, KT-1-kt) younger stages are still in flight
            const int younger = KT - 1 - kt;
            if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LOADS>();
            else if (STAGES >= 3 && younger >= 1) wait_vmcnt<LOADS>();
            else wait_vmcnt<0>();
            block_barrier();      // everyone's pieces of tile kt are in LDS; everyone is done reading tile kt-1
            if (kt + STAGES - 1 < KT) stage(kt + STAGES - 1, (buf + STAGES - 1) % STAGES);
            const char* wt = smem + buf * STAGE + (ww * (TW / WW) + frow) * RB;
            const char* xt = smem + buf * STAGE + TW * RB + (wx * (TX / WX) + frow) * RB;
            // fragments of step kk+1 are read from LDS while the MFMAs of step kk execute
            bf16x8_t wf[2][FW], xf[2][FX];
            auto load_frags = [&](int kk, int par) {
                const int off = ((kk * 2 + fhi) ^ fswz) * 16;
#pragma unroll
                for (int i = 0; i < FW; ++i) wf[par][i] = *(const bf16x8_t*)(wt + i * 32 * RB + off);
#pragma unroll
                for (int j = 0; j < FX; ++j) xf[par][j] = *(const bf16x8_t*)(xt + j * 32 * RB + off);
            };
            load_frags(0, 0);
#pragma unroll
            for (int kk = 0; kk < KB / 16; ++kk) {
                if (kk + 1 < KB / 16) load_frags(kk + 1, (kk + 1) & 1);
                if (a.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int j = 0; j < FX; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk & 1][i], xf[kk & 1][j], acc[i][j], 0, 0, 0);
                if (a.prio) __builtin_amdgcn_s_setprio(0);
            }
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
    }

    // ---- PERSIST: next tile's first stages go out before this tile's stores ---------------------
    const int en0 = n0, em0 = m0;                        // the epilogue below belongs to the tile just finished
    bool has_next = false;
    if constexpr (PERSIST) {
        static_assert(PP && !GROUPED, "persistent tiles are implemented for the dense ping-pong schedule");
        const int total = a.n_tiles_w * a.n_tiles_x;
        has_next = t_lin + (int)gridDim.x < total;
        if (has_next) {
            t_lin += gridDim.x;
            const int tile = xcd_remap(t_lin, total);
            n0 = (tile % a.n_tiles_w) * NPT; m0 = (tile / a.n_tiles_w) * TX;
            set_sources();
            prologue();
        }
    }
    // ---- epilogue: lane holds, per fragment, 4 groups of 4 consecutive features of one row ---
#pragma unroll
    for (int j = 0; j < FX; ++j) {
        const int m = em0 + wx * (TX / WX) + j * 32 + frow;
        if (m >= a.M) continue;
        if constexpr (EPI == EPI_SWIGLU) {
            bf16_t* gu = a.out2 ? (bf16_t*)a.out2 + (size_t)m * a.ldo2 : nullptr;      // (g | u) only when a backward will need it
            bf16_t* ao = (bf16_t*)a.out + (size_t)m * a.ldo;
            const bool wide = ((N | a.Hp | a.ldo | (gu ? a.ldo2 : 0)) & 7) == 0 && (((uintptr_t)a.out | (uintptr_t)a.out2) & 15) == 0;
            // all values of this row first, then the stores output by output: the 16-byte pieces of one 128-byte line
            // (one line per row and output for a 64-hidden wave tile) leave back to back and merge in L2
            uint2 pg_[FW / 2][4], pu_[FW / 2][4], pa_[FW / 2][4];
#pragma unroll
            for (int ip = 0; ip < FW / 2; ++ip) {
                // fragment pair (2ip, 2ip+1) = (g, u) of hidden units n0 + (tile row / 64) * 32 + ...
                const int hb = en0 + (ww * (TW / WW) / 64 + ip) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int h = hb + 8 * g + 4 * fhi;
                    float gv[4], uv[4], av[4], b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b1[e] = bfround(a.bias[min(h + e, N - 1)]);
                    }
                    if (a.bias2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b2[e] = bfround(a.bias2[min(h + e, N - 1)]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool live = h + e < N;          // hidden sizes need not be multiples of 4 (2730)
                        gv[e] = live ? bfround(acc[2 * ip][j][4 * g + e] + b1[e]) : 0.f;
                        uv[e] = live ? bfround(acc[2 * ip + 1][j][4 * g + e] + b2[e]) : 0.f;
                        av[e] = bfround(silu_f(gv[e])) * uv[e];
                    }
                    pg_[ip][g] = make_uint2(pack2bf(gv[0], gv[1]), pack2bf(gv[2], gv[3]));
                    pu_[ip][g] = make_uint2(pack2bf(uv[0], uv[1]), pack2bf(uv[2], uv[3]));
                    pa_[ip][g] = make_uint2(pack2bf(av[0], av[1]), pack2bf(av[2], av[3]));
                }
            }
#pragma unroll
            for (int which = 0; which < 3; ++which) {
                if (which < 2 && !gu) continue;
                bf16_t* dst = which == 0 ? gu : which == 1 ? gu + a.Hp : ao;
#pragma unroll
                for (int ip = 0; ip < FW / 2; ++ip) {
                    const int hb = en0 + (ww * (TW / WW) / 64 + ip) * 32;
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const uint2 lo = which == 0 ? pg_[ip][g] : which == 1 ? pu_[ip][g] : pa_[ip][g];
                        const uint2 hi = which == 0 ? pg_[ip][g + 1] : which == 1 ? pu_[ip][g + 1] : pa_[ip][g + 1];
                        store_bf16_groups(dst, hb + 8 * g, lo, hi, fhi, N, wide);
                    }
                }
            }
        } else if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_TANH) {
            bf16_t* orow = (bf16_t*)a.out + (size_t)m * a.ldo;
            bf16_t* prow = (EPI == EPI_GELU && a.out2) ? (bf16_t*)a.out2 + (size_t)m * a.ldo2 : nullptr;
            const bool wide = ((N | a.ldo | (prow ? a.ldo2 : 0)) & 7) == 0 && (((uintptr_t)a.out | (uintptr_t)(prow ? a.out2 : nullptr)) & 15) == 0;
#pragma unroll
            for (int i = 0; i < FW; ++i) {
                const int nb = en0 + ww * (TW / WW) + i * 32;
                uint2 po[4], pp[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + 8 * g + 4 * fhi;
                    float v[4], b[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias && n < N) {
                        const float4 t = *(const float4*)(a.bias + n);     // n % 4 == 0, bias 16-B aligned
                        b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bfround(b[e]);
                    if constexpr (EPI == EPI_GELU) {
                        pp[g] = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_f(bfround(v[e]));
                    } else if constexpr (EPI == EPI_TANH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = tanhf(bfround(v[e]));
                    }
                    po[g] = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                }
                // one output after the other (the pieces of a 128-byte line leave back to back; see the SwiGLU epilogue)
#pragma unroll
                for (int g = 0; g < 4; g += 2) store_bf16_groups(orow, nb + 8 * g, po[g], po[g + 1], fhi, N, wide);
                if constexpr (EPI == EPI_GELU) {
                    if (prow) {
#pragma unroll
                        for (int g = 0; g < 4; g += 2) store_bf16_groups(prow, nb + 8 * g, pp[g], pp[g + 1], fhi, N, wide);
                    }
                }
            }
        } else if constexpr (EPI == EPI_SWIGLU_BWD || EPI == EPI_GELU_BWD) {
            // acc = d(act); res = saved (g | u) or pre-activation (bf16); out = (dg | du) or d(pre) (bf16)
            // (GatedMlp fm_utils.py:142-144 / Mlp :121-126).  16-byte loads and stores, see load/store_bf16_groups.
            const bf16_t* srow = (const bf16_t*)a.res + (size_t)m * a.ldr;
            bf16_t* orow = (bf16_t*)a.out + (size_t)m * a.ldo;
            const bool wide = ((N | a.Hp | a.ldo | a.ldr) & 7) == 0 && (((uintptr_t)a.out | (uintptr_t)a.res) & 15) == 0;
            uint2 o1[FW][4], o2[FW][4];
#pragma unroll
            for (int i = 0; i < FW; ++i) {
                const int nb = en0 + ww * (TW / WW) + i * 32;
                uint2 sg_[4], su_[4];
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    if (ACT_PREFETCH && act_wide) {              // requested at tile start
                        split_bf16_groups(apre[ACT_PREFETCH ? i : 0][ACT_PREFETCH ? j : 0][g / 2][0], sg_[g], sg_[g + 1]);
                        if constexpr (EPI == EPI_SWIGLU_BWD) split_bf16_groups(apre[ACT_PREFETCH ? i : 0][ACT_PREFETCH ? j : 0][g / 2][NSRC - 1], su_[g], su_[g + 1]);
                    } else {
                        load_bf16_groups(srow, nb + 8 * g, fhi, N, wide, sg_[g], sg_[g + 1]);
                        if constexpr (EPI == EPI_SWIGLU_BWD) load_bf16_groups(srow + a.Hp, nb + 8 * g, fhi, N, wide, su_[g], su_[g + 1]);
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + 8 * g + 4 * fhi;
                    float xv[4], r1[4], r2[4];
                    unpack_bf4(sg_[g], xv);
                    if constexpr (EPI == EPI_SWIGLU_BWD) {
                        float uv[4];
                        unpack_bf4(su_[g], uv);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = bfround(acc[i][j][4 * g + e]);
                            const float sg = 1.0f / (1.0f + __expf(-xv[e]));
                            const float sl = bfround(xv[e] * sg);
                            const float ds = bfround(d * uv[e]);
                            const bool live = n + e < N;
                            r2[e] = live ? d * sl : 0.f;
                            r1[e] = live ? ds * (sg * (1.0f + xv[e] * (1.0f - sg))) : 0.f;
                        }
                        o2[i][g] = make_uint2(pack2bf(r2[0], r2[1]), pack2bf(r2[2], r2[3]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float cdf = 0.5f * (1.0f + erf_fast(xv[e] * 0.70710678118654752f));
                            const float pdf = 0.3989422804014327f * __expf(-0.5f * xv[e] * xv[e]);
                            r1[e] = (n + e < N) ? bfround(acc[i][j][4 * g + e]) * (cdf + xv[e] * pdf) : 0.f;
                        }
                    }
                    o1[i][g] = make_uint2(pack2bf(r1[0], r1[1]), pack2bf(r1[2], r1[3]));
                }
            }
            // one output after the other: the pieces of a row's 128-byte line leave back to back (see the SwiGLU epilogue)
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
                for (int g = 0; g < 4; g += 2) store_bf16_groups(orow, en0 + ww * (TW / WW) + i * 32 + 8 * g, o1[i][g], o1[i][g + 1], fhi, N, wide);
            if constexpr (EPI == EPI_SWIGLU_BWD) {
#pragma unroll
                for (int i = 0; i < FW; ++i)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) store_bf16_groups(orow + a.Hp, en0 + ww * (TW / WW) + i * 32 + 8 * g, o2[i][g], o2[i][g + 1], fhi, N, wide);
            }
        } else {
#pragma unroll
            for (int i = 0; i < FW; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = en0 + ww * (TW / WW) + i * 32 + 8 * g + 4 * fhi;
                    if (n >= N) continue;
                    float v[4], b[4] = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) {
                        const float4 t = *(const float4*)(a.bias + n);     // n % 4 == 0, bias 16-B aligned
                        b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + ((EPI == EPI_F32) ? b[e] : bfround(b[e]));
                    if constexpr (EPI == EPI_RES) {
                        float4 r;
                        if constexpr (RES_PREFETCH) r = rpre[i][j][g];
                        else r = *(const float4*)(a.res + (size_t)m * a.ldr + n);
                        float4 o = make_float4(r.x + bfround(v[0]), r.y + bfround(v[1]), r.z + bfround(v[2]), r.w + bfround(v[3]));
                        *(float4*)((float*)a.out + (size_t)m * a.ldo + n) = o;
                    } else {  // EPI_F32 (optionally + res, no rounding)
                        if (a.res) {
                            const float4 r = *(const float4*)(a.res + (size_t)m * a.ldr + n);
                            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                        }
                        *(float4*)((float*)a.out + (size_t)m * a.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
        }
    }
    if (!has_next) break;
    first_tile = false;
    load_res();                                          // n0 / m0 already name the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// TN kernel:  out[n][k] += sum_r A[r][n] * B[r][k]      (fp32 atomic accumulate, split over r)
// ------------------------------------------------------------------------------------------------
struct TNArgs {
    const bf16_t* A; const bf16_t* B; float* out;
    int R, N, K, lda, ldb, ldo, splits;
    int a_cols, b_cols;                       // readable columns of A / B (clamp for the tile loads)
    const fm_gemm_group* groups; const int* seg_start; const int* seg_count;   // grouped (per-modality rows)
    int n_tiles_a, n_tiles_b;
};

// Reduction rows past the live row count contribute nothing: their LDS-DMA pieces are fetched from this 16-byte zero block
// instead of the operand (the caller's buffers need no zeroed padding rows, and a workspace reused with fewer live rows
// cannot leak stale rows into a weight gradient).

// element (row, col) of a row-major [64][cols] bf16 LDS tile with RB bytes per row; 16-byte chunks are
// XOR-swizzled by (row & 3) << 2 so that the 4 rows a transpose read touches fall on different banks
template <int RB>
__device__ __forceinline__ int tn_off(int row, int col) { return row * RB + ((((col >> 3) ^ ((row & 3) << 2))) << 4) + (col & 7) * 2; }
template <int RB>
__device__ __forceinline__ const char* tn_addr(const char* tile, int row, int col) { return tile + tn_off<RB>(row, col); }

// one half (4 rows x 16 columns per 16-lane group) of a transpose-read fragment at a compile-time byte offset
template <int OFF>
__device__ __forceinline__ s16x4_t tr_read(uint32_t lds_addr) {
    s16x4_t h;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h) : "v"(lds_addr), "n"(OFF));
    return h;
}

// PP = the ping-pong schedule of gemm_nt_kernel (wave rows one barrier apart; see there) on the TN operands: the
// fragment addresses of one lane differ between k-steps only by constants (the swizzle key (row & 3) does not
// change), so a K-tile's 8 * KB/16 transpose reads use immediate offsets on four per-lane base addresses.
// MASKED: reduction rows >= the live row count are fetched from a zero block (grouped segments, R % KB != 0); the unmasked
// instantiation carries no per-lane select and no extra branch in its DMA path.
template <bool TR, bool GROUPED, int TA, int TB, int WA, int WB, int KB, int STAGES, bool PP = false, bool MASKED = true>
__global__ __launch_bounds__(WA * WB * 64) void gemm_tn_kernel(TNArgs a) {
    constexpr int NWAVES = WA * WB;
    constexpr int RBA = TA * 2, RBB = TB * 2;                         // bytes per LDS tile row
    constexpr int PA = KB * RBA / 1024, PB = KB * RBB / 1024;         // 1-KiB DMA pieces per tile
    constexpr int LOADS = (PA + PB) / NWAVES;                         // LDS-DMA instructions per wave per stage
    constexpr int STAGE = KB * (RBA + RBB);
    static_assert(TA / WA == 64 && TB / WB == 64 && PA % NWAVES == 0 && PB % NWAVES == 0, "wave tile is 64x64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave / WB, wb = wave % WB;

    // Workgroups that reduce the same rows (same split) read the same A / B row panels, each a different
    // column slice pair.  Work items are ordered split-major and cut into 8 contiguous runs, one per XCD
    // (dispatch puts workgroup b on XCD b % 8): an XCD sees one or two splits, so a panel is fetched from HBM
    // once (at most twice) and shared through that XCD's L2, for any split count.
    //
    // Grouped (per-modality heads): groups have very different widths, so contiguous runs would leave most XCDs
    // idle on the narrow ones; there the order is cut into chunks of 4 A-tiles x all B-tiles (they share their
    // panels) dealt round-robin to the XCDs.
    const int nwg = a.n_tiles_a * a.n_tiles_b;
    const int total = nwg * a.splits;
    int item;
    if constexpr (GROUPED) {
        const int chunk_sz = 4 * a.n_tiles_b, j = blockIdx.x / 8;
        item = ((j / chunk_sz) * 8 + blockIdx.x % 8) * chunk_sz + j % chunk_sz;
        if (item >= total) return;
    } else {
        const int per_xcd = (total + 7) / 8;
        item = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
        if (blockIdx.x / 8 >= per_xcd || item >= total) return;
    }
    const int split = item / nwg, tile = item % nwg;
    const int ta = tile / a.n_tiles_b, tb = tile % a.n_tiles_b;
    int N = a.N, r_begin = 0, r_end = (a.R + KB - 1) / KB * KB, r_lim = a.R;
    float* out = a.out;
    if constexpr (GROUPED) {
        const int g = blockIdx.z;
        N = a.groups[g].N; out = (float*)a.groups[g].out;
        r_begin = a.seg_start[g];
        r_lim = r_begin + a.seg_count[g];
        r_end = r_begin + ((a.seg_count[g] + 63) / 64) * 64;
    }
    const bf16_t* zsrc = (const bf16_t*)g_zero16;
    const int n0 = ta * TA, k0 = tb * TB;
    if (n0 >= N || k0 >= a.K) return;
    const int nt = (r_end - r_begin) / KB;                      // reduction tiles in total
    const int per = (nt + a.splits - 1) / a.splits;
    const int t_begin = split * per, t_end = min(nt, t_begin + per);
    if (t_begin >= t_end) return;

    // piece q of K-tile t (q < PA/NWAVES: A rows, else B rows)
    auto stage_piece = [&](int t, int buf, int q) {
        char* base = smem + buf * STAGE;
        const int r0 = r_begin + t * KB;
        if (q < PA / NWAVES) {
            constexpr int LPR = RBA / 16, RPP = 1024 / RBA;        // lanes per row, rows per piece
            const int piece = q * NWAVES + wave;
            const int row = piece * RPP + lane / LPR;
            const int lc = (lane % LPR) ^ ((row & 3) << 2);
            int ca = n0 + lc * 8; ca = ca <= a.a_cols - 8 ? ca : a.a_cols - 8;
            const bf16_t* src = a.A + (size_t)(r0 + row) * a.lda + ca;
            if constexpr (MASKED) src = r0 + row < r_lim ? src : zsrc;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + piece * 1024), 16, 0, 0);
        } else {
            constexpr int LPR = RBB / 16, RPP = 1024 / RBB;
            const int piece = (q - PA / NWAVES) * NWAVES + wave;
            const int row = piece * RPP + lane / LPR;
            const int lc = (lane % LPR) ^ ((row & 3) << 2);
            int cb = k0 + lc * 8; cb = cb <= a.b_cols - 8 ? cb : a.b_cols - 8;
            const bf16_t* src = a.B + (size_t)(r0 + row) * a.ldb + cb;
            if constexpr (MASKED) src = r0 + row < r_lim ? src : zsrc;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + KB * RBA + piece * 1024), 16, 0, 0);
        }
    };
    constexpr int NP = PA / NWAVES + PB / NWAVES;
    auto stage = [&](int t, int buf) {
#pragma unroll
        for (int q = 0; q < NP; ++q) stage_piece(t, buf, q);
    };
    // the same pieces, a share of them per call (ping-pong: spread between the MFMA groups of the leading wave row)
    auto stage_part = [&](int t, int buf, int part, int nparts) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (q >= part * NP / nparts && q < (part + 1) * NP / nparts) stage_piece(t, buf, q);
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int NT_ = t_end - t_begin;
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (p < NT_) stage(t_begin + p, p);
    const int fhi = lane >> 5;
    if constexpr (PP) {
        static_assert(TR && WA == 2 && NWAVES == 8 && STAGES == 3, "ping-pong schedule: transpose reads, 2 wave rows of 4 waves, 3-deep ring");
        constexpr int KS = KB / 16;
        const bool lead = wa == 0;
        // per-lane base addresses (LDS byte offsets inside a stage) of the 2 + 2 fragments at k-step 0
        const int li = lane & 15;
        const int frow0 = fhi * 8 + (li >> 2), fcol = ((lane >> 4) & 1) * 16 + (li & 3) * 4;
        const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
        uint32_t baseA[2], baseB[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            baseA[i] = (uint32_t)tn_off<RBA>(frow0, wa * 64 + i * 32 + fcol);
            baseB[i] = (uint32_t)tn_off<RBB>(frow0, wb * 64 + i * 32 + fcol) + KB * RBA;
        }
        if (NT_ > 1) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
        block_barrier();
        if (!lead) block_barrier();
        int buf = 0;
        for (int it = 0; it < NT_; ++it) {
            const bool more = it + 2 < NT_;
            const int nbuf = buf >= 1 ? buf - 1 : 2;
            const uint32_t st = smem_lds + buf * STAGE;
            union Frag { bf16x8_t v; s16x4_t h[2]; };
            Frag af[KS][2], bfr[KS][2];
            auto read_k = [&](auto kk_c) {
                constexpr int kk = decltype(kk_c)::value;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[kk][i].h[0] = tr_read<kk * 16 * RBA>(st + baseA[i]);
                    af[kk][i].h[1] = tr_read<kk * 16 * RBA + 4 * RBA>(st + baseA[i]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bfr[kk][j].h[0] = tr_read<kk * 16 * RBB>(st + baseB[j]);
                    bfr[kk][j].h[1] = tr_read<kk * 16 * RBB + 4 * RBB>(st + baseB[j]);
                }
            };
            read_k(std::integral_constant<int, 0>{});
            if constexpr (KS > 1) read_k(std::integral_constant<int, 1>{});
            if constexpr (KS > 2) read_k(std::integral_constant<int, 2>{});
            if constexpr (KS > 3) read_k(std::integral_constant<int, 3>{});
            if (!lead) {
                if (more) { stage(t_begin + it + 2, nbuf); wait_vmcnt<LOADS>(); } else wait_vmcnt<0>();
            }
            wait_lgkmcnt<0>();
            block_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i].v, bfr[kk][j].v, acc[i][j], 0, 0, 0);
                if (lead && more) stage_part(t_begin + it + 2, nbuf, kk, KS);
            }
            __builtin_amdgcn_s_setprio(0);
            if (lead) { if (more) wait_vmcnt<LOADS>(); else wait_vmcnt<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            if (lead || it + 1 < NT_) block_barrier();
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
    } else {
    int buf = 0;
    for (int it = 0; it < NT_; ++it) {
        const int younger = NT_ - 1 - it;
        if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LOADS>();
        else if (STAGES >= 3 && younger >= 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        block_barrier();
        if (it + STAGES - 1 < NT_) stage(t_begin + it + STAGES - 1, (buf + STAGES - 1) % STAGES);
        const char* at = smem + buf * STAGE;
        const char* bt = at + KB * RBA;
        if constexpr (TR) {
            // fragments of step kk+1 are in flight while the MFMAs of step kk run (8 transpose reads per step)
            bf16x8_t af[2][2], bfr[2][2];
            auto load = [&](int kk, int par) {
                const int rA = kk * 16 + fhi * 8, rB = rA + 4;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[par][i] = lds_col_frag_tr_async([&](int r, int c) { return tn_addr<RBA>(at, r, c); }, rA, rB, wa * 64 + i * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bfr[par][j] = lds_col_frag_tr_async([&](int r, int c) { return tn_addr<RBB>(bt, r, c); }, rA, rB, wb * 64 + j * 32);
            };
            load(0, 0);
#pragma unroll
            for (int kk = 0; kk < KB / 16; ++kk) {
                if (kk + 1 < KB / 16) { load(kk + 1, (kk + 1) & 1); wait_lgkmcnt<8>(); }
                else wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][i], bfr[kk & 1][j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KB / 16; ++kk) {
                const int rA = kk * 16 + fhi * 8, rB = rA + 4;
                bf16x8_t af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    af[i] = lds_col_frag<false>([&](int r, int c) { return tn_addr<RBA>(at, r, c); }, rA, rB, wa * 64 + i * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bfr[j] = lds_col_frag<false>([&](int r, int c) { return tn_addr<RBB>(bt, r, c); }, rA, rB, wb * 64 + j * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
    }
    // accumulator: rows <-> n (A columns), cols <-> k (B columns); lane = k, regs = n
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + wb * 64 + j * 32 + (lane & 31);
        if (k >= a.K) continue;   // (TB-wide tile: wb in [0, WB))
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wa * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                if (n < N) unsafeAtomicAdd(out + (size_t)n * a.ldo + k, acc[i][j][r]);
            }
    }
}


// ------------------------------------------------------------------------------------------------
// TN job list: every weight-gradient GEMM of one transformer layer in ONE launch
// ------------------------------------------------------------------------------------------------
// The trunk's backward produces 5 (encoder) / 8 (decoder layer) weight gradients out[n][k] += sum_r dY[r][n] X[r][k] with a few
// dozen 128 x 256 output tiles each and a 32k-row reduction.  Launched one by one, each must be split ~14 ways over the rows to
// occupy 256 CUs: 14 x the output in fp32 atomics per GEMM and a ~36-step main loop per workgroup.  Here the jobs of a layer form
// one tile list (216 tiles for a 4M-B encoder layer, 288 for a decoder layer) on a grid of one workgroup per CU:
//   * while at least gridDim tiles remain, workgroup w reduces tile f * gridDim + w over ALL its rows (no split at all);
//   * the last rem < gridDim tiles are cut once: workgroup i < rem ("main") takes the first q = KT * rem / gridDim k-tiles of
//     tile i, the other gridDim - rem workgroups ("tail") share the remaining k-tiles evenly, each walking a contiguous run of them
//     (a handful of tile tails); every workgroup ends up with ~ total k-tiles / gridDim.
// Main workgroups of neighbouring tiles (same XCD: logical index = (blockIdx % 8) * gridDim / 8 + blockIdx / 8) run in lock step
// over the same rows, so the operand panels they share come from that XCD's L2 once - a stream-K cut at arbitrary offsets would
// have every workgroup at a different row and lose that.  Atomic traffic: (tiles + gridDim) x 128 KB per LAYER.
// The main loop is gemm_tn_kernel's ping-pong schedule (K-step 64, 3-stage LDS-DMA ring, transpose reads), restarted per segment.
// TA = 128, KB = 64: 64 x 64 wave tiles, 144 KB of LDS;  TA = 256, KB = 32: 128 x 64 wave tiles (8 accumulators per wave), 96 KB - per MFMA
// 2/3 of the LDS-DMA pieces and 3/8 of the transpose reads of the small tile.
// LS ("lock step", TA = 256, KB = 64, two stages = 128 KB): the structure of gemm_nt3.hip on the TN operands - ONE barrier per K-tile,
// placed before its last k-step (whose fragments are already in registers), every transpose read of the next k-step and every LDS-DMA
// piece of the next K-tile issued BETWEEN the MFMAs (a burst of reads behind a barrier costs ~200 cycles, tools/ubench.hip), rows past
// the live count read as zero through the buffer descriptor's bounds check instead of a per-lane select.
template <bool MASKED, int TA = 128, int KB = 64, bool LS = false>
__global__ __launch_bounds__(512) void gemm_tn_multi_kernel(TNMultiArgs a) {
    constexpr int TB = 256, WB = 4, STAGES = LS ? 2 : 3, NWAVES = 8;
    constexpr int FWA = TA / 2 / 32;                  // 32-column A fragments per wave
    constexpr int RBA = TA * 2, RBB = TB * 2;
    constexpr int PA = KB * RBA / 1024, PB = KB * RBB / 1024;
    constexpr int LOADS = (PA + PB) / NWAVES;
    constexpr int STAGE = KB * (RBA + RBB);
    constexpr int KS = KB / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave / WB, wb = wave % WB;
    const bool lead = wa == 0;
    const int fhi = lane >> 5;
    const bf16_t* zsrc = (const bf16_t*)g_zero16;
    const int G = gridDim.x;
    const int w = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;

    const int li = lane & 15;
    const int frow0 = fhi * 8 + (li >> 2), fcol = ((lane >> 4) & 1) * 16 + (li & 3) * 4;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    uint32_t baseA[FWA], baseB[2];
#pragma unroll
    for (int i = 0; i < FWA; ++i) baseA[i] = (uint32_t)tn_off<RBA>(frow0, wa * (TA / 2) + i * 32 + fcol);
#pragma unroll
    for (int i = 0; i < 2; ++i) baseB[i] = (uint32_t)tn_off<RBB>(frow0, wb * 64 + i * 32 + fcol) + KB * RBA;

    // one segment: k-tiles [t_begin, t_end) of global tile `tile`, accumulated into its job's output
    auto run = [&](int tile, int t_begin, int t_end) {
        if (t_begin >= t_end) return;
        int j = 0;
        while (j + 1 < a.n_jobs && tile >= a.job[j + 1].tile_start) ++j;
        const TNJob& jb = a.job[j];
        const int local = tile - jb.tile_start;
        const int n0 = (local / jb.n_tiles_b) * TA, k0 = (local % jb.n_tiles_b) * TB;
        const bf16_t* A = jb.A; const bf16_t* B = jb.B;
        const int lda = jb.lda, ldb = jb.ldb, a_cols = jb.a_cols, b_cols = jb.b_cols, r_lim = jb.R;

#if defined(__HIP_DEVICE_COMPILE__)      // (buffer-descriptor builtins: the host pass only needs the kernel stub)
        if constexpr (LS) {
            static_assert(!LS || (TA == 256 && KB == 64), "lock-step form: 256 x 256 tiles, K-step 64");
            // ---- DMA: 8 pieces of 1 KiB (2 rows x 512 B) per wave and stage; source rows >= R fall outside the descriptor: zeros --------
            constexpr int NPA = PA / NWAVES, NPB = PB / NWAVES, NP = NPA + NPB;          // 4 + 4
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)min((size_t)r_lim * lda * 2, (size_t)0x7fffffff), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)min((size_t)r_lim * ldb * 2, (size_t)0x7fffffff), 0x00020000);
            uint32_t offp[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const bool isa = q < NPA;
                const int piece = (isa ? q : q - NPA) * NWAVES + wave;
                const int row = piece * 2 + lane / 32;
                const int lc = (lane % 32) ^ ((row & 3) << 2);
                int c = (isa ? n0 : k0) + lc * 8;
                const int cols = isa ? a_cols : b_cols;
                c = c <= cols - 8 ? c : cols - 8;
                offp[q] = (uint32_t)row * (uint32_t)(isa ? lda : ldb) * 2u + (uint32_t)c * 2u;
            }
            auto dma = [&](int t, int buf, int q) __attribute__((always_inline)) {
                char* base = smem + buf * STAGE;
                if (q < NPA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, LDS_PTR(base + (q * NWAVES + wave) * 1024), 16, offp[q], t * (KB * lda * 2), 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, LDS_PTR(base + KB * RBA + ((q - NPA) * NWAVES + wave) * 1024), 16, offp[q], t * (KB * ldb * 2), 0, 0);
            };
            f32x16_t acc[FWA][2];
#pragma unroll
            for (int i = 0; i < FWA; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
            const int NT_ = t_end - t_begin;
#pragma unroll
            for (int q = 0; q < NP; ++q) dma(t_begin, 0, q);
            if (NT_ > 1) {
#pragma unroll
                for (int q = 0; q < NP; ++q) dma(t_begin + 1, 1, q);
                wait_vmcnt<NP>();
            } else wait_vmcnt<0>();
            block_barrier();
            union Frag { bf16x8_t v; s16x4_t h[2]; };
            Frag af[2][FWA], bfr[2][2];
            // one half (4 rows) of fragment f (f < FWA: A, else B) of k-step kk in stage `st`, into register set PAR
            auto rd = [&](uint32_t st, auto kk_c, auto f_c, auto h_c, auto par_c) __attribute__((always_inline)) {
                constexpr int kk = decltype(kk_c)::value, f = decltype(f_c)::value, h = decltype(h_c)::value, PAR = decltype(par_c)::value;
                if constexpr (f < FWA) af[PAR][f].h[h] = tr_read<kk * 16 * RBA + h * 4 * RBA>(st + baseA[f]);
                else bfr[PAR][f - FWA].h[h] = tr_read<kk * 16 * RBB + h * 4 * RBB>(st + baseB[f - FWA]);
            };
            constexpr int NF = FWA + 2, NMF = FWA * 2;                               // 6 fragments (12 reads), 8 MFMAs per k-step
            static_assert(!LS || NMF == 8, "k-step of the lock-step form: 8 MFMAs");
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
            // k-step with the fragments of register set PAR: MFMA m, then both halves of fragment m (m < 6) of k-step KN of stage `st_next`
            // into the other set, and (DN > 0) DMA piece D0 + m of K-tile t_dma behind MFMA m < DN
            auto k_step = [&](uint32_t st_next, auto kn_c, auto par_c, int t_dma, int buf_dma, auto d0_c, auto dn_c) __attribute__((always_inline)) {
                constexpr int PAR = decltype(par_c)::value, d0 = decltype(d0_c)::value, dn = decltype(dn_c)::value;
                wait_lgkmcnt<0>();
                __builtin_amdgcn_sched_barrier(0);
                auto one = [&](auto m_c) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_c)::value;
                    constexpr int i = m / 2, jj = m % 2;
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PAR][i].v, bfr[PAR][jj].v, acc[i][jj], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m < NF) {
                        rd(st_next, kn_c, std::integral_constant<int, (m < NF ? m : 0)>{}, I0{}, std::integral_constant<int, PAR ^ 1>{});
                        rd(st_next, kn_c, std::integral_constant<int, (m < NF ? m : 0)>{}, I1{}, std::integral_constant<int, PAR ^ 1>{});
                    }
                    if constexpr (m < dn) { if (t_dma >= 0) dma(t_dma, buf_dma, d0 + m); }
                    __builtin_amdgcn_sched_barrier(0);
                };
                one(I0{}); one(I1{}); one(I2{}); one(I3{}); one(I4{}); one(I5{}); one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
            };
            // fragments of (stage 0, k-step 0)
            {
                const uint32_t st0 = smem_lds;
                rd(st0, I0{}, I0{}, I0{}, I0{}); rd(st0, I0{}, I0{}, I1{}, I0{}); rd(st0, I0{}, I1{}, I0{}, I0{}); rd(st0, I0{}, I1{}, I1{}, I0{});
                rd(st0, I0{}, I2{}, I0{}, I0{}); rd(st0, I0{}, I2{}, I1{}, I0{}); rd(st0, I0{}, I3{}, I0{}, I0{}); rd(st0, I0{}, I3{}, I1{}, I0{});
                rd(st0, I0{}, I4{}, I0{}, I0{}); rd(st0, I0{}, I4{}, I1{}, I0{}); rd(st0, I0{}, I5{}, I0{}, I0{}); rd(st0, I0{}, I5{}, I1{}, I0{});
            }
            int buf = 0;
            using NA = std::integral_constant<int, NPA>; using NB = std::integral_constant<int, NPB>;
            for (int it = 0; it < NT_; ++it) {
                const uint32_t st = smem_lds + buf * STAGE, stn = smem_lds + (buf ^ 1) * STAGE;
                const bool next = it + 1 < NT_, more = it + 2 < NT_;
                // k-step 0 carries the B pieces of stage it + 1 (its A pieces went out behind the previous barrier); k-steps 1, 2 none
                k_step(st, I1{}, I0{}, (it > 0 && next) ? t_begin + it + 1 : -1, buf ^ 1, NA{}, NB{});
                k_step(st, I2{}, I1{}, 0, 0, I0{}, I0{});
                k_step(st, I3{}, I0{}, 0, 0, I0{}, I0{});
                // stage it + 1 landed for everyone, stage `it` read by everyone (k-step 3's fragments are in registers)
                if (next) wait_vmcnt<0>();
                wait_lgkmcnt<0>();
                block_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // k-step 3: first fragments of stage it + 1 (stale bytes, unused, after the last stage), the A pieces of stage it + 2
                k_step(stn, I0{}, I1{}, more ? t_begin + it + 2 : -1, buf, I0{}, NA{});
                buf ^= 1;
            }
            // accumulator: rows <-> n (A columns), cols <-> k (B columns); lane = k, regs = n
            float* out = jb.out;
            const int N = jb.N, K = jb.K, ldo = jb.ldo;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int k = k0 + wb * 64 + jj * 32 + (lane & 31);
                if (k >= K) continue;
#pragma unroll
                for (int i = 0; i < FWA; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + wa * (TA / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                        if (n < N) unsafeAtomicAdd(out + (size_t)n * ldo + k, acc[i][jj][r]);
                    }
            }
            return;
        }
#endif

        auto stage_piece = [&](int t, int buf, int q) {
            char* base = smem + buf * STAGE;
            const int r0 = t * KB;
            if (q < PA / NWAVES) {
                constexpr int LPR = RBA / 16, RPP = 1024 / RBA;
                const int piece = q * NWAVES + wave;
                const int row = piece * RPP + lane / LPR;
                const int lc = (lane % LPR) ^ ((row & 3) << 2);
                int ca = n0 + lc * 8; ca = ca <= a_cols - 8 ? ca : a_cols - 8;
                const bf16_t* src = A + (size_t)(r0 + row) * lda + ca;
                if constexpr (MASKED) src = r0 + row < r_lim ? src : zsrc;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + piece * 1024), 16, 0, 0);      // (non-temporal operand loads measured slower: 11.75 -> 12.0 / 12.3 ms per step for B / A + B)
            } else {
                constexpr int LPR = RBB / 16, RPP = 1024 / RBB;
                const int piece = (q - PA / NWAVES) * NWAVES + wave;
                const int row = piece * RPP + lane / LPR;
                const int lc = (lane % LPR) ^ ((row & 3) << 2);
                int cb = k0 + lc * 8; cb = cb <= b_cols - 8 ? cb : b_cols - 8;
                const bf16_t* src = B + (size_t)(r0 + row) * ldb + cb;
                if constexpr (MASKED) src = r0 + row < r_lim ? src : zsrc;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + KB * RBA + piece * 1024), 16, 0, 0);
            }
        };
        constexpr int NP = PA / NWAVES + PB / NWAVES;
        auto stage = [&](int t, int buf) {
#pragma unroll
            for (int q = 0; q < NP; ++q) stage_piece(t, buf, q);
        };
        auto stage_part = [&](int t, int buf, int part, int nparts) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
                if (q >= part * NP / nparts && q < (part + 1) * NP / nparts) stage_piece(t, buf, q);
        };

        f32x16_t acc[FWA][2];
#pragma unroll
        for (int i = 0; i < FWA; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

        const int NT_ = t_end - t_begin;
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p)
            if (p < NT_) stage(t_begin + p, p);
        if (NT_ > 1) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
        block_barrier();
        if (!lead) block_barrier();
        int buf = 0;
        for (int it = 0; it < NT_; ++it) {
            const bool more = it + 2 < NT_;
            const int nbuf = buf >= 1 ? buf - 1 : 2;
            const uint32_t st = smem_lds + buf * STAGE;
            union Frag { bf16x8_t v; s16x4_t h[2]; };
            Frag af[KS][FWA], bfr[KS][2];
            auto read_k = [&](auto kk_c) {
                constexpr int kk = decltype(kk_c)::value;
#pragma unroll
                for (int i = 0; i < FWA; ++i) {
                    af[kk][i].h[0] = tr_read<kk * 16 * RBA>(st + baseA[i]);
                    af[kk][i].h[1] = tr_read<kk * 16 * RBA + 4 * RBA>(st + baseA[i]);
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    bfr[kk][jj].h[0] = tr_read<kk * 16 * RBB>(st + baseB[jj]);
                    bfr[kk][jj].h[1] = tr_read<kk * 16 * RBB + 4 * RBB>(st + baseB[jj]);
                }
            };
            read_k(std::integral_constant<int, 0>{});
            read_k(std::integral_constant<int, 1>{});
            if constexpr (KS > 2) read_k(std::integral_constant<int, 2>{});
            if constexpr (KS > 3) read_k(std::integral_constant<int, 3>{});
            if (!lead) {
                if (more) { stage(t_begin + it + 2, nbuf); wait_vmcnt<LOADS>(); } else wait_vmcnt<0>();
            }
            wait_lgkmcnt<0>();
            block_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
                for (int i = 0; i < FWA; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk][i].v, bfr[kk][jj].v, acc[i][jj], 0, 0, 0);
                if (lead && more) stage_part(t_begin + it + 2, nbuf, kk, KS);
            }
            __builtin_amdgcn_s_setprio(0);
            if (lead) { if (more) wait_vmcnt<LOADS>(); else wait_vmcnt<0>(); }
            __builtin_amdgcn_sched_barrier(0);
            if (lead || it + 1 < NT_) block_barrier();
            buf = buf + 1 == STAGES ? 0 : buf + 1;
        }
        // accumulator: rows <-> n (A columns), cols <-> k (B columns); lane = k, regs = n
        float* out = jb.out;
        const int N = jb.N, K = jb.K, ldo = jb.ldo;
        if (a.lab & 1) return;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int k = k0 + wb * 64 + jj * 32 + (lane & 31);
            if (k >= K) continue;
#pragma unroll
            for (int i = 0; i < FWA; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wa * (TA / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
                    if (n < N) unsafeAtomicAdd(out + (size_t)n * ldo + k, acc[i][jj][r]);
                }
        }
    };
    tn_multi_walk(a, w, G, run);
}

int g_nt_config = 9, g_nt_prio = 1;
// experiment knobs (tools/gemm_lab via fm_lab_set): [0] de-phase groups, [1] de-phase step (x 2048 cycles), [2] gemm_nt3 mode (0 off,
// 1 = 256-wide tiles, 2 = 192-wide, 3 = by shape: the default; FOURM_NT3=0 turns it off), [3] gemm_nt3 experiment flags
int g_lab[16] = {0, 0, [] { const char* e = getenv("FOURM_NT3"); return e ? atoi(e) : 3; }(), [] { const char* e = getenv("FOURM_NT3_LAB"); return e ? atoi(e) : 0; }(),
                 [] { const char* e = getenv("FOURM_NT4"); return e ? atoi(e) : 1; }(),       // [4] gemm_nt4 mode (gemm_nt4.hip; FOURM_NT4=0 turns it off)
                 [] { const char* e = getenv("FOURM_TN4"); return e ? atoi(e) : 0; }(), 0, 0, 0,
                 [] { const char* e = getenv("FOURM_NT_SMALL"); return e ? atoi(e) : 1; }(),
                 [] { const char* e = getenv("FOURM_CONV_K32"); return e ? atoi(e) : 1; }()};     // [10] implicit convolutions on K-step 32 (48 KB of LDS: 3 workgroups per CU cover the gather's address arithmetic; FOURM_CONV_K32=0: K-step 64)      // [9] small-grid policy of fm_gemm_nt (128 x 128 tiles / split-K; FOURM_NT_SMALL=0: off)      // [5] gemm_tn4.hip for the dW job lists (FOURM_TN4=1 turns it on; [6] its lab flags, [7] / [8] its planner constants)
int g_nt_swiglu = 12;
int g_nt_auto[2] = {11, 10};        // automatic choice: short reductions / long ones (K >= 1536) and the reading epilogues

// Compute units the persistent GEMM grids may occupy.  With a gradient exchange in flight (fourm.parallel.DataParallel) RCCL's own
// kernels need CUs: a persistent grid that holds every CU for the whole launch would serialise the collective behind each GEMM
// instead of overlapping it, so the data-parallel wrapper reserves a few (fm_set_reserved_cus; multiples of 8 = whole XCD slices).
int g_reserved_cus = [] { const char* e = getenv("FOURM_RESERVED_CUS"); return e ? atoi(e) : 0; }();
static int n_compute_units() {
    static int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    const int avail = n - g_reserved_cus;
    return avail >= 64 ? avail / 8 * 8 : n;
}

}  // namespace
int fm_grid_cus() { return n_compute_units(); }      // (gemm_nt3.hip sizes its persistent grid with the same reservation)
namespace {

template <int TW, int TX, int WW, int WX, int KB, int STAGES, int EPI, bool GROUPED, bool PP = false, bool PERSIST = false, bool CONV = false>
int launch_nt_cfg(NTArgs a, int max_n, hipStream_t s) {
    constexpr int NPT = (EPI == EPI_SWIGLU) ? TW / 2 : TW;
    a.n_tiles_w = (max_n + NPT - 1) / NPT;
    a.n_tiles_x = (a.M + TX - 1) / TX;
    int grid = a.n_tiles_w * a.n_tiles_x;
    if (GROUPED) {
        const int gw_max = TW > 128 ? 8 : 16;            // one L2 working set of W tiles (<= 3.1 MB at K = 768)
        a.group_w = a.n_tiles_w < gw_max ? a.n_tiles_w : gw_max;
        const int chunks = ((a.n_tiles_w + a.group_w - 1) / a.group_w) * ((a.n_tiles_x + 3) / 4);
        grid = (chunks + 7) / 8 * 8 * a.group_w * 4;
    }
    const size_t lds = (size_t)STAGES * (TW + TX) * KB * 2;
    if (PERSIST) {                                        // one resident workgroup per LDS slot, a multiple of the 8 XCDs
        const int slots = n_compute_units() * (int)(160 * 1024 / lds) / 8 * 8;
        if (grid > slots) grid = slots;
    }
    auto k = gemm_nt_kernel<TW, TX, WW, WX, KB, STAGES, EPI, GROUPED, PP, PERSIST, CONV>;
    static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(k, dim3(grid, a.split_k > 1 ? a.split_k : 1), dim3(WW * WX * 64), lds, s, a);
    FM_CHECK_LAUNCH("fm_gemm_nt");
    return 0;
}

// out(bf16)[m][n] = bf16(sum over the K-slices of partial[z][m][n] + bf16(bias[n])): the second pass of a split-K launch
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long stride, int M, int N4, int ldp, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ out, int ldo) {
    const long long total = (long long)M * N4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int m = (int)(e / N4), n = (int)(e % N4) * 4;
        float4 acc = *(const float4*)(ws + (size_t)m * ldp + n);
        for (int z = 1; z < S; ++z) {
            const float4 t = *(const float4*)(ws + (size_t)z * stride + (size_t)m * ldp + n);
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (bias) { const float4 b = *(const float4*)(bias + n); acc.x += bfround(b.x); acc.y += bfround(b.y); acc.z += bfround(b.z); acc.w += bfround(b.w); }
        *(uint2*)(out + (size_t)m * ldo + n) = make_uint2(pack2bf(acc.x, acc.y), pack2bf(acc.z, acc.w));
    }
}

// Tile configurations (fm_set_gemm_nt_config):
//   0  128(feat) x 128(rows), 4 waves, K-step 64, 3 stages
//   1  128 x 256, 8 waves (64x64 wave tiles), K-step 64, 3 stages, 144 KB  (1 workgroup / CU)
//   2  128 x 256, 8 waves, K-step 32, 3 stages, 72 KB                      (2 workgroups / CU: epilogue overlap)
//   3  256 x 256, 8 waves (128x64 wave tiles), K-step 32, 3 stages, 96 KB, ping-pong schedule
//   4  256 x 256, 8 waves, K-step 32, 3 stages, 96 KB
//   5  256 x 256, 8 waves, K-step 64, 2 stages, 128 KB
//   6  128 x 128, 4 waves, K-step 32, 3 stages, 48 KB                      (3 workgroups / CU; grouped path)
//   7  as 1 with the ping-pong schedule (wave rows one barrier apart: LDS reads of one row under the MFMAs of the other)
//   8  as 2 with the ping-pong schedule
//  10  as 7, persistent workgroups (next tile's first stages issued before the store epilogue)
//  11  as 8, persistent workgroups
//  12  as 3 (256 x 256), persistent workgroups
//   +256: s_setprio(1) around the MFMA clusters
//   9  automatic (default): 10 for long reductions (K >= 1536) and the reading epilogues, 2 for SwiGLU, else 11
//      (measured: profiles/r01_v5_nt_config_sweep.txt, profiles/r01_v8_nt_config_sweep.txt)
template <int EPI, bool GROUPED>
int launch_nt(const NTArgs& a, int max_n, hipStream_t s) {
    if constexpr (GROUPED) {
        // rows are segmented in FM_SEG_ROWS (= 256 = the X tile) per group; a.K = upper bound of the groups' K
        static_assert(FM_SEG_ROWS == 256, "the grouped configurations use a 256-row X tile");
        if constexpr (EPI == EPI_BF16) {      // lab (tools/heads_bench.py): 256 x 256 tiles for the dY GEMM of the heads
            static const int heads_cfg_k = [] { const char* e = getenv("FOURM_HEADS_NT_CFG"); return e ? atoi(e) : 0; }();
            if (a.K >= 1536 && heads_cfg_k == 3) return launch_nt_cfg<256, 256, 2, 4, 32, 3, EPI, true, true>(a, max_n, s);
            if (a.K >= 1536 && heads_cfg_k == 4) return launch_nt_cfg<256, 256, 2, 4, 64, 2, EPI, true>(a, max_n, s);
        }
        if (a.K >= 1536) return launch_nt_cfg<128, 256, 2, 4, 64, 3, EPI, true, true>(a, max_n, s);
        if constexpr (EPI == EPI_BF16) {      // lab (tools/heads_bench.py): 256 x 256 ping-pong tiles for the logits GEMM
            static const int heads_cfg = [] { const char* e = getenv("FOURM_HEADS_NT_CFG"); return e ? atoi(e) : 0; }();
            if (heads_cfg == 1) return launch_nt_cfg<256, 256, 2, 4, 32, 3, EPI, true, true>(a, max_n, s);
            if (heads_cfg == 2) return launch_nt_cfg<128, 256, 2, 4, 32, 3, EPI, true, true>(a, max_n, s);
        }
        return launch_nt_cfg<128, 256, 2, 4, 32, 3, EPI, true>(a, max_n, s);
    }
    if (a.M <= 128) return launch_nt_cfg<128, 128, 2, 2, 32, 3, EPI, GROUPED>(a, max_n, s);
    if constexpr (!GROUPED) {
        int cfg = g_nt_config;
        if (cfg == 9) {  // epilogues that READ and long reductions: one workgroup per CU (10); SwiGLU: 256 x 256 tiles (12)
            cfg = EPI == EPI_SWIGLU ? g_nt_swiglu : g_nt_auto[(a.K >= 1536 || EPI == EPI_RES || EPI == EPI_SWIGLU_BWD || EPI == EPI_GELU_BWD) ? 1 : 0];
            // short reductions whose 256 x 256 tiling fills the chip in whole rounds (N = 2048, 1536 at 32768 rows): +5 %
            const long t256 = (long)((max_n + 255) / 256) * ((a.M + 255) / 256);
            if (cfg == g_nt_auto[0] && g_nt_auto[0] == 11 && max_n % 256 == 0 && t256 % n_compute_units() == 0) cfg = 12;
        }
        switch (cfg) {
            case 0: return launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI, false>(a, max_n, s);
            case 1: return launch_nt_cfg<128, 256, 2, 4, 64, 3, EPI, false>(a, max_n, s);
            case 3: return launch_nt_cfg<256, 256, 2, 4, 32, 3, EPI, false, true>(a, max_n, s);
            case 4: return launch_nt_cfg<256, 256, 2, 4, 32, 3, EPI, false>(a, max_n, s);
            case 5: return launch_nt_cfg<256, 256, 2, 4, 64, 2, EPI, false>(a, max_n, s);
            case 6: return launch_nt_cfg<128, 128, 2, 2, 32, 3, EPI, false>(a, max_n, s);
            case 7: return launch_nt_cfg<128, 256, 2, 4, 64, 3, EPI, false, true>(a, max_n, s);
            case 8: return launch_nt_cfg<128, 256, 2, 4, 32, 3, EPI, false, true>(a, max_n, s);
            case 10: return launch_nt_cfg<128, 256, 2, 4, 64, 3, EPI, false, true, true>(a, max_n, s);
            case 11: return launch_nt_cfg<128, 256, 2, 4, 32, 3, EPI, false, true, true>(a, max_n, s);
            case 12: return launch_nt_cfg<256, 256, 2, 4, 32, 3, EPI, false, true, true>(a, max_n, s);
            case 13: if constexpr (EPI == EPI_BF16) return launch_nt_cfg<256, 256, 2, 2, 64, 2, EPI, false>(a, max_n, s); else break;
            case 14: if constexpr (EPI == EPI_BF16) return launch_nt_cfg<256, 256, 2, 2, 32, 3, EPI, false>(a, max_n, s); else break;
            default: return launch_nt_cfg<128, 256, 2, 4, 32, 3, EPI, false>(a, max_n, s);
        }
    }
    return -1;
}

}  // namespace

extern "C" int fm_gemm_nt(const fm_gemm_nt_args* p, void* stream) {
    FM_CHECK_ARG(p && p->X && p->out, "fm_gemm_nt: null pointer");
    const bool grouped = p->groups != nullptr;
    FM_CHECK_ARG(grouped || p->W, "fm_gemm_nt: W is null");
    FM_CHECK_ARG(p->M > 0 && (grouped || (p->N > 0 && p->K > 0)), "fm_gemm_nt: bad shape M=%d N=%d K=%d", p->M, p->N, p->K);
    FM_CHECK_ARG(grouped || p->K % BK == 0, "fm_gemm_nt: K=%d must be a multiple of %d (zero padded)", p->K, BK);
    FM_CHECK_ARG(p->ldx % 8 == 0 && (grouped || p->ldw % 8 == 0), "fm_gemm_nt: leading dims must be multiples of 8");
    // a lane stores 4 consecutive features: when N % 4 != 0 the last group spills into [N, roundup4(N))
    FM_CHECK_ARG(grouped || p->epilogue == FM_EPI_SWIGLU || p->ldo >= (p->N + 3) / 4 * 4, "fm_gemm_nt: ldo=%d too small for N=%d", p->ldo, p->N);
    FM_CHECK_ARG(!p->bias || (((uintptr_t)p->bias) & 15) == 0, "fm_gemm_nt: bias must be 16-byte aligned");
    FM_CHECK_ARG(p->ldo % 4 == 0, "fm_gemm_nt: ldo must be a multiple of 4");
    NTArgs a{};
    a.W = (const bf16_t*)p->W; a.W2 = (const bf16_t*)p->W2; a.X = (const bf16_t*)p->X;
    a.out = p->out; a.out2 = p->out2; a.res = (const float*)p->res; a.bias = (const float*)p->bias; a.bias2 = (const float*)p->bias2;
    a.M = p->M; a.N = p->N; a.K = p->K; a.ldw = p->ldw; a.ldx = p->ldx; a.ldo = p->ldo; a.ldo2 = p->ldo2; a.ldr = p->ldr; a.Hp = p->Hp;
    a.groups = p->groups; a.tile_group = p->tile_group;
    a.prio = g_nt_prio;
    a.dephase_groups = g_lab[0]; a.dephase_step = g_lab[1]; a.lab = g_lab[3];
    hipStream_t s = (hipStream_t)stream;
    a.m_dev = p->m_dev; a.row0_dev = p->row0_dev;
    if (p->m_dev || p->row0_dev) {            // row range in device memory (one dense launch per modality head): gemm_nt3 only
        FM_CHECK_ARG(p->m_dev && p->row0_dev && !grouped, "fm_gemm_nt: m_dev and row0_dev go together (dense launches only)");
        const int r = fm_launch_nt3(a, p->epilogue, g_lab[2] ? g_lab[2] : 3, s);
        if (r < 0) { fm_set_error("fm_gemm_nt (nt3, device-side rows): launch failed"); return -2; }
        if (r > 0) return 0;
        fm_set_error("fm_gemm_nt: a device-side row range needs FM_EPI_BF16 without bias, N %% 8 == 0, K %% 64 == 0, ldo %% 64 == 0 and a 128-byte aligned out "
                     "(N=%d K=%d ldo=%d epilogue=%d)", p->N, p->K, p->ldo, p->epilogue);
        return -1;
    }
    const int max_n = grouped ? p->max_N : p->N;
    if (!grouped && p->M <= 32 && p->conv_C == 0) {             // a handful of rows (a decoding step): the weight-streaming kernel of gemm_skinny.hip
        const int r = fm_launch_nt_skinny(a, p->epilogue, s);
        if (r < 0) { fm_set_error("fm_gemm_nt (skinny): launch failed"); return -2; }
        if (r > 0) return 0;
    }
    a.conv_C = p->conv_C; a.conv_H = p->conv_H; a.conv_W = p->conv_W; a.conv_Ho = p->conv_Ho; a.conv_Wo = p->conv_Wo; a.conv_stride = p->conv_stride; a.conv_up = p->conv_up;
    if (p->conv_C > 0) {          // implicit 3 x 3 convolution: the gathering instantiation of the 128 x 128 kernel, split over K when the grid is small
        FM_CHECK_ARG(!grouped && !p->m_dev && !p->out2 && !p->res && (p->epilogue == FM_EPI_BF16 || p->epilogue == FM_EPI_F32), "fm_gemm_nt (conv): dense FM_EPI_BF16 / FM_EPI_F32 only");
        FM_CHECK_ARG(p->conv_C % 64 == 0 && p->K == 9 * p->conv_C && (p->conv_stride == 1 || p->conv_stride == 2) && (p->conv_up == 0 || p->conv_up == 1) &&
                     p->conv_H > 0 && p->conv_W > 0 && p->conv_Ho > 0 && p->conv_Wo > 0 && p->M % (p->conv_Ho * p->conv_Wo) == 0,
                     "fm_gemm_nt (conv): C=%d K=%d stride=%d up=%d grid %dx%d -> %dx%d M=%d", p->conv_C, p->K, p->conv_stride, p->conv_up, p->conv_H, p->conv_W, p->conv_Ho, p->conv_Wo, p->M);
        FM_CHECK_ARG(p->conv_Ho == (p->conv_H + 2 - 3) / p->conv_stride + 1 && p->conv_Wo == (p->conv_W + 2 - 3) / p->conv_stride + 1, "fm_gemm_nt (conv): output grid does not match the input grid");
        if (p->epilogue == FM_EPI_F32) return launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI_F32, false, false, false, true>(a, max_n, s);
        const int cus = n_compute_units();
        const long t128 = (long)((p->M + 127) / 128) * ((p->N + 127) / 128);
        const int kt = p->K / 64;
        const long ldp = (p->N + 3) / 4 * 4;
        int S = 1;
        if (g_lab[9] && p->splitk_ws && t128 * 2 <= cus && p->N % 4 == 0 && (((uintptr_t)p->splitk_ws) & 15) == 0 && (((uintptr_t)p->out) & 7) == 0)
            S = (int)std::min<long>({cus / t128, (long)kt / 4, 16L, (long)(p->splitk_ws_bytes / ((long long)p->M * ldp * 4))});
        if (S >= 2) {
            const int ks = (kt + S - 1) / S * 64;
            S = (p->K + ks - 1) / ks;
            NTArgs b = a;
            b.out = p->splitk_ws; b.ldo = (int)ldp; b.bias = nullptr;
            b.split_k = S; b.k_slice = ks; b.split_stride = (long long)p->M * ldp;
            const int r = launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI_F32, false, false, false, true>(b, max_n, s);
            if (r != 0) return r;
            const long long quads = (long long)p->M * (ldp / 4);
            const int rgrid = (int)std::min<long long>((quads + 255) / 256, 2048);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, s, (const float*)p->splitk_ws, S, b.split_stride, p->M, (int)(ldp / 4), (int)ldp,
                               (const float*)p->bias, (bf16_t*)p->out, p->ldo);
            FM_CHECK_LAUNCH("fm_gemm_nt (conv, split-K reduction)");
            return 0;
        }
        if (g_lab[10]) return launch_nt_cfg<128, 128, 2, 2, 32, 3, EPI_BF16, false, false, false, true>(a, max_n, s);      // K-step 32, 48 KB of LDS = 3 workgroups per CU (43.3 -> 44.6 images/s)
        return launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI_BF16, false, false, false, true>(a, max_n, s);
    }
    // Small grids (FOURM_NT_SMALL=0 / fm_lab_set(9, 0): off): a dense bf16 launch whose 256 x 256 tiling would occupy less than half of the CUs
    // (the convolutions of the DiVAE UNet: M = batch x 56^2 ... batch x 7^2 rows, N = 256 / 512, K up to 9216) runs on 128 x 128 tiles; when
    // even those leave half of the chip idle and the reduction is long, K is cut into slices on gridDim.y (fp32 partial tiles in the caller's
    // scratch, fm_gemm_nt_args.splitk_ws) and one reduction pass adds the bias and rounds: M = 392, N = 512, K = 4608: 64 -> ~15 us.
    if (!grouped && g_lab[9] && p->epilogue == FM_EPI_BF16 && p->M > 32 && !p->out2 && !p->res && p->N % 4 == 0 && (((uintptr_t)p->out) & 7) == 0) {
        const int cus = n_compute_units();
        const long t256 = (long)((p->M + 255) / 256) * ((p->N + 255) / 256);
        if (t256 * 2 <= cus) {
            const long t128 = (long)((p->M + 127) / 128) * ((p->N + 127) / 128);
            const int kt = p->K / 64;
            int S = 1;
            const long ldp = (p->N + 3) / 4 * 4;
            if (p->splitk_ws && t128 * 2 <= cus && kt >= 8 && p->K % 64 == 0 && (((uintptr_t)p->splitk_ws) & 15) == 0) {
                S = (int)std::min<long>({cus / t128, (long)kt / 4, 16L, (long)(p->splitk_ws_bytes / ((long long)p->M * ldp * 4))});
            }
            if (S >= 2) {
                const int ks = (kt + S - 1) / S * 64;             // K elements per slice (whole K-tiles); the last slice may be shorter
                S = (p->K + ks - 1) / ks;
                NTArgs b = a;
                b.out = p->splitk_ws; b.ldo = (int)ldp; b.bias = nullptr;
                b.split_k = S; b.k_slice = ks; b.split_stride = (long long)p->M * ldp;
                const int r = launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI_F32, false>(b, max_n, s);
                if (r != 0) return r;
                const long long quads = (long long)p->M * (ldp / 4);
                const int rgrid = (int)std::min<long long>((quads + 255) / 256, 2048);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(rgrid), dim3(256), 0, s, (const float*)p->splitk_ws, S, b.split_stride, p->M, (int)(ldp / 4), (int)ldp,
                                   (const float*)p->bias, (bf16_t*)p->out, p->ldo);
                FM_CHECK_LAUNCH("fm_gemm_nt (split-K reduction)");
                return 0;
            }
            if (p->K >= 1536) return launch_nt_cfg<128, 128, 2, 2, 64, 3, EPI_BF16, false>(a, max_n, s);
            return launch_nt_cfg<128, 128, 2, 2, 32, 3, EPI_BF16, false>(a, max_n, s);
        }
    }
    if (!grouped && g_lab[4]) {               // the 4-wave 256 x 384-tile kernel (gemm_nt4.hip) takes the plain bf16 launches it tiles exactly
        const int r = fm_launch_nt4(a, p->epilogue, g_lab[4], s);
        if (r < 0) { fm_set_error("fm_gemm_nt (nt4): launch failed"); return -2; }
        if (r > 0) return 0;
    }
    if (!grouped && g_lab[2]) {               // the lock-step large-tile kernel (gemm_nt3.hip) takes the dense launches it handles
        const int r = fm_launch_nt3(a, p->epilogue, g_lab[2], s);
        if (r < 0) { fm_set_error("fm_gemm_nt (nt3): launch failed"); return -2; }
        if (r > 0) return 0;
    }
    if (!grouped && g_nt_config == 9) {       // the flattened persistent kernel takes the big dense launches it handles
        const int r = fm_launch_nt_flat(a, p->epilogue, s);
        if (r < 0) { fm_set_error("fm_gemm_nt (flat): launch failed"); return -2; }
        if (r > 0) return 0;
    }
    if (grouped) {
        FM_CHECK_ARG(p->tile_group && p->max_N > 0, "fm_gemm_nt: grouped mode needs tile_group and max_N");
        FM_CHECK_ARG(p->epilogue == FM_EPI_BF16, "fm_gemm_nt: grouped mode supports FM_EPI_BF16 only");
        return launch_nt<EPI_BF16, true>(a, max_n, s);
    }
    switch (p->epilogue) {
        case FM_EPI_BF16: return launch_nt<EPI_BF16, false>(a, max_n, s);
        case FM_EPI_GELU: return launch_nt<EPI_GELU, false>(a, max_n, s);
        case FM_EPI_RESIDUAL:
            FM_CHECK_ARG(p->res && p->ldr % 4 == 0, "fm_gemm_nt: residual epilogue needs res / ldr%%4==0");
            return launch_nt<EPI_RES, false>(a, max_n, s);
        case FM_EPI_SWIGLU:
            FM_CHECK_ARG(p->W2 && p->Hp % 4 == 0 && (!p->out2 || p->ldo2 % 4 == 0), "fm_gemm_nt: SwiGLU epilogue needs W2, Hp%%4==0");
            return launch_nt<EPI_SWIGLU, false>(a, max_n, s);
        case FM_EPI_F32: return launch_nt<EPI_F32, false>(a, max_n, s);
        case FM_EPI_TANH: return launch_nt<EPI_TANH, false>(a, max_n, s);
        case FM_EPI_SWIGLU_BWD:
            FM_CHECK_ARG(p->res && p->ldr % 4 == 0 && p->Hp % 4 == 0 && p->Hp >= p->N && p->ldo >= 2 * p->Hp && p->ldr >= 2 * p->Hp,
                         "fm_gemm_nt: SwiGLU-backward epilogue needs res=(g|u) and out=(dg|du) of width 2*Hp");
            return launch_nt<EPI_SWIGLU_BWD, false>(a, max_n, s);
        case FM_EPI_GELU_BWD:
            FM_CHECK_ARG(p->res && p->ldr % 4 == 0, "fm_gemm_nt: GELU-backward epilogue needs res = pre-activation");
            return launch_nt<EPI_GELU_BWD, false>(a, max_n, s);
    }
    fm_set_error("fm_gemm_nt: unknown epilogue %d", p->epilogue);
    return -1;
}

extern int g_nt_flat;
extern "C" void fm_set_gemm_nt_config(int cfg) {
    g_nt_config = cfg & 0xff; g_nt_prio = (cfg >> 8) & 1;
    g_nt_flat = (cfg >> 29) & 1;                                                                         // bit 29: flattened persistent kernel (gemm_nt_flat.hip) where it applies
    if ((cfg >> 16) & 0xff) { g_nt_auto[0] = (cfg >> 16) & 0xf; g_nt_auto[1] = (cfg >> 20) & 0xf; }   // bits 16-19 / 20-23: automatic pair
    if ((cfg >> 24) & 0xf) g_nt_swiglu = (cfg >> 24) & 0xf;                                             // bits 24-27: the SwiGLU choice
}
extern "C" int fm_get_gemm_nt_config(void) { return g_nt_config; }
extern "C" void fm_lab_set(int key, int value) { if (key >= 0 && key < 16) g_lab[key] = value; }
extern "C" void fm_set_reserved_cus(int n) { g_reserved_cus = n < 0 ? 0 : n; }
extern "C" int fm_get_reserved_cus(void) { return g_reserved_cus; }
static int g_tn_config = [] { const char* e = getenv("FOURM_TN_CONFIG"); return e ? atoi(e) : 1; }();      // 0 lock-step K32 x 2 WG/CU, 1 ping-pong (dW list: 256 x 256 / K32), 3 as 1 with 128 x 256 list tiles, 4 as 1 with the lock-step list kernel (equal in situ: 11.6 ms per 4M-B step either way, profiles/r03_ab_tn_lockstep.txt)
extern "C" void fm_set_gemm_tn_config(int cfg) { g_tn_config = cfg; }
extern "C" int fm_get_gemm_tn_config(void) { return g_tn_config; }
static int g_tn_use_tr = 1;   // ds_read_b64_tr_b16 semantics verified on hardware (tools/probe_gfx950.hip)
extern "C" void fm_set_tn_transpose_read(int on) { g_tn_use_tr = on; }
extern "C" int fm_get_tn_transpose_read(void) { return g_tn_use_tr; }

extern "C" int fm_gemm_tn(const fm_gemm_tn_args* p, void* stream) {
    FM_CHECK_ARG(p && p->A && p->B, "fm_gemm_tn: null pointer");
    const bool grouped = p->groups != nullptr;
    FM_CHECK_ARG(grouped || p->out, "fm_gemm_tn: out is null");
    FM_CHECK_ARG(p->K > 0 && (grouped || (p->N > 0 && p->R > 0)), "fm_gemm_tn: bad shape");
    FM_CHECK_ARG(p->lda % 8 == 0 && p->ldb % 8 == 0, "fm_gemm_tn: leading dims must be multiples of 8");
    TNArgs a{};
    a.A = (const bf16_t*)p->A; a.B = (const bf16_t*)p->B; a.out = (float*)p->out;
    a.R = p->R; a.N = p->N; a.K = p->K; a.lda = p->lda; a.ldb = p->ldb; a.ldo = p->ldo;
    a.a_cols = p->a_cols > 0 ? p->a_cols : p->lda; a.b_cols = p->b_cols > 0 ? p->b_cols : p->ldb;
    FM_CHECK_ARG(a.a_cols >= 8 && a.b_cols >= 8, "fm_gemm_tn: operands need at least 8 readable columns");
    a.groups = p->groups; a.seg_start = p->seg_start; a.seg_count = p->seg_count;
    const int max_n = grouped ? p->max_N : p->N;
    // configuration 0: K-step 32, 72 KB of LDS, two workgroups per CU, lock-step waves
    // configuration 1: K-step 64, 144 KB, ONE workgroup per CU, ping-pong schedule: half the workgroups, so half the
    //                  fp32 atomic traffic of the epilogue (workgroups x 128 KB per launch, ~25 us at 512 workgroups)
    constexpr int TN_TA = 128, TN_TB = 256, TN_STAGES = 3;
    // (the grouped head GEMM keeps configuration 0: short, uneven reductions - 940 vs 1110 us at the 4M-B shapes)
    const bool pp = (g_tn_config == 1 || g_tn_config == 3 || g_tn_config == 4) && !grouped && p->force_tr != 0;
    const int kb = pp ? 64 : 32, slots = (pp ? 1 : 2) * n_compute_units();
    a.n_tiles_a = (max_n + TN_TA - 1) / TN_TA; a.n_tiles_b = (p->K + TN_TB - 1) / TN_TB;
    int splits = p->splits;
    if (splits <= 0) {
        // Fill the chip in ONE round (e.g. 66 tiles on 512 slots take 7 splits = 462 workgroups; 8 would leave 16
        // workgroups for a second round).
        const int tiles = a.n_tiles_a * a.n_tiles_b;
        const int nt = grouped ? (p->max_R + kb - 1) / kb : (p->R + kb - 1) / kb;
        splits = tiles >= slots ? 1 : slots / tiles;
        while (splits > 1 && nt / splits < 8) --splits;                // keep >= 8 reduction tiles per workgroup
        if (splits > 64) splits = 64;
        if (nt < 16) splits = 1;                                       // tiny problems
    }
    a.splits = splits;
    const size_t lds = (size_t)TN_STAGES * kb * (TN_TA + TN_TB) * 2;
    const int total_items = a.n_tiles_a * a.n_tiles_b * splits;
    const int deal = grouped ? 8 * 4 * a.n_tiles_b : 8;                // grouped: chunks of 4 A-tiles x all B-tiles per XCD
    dim3 grid((total_items + deal - 1) / deal * deal, 1, grouped ? p->n_groups : 1);
    hipStream_t s = (hipStream_t)stream;
    const bool masked = grouped || p->R % kb != 0;
#define LAUNCH_TN(TR, G, KBV, PPV)                                                                  \
    if (masked) LAUNCH_TN2(TR, G, KBV, PPV, true) else LAUNCH_TN2(TR, G, KBV, PPV, false)
#define LAUNCH_TN2(TR, G, KBV, PPV, MK)                                                             \
    {                                                                                               \
        auto k = gemm_tn_kernel<TR, G, TN_TA, TN_TB, 2, 4, KBV, TN_STAGES, PPV, MK>;                \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true); \
        (void)once;                                                                                 \
        hipLaunchKernelGGL(k, grid, dim3(512), lds, s, a);                                          \
    }
    const int tr = p->force_tr >= 0 ? p->force_tr : g_tn_use_tr;
    if (pp) { if (grouped) { LAUNCH_TN(true, true, 64, true); } else { LAUNCH_TN(true, false, 64, true); } }
    else if (tr) { if (grouped) { LAUNCH_TN(true, true, 32, false); } else { LAUNCH_TN(true, false, 32, false); } }
    else { if (grouped) { LAUNCH_TN(false, true, 32, false); } else { LAUNCH_TN(false, false, 32, false); } }
#undef LAUNCH_TN
#undef LAUNCH_TN2
    FM_CHECK_LAUNCH("fm_gemm_tn");
    return 0;
}

extern "C" int fm_gemm_tn_multi(const fm_gemm_tn_job* jobs, int n_jobs, void* stream) {
    FM_CHECK_ARG(jobs && n_jobs > 0 && n_jobs <= FM_TN_MAX_JOBS, "fm_gemm_tn_multi: 1..FM_TN_MAX_JOBS jobs");
    TNMultiArgs a{};
    bool masked = false;
    int tiles = 0;
    long long units = 0;
    // 256 x 256 tiles with K-step 32 (default) or 128 x 256 with K-step 64 (fm_set_gemm_tn_config(3); FOURM_TN_MULTI_TILE=128): the
    // large tile needs 2/3 of the LDS-DMA pieces and 3/8 of the transpose reads per MFMA - 4M-B decoder layer 646 -> 574 us, encoder
    // layer 478 -> 461 us (profiles/r02_lab_tn_multi_tiles.txt)
    static const bool small_env = [] { const char* e = getenv("FOURM_TN_MULTI_TILE"); return e && atoi(e) == 128; }();
    const bool big = !(small_env || g_tn_config == 3);
    // gemm_tn4.hip (4 waves, 512 registers, 256 x 384 tiles, K-step 64): when every job fits its shape rules and the tiling wastes < 8 % of
    // its MFMAs on columns past the operands.  OFF by default (FOURM_TN4=1 / fm_lab_set(5, 1): on): whole rounds of whole tiles run 10 - 13 %
    // faster than on the 8-wave kernel (1250 - 1320 against 1134 - 1163 TFLOP/s), but a 4M-B layer is 74 / 96 tiles on 256 CUs - every tile cut
    // 3.5 ways, every segment ending in 384 KB of fp32 atomics instead of 256 KB - and lands where the 8-wave kernel does (encoder / decoder layer
    // 430 / 551 against 433 / 559 us; without the atomics 393 / 514 against 413 / 529: profiles/r06_lab_tn4.txt)
    bool t4 = big && g_lab[5] != 0 && (g_tn_config == 1 || g_tn_config == 4);
    if (t4) {
        double useful = 0, padded = 0;
        for (int i = 0; i < n_jobs && t4; ++i) {
            const fm_gemm_tn_job& p = jobs[i];
            const int ac = p.a_cols > 0 ? p.a_cols : p.lda, bc = p.b_cols > 0 ? p.b_cols : p.ldb;
            t4 = p.R % 64 == 0 && p.R >= 256 && p.N % 128 == 0 && p.K % 128 == 0 && ac >= 128 && bc >= 128 && ac % 8 == 0 && bc % 8 == 0 &&
                 (((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0 && ((size_t)p.R * p.lda + ac) * 2 < 0x7fffffffull && ((size_t)p.R * p.ldb + bc) * 2 < 0x7fffffffull;
            useful += (double)p.N * p.K * p.R;
            padded += (double)((p.N + 255) / 256 * 256) * ((p.K + 383) / 384 * 384) * p.R;
        }
        t4 = t4 && useful >= 0.92 * padded;
    }
    const bool ls = big && (g_tn_config == 4 || t4);         // lock-step form: 256 x 256 tiles, K-step 64, two stages (the planner constants of t4 too)
    const int ta = big ? 256 : 128, kb = (big && !ls) ? 32 : 64, tb = t4 ? 384 : 256;
    for (int i = 0; i < n_jobs; ++i) {
        const fm_gemm_tn_job& p = jobs[i];
        FM_CHECK_ARG(p.A && p.B && p.out, "fm_gemm_tn_multi: null pointer");
        FM_CHECK_ARG(p.K > 0 && p.N > 0 && p.R > 0, "fm_gemm_tn_multi: bad shape");
        FM_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0, "fm_gemm_tn_multi: leading dims must be multiples of 8");
        TNJob& j = a.job[i];
        j.A = (const bf16_t*)p.A; j.B = (const bf16_t*)p.B; j.out = (float*)p.out;
        j.R = p.R; j.N = p.N; j.K = p.K; j.lda = p.lda; j.ldb = p.ldb; j.ldo = p.ldo;
        j.a_cols = p.a_cols > 0 ? p.a_cols : p.lda; j.b_cols = p.b_cols > 0 ? p.b_cols : p.ldb;
        FM_CHECK_ARG(j.a_cols >= 8 && j.b_cols >= 8, "fm_gemm_tn_multi: operands need at least 8 readable columns");
        j.n_tiles_b = (p.K + tb - 1) / tb;
        j.tiles = ((p.N + ta - 1) / ta) * j.n_tiles_b;
        j.tile_start = tiles;
        j.kt = (p.R + kb - 1) / kb;
        tiles += j.tiles;
        units += (long long)j.tiles * j.kt;
        masked = masked || p.R % kb != 0;
    }
    a.n_jobs = n_jobs; a.tiles = tiles; a.lab = g_lab[6];
    int grid = n_compute_units();
    // tiny lists: no more workgroups than 8-k-tile shares (a multiple of the 8 XCDs)
    if (units / 8 < grid) grid = (int)((units / 8 + 7) / 8 * 8);
    if (grid < 8) grid = 8;
    {   // The cut of the last partial round.  A segment costs its k-tiles + c (ring fill + the atomic epilogue; c fitted on
        // profiles/r02_lab_tn_multi.txt, r02_lab_tn_multi_tiles.txt); main and the busiest tail are balanced on that.  With at least as many remaining
        // tiles as tail workgroups, whole tails are dealt round-robin: the tails then walk neighbouring tiles over the SAME rows
        // in lock step and share operand panels in L2 like the mains do (4M-B encoder layer, 216 tiles: 476 us against 512 us for
        // contiguous runs).  Otherwise each tail is cut into contiguous runs, ~ ntail / rem per tile.
        // (in k-tiles of this configuration; the 256 KB atomic epilogue and the refill of the large tile weigh more when a workgroup's
        // share is a fraction of a tile, i.e. in the contiguous cut)
        const int rem = tiles % grid, ntail = grid - rem;
        a.tail_rr = rem >= ntail;
        double c = ls ? (a.tail_rr ? 12.0 : 64.0) : big ? (a.tail_rr ? 24.0 : 128.0) : 8.0;     // (in k-tiles of this configuration)
        if (t4 && g_lab[7] > 0) c = g_lab[7];                                                     // (lab: fm_lab_set 7 / 8 = c / cb of the 256 x 384 kernel)
        // Contiguous cut with bands (rem < ntail): tail workgroup i takes ONE band [q, q + lb) of tile i - the first rem tails start
        // together on neighbouring tiles over the same rows and share operand panels in L2 like the mains (as plain contiguous runs
        // the tails re-read 1.9 x the operands: profiles/r02_v6_traffic_table.txt) - the other ntail - rem walk the rest.  4M-B encoder
        // layer (108 tiles): 474 -> 420 us (profiles/r02_lab_tn_multi_tiles.txt).  FOURM_TN_BANDS=0: plain contiguous runs (A/B).
        double cb = ls ? 16.0 : 32.0;
        if (t4 && g_lab[8] > 0) cb = g_lab[8];
        static const bool band_off = [] { const char* e = getenv("FOURM_TN_BANDS"); return e && atoi(e) == 0; }();
        a.banded = !a.tail_rr && rem > 0 && ntail > rem && !band_off;
        static const bool hybrid_off = [] { const char* e = getenv("FOURM_TN_HYBRID"); return e && atoi(e) == 0; }();
        a.hybrid = a.tail_rr && rem > ntail && ntail > 0 && !hybrid_off;
        for (int i = 0; i < n_jobs; ++i) {
            TNJob& j = a.job[i];
            j.lb = 0;
            if (rem == 0) { j.q = j.kt; continue; }
            if (a.banded) {       // mains and band tails: one segment of q (+ c); walkers: rem (kt - 2 q) / nwalk + (rem / nwalk + 1) c
                const double rw = (double)rem / (ntail - rem);
                double q = (rw * j.kt + rw * cb) / (1.0 + 2.0 * rw);
                if (q < 0) q = 0;
                if (2 * q > j.kt) q = j.kt / 2;
                j.q = j.lb = (int)q;
                continue;
            }
            if (a.hybrid) {       // main: q + c;  tail: one whole tail + its share of the other rem - ntail:  (kt - q) rem / ntail + 2 c
                const double r = (double)rem / ntail;
                j.q = (int)((j.kt * r + c) / (1.0 + r));
            } else if (a.tail_rr) {      // the busiest tail workgroup has n = ceil(rem / ntail) tails:  q + c = n (kt - q + c)
                const int n = (rem + ntail - 1) / ntail;
                double left = (j.kt - (n - 1) * c) / (n + 1.0);
                j.q = j.kt - (left > 0 ? (int)left : 0);
            } else {              // r = rem / ntail tails per tail workgroup on average:  q + c = (kt - q) r + (r + 1) c
                const double r = (double)rem / ntail;
                j.q = (int)((j.kt * r + r * c) / (1.0 + r));
            }
            if (j.q > j.kt) j.q = j.kt;
            if (j.q < 0) j.q = 0;
        }
    }
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TNM(MK, TAV, KBV, ...)                                                                                         \
    {                                                                                                                         \
        constexpr bool lsv = sizeof(#__VA_ARGS__) > 1;                                                                        \
        constexpr size_t lds = (size_t)(lsv ? 2 : 3) * KBV * (TAV + 256) * 2;                                                 \
        auto k = gemm_tn_multi_kernel<MK, TAV, KBV, lsv>;                                                                     \
        static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true); \
        (void)once;                                                                                                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, a);                                                              \
    }
    if (t4) fm_launch_tn4_multi(a, grid, s);
    else if (ls) LAUNCH_TNM(false, 256, 64, ls)        // (row masking through the buffer descriptor)
    else if (big) { if (masked) LAUNCH_TNM(true, 256, 32) else LAUNCH_TNM(false, 256, 32) }
    else { if (masked) LAUNCH_TNM(true, 128, 64) else LAUNCH_TNM(false, 128, 64) }
#undef LAUNCH_TNM
    FM_CHECK_LAUNCH("fm_gemm_tn_multi");
    return 0;
}
