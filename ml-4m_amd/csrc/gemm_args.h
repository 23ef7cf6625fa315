// Argument record and epilogue helpers shared by the NT GEMM kernels (gemm.hip, gemm_nt_flat.hip).
#pragma once
#include "common.h"
#include "fourm_hip.h"

namespace fmk {

enum { EPI_BF16 = FM_EPI_BF16, EPI_GELU = FM_EPI_GELU, EPI_RES = FM_EPI_RESIDUAL, EPI_SWIGLU = FM_EPI_SWIGLU,
       EPI_F32 = FM_EPI_F32, EPI_TANH = FM_EPI_TANH, EPI_SWIGLU_BWD = FM_EPI_SWIGLU_BWD, EPI_GELU_BWD = FM_EPI_GELU_BWD };

struct NTArgs {
    const bf16_t* W; const bf16_t* W2; const bf16_t* X;
    void* out; void* out2; const float* res; const float* bias; const float* bias2;
    int M, N, K, ldw, ldx, ldo, ldo2, ldr, Hp;
    const fm_gemm_group* groups; const int* tile_group;   // grouped mode (may be null)
    const int* m_dev; const int* row0_dev;                // gemm_nt3 DEVM: row count / first row of this launch in device memory (may be null)
    int n_tiles_w, n_tiles_x;
    int group_w;                                          // grouped: W-tiles per column block
    int prio;                                             // raise the wave priority around the MFMA clusters
    int dephase_groups, dephase_step;                     // experiment (fm_lab_set): staggered workgroup start
    int reverse;                                          // gemm_nt3: tiles from the last row block to the first - a consumer that starts with the rows its producer wrote LAST finds them in the Infinity Cache (LRU: a forward walk over more than 256 MB of freshly written data meets the oldest, evicted, rows first)
    // implicit 3 x 3 convolution (gemm_nt_kernel<..., CONV>): X = a (B, conv_H >> conv_up, conv_W >> conv_up, conv_C) bf16 feature map in rows, read at
    // (y >> conv_up, x >> conv_up); output row m = (b, oy, ox) of a (conv_Ho, conv_Wo) grid; reduction index k = tap * conv_C + c reads
    // in[b][oy * stride + tap / 3 - 1][ox * stride + tap % 3 - 1][c] (zero outside the (conv_H, conv_W) grid) - fm_unet_im2col's rows without the round trip
    int conv_C, conv_H, conv_W, conv_Ho, conv_Wo, conv_stride, conv_up;
    int split_k, k_slice; long long split_stride;         // gemm_nt_kernel with FM_EPI_F32 on gridDim.y = split_k K-slices of k_slice elements: slice z writes fp32 partials at out + z * split_stride
    int lab;                                              // experiment flags of gemm_nt3 (fm_lab_set 3): 1 no wait for the DMA, 2 no DMA, 4 no stores, 16 all DMA pieces in one k-step, 256 take the residual epilogue
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }


// ---- TN job list (fm_gemm_tn_multi): the tiles of every weight-gradient GEMM of one transformer layer on one grid; kernels: gemm.hip
// (gemm_tn_multi_kernel, 8 waves, 128 / 256 x 256 tiles) and gemm_tn4.hip (4 waves, 512 registers, 256 x 384 tiles) -------------------
struct TNJob {
    const bf16_t* A; const bf16_t* B; float* out;
    int R, N, K, lda, ldb, ldo, a_cols, b_cols;
    int n_tiles_b, tiles, tile_start, kt;        // kt = reduction tiles (of 64 rows) of this job
    int q;                                       // main share of a tile of the last partial round
    int lb;                                      // banded cut: k-tiles [q, q + lb) of tile i go to tail workgroup i (0: no band)
};
struct TNMultiArgs {
    TNJob job[FM_TN_MAX_JOBS];
    int n_jobs, tiles, tail_rr, banded;
    int hybrid;             // round-robin tails with rem > ntail: one whole tail per tail workgroup, the other rem - ntail tails walked by all of them
    int lab;                // experiment flags (fm_lab_set 6): 1 = no atomic epilogue (timing only)
};


// The segments (tile, k-tiles [t0, t1)) of workgroup w (logical index) of a G-workgroup grid, in execution order: whole tiles while a full
// round of tiles remains, then this workgroup's share of the cut of the last partial round (main / band / round-robin tail / walked run;
// planned on the host by fm_gemm_tn_multi).  `run(tile, t0, t1)` is the one call site of the kernel's main loop.
template <typename Run>
__device__ __forceinline__ void tn_multi_walk(const TNMultiArgs& a, int w, int G, Run&& run) {
    auto job_of = [&](int tile) {
        int j = 0;
        while (j + 1 < a.n_jobs && tile >= a.job[j + 1].tile_start) ++j;
        return j;
    };

    // the segments of this workgroup, one call site for the main loop
    const int full = a.tiles / G, T0 = full * G, rem = a.tiles - T0, ntail = G - rem;
    auto q_of = [&](int j) { return a.job[j].q; };
    // Banded cut (contiguous mode): the first `rem` tail workgroups take ONE band [q, q + lb) of "their" tile each - they start
    // together on neighbouring tiles over the same rows, so they share operand panels in L2 like the mains - and only the remaining
    // ntail - rem workgroups walk what is left ([q + lb, kt) of every tile) as contiguous runs.
    // Hybrid cut (round-robin mode with more tails than tail workgroups, e.g. a 4M-B decoder layer: 144 tiles, 112 tail workgroups):
    // tail workgroup j takes the WHOLE tail of tile j (aligned with its neighbours, like the mains) and the tails of the other
    // rem - ntail tiles are walked by ALL tail workgroups as contiguous runs - instead of 32 of them taking a second whole tail
    // (makespan 691 k-tiles against a mean of 576).
    const int nband = a.hybrid ? 0 : (a.banded ? rem : 0), nwalk = ntail - nband;
    const int walk_T0 = a.hybrid ? T0 + ntail : T0;                         // first tile of the walked region
    auto q2_of = [&](int j) { return a.job[j].q + ((a.banded && !a.hybrid) ? a.job[j].lb : 0); };
    long long u0 = 0, u1 = 0;
    if (rem > 0 && w >= rem + nband && (!a.tail_rr || a.hybrid)) {
        long long Lsum = 0;
        for (int j = 0; j < a.n_jobs; ++j) {
            const int lo = max(a.job[j].tile_start, walk_T0), hi = a.job[j].tile_start + a.job[j].tiles;
            if (hi > lo) Lsum += (long long)(hi - lo) * (a.job[j].kt - q2_of(j));
        }
        u0 = Lsum * (w - rem - nband) / nwalk; u1 = Lsum * (w - rem - nband + 1) / nwalk;
    }
    int phase = 0, f = 0, tj = 0, ci = -1;
    long long P = 0;
    for (;;) {
        int tile = 0, t0 = 0, t1 = 0;
        bool have = false;
        if (phase == 0) {
            if (f < full) { tile = f * G + w; t1 = a.job[job_of(tile)].kt; ++f; have = true; }
            else { phase = rem == 0 ? 3 : (w < rem ? 1 : ((!a.tail_rr && w < rem + nband) ? 4 : 2)); f = 0; }
        } else if (phase == 1) {
            tile = T0 + w; t1 = q_of(job_of(tile)); phase = 3; have = true;
        } else if (phase == 4) {
            tile = T0 + (w - rem); const int j = job_of(tile); t0 = q_of(j); t1 = t0 + a.job[j].lb; phase = 3; have = true;
        } else if (phase == 2 && a.tail_rr && !(a.hybrid && f > 0)) {
            const int sgm = (w - rem) + f * ntail;          // f counts this tail's segments here
            if (sgm >= rem) phase = 3;
            else { tile = T0 + sgm; const int j = job_of(tile); t0 = q_of(j); t1 = a.job[j].kt; ++f; have = true; }
        } else if (phase == 2) {                            // (hybrid: after the one whole tail, this workgroup's share of the walk)
            if (tj >= a.n_jobs) phase = 3;
            else {
                const int lo = max(a.job[tj].tile_start, walk_T0), hi = a.job[tj].tile_start + a.job[tj].tiles;
                const int q = q2_of(tj), left = a.job[tj].kt - q;
                bool advance = true;
                if (hi > lo && left > 0) {
                    const long long Pn = P + (long long)(hi - lo) * left;
                    if (u1 > P && u0 < Pn) {
                        const long long s0 = max(u0, P) - P, s1 = min(u1, Pn) - P;       // leftover k-tiles of this job: [s0, s1)
                        if (ci < 0) ci = (int)(s0 / left);
                        const long long base = (long long)ci * left;
                        if (base < s1) {
                            tile = lo + ci; t0 = q + (int)(max(s0, base) - base); t1 = q + (int)(min(s1, base + left) - base);
                            ++ci; have = true; advance = false;
                        }
                    }
                    if (advance) P = Pn;
                }
                if (advance) { ++tj; ci = -1; }
            }
        } else break;
        if (have) run(tile, t0, t1);
    }
}

}  // namespace fmk

// gemm_nt_flat.hip: the flattened persistent kernel.  Returns 1 when it took the launch, 0 when the arguments are outside what it
// handles (the caller then uses the tile-at-a-time kernels of gemm.hip), < 0 on a launch error.
int fm_launch_nt_flat(const fmk::NTArgs& a, int epilogue, hipStream_t s);
// gemm_nt3.hip: the lock-step large-tile kernel (same return convention; mode 1 = 256-wide tiles, 2 = 192-wide, 3 = by shape)
int fm_launch_nt3(const fmk::NTArgs& a, int epilogue, int mode, hipStream_t s);
// gemm_nt4.hip: the 4-wave / 512-register kernel on 256 x 384 tiles (same return convention; mode bits: see the file)
int fm_launch_nt4(const fmk::NTArgs& a, int epilogue, int mode, hipStream_t s);
// gemm_skinny.hip: M <= 32 rows (decoding steps): one workgroup per 32 output features, the 4 waves split K (same return convention)
int fm_launch_nt_skinny(const fmk::NTArgs& a, int epilogue, hipStream_t s);
// compute units the persistent GEMM grids may occupy (all of them minus fm_set_reserved_cus, a multiple of 8)
int fm_grid_cus();
// gemm_tn4.hip: the 4-wave / 512-register form of the TN job list on 256 x 384 tiles (the planner of fm_gemm_tn_multi in gemm.hip fills the list)
int fm_launch_tn4_multi(const fmk::TNMultiArgs& a, int grid, hipStream_t s);
