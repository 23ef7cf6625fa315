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
    int lab;                                              // experiment flags of gemm_nt3 (fm_lab_set 3): 1 no wait for the DMA, 2 no DMA, 4 no stores, 16 all DMA pieces in one k-step, 256 take the residual epilogue
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }


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
