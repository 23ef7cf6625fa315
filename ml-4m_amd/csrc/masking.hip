// Input / target masks of image-like modalities on the device (first slice).
//
// Replaces, per sample and modality, UnifiedMasking.image_mask (fourm/data/masking.py:237-266): argsort of a uniform noise vector,
// two gathers of budget step functions through that permutation, and the decoder_attention_mask entry that carries the number of
// targets at the first target position.  The loader runs it per sample in worker processes on the host; here one workgroup per
// (sample, modality) ranks the noise in LDS.  The noise is the caller's (torch.rand on the device), so the result is a pure function
// of (noise, budgets): bit-exact against the oracle, which is pinned to upstream's function under the same seed.
#include "common.h"
#include "fourm_hip.h"

namespace {

constexpr int MASK_MAX_L = 4096;

// ids[r] = index of the r-th smallest noise value (ties: lower index first, i.e. a stable argsort)
__global__ __launch_bounds__(256) void image_mask_kernel(const float* __restrict__ noise, const int* __restrict__ in_budget,
                                                         const int* __restrict__ tgt_budget, int L, uint8_t* __restrict__ input_mask,
                                                         uint8_t* __restrict__ target_mask, int* __restrict__ dam) {
    extern __shared__ float sh[];                     // noise (L) | ids (L, as int)
    float* nz = sh;
    int* ids = (int*)(sh + L);
    __shared__ int first_target, n_target;
    const int b = blockIdx.x;
    const float* src = noise + (size_t)b * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) nz[i] = src[i];
    if (threadIdx.x == 0) { first_target = L; n_target = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        const float v = nz[j];
        int r = 0;
        for (int k = 0; k < L; ++k) { const float w = nz[k]; r += (w < v) || (w == v && k < j); }
        ids[r] = j;
    }
    __syncthreads();
    const int kin = in_budget[b], kt = tgt_budget ? tgt_budget[b] : -1;
    int cnt = 0, first = L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int id = ids[i];
        const bool in_vis = id < kin;                                        // an input token
        const bool tgt_vis = kt < 0 ? !in_vis : (id >= kin && id < kin + kt);   // a target token (no target budget: every non-input)
        input_mask[(size_t)b * L + i] = in_vis ? 0 : 1;                      // masks: 1 = masked out
        target_mask[(size_t)b * L + i] = tgt_vis ? 0 : 1;
        dam[(size_t)b * L + i] = 0;
        if (tgt_vis) { ++cnt; first = min(first, i); }
    }
    atomicAdd(&n_target, cnt);
    atomicMin(&first_target, first);
    __syncthreads();
    if (threadIdx.x == 0) {
        // upstream: argmin(target_mask + arange * 1e-6): the first unmasked target position, or position 0 when there is none
        const int pos = first_target < L ? first_target : 0;
        dam[(size_t)b * L + pos] = n_target;
    }
}

}  // namespace

extern "C" int fm_image_mask(const void* noise, const int32_t* input_budget, const int32_t* target_budget, int B, int L, void* input_mask,
                             void* target_mask, int32_t* decoder_attention_mask, void* stream) {
    FM_CHECK_ARG(noise && input_budget && input_mask && target_mask && decoder_attention_mask, "fm_image_mask: null pointer");
    FM_CHECK_ARG(B > 0 && L > 0 && L <= MASK_MAX_L, "fm_image_mask: 1 <= L <= %d", MASK_MAX_L);
    hipLaunchKernelGGL(image_mask_kernel, dim3(B), dim3(256), (size_t)L * 8, (hipStream_t)stream, (const float*)noise, input_budget, target_budget, L,
                       (uint8_t*)input_mask, (uint8_t*)target_mask, decoder_attention_mask);
    FM_CHECK_LAUNCH("fm_image_mask");
    return 0;
}
