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

// ---- compact host-to-device batch format: what crosses PCIe is uint8 pixels, uint16 ids and bit-packed masks ----------------------

// (B, H, W, C) uint8 -> (B, C, H, W) f32 = ((x / 255) - mean[c]) / std[c]: torchvision's to_tensor + normalize (modality_transforms.py:
// 215-218), IEEE divisions in the same order, so the result is bit-identical to the loader's float pipeline
__global__ __launch_bounds__(256) void unpack_image_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int C,
                                                           float m0, float m1, float m2, float m3, float s0, float s1, float s2, float s3) {
    const float mean[4] = {m0, m1, m2, m3}, stdv[4] = {s0, s1, s2, s3};
    const size_t hw = (size_t)H * W, total = (size_t)B * hw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / hw, p = e % hw;
        for (int c = 0; c < C; ++c) {
            const float t = __fdiv_rn((float)src[e * C + c], 255.0f);
            dst[(b * C + c) * hw + p] = __fdiv_rn(t - mean[c], stdv[c]);
        }
    }
}

__global__ __launch_bounds__(256) void unpack_ids_kernel(const uint16_t* __restrict__ src, long long* __restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = (long long)src[e];
}

// bits: row b occupies ceil(L / 8) bytes, bit (i & 7) of byte (i >> 3) = mask[b][i]  (numpy.packbits(..., bitorder="little"))
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint8_t* __restrict__ bits, uint8_t* __restrict__ dst, int B, int L) {
    const int bytes = (L + 7) / 8;
    const size_t total = (size_t)B * L;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t b = e / L; const int i = (int)(e % L);
        dst[e] = (bits[b * bytes + (i >> 3)] >> (i & 7)) & 1;
    }
}

// decoder_attention_mask of an image-like modality from its target_mask: the target count at the first target position (masking.py:262-264)
__global__ __launch_bounds__(256) void dam_from_target_kernel(const uint8_t* __restrict__ target_mask, int* __restrict__ dam, int L) {
    __shared__ int first, cnt;
    if (threadIdx.x == 0) { first = L; cnt = 0; }
    __syncthreads();
    const uint8_t* t = target_mask + (size_t)blockIdx.x * L;
    int c = 0, f = L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        dam[(size_t)blockIdx.x * L + i] = 0;
        if (!t[i]) { ++c; f = min(f, i); }
    }
    atomicAdd(&cnt, c); atomicMin(&first, f);
    __syncthreads();
    if (threadIdx.x == 0) dam[(size_t)blockIdx.x * L + (first < L ? first : 0)] = cnt;
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

extern "C" int fm_unpack_image_u8(const void* src, void* dst, int B, int H, int W, int C, const float* mean, const float* stdv, void* stream) {
    FM_CHECK_ARG(src && dst && mean && stdv && B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, "fm_unpack_image_u8: bad argument (C <= 4; mean / std are HOST arrays)");
    float m[4] = {0, 0, 0, 0}, sd[4] = {1, 1, 1, 1};
    for (int c = 0; c < C; ++c) { m[c] = mean[c]; sd[c] = stdv[c]; }
    size_t blocks = ((size_t)B * H * W + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (float*)dst, B, H, W, C,
                       m[0], m[1], m[2], m[3], sd[0], sd[1], sd[2], sd[3]);
    FM_CHECK_LAUNCH("fm_unpack_image_u8");
    return 0;
}

extern "C" int fm_unpack_ids_u16(const void* src, int64_t* dst, int64_t n, void* stream) {
    FM_CHECK_ARG(src && dst && n > 0, "fm_unpack_ids_u16: bad argument");
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_ids_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (long long*)dst, (size_t)n);
    FM_CHECK_LAUNCH("fm_unpack_ids_u16");
    return 0;
}

extern "C" int fm_unpack_mask_bits(const void* bits, void* dst, int B, int L, void* stream) {
    FM_CHECK_ARG(bits && dst && B > 0 && L > 0, "fm_unpack_mask_bits: bad argument");
    size_t blocks = ((size_t)B * L + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)bits, (uint8_t*)dst, B, L);
    FM_CHECK_LAUNCH("fm_unpack_mask_bits");
    return 0;
}

extern "C" int fm_decoder_attention_from_target(const void* target_mask, int32_t* decoder_attention_mask, int B, int L, void* stream) {
    FM_CHECK_ARG(target_mask && decoder_attention_mask && B > 0 && L > 0, "fm_decoder_attention_from_target: bad argument");
    hipLaunchKernelGGL(dam_from_target_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)target_mask, decoder_attention_mask, L);
    FM_CHECK_LAUNCH("fm_decoder_attention_from_target");
    return 0;
}
