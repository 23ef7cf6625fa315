// Hardware fact probe for gfx950: prints the lane<->element maps the kernels rely on.
//   1. ds_read_b64_tr_b16: which LDS element each lane receives
//   2. v_mfma_f32_32x32x16_bf16: accumulator row/column of every (lane, register)
//   3. global_load_lds_dwordx4: where lane i's 16 bytes land
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_gfx950.hip -o tools/bin/probe_gfx950
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__global__ void tr_probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;     // element index = row*64 + col
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15;
    // every 16-lane group g reads the 4x16 block at rows 4g.., columns 0..15 of a [64][64] tile
    const int row = 4 * (lane >> 4) + (i >> 2), col = (i & 3) * 4;
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

__global__ void mfma_probe(float* out) {
    const int lane = threadIdx.x;
    // A[i][k] = (i == k'), B[k][j] chosen so that D[i][j] = 100*i + j  (uses k = i % 16 only)
    bf16x8_t a, b;
    f32x16_t acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int pass = 0; pass < 2; ++pass) {   // rows 0-15 then rows 16-31 through the 16-wide K
        for (int e = 0; e < 8; ++e) {
            const int k = (lane >> 5) * 8 + e;           // assumed: lane holds k = 8*(lane>>5) + e
            const int i = lane & 31, j = lane & 31;
            a[e] = (__bf16)((i == k + 16 * pass) ? 1.0f : 0.0f);
            b[e] = (__bf16)(float)(((k + 16 * pass) % 8) * 32 + j);      // exactly representable (< 256)
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}

__global__ void glds_probe(const uint32_t* src, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x * 4),
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    uint16_t* d16; float* df; uint32_t *ds, *du;
    hipMalloc(&d16, 256 * 2); hipMalloc(&df, 64 * 16 * 4); hipMalloc(&ds, 1024); hipMalloc(&du, 1024);
    uint16_t h16[256]; float hf[1024]; uint32_t hs[256], hu[256];
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, d16);
    hipMemcpy(h16, d16, sizeof(h16), hipMemcpyDeviceToHost);
    printf("== ds_read_b64_tr_b16: lane -> (row,col) of the 4 received elements; expected lane i of a group: rows 4g+0..3, col i\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            printf(" (%d,%d)", h16[l * 4 + j] / 64, h16[l * 4 + j] % 64);
            if (h16[l * 4 + j] / 64 != 4 * (l >> 4) + j || h16[l * 4 + j] % 64 != (l & 15)) ok = 0;
        }
        printf("\n");
    }
    printf("TR_SEMANTICS_AS_ASSUMED=%d\n", ok);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, df);
    hipMemcpy(hf, df, sizeof(hf), hipMemcpyDeviceToHost);
    // D[i][j] = sum_k A[i][k] B[k][j] = B[i][j] = (i%8)*32 + j ; expected at lane = j + 32*hi, reg r: i = (r&3)+8*(r>>2)+4*hi
    ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
            if (hf[l * 16 + r] != (float)((i % 8) * 32 + j)) ok = 0;
        }
    printf("== mfma_32x32x16 accumulator map as assumed: %d   (lane0 regs:", ok);
    for (int r = 0; r < 16; ++r) printf(" %.0f", hf[r]);
    printf(" ; lane33 regs:");
    for (int r = 0; r < 16; ++r) printf(" %.0f", hf[33 * 16 + r]);
    printf(")\nMFMA_MAP_AS_ASSUMED=%d\n", ok);
    for (int i = 0; i < 256; ++i) hs[i] = i;
    hipMemcpy(ds, hs, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(glds_probe, dim3(1), dim3(64), 0, 0, ds, du);
    hipMemcpy(hu, du, 1024, hipMemcpyDeviceToHost);
    ok = 1;
    for (int i = 0; i < 256; ++i) if (hu[i] != (uint32_t)i) ok = 0;
    printf("GLDS_LANE_LINEAR=%d (lds[0..7] = %u %u %u %u %u %u %u %u)\n", ok, hu[0], hu[1], hu[2], hu[3], hu[4], hu[5], hu[6], hu[7]);
    return 0;
}
