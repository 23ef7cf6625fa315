// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the 4M hot path.
// Wavefront = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error reporting (host side) -----------------------------------------------------------
extern "C" const char* fm_last_error(void);
void fm_set_error(const char* fmt, ...);
#define FM_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            fm_set_error(__VA_ARGS__);          \
            return -1;                          \
        }                                       \
    } while (0)
#define FM_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            fm_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return -2;                                                               \
        }                                                                            \
    } while (0)

// ---- bf16 <-> f32 ---------------------------------------------------------------------------
// float -> bf16 uses the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved —
// the same rounding as torch's .to(bfloat16)); bf16 -> float is a 16-bit shift.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bfround(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- wave / block reductions -----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ int wave_scan_incl(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// XCD-aware remap of a linear workgroup id: consecutive remapped ids run on the same XCD (and
// therefore share one L2).  Bijective for every grid size (guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int xcd = bid % NX, q = nwg / NX, r = nwg % NX;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / NX;
}

// Read a "column fragment" from a row-major bf16 LDS tile: 8 values T[rA+0..3][c], T[rB+0..3][c]
// for this lane's column.  Two implementations, selected at compile time:
//   TR = true : two ds_read_b64_tr_b16 (hardware 4x16 transpose read); the 16 lanes of a group
//               address one 4-row x 16-column block each (lane i -> row i>>2, columns (i&3)*4..+3)
//   TR = false: eight 2-byte LDS reads (slow reference path, layout-agnostic)
// `addr_of(row, col)` returns a (generic) pointer to element (row, col) inside the (swizzled) LDS
// tile; 4 consecutive columns starting at a multiple of 4 must be contiguous.
template <bool TR, typename AddrFn>
__device__ __forceinline__ bf16x8_t lds_col_frag(AddrFn addr_of, int rA, int rB, int col_base32) {
    const int lane = threadIdx.x & 63;
    union { bf16x8_t v; s16x4_t h[2]; uint16_t e[8]; } u;
    if constexpr (TR) {
        const int i = lane & 15;
        const int c = col_base32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(addr_of(rA + (i >> 2), c)));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(addr_of(rB + (i >> 2), c)));
    } else {
        const int c = col_base32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u.e[j] = *(const uint16_t*)addr_of(rA + j, c);
            u.e[4 + j] = *(const uint16_t*)addr_of(rB + j, c);
        }
    }
    return u.v;
}

// The same transpose-read fragment issued through inline asm: hipcc (ROCm 7.2) orders a ds_read_tr
// *builtin* behind every outstanding LDS-DMA with a full `s_waitcnt vmcnt(0)`, which drains a
// multi-stage prefetch ring.  The asm form is invisible to that pass; the caller owns the waits
// (`wait_lgkmcnt<N>()` + `__builtin_amdgcn_sched_barrier(0)` before the first consumer, guide 5.7 / rule 18).
template <typename AddrFn>
__device__ __forceinline__ bf16x8_t lds_col_frag_tr_async(AddrFn addr_of, int rA, int rB, int col_base32) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15;
    const int c = col_base32 + ((lane >> 4) & 1) * 16 + (i & 3) * 4;
    union { bf16x8_t v; s16x4_t h[2]; } u;
    const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(addr_of(rA + (i >> 2), c));
    const uint32_t a1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)(addr_of(rB + (i >> 2), c));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[0]) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[1]) : "v"(a1));
    return u.v;
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// erf to ~1.5e-7 absolute (Abramowitz & Stegun 7.1.26): one v_exp + one v_rcp + a degree-5 polynomial instead of
// libdevice's erff (several times the VALU work); exact-GELU epilogues and their backward use it.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return x < 0.f ? -r : r;
}

// ---- 16-byte accesses for MFMA 32x32 accumulator rows (used by the GEMM epilogues and attention) ----------------
// bf16 epilogue store of two adjacent 8-feature groups of one row.  After the 32x32 MFMA chain lane l holds
// features [8g+4h, 8g+4h+4) of its row (h = l >> 5; lanes l and l+32 share the row), packed in `pg` for group g
// and `pg1` for group g+1.  Wide form: two v_permlane32_swap per pair hand each half-wave 16 contiguous bytes
// (lower half: group g, upper half: group g+1) -> ONE 16-byte store instead of two 8-byte ones (the store tail is
// issue bound, not byte bound).  `col` = first feature of group g; features >= N are not written.
// The read-side mirror of store_bf16_groups: one 16-byte load per half-wave (lower half: group g, upper half:
// group g+1), two swaps, and every lane has its 4 features of both groups.  Features >= N read as zero.
// second half of the wide load_bf16_groups: a 16-byte value loaded by each half-wave -> this lane's 4 features of both groups
__device__ __forceinline__ void split_bf16_groups(uint4 t, uint2& pg, uint2& pg1) {
    const auto x = __builtin_amdgcn_permlane32_swap(t.x, t.z, false, false);
    const auto y = __builtin_amdgcn_permlane32_swap(t.y, t.w, false, false);
    pg = make_uint2(x[0], y[0]); pg1 = make_uint2(x[1], y[1]);
}

__device__ __forceinline__ void load_bf16_groups(const bf16_t* row, int col, int fhi, int N, bool wide, uint2& pg, uint2& pg1) {
    if (wide) {
        const int c = col + 8 * fhi;
        split_bf16_groups(c < N ? *(const uint4*)(row + c) : make_uint4(0u, 0u, 0u, 0u), pg, pg1);
    } else {
        const int c = col + 4 * fhi;
        pg = c < N ? *(const uint2*)(row + c) : make_uint2(0u, 0u);
        pg1 = c + 8 < N ? *(const uint2*)(row + c + 8) : make_uint2(0u, 0u);
    }
}

__device__ __forceinline__ void unpack_bf4(uint2 p, float (&v)[4]) {
    v[0] = bf2f((bf16_t)(p.x & 0xffff)); v[1] = bf2f((bf16_t)(p.x >> 16)); v[2] = bf2f((bf16_t)(p.y & 0xffff)); v[3] = bf2f((bf16_t)(p.y >> 16));
}

__device__ __forceinline__ void store_bf16_groups(bf16_t* row, int col, uint2 pg, uint2 pg1, int fhi, int N, bool wide) {
    if (wide) {
        const auto x = __builtin_amdgcn_permlane32_swap(pg.x, pg1.x, false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pg.y, pg1.y, false, false);
        const int c = col + 8 * fhi;
        if (c < N) *(uint4*)(row + c) = make_uint4(x[0], y[0], x[1], y[1]);
    } else {
        const int c = col + 4 * fhi;
        if (c < N) *(uint2*)(row + c) = pg;
        if (c + 8 < N) *(uint2*)(row + c + 8) = pg1;
    }
}

// wait until at most N of this wave's vector-memory operations (here: LDS-DMA pieces) are outstanding
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


