// Stand-alone timing of fm_gemm_nt / fm_gemm_tn at the 4M-B shapes (no Python, no torch: a GPU-box call costs seconds).
//   hipcc --offload-arch=gfx950 -O2 -I include tools/gemm_lab.cpp -L ml-4m_amd/fourm/_lib -lfourm_hip -o tools/bin/gemm_lab
//   tools/bin/gemm_lab nt <cfg,cfg,...>  |  tools/bin/gemm_lab tn  |  FOURM_NT_ABLATE=1 tools/bin/gemm_lab nt 10
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>
#include <string>
#include "fourm_hip.h"
extern "C" void fm_lab_set(int key, int value);      // experiment knobs of libfourm_hip.so (not part of the ABI header)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static void* dev_rand_bf16(size_t n, unsigned seed, float scale = 0.5f) {
    std::vector<uint16_t> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = f2bf(scale * (((s >> 8) & 0xffff) / 32768.0f - 1.0f)); }
    void* d; CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice)); return d;
}
static void* dev_zero(size_t bytes) { void* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes)); return d; }

template <typename F> static double time_us(F fn, int iters = 20, int warm = 3) {
    for (int i = 0; i < warm; ++i) fn();
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / iters;
}

struct NTCase { const char* name; int N, K, epi; };
// configurations >= 2000: gemm_nt4.hip in mode (cfg - 2000) % 10 where it applies (nt3 by shape elsewhere), lab flags (cfg - 2000) / 10
// configurations >= 1000: the lock-step large-tile kernel (gemm_nt3.hip) in mode cfg - 1000 over the default choice
static void set_cfg(int cfg) { fm_lab_set(4, 0); if (cfg >= 2000) { fm_lab_set(4, (cfg - 2000) % 10); fm_lab_set(2, 3); fm_lab_set(3, (cfg - 2000) / 10); fm_set_gemm_nt_config(9); } else if (cfg >= 1000) { fm_lab_set(2, (cfg - 1000) % 10); fm_lab_set(3, (cfg - 1000) / 10); fm_set_gemm_nt_config(9); } else { fm_lab_set(2, 0); fm_set_gemm_nt_config(cfg); } }

// the clocks of an idle box ramp up over hundreds of milliseconds: spin a GEMM before the first timing
static void warm_gpu(int ms_target) {
    const int R = 8192, N = 2048, K = 2048;
    void* W = dev_rand_bf16((size_t)N * K, 7), *X = dev_rand_bf16((size_t)R * K, 8), *out = dev_zero((size_t)R * N * 2);
    fm_gemm_nt_args a{};
    a.W = W; a.X = X; a.out = out; a.M = R; a.N = N; a.K = K; a.ldw = K; a.ldx = K; a.ldo = N; a.epilogue = FM_EPI_BF16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    CK(hipEventRecord(e0, 0));
    while (ms < ms_target) { for (int i = 0; i < 50; ++i) fm_gemm_nt(&a, 0); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(out));
}

int main(int argc, char** argv) {
    const int R = 256 * 128;
    std::string mode = argc > 1 ? argv[1] : "nt";
    if (!getenv("GEMM_LAB_NOWARM")) warm_gpu(600);      // (PMC runs: skip the warm-up so that dispatches can be told apart by order)
    if (mode == "nt") {
        std::vector<int> cfgs;
        if (argc > 2) { char* t = strtok(argv[2], ","); while (t) { cfgs.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        else cfgs = {265};
        NTCase cases[] = {{"qkv      N2304 K768 ", 2304, 768, FM_EPI_BF16}, {"proj/dX  N768  K768 ", 768, 768, FM_EPI_BF16},
                          {"dX fc2   N2048 K768 ", 2048, 768, FM_EPI_BF16}, {"kv       N1536 K768 ", 1536, 768, FM_EPI_BF16},
                          {"dX fc13  N768  K4096", 768, 4096, FM_EPI_BF16}, {"dX qkv   N768  K2304", 768, 2304, FM_EPI_BF16},
                          {"fc2+res  N768  K2048", 768, 2048, FM_EPI_RESIDUAL}, {"proj+res N768  K768 ", 768, 768, FM_EPI_RESIDUAL},
                          {"swiglu   N2x2048 K768", 2048, 768, FM_EPI_SWIGLU}, {"fc2dX+act' N2048 K768", 2048, 768, FM_EPI_SWIGLU_BWD}};
        double tot[32] = {0};
        const double per_step[] = {24, 61, 24, 12, 24, 24, 24, 38, 24, 0};       // launches per 4M-B train step (profiles/r01_v10_shape_table.txt)
        int ci = 0;
        for (auto& c : cases) {
            void* W = dev_rand_bf16((size_t)c.N * c.K, 1), *W2 = dev_rand_bf16((size_t)c.N * c.K, 2), *X = dev_rand_bf16((size_t)R * c.K, 3);
            const bool f32out = c.epi == FM_EPI_RESIDUAL, actbwd = c.epi == FM_EPI_SWIGLU_BWD;
            const int two = c.epi == FM_EPI_SWIGLU ? 2 : 1;
            void* out = dev_zero((size_t)R * c.N * (f32out ? 4 : actbwd ? 4 : 2));       // (dg | du) for the activation backward
            void* out2 = c.epi == FM_EPI_SWIGLU ? dev_zero((size_t)R * c.N * 2 * 2) : nullptr;
            void* res = f32out ? dev_zero((size_t)R * c.N * 4) : actbwd ? dev_rand_bf16((size_t)R * c.N * 2, 9, 2.0f) : nullptr;
            fm_gemm_nt_args a{};
            a.W = W; a.W2 = c.epi == FM_EPI_SWIGLU ? W2 : nullptr; a.X = X; a.out = out; a.out2 = out2; a.res = res;
            a.M = R; a.N = c.N; a.K = c.K; a.ldw = c.K; a.ldx = c.K; a.ldo = actbwd ? 2 * c.N : c.N; a.ldo2 = 2 * c.N; a.ldr = actbwd ? 2 * c.N : c.N; a.Hp = c.N; a.epilogue = c.epi;
            printf("%s |", c.name);
            std::vector<double> best(cfgs.size(), 1e30);
            {   // every configuration against the first one (bit patterns of the primary output)
                const size_t nbytes = (size_t)R * c.N * ((f32out || actbwd) ? 4 : 2);
                std::vector<uint16_t> ref(nbytes / 2), got(nbytes / 2);
                set_cfg(cfgs[0]); CK(hipMemset(out, 0, nbytes)); fm_gemm_nt(&a, 0); CK(hipDeviceSynchronize());
                CK(hipMemcpy(ref.data(), out, nbytes, hipMemcpyDeviceToHost));
                for (size_t k = 1; k < cfgs.size(); ++k) {
                    set_cfg(cfgs[k]); CK(hipMemset(out, 0, nbytes));
                    if (fm_gemm_nt(&a, 0) != 0) continue;
                    CK(hipDeviceSynchronize()); CK(hipMemcpy(got.data(), out, nbytes, hipMemcpyDeviceToHost));
                    size_t bad = 0; for (size_t e = 0; e < ref.size(); ++e) bad += ref[e] != got[e];
                    if (bad) printf(" [c%d: %zu of %zu halfwords differ from c%d]", cfgs[k], bad, ref.size(), cfgs[0]);
                }
            }
            for (int rep = 0; rep < 3; ++rep)                   // interleaved A/B: every configuration sees the same clock history
                for (size_t k = 0; k < cfgs.size(); ++k) {
                    set_cfg(cfgs[k]);
                    if (fm_gemm_nt(&a, 0) != 0) { printf(" cfg%d: %s", cfgs[k], fm_last_error()); continue; }
                    double us = time_us([&] { fm_gemm_nt(&a, 0); }, 20, 2);
                    if (us < best[k]) best[k] = us;
                }
            for (size_t k = 0; k < cfgs.size(); ++k) {
                printf("  c%d: %7.1f us %5.0f TF", cfgs[k], best[k], 2.0 * R * c.N * c.K * two / best[k] / 1e6);
                tot[k] += best[k] * per_step[ci];
            }
            printf("\n"); fflush(stdout);
            CK(hipFree(W)); CK(hipFree(W2)); CK(hipFree(X)); CK(hipFree(out)); if (out2) CK(hipFree(out2)); if (res) CK(hipFree(res));
            ++ci;
        }
        printf("sum over a 4M-B step (ms):");
        for (size_t k = 0; k < cfgs.size(); ++k) printf("  c%d: %.2f", cfgs[k], tot[k] / 1e3);
        printf("\n");
    } else if (mode == "ldpad") {
        // Do the row strides of the operands / outputs matter (L2 channel or DRAM bank conflicts)?  The same shapes with leading
        // dimensions K + pad (X, W) and N + pad (out): tools/bin/gemm_lab ldpad <cfg> <pad,pad,...>  (pad in elements, multiples of 64)
        const int cfg = argc > 2 ? atoi(argv[2]) : 1003;
        std::vector<int> pads = {0, 64, 128, 192};
        if (argc > 3) { pads.clear(); char* t = strtok(argv[3], ","); while (t) { pads.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        NTCase cases[] = {{"qkv      N2304 K768 ", 2304, 768, FM_EPI_BF16}, {"proj/dX  N768  K768 ", 768, 768, FM_EPI_BF16},
                          {"dX fc2   N2048 K768 ", 2048, 768, FM_EPI_BF16}, {"dX fc13  N768  K4096", 768, 4096, FM_EPI_BF16},
                          {"dX qkv   N768  K2304", 768, 2304, FM_EPI_BF16}, {"fc2      N768  K2048", 768, 2048, FM_EPI_BF16},
                          {"swiglu   N2x2048 K768", 2048, 768, FM_EPI_SWIGLU}};
        set_cfg(cfg);
        for (auto& c : cases) {
            printf("%s |", c.name);
            for (int variant = 0; variant < 3; ++variant) {      // 0: pad X, W and out; 1: pad X and W only; 2: pad out only
                for (int pad : pads) {
                    if (variant && pad == 0) continue;
                    const int pin = variant == 2 ? 0 : pad, pout = variant == 1 ? 0 : pad;
                    const int ldk = c.K + pin, ldn = c.N + pout;
                    void* W = dev_rand_bf16((size_t)c.N * ldk, 1), *W2 = dev_rand_bf16((size_t)c.N * ldk, 2), *X = dev_rand_bf16((size_t)R * ldk, 3);
                    void* out = dev_zero((size_t)R * ldn * 2);
                    void* out2 = c.epi == FM_EPI_SWIGLU ? dev_zero((size_t)R * (2 * c.N + pout) * 2) : nullptr;
                    fm_gemm_nt_args a{};
                    a.W = W; a.W2 = c.epi == FM_EPI_SWIGLU ? W2 : nullptr; a.X = X; a.out = out; a.out2 = out2;
                    a.M = R; a.N = c.N; a.K = c.K; a.ldw = ldk; a.ldx = ldk; a.ldo = ldn; a.ldo2 = 2 * c.N + pout; a.Hp = c.N; a.epilogue = c.epi;
                    if (fm_gemm_nt(&a, 0) != 0) { printf(" %s", fm_last_error()); continue; }
                    double best = 1e30;
                    for (int rep = 0; rep < 3; ++rep) { double us = time_us([&] { fm_gemm_nt(&a, 0); }, 20, 2); if (us < best) best = us; }
                    printf("  %s+%d: %6.1f us", variant == 0 ? "all" : variant == 1 ? "in" : "out", pad, best);
                    CK(hipFree(W)); CK(hipFree(W2)); CK(hipFree(X)); CK(hipFree(out)); if (out2) CK(hipFree(out2));
                }
            }
            printf("\n"); fflush(stdout);
        }
    } else if (mode == "dephase") {
        // staggered workgroup start (fm_lab_set 0/1): the store epilogue of one workgroup under the main loop of another
        std::vector<int> cfgs = {266, 267, 268};
        if (argc > 2) { cfgs.clear(); char* t = strtok(argv[2], ","); while (t) { cfgs.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        NTCase cases[] = {{"qkv      N2304 K768 ", 2304, 768, FM_EPI_BF16}, {"proj/dX  N768  K768 ", 768, 768, FM_EPI_BF16}, {"dX fc2   N2048 K768 ", 2048, 768, FM_EPI_BF16},
                          {"dX qkv   N768  K2304", 768, 2304, FM_EPI_BF16}, {"fc2+res  N768  K2048", 768, 2048, FM_EPI_RESIDUAL}, {"proj+res N768  K768 ", 768, 768, FM_EPI_RESIDUAL},
                          {"swiglu   N2x2048 K768", 2048, 768, FM_EPI_SWIGLU}};
        const int knobs[][2] = {{0, 0}, {2, 6}, {2, 12}, {2, 24}, {4, 4}, {4, 8}, {8, 2}, {8, 4}};
        for (auto& c : cases) {
            void* W = dev_rand_bf16((size_t)c.N * c.K, 1), *W2 = dev_rand_bf16((size_t)c.N * c.K, 2), *X = dev_rand_bf16((size_t)R * c.K, 3);
            const bool f32out = c.epi == FM_EPI_RESIDUAL, swi = c.epi == FM_EPI_SWIGLU;
            void* out = dev_zero((size_t)R * c.N * (f32out ? 4 : 2));
            void* out2 = swi ? dev_zero((size_t)R * c.N * 2 * 2) : nullptr;
            void* res = f32out ? dev_zero((size_t)R * c.N * 4) : nullptr;
            fm_gemm_nt_args a{};
            a.W = W; a.W2 = swi ? W2 : nullptr; a.X = X; a.out = out; a.out2 = out2; a.res = res; a.M = R; a.N = c.N; a.K = c.K; a.ldw = c.K; a.ldx = c.K; a.ldo = c.N; a.ldo2 = 2 * c.N;
            a.ldr = c.N; a.Hp = c.N; a.epilogue = c.epi;
            for (int cfg : cfgs) {
                set_cfg(cfg);
                printf("%s c%d |", c.name, cfg);
                double best[8]; for (auto& b : best) b = 1e30;
                for (int rep = 0; rep < 3; ++rep)
                    for (int k = 0; k < 8; ++k) {
                        fm_lab_set(0, knobs[k][0]); fm_lab_set(1, knobs[k][1]);
                        best[k] = std::min(best[k], time_us([&] { fm_gemm_nt(&a, 0); }, 20, 2));
                    }
                for (int k = 0; k < 8; ++k) printf("  g%d s%-2d %6.1f", knobs[k][0], knobs[k][1], best[k]);
                printf("  us\n"); fflush(stdout);
            }
            fm_lab_set(0, 0); fm_lab_set(1, 0);
            CK(hipFree(W)); CK(hipFree(W2)); CK(hipFree(X)); CK(hipFree(out)); if (out2) CK(hipFree(out2)); if (res) CK(hipFree(res));
        }
    } else if (mode == "pmc") {
        // counter runs (rocprofv3 --pmc, GEMM_LAB_NOWARM=1): exactly 5 launches per (case, configuration), case-major, so that the
        // dispatches can be told apart by their order; prints the order it ran
        std::vector<int> cfgs;
        if (argc > 2) { char* t = strtok(argv[2], ","); while (t) { cfgs.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        else cfgs = {1003};
        NTCase cases[] = {{"qkv_N2304_K768", 2304, 768, FM_EPI_BF16}, {"proj_N768_K768", 768, 768, FM_EPI_BF16}, {"dXfc2_N2048_K768", 2048, 768, FM_EPI_BF16},
                          {"dXfc13_N768_K4096", 768, 4096, FM_EPI_BF16}, {"swiglu_N2x2048_K768", 2048, 768, FM_EPI_SWIGLU}};
        for (auto& c : cases) {
            void* W = dev_rand_bf16((size_t)c.N * c.K, 1), *W2 = dev_rand_bf16((size_t)c.N * c.K, 2), *X = dev_rand_bf16((size_t)R * c.K, 3);
            void* out = dev_zero((size_t)R * c.N * 2);
            void* out2 = c.epi == FM_EPI_SWIGLU ? dev_zero((size_t)R * c.N * 2 * 2) : nullptr;
            fm_gemm_nt_args a{};
            a.W = W; a.W2 = c.epi == FM_EPI_SWIGLU ? W2 : nullptr; a.X = X; a.out = out; a.out2 = out2;
            a.M = R; a.N = c.N; a.K = c.K; a.ldw = c.K; a.ldx = c.K; a.ldo = c.N; a.ldo2 = 2 * c.N; a.Hp = c.N; a.epilogue = c.epi;
            for (int cfg : cfgs) {
                set_cfg(cfg);
                for (int i = 0; i < 5; ++i) if (fm_gemm_nt(&a, 0) != 0) { printf("cfg%d: %s\n", cfg, fm_last_error()); return 1; }
                CK(hipDeviceSynchronize());
                printf("%s c%d\n", c.name, cfg);
            }
            CK(hipFree(W)); CK(hipFree(W2)); CK(hipFree(X)); CK(hipFree(out)); if (out2) CK(hipFree(out2));
        }
    } else if (mode == "ksweep") {
        // T(K) = fixed + slope * K at fixed M, N: separates per-tile costs from the main-loop rate
        std::vector<int> cfgs = {265, 266, 267};
        if (argc > 2) { cfgs.clear(); char* t = strtok(argv[2], ","); while (t) { cfgs.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        const int Ks[] = {512, 768, 1536, 3072, 6144};
        for (int N : {2304, 768}) {
            void* W = dev_rand_bf16((size_t)N * 6144, 1), *X = dev_rand_bf16((size_t)R * 6144, 3), *out = dev_zero((size_t)R * N * 2);
            std::vector<std::vector<double>> t(cfgs.size(), std::vector<double>(5, 1e30));
            for (int rep = 0; rep < 3; ++rep)
                for (int ki = 0; ki < 5; ++ki)
                    for (size_t c = 0; c < cfgs.size(); ++c) {
                        fm_gemm_nt_args a{};
                        a.W = W; a.X = X; a.out = out; a.M = R; a.N = N; a.K = Ks[ki]; a.ldw = 6144; a.ldx = 6144; a.ldo = N; a.epilogue = FM_EPI_BF16;
                        set_cfg(cfgs[c]);
                        if (fm_gemm_nt(&a, 0) != 0) { printf("cfg%d: %s\n", cfgs[c], fm_last_error()); continue; }
                        double us = time_us([&] { fm_gemm_nt(&a, 0); }, 20, 2);
                        if (us < t[c][ki]) t[c][ki] = us;
                    }
            for (size_t c = 0; c < cfgs.size(); ++c) {
                printf("N=%d c%d:", N, cfgs[c]);
                for (int ki = 0; ki < 5; ++ki) printf("  K%d: %7.1f us", Ks[ki], t[c][ki]);
                const double slope = (t[c][4] - t[c][1]) / (Ks[4] - Ks[1]), fixed = t[c][1] - slope * Ks[1];
                printf(" | fixed %6.1f us, main loop %5.0f TF\n", fixed, 2.0 * R * N / slope / 1e6);
            }
            CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(out));
        }
    } else if (mode == "tn") {
        struct { const char* name; int N, K; double per_step; } cases[] = {{"dW fc1/3 N2048 K768 ", 2048, 768, 48}, {"dW proj  N768  K768 ", 768, 768, 50},
            {"dW qkv   N2304 K768 ", 2304, 768, 24}, {"dW fc2   N768  K2048", 768, 2048, 24}, {"dW kv    N1536 K768 ", 1536, 768, 12}};
        std::vector<int> cfgs = {1, 0};
        if (argc > 2) { cfgs.clear(); char* t = strtok(argv[2], ","); while (t) { cfgs.push_back(atoi(t)); t = strtok(nullptr, ","); } }
        double tot[8] = {0};
        for (auto& c : cases) {
            void* A = dev_rand_bf16((size_t)R * c.N, 4), *B = dev_rand_bf16((size_t)R * c.K, 5);
            void* out = dev_zero((size_t)c.N * c.K * 4);
            fm_gemm_tn_args a{};
            a.A = A; a.B = B; a.out = out; a.R = R; a.N = c.N; a.K = c.K; a.lda = c.N; a.ldb = c.K; a.ldo = c.K; a.force_tr = -1;
            printf("%s |", c.name);
            std::vector<double> best(cfgs.size(), 1e30);
            for (int rep = 0; rep < 3; ++rep)
                for (size_t k = 0; k < cfgs.size(); ++k) {
                    fm_set_gemm_tn_config(cfgs[k]);
                    if (fm_gemm_tn(&a, 0) != 0) { printf(" cfg%d: %s", cfgs[k], fm_last_error()); continue; }
                    double us = time_us([&] { fm_gemm_tn(&a, 0); }, 20, 2);
                    if (us < best[k]) best[k] = us;
                }
            for (size_t k = 0; k < cfgs.size(); ++k) {
                printf("  c%d: %7.1f us %5.0f TF", cfgs[k], best[k], 2.0 * R * c.N * c.K / best[k] / 1e6);
                tot[k] += best[k] * c.per_step;
            }
            printf("\n"); fflush(stdout);
            CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(out));
        }
        printf("sum over a 4M-B step (ms):");
        for (size_t k = 0; k < cfgs.size(); ++k) printf("  c%d: %.2f", cfgs[k], tot[k] / 1e3);
        printf("\n");
    } else if (mode == "cold") {
        // the same GEMM on operands that are resident in the Infinity Cache (one buffer set reused) vs streamed from HBM (a ring of
        // buffer sets larger than the 256 MB cache): how much of the in-situ / lab gap is operand residency
        struct { const char* name; int N, K; } cases[] = {{"N768 K768", 768, 768}, {"N2304 K768", 2304, 768}, {"N768 K4096", 768, 4096}};
        for (auto& c : cases) {
            const int SETS = 12;
            std::vector<void*> Xs, Os;
            void* W = dev_rand_bf16((size_t)c.N * c.K, 1);
            for (int i = 0; i < SETS; ++i) { Xs.push_back(dev_rand_bf16((size_t)R * c.K, 3 + i)); Os.push_back(dev_zero((size_t)R * c.N * 2)); }
            fm_gemm_nt_args a{};
            a.W = W; a.M = R; a.N = c.N; a.K = c.K; a.ldw = c.K; a.ldx = c.K; a.ldo = c.N; a.epilogue = FM_EPI_BF16;
            fm_set_gemm_nt_config(265);
            double warm = 1e30, cold = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                a.X = Xs[0]; a.out = Os[0];
                warm = std::min(warm, time_us([&] { fm_gemm_nt(&a, 0); }, 24, 2));
                int i = 0;
                cold = std::min(cold, time_us([&] { a.X = Xs[i % SETS]; a.out = Os[i % SETS]; ++i; fm_gemm_nt(&a, 0); }, 24, 2));
            }
            // third case: X was just WRITTEN by the previous kernel on the stream (a device-to-device copy), as in the train step
            double written = 1e30;
            {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                a.X = Xs[0]; a.out = Os[0];
                for (int rep = 0; rep < 12; ++rep) {
                    CK(hipMemcpyAsync(Xs[0], Xs[1 + rep % (SETS - 1)], (size_t)R * c.K * 2, hipMemcpyDeviceToDevice, 0));
                    CK(hipEventRecord(e0, 0));
                    fm_gemm_nt(&a, 0);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 2) written = std::min(written, (double)ms * 1e3);
                }
            }
            printf("%-12s cache-resident operands %7.1f us   streamed from HBM %7.1f us   X just written by the previous kernel %7.1f us (single launches)\n",
                   c.name, warm, cold, written);
            CK(hipFree(W)); for (auto p : Xs) CK(hipFree(p)); for (auto p : Os) CK(hipFree(p));
        }
    } else if (mode == "tnrate") {
        // main-loop rate of the job-list kernels on whole rounds of whole tiles (no cut of a partial round): one job N x K, R rows
        const int N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 12288, Rr = argc > 4 ? atoi(argv[4]) : 16384;
        void* A = dev_rand_bf16((size_t)Rr * N, 4), *B = dev_rand_bf16((size_t)Rr * K, 5);
        void* out = dev_zero((size_t)N * K * 4);
        fm_gemm_tn_job j{}; j.A = A; j.B = B; j.out = out; j.R = Rr; j.N = N; j.K = K; j.lda = N; j.ldb = K; j.ldo = K;
        for (int v : {0, 1, 0, 1}) {
            fm_lab_set(5, v);
            if (fm_gemm_tn_multi(&j, 1, 0) != 0) { printf("multi: %s\n", fm_last_error()); return 1; }
            const double us = time_us([&] { fm_gemm_tn_multi(&j, 1, 0); }, 5, 1);
            printf("N=%d K=%d R=%d tn4=%d: %8.1f us %5.0f TF\n", N, K, Rr, v, us, 2.0 * Rr * N * K / us / 1e6);
        }
        fm_lab_set(5, 0);
    } else if (mode == "tnmulti") {
        // all weight-gradient GEMMs of one 4M-B layer: one fm_gemm_tn launch each  vs  ONE fm_gemm_tn_multi launch
        struct Shape { int N, K; };
        std::vector<Shape> enc = {{2048, 768}, {2048, 768}, {768, 2048}, {768, 768}, {2304, 768}};
        std::vector<Shape> dec = enc; dec.push_back({768, 768}); dec.push_back({768, 768}); dec.push_back({1536, 768});
        const int Rm = argc > 2 ? atoi(argv[2]) : R;
        const int multi_cfg = argc > 3 ? atoi(argv[3]) : 1;      // fm_set_gemm_tn_config for the one-launch form (2 = 256 x 256 tiles)
        const int layers_per_list = argc > 4 ? atoi(argv[4]) : 1; // > 1: the dW GEMMs of several layers in ONE list (distinct operands)
        if (argc > 5) fm_lab_set(5, atoi(argv[5]));              // 0 (default): the list on gemm.hip's 8-wave kernel, 1: gemm_tn4.hip where it applies
        if (argc > 6) fm_lab_set(6, atoi(argv[6]));              // 1: no atomic epilogue (timing only; the "rel diff" is then meaningless)
        if (argc > 7) fm_lab_set(7, atoi(argv[7]));              // planner constants of the 256 x 384 kernel: segment cost c ...
        if (argc > 8) fm_lab_set(8, atoi(argv[8]));              // ... and band cost cb, in K-tiles
        std::vector<Shape> enc_n, dec_n;
        for (int l = 0; l < layers_per_list; ++l) { enc_n.insert(enc_n.end(), enc.begin(), enc.end()); dec_n.insert(dec_n.end(), dec.begin(), dec.end()); }
        if (layers_per_list > 1) printf("(%d layers per list: divide the times by %d)\n", layers_per_list, layers_per_list);
        for (auto* layer : {&enc_n, &dec_n}) {
            if ((int)layer->size() > FM_TN_MAX_JOBS) continue;
            std::vector<fm_gemm_tn_job> jobs; std::vector<fm_gemm_tn_args> args; std::vector<void*> outs, outs2;
            double flops = 0;
            for (auto& sh : *layer) {
                void* A = dev_rand_bf16((size_t)R * sh.N, 4 + (int)jobs.size()), *B = dev_rand_bf16((size_t)R * sh.K, 50 + (int)jobs.size());
                void* o1 = dev_zero((size_t)sh.N * sh.K * 4), *o2 = dev_zero((size_t)sh.N * sh.K * 4);
                fm_gemm_tn_job j{}; j.A = A; j.B = B; j.out = o2; j.R = Rm; j.N = sh.N; j.K = sh.K; j.lda = sh.N; j.ldb = sh.K; j.ldo = sh.K;
                fm_gemm_tn_args a{}; a.A = A; a.B = B; a.out = o1; a.R = Rm; a.N = sh.N; a.K = sh.K; a.lda = sh.N; a.ldb = sh.K; a.ldo = sh.K; a.force_tr = -1;
                jobs.push_back(j); args.push_back(a); outs.push_back(o1); outs2.push_back(o2);
                flops += 2.0 * Rm * sh.N * sh.K;
            }
            // correctness: one pass each into zeroed outputs
            for (auto& a : args) if (fm_gemm_tn(&a, 0) != 0) { printf("tn: %s\n", fm_last_error()); return 1; }
            fm_set_gemm_tn_config(multi_cfg);
            if (fm_gemm_tn_multi(jobs.data(), (int)jobs.size(), 0) != 0) { printf("multi: %s\n", fm_last_error()); return 1; }
            fm_set_gemm_tn_config(1);
            CK(hipDeviceSynchronize());
            double worst = 0;
            for (size_t i = 0; i < jobs.size(); ++i) {
                const size_t n = (size_t)(*layer)[i].N * (*layer)[i].K;
                std::vector<float> h1(n), h2(n);
                CK(hipMemcpy(h1.data(), outs[i], n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), outs2[i], n * 4, hipMemcpyDeviceToHost));
                double num = 0, den = 0;
                for (size_t e = 0; e < n; ++e) { num += (double)(h1[e] - h2[e]) * (h1[e] - h2[e]); den += (double)h1[e] * h1[e]; }
                worst = std::max(worst, std::sqrt(num / (den + 1e-30)));
            }
            double t1 = 1e30, t2 = 1e30;
            for (int rep = 0; rep < 4; ++rep) {
                t1 = std::min(t1, time_us([&] { for (auto& a : args) fm_gemm_tn(&a, 0); }, 10, 2));
                fm_set_gemm_tn_config(multi_cfg);
                t2 = std::min(t2, time_us([&] { fm_gemm_tn_multi(jobs.data(), (int)jobs.size(), 0); }, 10, 2));
                fm_set_gemm_tn_config(1);
            }
            printf("%s layer (%zu dW GEMMs, R=%d): separate %7.1f us %5.0f TF | one launch %7.1f us %5.0f TF | rel diff %.2e\n",
                   layer == &enc_n ? "encoder" : "decoder", jobs.size(), Rm, t1, flops / t1 / 1e6, t2, flops / t2 / 1e6, worst);
            for (size_t i = 0; i < jobs.size(); ++i) { CK(hipFree((void*)jobs[i].A)); CK(hipFree((void*)jobs[i].B)); CK(hipFree(outs[i])); CK(hipFree(outs2[i])); }
        }
    }
    return 0;
}
