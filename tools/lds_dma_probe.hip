// Developer probe (round 4): what does ISSUING a global_load_lds_dwordx4 cost a wave on gfx950, and does a
// write of M0 (the LDS base of the next LDS-DMA instruction) between two of them wait for the one in flight?
// k_coadd_fused_dma spends ~1 000 shader clocks per LDS-DMA instruction in its issue phase (phase clocks:
// 667 k cycles per wave for ~640 instructions) although the loop around them is twenty vector instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o tools/_build/lds_dma_probe && tools/_build/lds_dma_probe
// Variants, each timed around the ISSUE only (s_memtime before / after, no s_waitcnt vmcnt in between):
//   same   - NI instructions with one M0 value and immediate offsets 0, 1024, 2048, 3072 (4 KB of LDS)
//   move   - NI instructions, M0 rewritten before each one
//   vgpr   - the same traffic as plain global_load_dwordx4 into registers (no LDS-DMA)
//   box    - LDS-DMA with the fused kernel's addresses: 16-byte pieces along box rows of 19 quads, rows 12 KB apart
// for 1 wave alone, and for 8 waves x 2 workgroups per CU on every CU (the fused kernel's occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NI 4
#define REPS 64
typedef const void __attribute__((address_space(1)))* gptr;

template <int MODE>
__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ src, size_t span, long long* __restrict__ out,
                                               float* __restrict__ sink) {
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = reinterpret_cast<char*>(lds) + (size_t)wave * NI * 1024;
    long long issue = 0, total = 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t off = ((size_t)blockIdx.x * 8 + wave) * 65536 % span;
    for (int r = 0; r < REPS; ++r) {
        const float* p = src + (off + (size_t)r * 16384 * 7) % span + lane * 4;
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
            // (the immediate offset adds to the global address too: taken off the pointer)
            __builtin_amdgcn_global_load_lds((gptr)(p), (__attribute__((address_space(3))) void*)base, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(p + 4096 - 256), (__attribute__((address_space(3))) void*)base, 16, 1024, 0);
            __builtin_amdgcn_global_load_lds((gptr)(p + 2 * 4096 - 512), (__attribute__((address_space(3))) void*)base, 16, 2048, 0);
            __builtin_amdgcn_global_load_lds((gptr)(p + 3 * 4096 - 768), (__attribute__((address_space(3))) void*)base, 16, 3072, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
                __builtin_amdgcn_global_load_lds((gptr)(p + i * 4096), (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
        } else if (MODE == 3) {
            // the fused kernel's pattern: a piece = quad (lane % 19) of box row (lane / 19), rows 3072 floats apart
            const float* q = src + (off + (size_t)r * 16384 * 7) % span + (size_t)(lane / 19) * 3072 + (lane % 19) * 4;
#pragma unroll
            for (int i = 0; i < NI; ++i)
                __builtin_amdgcn_global_load_lds((gptr)(q + (size_t)i * 4 * 3072), (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(p + i * 4096);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        const long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_amdgcn_s_memtime();
        issue += t1 - t0;
        total += t2 - t0;
        __syncthreads();
        if (MODE != 2) {
            const float4 v = reinterpret_cast<const float4*>(base)[lane];
            acc.x += v.x; acc.y += v.y;
        }
    }
    if (lane == 0) {
        out[((size_t)blockIdx.x * 8 + wave) * 2] = issue;
        out[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = total;
    }
    if (acc.x == 12345.f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const size_t span = (size_t)256 << 20;           // floats: 1 GiB
    float *src = nullptr, *sink = nullptr;
    long long* out = nullptr;
    HIPCHECK(hipMalloc(&src, span * 4 + (1 << 20)));
    HIPCHECK(hipMemset(src, 0, span * 4 + (1 << 20)));
    HIPCHECK(hipMalloc(&sink, 64));
    const int maxwg = 512;
    HIPCHECK(hipMalloc(&out, sizeof(long long) * 2 * 8 * maxwg));
    const char* names[4] = {"same M0 + offsets", "M0 rewritten", "registers", "box rows (19 quads)"};
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int wgs = cfg == 0 ? 1 : 512, threads = cfg == 0 ? 64 : 512;
        printf("%s\n", cfg == 0 ? "one wave alone:" : "512 workgroups x 8 waves (two per CU):");
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                const size_t sh = 80 * 1024;
                if (mode == 0) { HIPCHECK(hipFuncSetAttribute((const void*)k_probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); hipLaunchKernelGGL(k_probe<0>, dim3(wgs), dim3(threads), sh, 0, src, span, out, sink); }
                if (mode == 1) { HIPCHECK(hipFuncSetAttribute((const void*)k_probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); hipLaunchKernelGGL(k_probe<1>, dim3(wgs), dim3(threads), sh, 0, src, span, out, sink); }
                if (mode == 2) { HIPCHECK(hipFuncSetAttribute((const void*)k_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); hipLaunchKernelGGL(k_probe<2>, dim3(wgs), dim3(threads), sh, 0, src, span, out, sink); }
                if (mode == 3) { HIPCHECK(hipFuncSetAttribute((const void*)k_probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); hipLaunchKernelGGL(k_probe<3>, dim3(wgs), dim3(threads), sh, 0, src, span, out, sink); }
                HIPCHECK(hipDeviceSynchronize());
            }
            const int nw = wgs * (threads / 64);
            std::vector<long long> h((size_t)2 * nw);
            HIPCHECK(hipMemcpy(h.data(), out, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
            double si = 0, st = 0;
            for (int w = 0; w < nw; ++w) { si += h[2 * w]; st += h[2 * w + 1]; }
            printf("  %-20s issue %8.1f clocks per instruction, issue + wait %8.1f per group of %d\n", names[mode],
                   si / nw / REPS / NI, st / nw / REPS, NI);
        }
    }
    return 0;
}
