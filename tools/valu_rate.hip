// Developer probe: issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 on gfx950 at 1, 2
// and 4 waves per SIMD (the resample kernel runs at 4).  Decides whether "packing" two fp32
// lanes into one instruction buys VALU throughput or only instruction count.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/_build/valu_rate && tools/_build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define ITER 2048
#define NACC 12

// shader-clock and 100 MHz wall-clock stamps of workgroup 0: the clock the chip holds during the loop
#define CLK_BEGIN if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64(); clk[1] = wall_clock64(); }
#define CLK_END if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = clock64(); clk[3] = wall_clock64(); }

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b, long long* clk) {
    CLK_BEGIN
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    CLK_END
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b, long long* clk) {
    CLK_BEGIN
    v2f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f + i};
    const v2f aa = (v2f){a, a * 1.0001f}, bb = (v2f){b, b * 0.999f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], aa, bb);
    }
    CLK_END
    v2f s = (v2f){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

__global__ __launch_bounds__(256) void k_pkmul(float* out, float a, float b, long long* clk) {
    CLK_BEGIN
    v2f acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f + i};
    const v2f aa = (v2f){a, a * 1.0001f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = acc[i] * aa;
    }
    CLK_END
    v2f s = (v2f){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// LDS: 6 x ds_read_b64 per iteration from consecutive lanes (the resample tap-row pattern)
__global__ __launch_bounds__(256) void k_ldsrow(float* out, int pitch, long long* clk) {
    CLK_BEGIN
    __shared__ float2 tile[4096];
    for (int e = threadIdx.x; e < 4096; e += 256) tile[e] = make_float2(e, -e);
    __syncthreads();
    v2f acc = (v2f){0.f, 0.f};
    unsigned base = (threadIdx.x & 63) + (threadIdx.x >> 6) * pitch;
    for (int it = 0; it < ITER; ++it) {
        const unsigned a = (unsigned)(size_t)(tile + ((base + it * pitch) & 2047));
        v2f r0, r1, r2, r3, r4, r5;
        asm volatile("ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:8\n\tds_read_b64 %2, %6 offset:16\n\t"
                     "ds_read_b64 %3, %6 offset:24\n\tds_read_b64 %4, %6 offset:32\n\tds_read_b64 %5, %6 offset:40\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5) : "v"(a) : "memory");
        acc += r0 + r1 + r2 + r3 + r4 + r5;
    }
    CLK_END
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

static long long* g_clk;
static double g_ghz;                      // shader clock of the last timed kernel (workgroup 0)

template <typename K, typename... A>
static double time_kernel(K k, int blocks, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, args...);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[4];
    hipMemcpy(h, g_clk, sizeof(h), hipMemcpyDeviceToHost);
    g_ghz = (double)(h[2] - h[0]) / ((double)(h[3] - h[1]) * 10.0);      // clocks per ns
    return ms / 5 * 1e-3;
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 16);
    hipMalloc(&g_clk, sizeof(long long) * 4);
    for (int wps : {1, 2, 4}) {            // waves per SIMD: one 256-thread block = 1 wave on each SIMD
        const int blocks = 256 * wps;
        const double n = (double)ITER * NACC * wps;      // instructions per SIMD
        double t1 = time_kernel(k_fma, blocks, out, 1.0001f, 1e-3f, g_clk);
        const double g1 = g_ghz;
        double t2 = time_kernel(k_pkfma, blocks, out, 1.0001f, 1e-3f, g_clk);
        const double g2 = g_ghz;
        double t3 = time_kernel(k_pkmul, blocks, out, 1.0001f, 1e-3f, g_clk);
        const double g3 = g_ghz;
        double t4 = time_kernel(k_ldsrow, blocks, out, 70, g_clk);
        const double g4 = g_ghz;
        printf("waves/SIMD %d: v_fma_f32 %.2f ns/instr/SIMD at %.2f GHz = %.2f cycles; v_pk_fma_f32 %.2f ns at %.2f GHz = %.2f; "
               "v_pk_mul_f32 %.2f ns at %.2f GHz = %.2f; 6 x ds_read_b64 row: %.2f ns per wave-row per CU at %.2f GHz = %.1f cycles\n",
               wps, t1 * 1e9 / n, g1, t1 * 1e9 / n * g1, t2 * 1e9 / n, g2, t2 * 1e9 / n * g2, t3 * 1e9 / n, g3,
               t3 * 1e9 / n * g3, t4 * 1e9 / ((double)ITER * 4 * wps), g4, t4 * 1e9 / ((double)ITER * 4 * wps) * g4);
    }
    hipFree(out);
    return 0;
}
