// Developer probe (not part of the library): where do workgroups land (XCC_ID), and what does a
// flag round trip between two workgroups cost with L2-scope (sc0) vs device-scope (sc1) accesses?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ inline unsigned ld_sc0(const unsigned* base, unsigned byte_off) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 1 << 20, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 1);
}
__device__ inline unsigned ld_sc1(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_where(unsigned* out) {
    extern __shared__ char big[];
    if (threadIdx.x == 0) { big[0] = 1; out[blockIdx.x] = xcc_id(); }
}

// mode 0: sc0 loads + sc0 stores; 1: sc1 atomics; 2: nontemporal loads + sc0 stores;
// 3: buffer_inv sc0 then a plain load + sc0 stores; 4: buffer_inv sc1 + plain load, sc1 store.  wgA / wgB ping-pong N times.
__global__ void k_ping(unsigned* flags, int wgA, int wgB, int N, int mode, long long* ticks, unsigned* fail) {
    extern __shared__ char big[];
    if (threadIdx.x != 0) return;
    big[0] = 1;
    const bool isA = (int)blockIdx.x == wgA, isB = (int)blockIdx.x == wgB;
    if (!isA && !isB) return;
    unsigned* mine = flags + (isA ? 0 : 64);
    unsigned* theirs = flags + (isA ? 64 : 0);
    long long t0 = wall_clock64();
    for (int i = 1; i <= N; ++i) {
        if (isA) {
            if (mode != 1 && mode != 4) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int spins = 0;
        while (true) {
            unsigned v;
            if (mode == 0) v = ld_sc0(flags, isA ? 256 : 0);
            else if (mode == 1) v = ld_sc1(theirs);
            else if (mode == 2) v = __builtin_nontemporal_load(theirs);
            else if (mode == 3) { asm volatile("buffer_inv sc0" ::: "memory"); v = *(volatile unsigned*)theirs; }
            else { asm volatile("buffer_inv sc1" ::: "memory"); v = *theirs; asm volatile("" ::: "memory"); }
            if (v >= (unsigned)i) break;
            if (++spins > (1 << 20)) { atomicAdd(fail, 1u); return; }
        }
        if (isB) {
            if (mode != 1 && mode != 4) __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (isA) ticks[0] = wall_clock64() - t0;
}

int main() {
    const int NWG = 256;
    unsigned *d_out, *d_flags, *d_fail;
    long long* d_ticks;
    CK(hipMalloc(&d_out, NWG * 4));
    CK(hipMalloc(&d_flags, 1 << 20));
    CK(hipMalloc(&d_fail, 4));
    CK(hipMalloc(&d_ticks, 8));
    const int lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_ping, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    std::vector<unsigned> h(NWG);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_where, dim3(NWG), dim3(256), lds, 0, d_out);
        CK(hipMemcpy(h.data(), d_out, NWG * 4, hipMemcpyDeviceToHost));
        int hist[16] = {0}, rr = 0;
        for (int i = 0; i < NWG; ++i) { hist[h[i]]++; rr += (h[i] == (unsigned)(i % 8)); }
        printf("xcc histogram:");
        for (int x = 0; x < 16; ++x) if (hist[x]) printf(" [%d]=%d", x, hist[x]);
        printf("   blockIdx %% 8 == xcc for %d of %d\n", rr, NWG);
    }
    printf("first 24 xcc:");
    for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
    printf("\n");
    // pairs: same XCD (0, 8), different XCD (0, 1)
    const int N = 2000;
    for (int mode = 0; mode < 5; ++mode)
        for (int pair = 0; pair < 2; ++pair) {
            int a = 0, b = pair == 0 ? 8 : 1;
            if (mode != 1 && mode != 4 && pair == 1) continue;    // L2-scope modes: same XCD only
            CK(hipMemset(d_flags, 0, 1 << 20));
            CK(hipMemset(d_fail, 0, 4));
            hipLaunchKernelGGL(k_ping, dim3(NWG), dim3(256), lds, 0, d_flags, a, b, N, mode, d_ticks, d_fail);
            CK(hipDeviceSynchronize());
            long long t; unsigned f;
            CK(hipMemcpy(&t, d_ticks, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&f, d_fail, 4, hipMemcpyDeviceToHost));
            printf("mode %s, wg %d <-> wg %d (xcc %u, %u): %.3f us per round trip pair (fail %u)\n",
                   mode == 0 ? "sc0 load" : mode == 1 ? "sc1 atomics" : mode == 2 ? "nt load" : mode == 3 ? "inv sc0 + volatile load" : "inv sc1 + load", a, b, h[a], h[b], t * 0.01 / N, f);
        }
    return 0;
}
