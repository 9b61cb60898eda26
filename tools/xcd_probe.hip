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

int bulk_main();
int pitch_main();
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
    bulk_main();
    pitch_main();
    return 0;
}

// ---- bulk: how long does a workgroup need to pull a 64 KB tile another workgroup just wrote? ----
// writers: workgroups with (blockIdx % 8 == 0, slot < nw) write tile[slot]; readers: the same
// workgroups read tile[(slot + 1) % nw] after a device-wide flag count.  mode 0: plain stores +
// nontemporal loads (one XCD's L2); mode 1: sc1 stores + sc1 loads (device scope).
__global__ void k_bulk(double* tiles, unsigned* flag, int nw, int mode, int allx, long long* ticks) {
    extern __shared__ char big[];
    big[0] = 1;
    const int xcd = blockIdx.x % 8;
    if (!allx && xcd != 0) return;
    const int slot = blockIdx.x / 8;
    if (slot >= nw) return;
    const int tid = threadIdx.x;
    tiles += (size_t)xcd * 32 * 8192;
    flag += xcd * 32;
    double* mine = tiles + (size_t)slot * 8192;
    const double* theirs = tiles + (size_t)((slot + 1) % nw) * 8192;
    for (int i = 0; i < 32; ++i) {
        const double v = slot * 100000.0 + tid + 256.0 * i;
        if (mode == 0) __hip_atomic_store(&mine[tid + 256 * i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(&mine[tid + 256 * i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nw && ++spins < (1 << 22)) {}
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc = 0.0;
    double r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (mode == 0) r[i] = __builtin_nontemporal_load(&theirs[tid + 256 * i]);
        else r[i] = __hip_atomic_load(&theirs[tid + 256 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += r[i] - (((slot + 1) % nw) * 100000.0 + tid + 256.0 * i);
    __syncthreads();
    const long long t1 = wall_clock64();
    if (tid == 0) { ticks[(xcd * 32 + slot) * 2] = t1 - t0; ticks[(xcd * 32 + slot) * 2 + 1] = (long long)(acc != 0.0); }
    if (acc == 12345.0) tiles[0] = acc;
}

int bulk_main() {
    const int NWG = 256, lds = 100 * 1024;
    double* d_tiles; unsigned* d_flag; long long* d_ticks;
    CK(hipMalloc(&d_tiles, 8 * 32 * 8192 * 8));
    CK(hipMalloc(&d_flag, 4 * 256));
    CK(hipMalloc(&d_ticks, 512 * 8));
    CK(hipFuncSetAttribute((const void*)k_bulk, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int allx = 0; allx < 2; ++allx)
        for (int nw : {2, 26, 32})
            for (int mode = 0; mode < 2; ++mode)
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(d_flag, 0, 4 * 256));
                    CK(hipMemset(d_tiles, 0, 8 * 32 * 8192 * 8));
                    CK(hipMemset(d_ticks, 0, 512 * 8));
                    hipLaunchKernelGGL(k_bulk, dim3(NWG), dim3(256), lds, 0, d_tiles, d_flag, nw, mode, allx, d_ticks);
                    CK(hipDeviceSynchronize());
                    long long h[512];
                    CK(hipMemcpy(h, d_ticks, sizeof(h), hipMemcpyDeviceToHost));
                    double mean = 0; long long mx = 0, bad = 0; int cnt = 0;
                    for (int x = 0; x < (allx ? 8 : 1); ++x)
                        for (int i = 0; i < nw; ++i) {
                            const long long t = h[(x * 32 + i) * 2];
                            mean += t; mx = t > mx ? t : mx; bad += h[(x * 32 + i) * 2 + 1]; ++cnt;
                        }
                    printf("bulk 64 KB tile, %2d workgroups on %s, %s: mean %.2f us, max %.2f us, wrong values in %lld workgroups\n",
                           nw, allx ? "each of 8 XCDs" : "one XCD", mode == 0 ? "sc0 store + nt load (L2)" : "sc1 store + sc1 load",
                           mean / cnt * 0.01, mx * 0.01, bad);
                }
    return 0;
}

// the same with the access pattern of k_chol_fused: a 64 x 64 tile and two 64 x 32 panels of a
// matrix with a 736-double row pitch (sc1 loads), per-lane indexing as in the MFMA layouts
__global__ void k_bulk_pitch(double* mats, unsigned* flag, int nw, long long* ticks) {
    extern __shared__ char big[];
    big[0] = 1;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    if (slot >= nw) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lda = 736;
    flag += xcd * 32;
    double* M = mats + ((size_t)xcd * 32 + slot) * 192 * lda;               // my matrix slab: 192 rows
    const double* T = mats + ((size_t)xcd * 32 + (slot + 1) % nw) * 192 * lda;
    for (int i = 0; i < 48; ++i) {                                           // write 192 x 64 doubles
        const int e = tid + 256 * i, r = e >> 6, c = e & 63;
        __hip_atomic_store(&M[(size_t)r * lda + c], (double)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nw && ++spins < (1 << 22)) {}
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    const int li = lane & 15, lk = lane >> 4;
    double r[32];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int i = 16 * wave + lk + 4 * rg, j = 16 * c + li;
            r[4 * c + rg] = __hip_atomic_load(&T[(size_t)i * lda + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = tid + 256 * q, rr = e >> 5, m = e & 31;
        r[16 + q] = __hip_atomic_load(&T[(size_t)(64 + rr) * lda + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r[24 + q] = __hip_atomic_load(&T[(size_t)(128 + rr) * lda + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += r[i];
    __syncthreads();
    const long long t1 = wall_clock64();
    if (tid == 0) ticks[(xcd * 32 + slot) * 2] = t1 - t0;
    if (acc == 12345.0) mats[0] = acc;
}

int pitch_main() {
    const int NWG = 256, lds = 100 * 1024;
    double* d_m; unsigned* d_flag; long long* d_ticks;
    CK(hipMalloc(&d_m, (size_t)256 * 192 * 736 * 8));
    CK(hipMalloc(&d_flag, 4 * 256));
    CK(hipMalloc(&d_ticks, 512 * 8));
    CK(hipFuncSetAttribute((const void*)k_bulk_pitch, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int nw : {2, 26, 32})
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(d_flag, 0, 4 * 256));
            CK(hipMemset(d_ticks, 0, 512 * 8));
            hipLaunchKernelGGL(k_bulk_pitch, dim3(NWG), dim3(256), lds, 0, d_m, d_flag, nw, d_ticks);
            CK(hipDeviceSynchronize());
            long long h[512];
            CK(hipMemcpy(h, d_ticks, sizeof(h), hipMemcpyDeviceToHost));
            double mean = 0; long long mx = 0; int cnt = 0;
            for (int x = 0; x < 8; ++x)
                for (int i = 0; i < nw; ++i) { const long long t = h[(x * 32 + i) * 2]; mean += t; mx = t > mx ? t : mx; ++cnt; }
            printf("pitched tile + 2 panels (64 KB, sc1), %2d workgroups on each of 8 XCDs: mean %.2f us, max %.2f us\n",
                   nw, mean / cnt * 0.01, mx * 0.01);
        }
    return 0;
}
