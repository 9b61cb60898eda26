// Exact median and 1.4826 x MAD of the unmasked pixels (quick_background_estimate, zuds/utils.py:32-53) by a
// SAMPLE-BRACKETED select (round 6; VERDICT r5 item 4c, the form DESIGN.md described and had not built).
//
// The three-pass radix select (api_subtract.hip) streams a frame six times and every pass histograms (nearly) every
// sky pixel in LDS: 6 x ~30 us per pair of 9.4 Mpx frames, 13 launches.  Here a frame is streamed TWICE:
//   k_rsel2_sample   one workgroup per image reads 4 096 stratified, jittered pixels (four per thread: one CU is
//                    1 / 256 of the GPU, whatever it does per thread is paid in full) and narrows - two in-LDS
//                    histogram steps - to a key interval [lo, lo + span) that holds the sample's ranks
//                    n/2 -+ (2.75 sqrt(n) + 1): 5.5 sigma of the binomial rank error, ~ 9 % of the pixels;
//   k_rsel2_pass     the whole frame, once: counts the valid keys below lo and files the keys of the interval under
//                    256 bins - ONE LDS atomic each: the count it returns is the key's slot in the workgroup's
//                    segment of that bin - with the smallest and largest key of every bin beside the counts;
//   k_rsel2_finish   one workgroup per image: finds the bin of each middle rank; a bin of one key (ties, narrow
//                    intervals) is the answer, otherwise the bin's ~3 000 keys come from the 256 segments into LDS
//                    and the rank is selected there, 8 bits per step; for the median it goes on to bracket
//                    |v - median| from the SAME sample (kept in HBM) for the second pass;
//                    after the MAD it writes the outputs; when a bracket missed its rank or the keys of a needed
//                    bin are not all there - a flag in the state - it first repeats the select with the three-pass
//                    form in ONE workgroup: slow (milliseconds), exact, and no launch depends on the host seeing
//                    the flag.
// Five launches, the same bits as numpy.median on float32 (tests/test_select_bracket_gpu.py forces the rescue with an
// adversarial frame built from the sample positions below).
#include "select_dev.h"

#define RS2_THREADS 1024
#define RS2_WAVES 16
#define RS2_NK 4                                // sampled pixels per thread: one float4 group
#define RS2_SAMPLE4 RS2_THREADS                 // 1 024 groups = 4 096 pixels per image
#define RS2_SBINS 2048
#define RS2_GRID 256
#define RS2_BINS 256
#define RS2_SUB 40                              // slots per workgroup, image and bin (expected ~13: the bins are equal
                                                // shares of the interval, not powers of two)
#define RS2_LIST 8192
#define RS2_UNROLL 2
#define RS2_FAIL_MED 1u
#define RS2_FAIL_MAD 2u

struct rs2_state {
    uint32_t lo, span_m1, mul, fail;                // bin = d * mul >> 32 (mul = 0: the interval has <= 256 keys, bin = d)
    unsigned long long below, valid, count;
    float centre, median;
    double out[3];
    unsigned int hist[RS2_BINS];
    uint32_t kmin[RS2_BINS], kmax[RS2_BINS];
    uint32_t ovf[RS2_BINS / 32];
};

__device__ inline int nbits(uint32_t x) { return x ? 32 - __clz(x) : 0; }

// position (in float4 groups) of sample j: one per stratum of `stride` groups, jittered by a hash so that the
// sample does not walk down a handful of columns (3072-px rows: 768 groups, stride 288 -> 8 columns without it)
__host__ __device__ inline int64_t rs2_sample_pos(int j, int64_t stride) {
    const uint32_t h = ((uint32_t)j * 2654435761u) ^ (((uint32_t)j * 40503u) >> 3);
    return (int64_t)j * stride + (int64_t)(h % (uint32_t)stride);
}

// Block-wide (1 024 threads): bins and exclusive counts of two ranks in a histogram of 2 048 bins.
// res: {bin0, excl0, bin1, excl1, total}.  A rank >= total leaves its pair untouched (callers initialise).
template <typename T>
__device__ inline void find_ranks_2048(const unsigned int* h, T r0, T r1, T* wtot, T* res) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int h0 = h[2 * tid], h1 = h[2 * tid + 1];
    const T loc = (T)h0 + (T)h1;
    T inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RS2_WAVES; ++w) {
        if (w < wave) off += wtot[w];
        tot += wtot[w];
    }
    const T excl = off + inc - loc;
    if (tid == 0) res[4] = tot;
    if (r0 >= excl && r0 < excl + loc) {
        if (r0 < excl + h0) { res[0] = (T)(2 * tid); res[1] = excl; }
        else { res[0] = (T)(2 * tid + 1); res[1] = excl + h0; }
    }
    if (r1 >= excl && r1 < excl + loc) {
        if (r1 < excl + h0) { res[2] = (T)(2 * tid); res[3] = excl; }
        else { res[2] = (T)(2 * tid + 1); res[3] = excl + h0; }
    }
    __syncthreads();
}

// Wave 0 only: bin and exclusive count of rank r in 256 bins.  res: {bin, excl, total}
__device__ inline void wave_find_256(const unsigned int* h, unsigned long long r, unsigned long long* res) {
    const int lane = threadIdx.x & 63;
    unsigned int hl[4];
    unsigned long long loc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { hl[j] = h[4 * lane + j]; loc += hl[j]; }
    unsigned long long inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    const unsigned long long tot = __shfl(inc, 63);
    unsigned long long excl = inc - loc;
    if (lane == 0) res[2] = tot;
    if (r >= excl && r < excl + loc) {
        int j = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (j == q && r >= excl + hl[q]) { excl += hl[q]; j = q + 1; }
        res[0] = (unsigned long long)(4 * lane + j);
        res[1] = excl;
    }
}

struct rs2_lds {
    unsigned int lh[RS2_SBINS];
    unsigned int wtot[RS2_WAVES];
    unsigned int res[5];
    unsigned int red[3][RS2_WAVES];
};

// One narrowing step over the keys a workgroup holds in registers (four per thread, `vm`: which are valid):
// [lo, lo + span_m1] -> the sub-interval between the bins of ranks r0 and r1 (ranks counted from lo).
__device__ inline void narrow_sample(const uint32_t (&key)[RS2_NK], uint32_t vm, uint32_t& lo, uint32_t& span_m1,
                                     unsigned int& r0, unsigned int& r1, int& shift_used, rs2_lds& L) {
    const int tid = threadIdx.x;
    const int shift = max(0, nbits(span_m1) - 11);
    for (int k = tid; k < RS2_SBINS; k += RS2_THREADS) L.lh[k] = 0;
    if (tid == 0) { L.res[0] = 0; L.res[1] = 0; L.res[2] = 0; L.res[3] = 0; }
    __syncthreads();
    unsigned int first = 0, cnt = 0;
    bool uniform = true;
#pragma unroll
    for (int j = 0; j < RS2_NK; ++j)
        if ((vm >> j) & 1u) {
            const uint32_t k = key[j];
            if (k >= lo && k - lo <= span_m1) {
                const unsigned int b = (k - lo) >> shift;
                if (cnt == 0) first = b;
                else if (b != first) uniform = false;
                ++cnt;
            }
        }
    // (a thread whose pixels share a bin - flat frames, ties - adds them at once: 64 lanes on one address cost 64 turns)
    if (cnt && uniform) atomicAdd(&L.lh[first], cnt);
    else if (cnt) {
#pragma unroll
        for (int j = 0; j < RS2_NK; ++j)
            if ((vm >> j) & 1u) {
                const uint32_t k = key[j];
                if (k >= lo && k - lo <= span_m1) atomicAdd(&L.lh[(k - lo) >> shift], 1u);
            }
    }
    __syncthreads();
    find_ranks_2048<unsigned int>(L.lh, r0, r1, L.wtot, L.res);
    const unsigned int b0 = L.res[0], e0 = L.res[1], b1 = L.res[2];
    const uint32_t new_lo = lo + (b0 << shift);
    const unsigned long long top = (unsigned long long)lo + (((unsigned long long)b1 + 1ull) << shift) - 1ull;
    const unsigned long long lim = (unsigned long long)lo + span_m1;
    const uint32_t hi = (uint32_t)(top < lim ? top : lim);
    r0 -= e0;
    r1 -= e0;
    span_m1 = hi - new_lo;
    lo = new_lo;
    shift_used = shift;
    __syncthreads();
}

// The bracket of one select from the four keys per thread: writes lo / span_m1 / shift of the state and clears what
// the pass accumulates.  Every thread of the workgroup calls it.
__device__ inline void bracket_from_sample(const uint32_t (&key)[RS2_NK], uint32_t vm, rs2_state* S, rs2_lds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned int nv = __popc(vm);
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < RS2_NK; ++j)
        if ((vm >> j) & 1u) { kmin = min(kmin, key[j]); kmax = max(kmax, key[j]); }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        nv += __shfl_xor(nv, o);
        kmin = min(kmin, (uint32_t)__shfl_xor(kmin, o));
        kmax = max(kmax, (uint32_t)__shfl_xor(kmax, o));
    }
    if (lane == 0) { L.red[0][wave] = nv; L.red[1][wave] = kmin; L.red[2][wave] = kmax; }
    __syncthreads();
    nv = 0; kmin = 0xffffffffu; kmax = 0u;
#pragma unroll
    for (int w = 0; w < RS2_WAVES; ++w) {
        nv += L.red[0][w];
        kmin = min(kmin, L.red[1][w]);
        kmax = max(kmax, L.red[2][w]);
    }
    __syncthreads();
    uint32_t lo = 0u, hi = 0xffffffffu;
    if (nv > 0) {
        const float half = 0.5f * (float)(nv - 1), d = 2.75f * sqrtf((float)nv) + 1.f;
        const unsigned int rl = (unsigned int)fmaxf(0.f, floorf(half - d));
        const unsigned int rh = (unsigned int)fminf((float)(nv - 1), ceilf(half + d));
        uint32_t rlo = kmin, span_m1 = kmax - kmin;
        unsigned int r0 = rl, r1 = rh;
        int sh = 0;
        narrow_sample(key, vm, rlo, span_m1, r0, r1, sh, L);
        if (sh > 0) narrow_sample(key, vm, rlo, span_m1, r0, r1, sh, L);
        lo = rlo;
        hi = rlo + span_m1;
        // an end of the sample is no bound: the interval is open on that side
        if (rl == 0) lo = 0u;
        if (rh == nv - 1) hi = 0xffffffffu;
    }
    for (int k = tid; k < RS2_BINS; k += RS2_THREADS) {
        S->hist[k] = 0;
        S->kmin[k] = 0xffffffffu;
        S->kmax[k] = 0u;
    }
    if (tid < RS2_BINS / 32) S->ovf[tid] = 0;
    if (tid == 0) {
        S->lo = lo;
        S->span_m1 = hi - lo;
        S->mul = (hi - lo) < (uint32_t)RS2_BINS ? 0u
                                                : (uint32_t)(((unsigned long long)RS2_BINS << 32) / ((unsigned long long)(hi - lo) + 1ull));
        S->below = 0;
        S->valid = 0;
    }
}

// grid: nimg.  The sample of each image -> samp (NaN where the pixel is not valid), the median's bracket -> st.
__global__ __launch_bounds__(RS2_THREADS) void k_rsel2_sample(const rs_batch B, rs2_state* __restrict__ st,
                                                              float* __restrict__ samp) {
    __shared__ rs2_lds L;
    const int im = blockIdx.x, tid = threadIdx.x;
    const float4* __restrict__ img = reinterpret_cast<const float4*>(B.im[im].img);
    const int4* __restrict__ mask = reinterpret_cast<const int4*>(B.im[im].mask);
    float4* __restrict__ sp = reinterpret_cast<float4*>(samp) + (size_t)im * RS2_SAMPLE4;
    const int64_t stride = (B.n / 4) / RS2_SAMPLE4;
    const int64_t q = rs2_sample_pos(tid, stride);
    const float4 v = img[q];
    const int4 m = mask ? mask[q] : make_int4(0, 0, 0, 0);
    uint32_t key[RS2_NK];
    uint32_t vm = 0;
    const float qnan = __uint_as_float(0x7fc00000u);
    const float a[4] = {v.x, v.y, v.z, v.w};
    const int b[4] = {m.x, m.y, m.z, m.w};
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool ok = b[c] == 0 && a[c] == a[c];
        key[c] = f2key(a[c]);
        if (ok) vm |= 1u << c;
        o[c] = ok ? a[c] : qnan;
    }
    sp[tid] = make_float4(o[0], o[1], o[2], o[3]);
    rs2_state* S = st + im;
    if (tid == 0) {
        S->fail = 0;
        S->count = 0;
        S->centre = 0.f;
        S->median = 0.f;
        S->out[0] = S->out[1] = S->out[2] = 0.0;
    }
    bracket_from_sample(key, vm, S, L);
}

// grid: (<= RS2_GRID, nimg).  mode 0: v = img[p]; mode 1: v = |img[p] - centre| (float32 arithmetic, as numpy).
// vbits / vmode as k_rsel_hist (api_subtract.hip): 1 = read the masks and write the validity bits, 2 = read the bits.
// seg: [image][workgroup][bin][RS2_SUB] keys (staged in LDS, written once); wcnt: [image][workgroup][bin] counts (what the workgroup saw, which
// may exceed RS2_SUB: the bin's bit in ovf says so).
// The loads of RS2_UNROLL iterations are issued together: with one in flight per thread a pass waited for memory
// nine times over (measured: 30 us for 78 MB).
template <int VMODE, int MODE>           // VMODE 0: no bit plane, 1: masks read, bits written, 2: bits read
__global__ __launch_bounds__(RS2_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_rsel2_pass(const rs_batch B, rs2_state* __restrict__ st, uint32_t* __restrict__ seg,
                  unsigned int* __restrict__ wcnt, unsigned long long* __restrict__ vbits) {
    __shared__ unsigned int lh[RS2_BINS];
    __shared__ uint32_t lmin[RS2_BINS], lmax[RS2_BINS];
    __shared__ unsigned int red[2][RS2_WAVES];
    __shared__ __attribute__((aligned(16))) uint32_t stage[RS2_BINS * RS2_SUB];
    const int im = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ img = B.im[im].img;
    const int32_t* __restrict__ mask = B.im[im].mask;
    rs2_state* S = st + im;
    const uint32_t lo = S->lo, span_m1 = S->span_m1, mul = S->mul;
    const float centre = S->centre;
    uint32_t* __restrict__ sg = seg + ((size_t)im * RS2_GRID + blockIdx.x) * (RS2_BINS * RS2_SUB);
    for (int k = tid; k < RS2_BINS; k += RS2_THREADS) { lh[k] = 0; lmin[k] = 0xffffffffu; lmax[k] = 0u; }
    __syncthreads();
    // Counts are kept per WAVE (ballot + scalar population count: the scalar unit is idle here, the vector unit is
    // what a pass waits for - at ~40 vector instructions per pixel the first form of this loop took 45 us).
    unsigned int nvalid = 0, nbelow = 0;
    // One float4 group: ok[c] = the wave's ballot of "pixel c of my group is valid".  The four atomics of a group
    // are issued before the first returned slot is needed.
    // Lane masks stay what they are - scalar register pairs: validity is the ballot the caller has, "below" and
    // "inside" are compares whose result IS a mask (v_cmp -> SGPR pair, combined and counted on the scalar unit), and
    // __builtin_amdgcn_inverse_ballot_w64 turns a mask back into the branch condition.  Per pixel the vector unit does:
    // [subtract / abs], two for the key, the difference from lo, two compares - and, where a wave holds a candidate
    // (nearly always: 9 % of 64 lanes), the bin, its atomic and the staging store.
    auto group = [&](const float4 v, const unsigned long long (&ok)[4]) {
        const float a[4] = {v.x, v.y, v.z, v.w};
        uint32_t d[4];
        unsigned long long cm[4], anym = 0ull;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x = MODE == 1 ? fabsf(a[c] - centre) : a[c];
            const uint32_t u = __float_as_uint(x);
            const uint32_t key = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);       // f2key
            nbelow += (unsigned int)__popcll(ok[c] & __builtin_amdgcn_ballot_w64(key < lo));
            d[c] = key - lo;                                     // (wraps above span_m1 where key < lo)
            cm[c] = ok[c] & __builtin_amdgcn_ballot_w64(d[c] <= span_m1);
            anym |= cm[c];
            nvalid += (unsigned int)__popcll(ok[c]);
        }
        if (anym) {
            unsigned int bin[4], slot[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bin[c] = mul ? __umulhi(d[c], mul) : d[c];
                slot[c] = 0u;
                if (__builtin_amdgcn_inverse_ballot_w64(cm[c])) slot[c] = atomicAdd(&lh[bin[c]], 1u);
            }
            if (mul) {                                           // (bins of one key each need no keys kept)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (__builtin_amdgcn_inverse_ballot_w64(cm[c])) {
                        const uint32_t key = d[c] + lo;
                        if (slot[c] < RS2_SUB) stage[bin[c] * RS2_SUB + slot[c]] = key;
                        else { atomicMin(&lmin[bin[c]], key); atomicMax(&lmax[bin[c]], key); }
                    }
            }
        }
    };
    const int64_t n = B.n;
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long* vb = VMODE ? vbits + (size_t)im * (size_t)((n4 + 63) / 64) * 4 : nullptr;
    // whole waves walk the loop together (the counts are per wave): the bound is rounded up to the wave
    const int64_t n4w = ((n4 + 63) / 64) * 64;
    // A batch = RS2_UNROLL float4 groups per thread (+ their masks or bit words).  The loads of batch k + 1 are in
    // flight while batch k is worked on (two register sets, A and B): with every wave loading, then waiting, then
    // computing, a pass took its memory time PLUS its compute time (28 us for 78 MB).
    auto load = [&](int64_t q0, float4 (&v)[RS2_UNROLL], int4 (&m)[RS2_UNROLL], uint32_t& xw) {
        // the bit words of the groups: eight 32-bit halves each, ONE load - lane l fetches half l & 7 of group
        // l >> 3 (q >> 6 is the same in every lane of a wave) - and readlane hands them to the scalar side later
        xw = 0u;
        if (VMODE == 2) {
            const int64_t qu = q0 + (int64_t)(lane >> 3) * stride;
            if (lane < 8 * RS2_UNROLL && qu < n4w)
                xw = reinterpret_cast<const uint32_t*>(vb)[(qu >> 6) * 8 + (lane & 7)];
        }
#pragma unroll
        for (int u = 0; u < RS2_UNROLL; ++u) {
            const int64_t q = q0 + u * stride;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            m[u] = make_int4(0, 0, 0, 0);
            if (q < n4) v[u] = reinterpret_cast<const float4*>(img)[q];
            if (VMODE != 2 && mask && q < n4) m[u] = reinterpret_cast<const int4*>(mask)[q];
        }
    };
    auto work = [&](int64_t q0, const float4 (&v)[RS2_UNROLL], const int4 (&m)[RS2_UNROLL], uint32_t xw) {
#pragma unroll
        for (int u = 0; u < RS2_UNROLL; ++u) {
            const int64_t q = q0 + u * stride;
            if (q >= n4w) break;                                 // (uniform within the wave)
            unsigned long long w[4];
            if (VMODE == 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    w[c] = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)xw, 8 * u + 2 * c + 1) << 32) |
                           (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)xw, 8 * u + 2 * c);
            } else {
                const unsigned long long inm = __builtin_amdgcn_ballot_w64(q < n4);
                w[0] = inm & __builtin_amdgcn_ballot_w64(m[u].x == 0) & __builtin_amdgcn_ballot_w64(v[u].x == v[u].x);
                w[1] = inm & __builtin_amdgcn_ballot_w64(m[u].y == 0) & __builtin_amdgcn_ballot_w64(v[u].y == v[u].y);
                w[2] = inm & __builtin_amdgcn_ballot_w64(m[u].z == 0) & __builtin_amdgcn_ballot_w64(v[u].z == v[u].z);
                w[3] = inm & __builtin_amdgcn_ballot_w64(m[u].w == 0) & __builtin_amdgcn_ballot_w64(v[u].w == v[u].w);
                if (VMODE == 1 && lane == 0) {
                    unsigned long long* wp = vb + (q >> 6) * 4;
                    wp[0] = w[0]; wp[1] = w[1]; wp[2] = w[2]; wp[3] = w[3];
                }
            }
            group(v[u], w);
        }
    };
    {
        float4 vA[RS2_UNROLL], vB[RS2_UNROLL];
        int4 mA[RS2_UNROLL], mB[RS2_UNROLL];
        uint32_t xA, xB;
        const int64_t step = RS2_UNROLL * stride;
        int64_t q = (int64_t)blockIdx.x * blockDim.x + tid;
        if (q < n4w) {                                           // (whole waves: n4w is a multiple of 64)
            load(q, vA, mA, xA);
            while (true) {
                const bool more = q + step < n4w;
                if (more) load(q + step, vB, mB, xB);
                work(q, vA, mA, xA);
                if (!more) break;
                q += step;
                const bool more2 = q + step < n4w;
                if (more2) load(q + step, vA, mA, xA);
                work(q, vB, mB, xB);
                if (!more2) break;
                q += step;
            }
        }
    }
    // the last n % 4 pixels: lanes 0 .. 2 of the first wave of the first workgroup
    if (blockIdx.x == 0 && wave == 0 && (n & 3)) {
        const int64_t p = 4 * n4 + lane;
        const bool in = lane < (int)(n & 3);
        const float x = in ? img[p] : 0.f;
        const int mk = (in && mask) ? mask[p] : 0;
        unsigned long long ok[4] = {__builtin_amdgcn_ballot_w64(in && mk == 0 && x == x), 0ull, 0ull, 0ull};
        group(make_float4(x, 0.f, 0.f, 0.f), ok);
    }
    if (lane == 0) { red[0][wave] = nvalid; red[1][wave] = nbelow; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long a = 0, b = 0;
        for (int w = 0; w < RS2_WAVES; ++w) { a += red[0][w]; b += red[1][w]; }
        if (a) atomicAdd(&S->valid, a);
        if (b) atomicAdd(&S->below, b);
    }
    unsigned int* __restrict__ wc = wcnt + ((size_t)im * RS2_GRID + blockIdx.x) * RS2_BINS;
    for (int k = tid; k < RS2_BINS; k += RS2_THREADS) {
        const unsigned int c = lh[k];
        wc[k] = c;
        if (c) {
            atomicAdd(&S->hist[k], c);
            if (mul && c > RS2_SUB) {
                // a bin with more keys than slots (ties, as a rule): its extremes over ALL the keys this workgroup
                // saw - the kept ones and those that found no slot - so that the finish can tell a bin of one key
                uint32_t mn = lmin[k], mx = lmax[k];
                for (unsigned int i = 0; i < RS2_SUB; ++i) {
                    const uint32_t kk = stage[k * RS2_SUB + i];
                    mn = min(mn, kk);
                    mx = max(mx, kk);
                }
                atomicMin(&S->kmin[k], mn);
                atomicMax(&S->kmax[k], mx);
                atomicOr(&S->ovf[k >> 5], 1u << (k & 31));
            }
        }
    }
    // the staged keys go out as whole lines
    if (mul) {
        uint4* __restrict__ o4 = reinterpret_cast<uint4*>(sg);
        const uint4* s4 = reinterpret_cast<const uint4*>(stage);
        for (int k = tid; k < RS2_BINS * RS2_SUB / 4; k += RS2_THREADS) o4[k] = s4[k];
    }
}

// Rank r among the keys of list[0 .. m) that lie in [base, base + 2^rem): 8 bits per step.  Every thread calls it;
// the key comes back in every thread.
__device__ inline uint32_t select_in_list(const uint32_t* list, unsigned int m, uint32_t base, int rem,
                                          unsigned long long r, unsigned int* h256, unsigned long long* res) {
    const int tid = threadIdx.x;
    while (rem > 0) {
        const int nb = rem < 8 ? rem : 8, sh = rem - nb;
        for (int k = tid; k < RS2_BINS; k += RS2_THREADS) h256[k] = 0;
        if (tid == 0) { res[0] = 0; res[1] = 0; }
        __syncthreads();
        const uint32_t width_m1 = (rem >= 32) ? 0xffffffffu : ((1u << rem) - 1u);
        for (unsigned int i = tid; i < m; i += RS2_THREADS) {
            const uint32_t k = list[i];
            if (k >= base && k - base <= width_m1) atomicAdd(&h256[(k - base) >> sh], 1u);
        }
        __syncthreads();
        if (tid < 64) wave_find_256(h256, r, res);
        __syncthreads();
        base += (uint32_t)res[0] << sh;
        r -= res[1];
        rem = sh;
        __syncthreads();
    }
    return base;
}

// The three-pass select in one workgroup (k_rsel_hist / k_rsel_scan of api_subtract.hip, restated for 1 024 threads):
// only where a bracket failed.
struct rs2_rescue_lds {
    unsigned int lh[2][RS2_SBINS];
    unsigned long long wtot[RS2_WAVES];
    unsigned long long res[5];
};

__device__ inline float rescue_select(const float* __restrict__ img, const int32_t* __restrict__ mask, int64_t n,
                                      int mode, float centre, unsigned long long& count, rs2_rescue_lds& L) {
    const int tid = threadIdx.x;
    uint32_t prefix[2] = {0u, 0u}, pmask = 0u;
    unsigned long long k[2] = {0ull, 0ull};
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass];
        const uint32_t bm = (1u << bits[pass]) - 1u;
        const bool two = prefix[1] != prefix[0];
        for (int q = tid; q < 2 * RS2_SBINS; q += RS2_THREADS) (&L.lh[0][0])[q] = 0;
        if (tid == 0) for (int q = 0; q < 5; ++q) L.res[q] = 0;
        __syncthreads();
        int cur = -1;
        unsigned int run = 0;
        for (int64_t p = tid; p < n; p += RS2_THREADS) {
            float v = img[p];
            if ((mask && mask[p] != 0) || !(v == v)) continue;
            if (mode == 1) v = fabsf(v - centre);
            const uint32_t key = f2key(v);
            const uint32_t hi = key & pmask;
            const int b = (int)((key >> shift) & bm);
            if (hi == prefix[0]) {
                if (b == cur) ++run;
                else {
                    if (run) atomicAdd(&L.lh[0][cur], run);
                    cur = b;
                    run = 1;
                }
            } else if (two && hi == prefix[1]) {
                atomicAdd(&L.lh[1][b], 1u);
            }
        }
        if (run) atomicAdd(&L.lh[0][cur], run);
        __syncthreads();
        if (pass == 0) {
            find_ranks_2048<unsigned long long>(L.lh[0], ~0ull, ~0ull, L.wtot, L.res);
            count = L.res[4];
            __syncthreads();
            if (count == 0) return 0.f;
            k[0] = (count - 1) / 2;
            k[1] = count / 2;
        }
        if (!two) {
            find_ranks_2048<unsigned long long>(L.lh[0], k[0], k[1], L.wtot, L.res);
            const unsigned long long b0 = L.res[0], e0 = L.res[1], b1 = L.res[2], e1 = L.res[3];
            __syncthreads();
            prefix[0] |= (uint32_t)b0 << shift;
            prefix[1] |= (uint32_t)b1 << shift;
            k[0] -= e0;
            k[1] -= e1;
        } else {
            find_ranks_2048<unsigned long long>(L.lh[0], k[0], ~0ull, L.wtot, L.res);
            const unsigned long long b0 = L.res[0], e0 = L.res[1];
            __syncthreads();
            find_ranks_2048<unsigned long long>(L.lh[1], k[1], ~0ull, L.wtot, L.res);
            const unsigned long long b1 = L.res[0], e1 = L.res[1];
            __syncthreads();
            prefix[0] |= (uint32_t)b0 << shift;
            prefix[1] |= (uint32_t)b1 << shift;
            k[0] -= e0;
            k[1] -= e1;
        }
        pmask |= bm << shift;
    }
    return 0.5f * (key2f_dev(prefix[0]) + key2f_dev(prefix[1]));
}

// grid: nimg.  Resolves the two middle ranks of the pass that ran; after the median (mode 0) it brackets the MAD;
// after the MAD (mode 1) it writes the outputs - behind a failed bracket, after the select in its three-pass form.
__global__ __launch_bounds__(RS2_THREADS) void k_rsel2_finish(int mode, const rs_batch B, rs2_state* __restrict__ st,
                                                              const uint32_t* __restrict__ seg,
                                                              const unsigned int* __restrict__ wcnt,
                                                              const float* __restrict__ samp, double* __restrict__ out) {
    __shared__ rs2_lds L;
    __shared__ rs2_rescue_lds R;
    __shared__ uint32_t lfail;
    __shared__ unsigned int h256[RS2_BINS];
    __shared__ uint32_t list[RS2_LIST];
    __shared__ unsigned int ln;
    __shared__ unsigned long long res[2][3];
    __shared__ unsigned long long res2[3];
    __shared__ int failed;
    const int im = blockIdx.x, tid = threadIdx.x;
    rs2_state* S = st + im;
    const uint32_t lo = S->lo, mul = S->mul;
    const unsigned long long below = S->below;
    const unsigned long long count = mode == 0 ? S->valid : S->count;
    const uint32_t failbit = mode ? RS2_FAIL_MAD : RS2_FAIL_MED;
    for (int k = tid; k < RS2_BINS; k += RS2_THREADS) h256[k] = S->hist[k];
    if (tid == 0) {
        failed = (S->fail & failbit) ? 1 : 0;
        for (int t = 0; t < 2; ++t) { res[t][0] = 0; res[t][1] = 0; res[t][2] = 0; }
    }
    __syncthreads();
    const unsigned long long k0 = count ? (count - 1) / 2 : 0, k1 = count ? count / 2 : 0;
    float med = 0.f;
    if (count > 0) {
        if (tid < 64) {
            wave_find_256(h256, k0 >= below ? k0 - below : ~0ull, res[0]);
            wave_find_256(h256, k1 >= below ? k1 - below : ~0ull, res[1]);
        }
        __syncthreads();
        const unsigned long long inside = res[0][2];
        if (tid == 0 && !(k0 >= below && k1 - below < inside)) failed = 1;
        __syncthreads();
        uint32_t keyout[2] = {0u, 0u};
        int have = -1;                                     // the bin whose keys are in `list`
        unsigned int m = 0;
        uint32_t bmin = 0u, bmax = 0u;
        bool whole = false;
        for (int t = 0; t < 2 && !failed; ++t) {
            if (t == 1 && k1 == k0) { keyout[1] = keyout[0]; break; }
            const unsigned int b = (unsigned int)res[t][0];
            const unsigned long long r = (t ? k1 : k0) - below - res[t][1];
            if (mul == 0) { keyout[t] = lo + b; continue; }
            if ((int)b != have) {
                // the keys of bin b from the 256 segments: four threads per segment, the first sixteen entries
                // requested at once; their extremes on the way (a bin of one key - ties - is its own answer)
                const bool over = (S->ovf[b >> 5] >> (b & 31)) & 1u;
                __syncthreads();
                if (tid == 0) ln = 0;
                __syncthreads();
                uint32_t mn = 0xffffffffu, mx = 0u;
                {
                    const int w = tid >> 2, part = tid & 3;
                    const uint32_t* __restrict__ sgb = seg + (((size_t)im * RS2_GRID + w) * RS2_BINS + b) * RS2_SUB;
                    unsigned int c = wcnt[((size_t)im * RS2_GRID + w) * RS2_BINS + b];
                    uint32_t first[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) first[j] = sgb[part + 4 * j];
                    c = c < RS2_SUB ? c : RS2_SUB;
                    auto take = [&](uint32_t kk) {
                        mn = min(mn, kk);
                        mx = max(mx, kk);
                        const unsigned int slot = atomicAdd(&ln, 1u);
                        if (slot < RS2_LIST) list[slot] = kk;
                    };
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((unsigned int)(part + 4 * j) < c) take(first[j]);
                    for (unsigned int i = part + 16; i < c; i += 4) take(sgb[i]);
                }
#pragma unroll
                for (int o = 32; o; o >>= 1) {
                    mn = min(mn, (uint32_t)__shfl_xor(mn, o));
                    mx = max(mx, (uint32_t)__shfl_xor(mx, o));
                }
                if ((tid & 63) == 0) { L.red[1][tid >> 6] = mn; L.red[2][tid >> 6] = mx; }
                __syncthreads();
                mn = 0xffffffffu; mx = 0u;
#pragma unroll
                for (int w = 0; w < RS2_WAVES; ++w) { mn = min(mn, L.red[1][w]); mx = max(mx, L.red[2][w]); }
                if (over) { mn = min(mn, S->kmin[b]); mx = max(mx, S->kmax[b]); }
                m = ln;
                bmin = mn;
                bmax = mx;
                whole = !over && m <= RS2_LIST;
                have = (int)b;
                __syncthreads();
            }
            if (bmin == bmax) { keyout[t] = bmin; continue; }
            if (!whole) {                                      // (keys missing from the list: not a tie, not resolvable here)
                __syncthreads();
                if (tid == 0) failed = 1;
                __syncthreads();
                break;
            }
            keyout[t] = select_in_list(list, m, bmin, nbits(bmax - bmin), r, h256, res2);
        }
        if (!failed) med = 0.5f * (key2f_dev(keyout[0]) + key2f_dev(keyout[1]));
    }
    __syncthreads();
    if (tid == 0) {
        lfail = S->fail | (failed ? failbit : 0u);
        if (failed) atomicOr(&S->fail, failbit);
        if (mode == 0) {
            S->count = count;
            S->median = med;
            S->centre = med;
            S->out[0] = med;
            S->out[2] = (double)count;
        } else {
            S->out[1] = 1.4826 * (double)med;
        }
    }
    if (mode == 1) {
        __syncthreads();
        double o[3] = {S->out[0], 1.4826 * (double)med, S->out[2]};
        if (lfail) {                                           // (uniform: shared)
            unsigned long long cnt = S->count;
            float m0 = S->median;
            if (lfail & RS2_FAIL_MED) m0 = rescue_select(B.im[im].img, B.im[im].mask, B.n, 0, 0.f, cnt, R);
            __syncthreads();
            unsigned long long c2 = 0;
            const float mad = cnt ? rescue_select(B.im[im].img, B.im[im].mask, B.n, 1, m0, c2, R) : 0.f;
            o[0] = m0;
            o[1] = 1.4826 * (double)mad;
            o[2] = (double)cnt;
            if (tid == 0) {
                S->count = cnt;
                S->median = m0;
                S->out[0] = o[0]; S->out[1] = o[1]; S->out[2] = o[2];
            }
        }
        if (tid < 3) out[3 * im + tid] = o[tid];
        return;
    }
    // the MAD's bracket from the same sample: |v - median| of its valid pixels
    const float4 v = (reinterpret_cast<const float4*>(samp) + (size_t)im * RS2_SAMPLE4)[tid];
    const float a[4] = {v.x, v.y, v.z, v.w};
    uint32_t key[RS2_NK];
    uint32_t vm = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (a[c] == a[c]) vm |= 1u << c;
        key[c] = f2key(fabsf(a[c] - med));
    }
    __syncthreads();
    bracket_from_sample(key, vm, S, L);
}

int zm_rs2_median_mad(zm_ctx* ctx, int nimg, const rs_batch& B, unsigned long long* d_vbits, double* out_dev) {
    ZM_CHECK(nimg >= 1 && nimg <= ZM_RS_MAXIMG && B.n >= ZM_RS2_MIN_N, "zm_rs2_median_mad: shape");
    rs2_state* d_st = nullptr;
    float* d_samp = nullptr;
    uint32_t* d_seg = nullptr;
    unsigned int* d_wcnt = nullptr;
    // (sized for the images of this call - 10.5 MB of candidate segments each; a later call with more of them grows
    // the buffers)
    ZM_TRY(ctx->get("rs2_state", sizeof(rs2_state) * ZM_RS_MAXIMG, (void**)&d_st));
    ZM_TRY(ctx->get("rs2_samp", sizeof(float) * 4 * RS2_SAMPLE4 * (size_t)nimg, (void**)&d_samp));
    ZM_TRY(ctx->get("rs2_seg", sizeof(uint32_t) * (size_t)RS2_SUB * RS2_BINS * RS2_GRID * (size_t)nimg, (void**)&d_seg));
    ZM_TRY(ctx->get("rs2_wcnt", sizeof(unsigned int) * (size_t)RS2_BINS * RS2_GRID * (size_t)nimg, (void**)&d_wcnt));
    const int grid = RS2_GRID;          // (every workgroup writes its row of counts: n >= ZM_RS2_MIN_N fills them all)
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_rsel2_sample, dim3(nimg), dim3(RS2_THREADS), 0, s, B, d_st, d_samp);
    if (d_vbits) hipLaunchKernelGGL((k_rsel2_pass<1, 0>), dim3(grid, nimg), dim3(RS2_THREADS), 0, s, B, d_st, d_seg, d_wcnt, d_vbits);
    else hipLaunchKernelGGL((k_rsel2_pass<0, 0>), dim3(grid, nimg), dim3(RS2_THREADS), 0, s, B, d_st, d_seg, d_wcnt, d_vbits);
    hipLaunchKernelGGL(k_rsel2_finish, dim3(nimg), dim3(RS2_THREADS), 0, s, 0, B, d_st, d_seg, d_wcnt, d_samp, out_dev);
    if (d_vbits) hipLaunchKernelGGL((k_rsel2_pass<2, 1>), dim3(grid, nimg), dim3(RS2_THREADS), 0, s, B, d_st, d_seg, d_wcnt, d_vbits);
    else hipLaunchKernelGGL((k_rsel2_pass<0, 1>), dim3(grid, nimg), dim3(RS2_THREADS), 0, s, B, d_st, d_seg, d_wcnt, d_vbits);
    hipLaunchKernelGGL(k_rsel2_finish, dim3(nimg), dim3(RS2_THREADS), 0, s, 1, B, d_st, d_seg, d_wcnt, d_samp, out_dev);
    ZM_HIP(hipGetLastError());
    return 0;
}
