// Robust statistics of the C-ABI: exact median and 1.4826 x MAD of the unmasked
// pixels (quick_background_estimate, zuds/utils.py:32-53), by a three-pass
// radix select on order-preserving integer keys (11 + 11 + 10 bits).
//
// Everything between the first histogram and the final answer stays on the
// device: a one-workgroup scan kernel turns each histogram into the next key
// prefix (for both middle ranks of an even count at once), so a median + MAD of
// a 9.4 Mpx frame is 6 streaming passes and one 24-byte copy back, batched over
// up to ZM_RS_MAXIMG images per launch (blockIdx.y).
#include "select_dev.h"

#define RS_BINS 2048

// per-image select state (device memory)
struct rs_state {
    uint32_t prefix[2];            // key prefixes of the two middle ranks
    uint32_t pmask;                // bits of the key already fixed
    uint32_t pad_;
    unsigned long long k[2];       // ranks within the current prefix
    unsigned long long count;      // number of selected values
    float centre;                  // MAD: values are |v - centre|
    float median;
    double out[3];                 // median, 1.4826 MAD, count
};

__global__ void k_rsel_init(rs_state* __restrict__ st, unsigned int* __restrict__ hist) {
    const int im = blockIdx.x;
    for (int k = threadIdx.x; k < 2 * RS_BINS; k += blockDim.x) hist[(size_t)im * 2 * RS_BINS + k] = 0;
    if (threadIdx.x == 0) {
        rs_state z;
        memset(&z, 0, sizeof(z));
        st[im] = z;
    }
}

// mode 0: v = img[p]; mode 1: v = |img[p] - centre| (float32 arithmetic, as numpy).
// hist[im][t][bin]: t = 0 counts the keys under prefix[0]; t = 1 those under prefix[1]
// when it differs (the two middle ranks straddle a bin boundary: rare).
// Round 5.  (1) 1 024 threads per workgroup, 256 workgroups (was 256 x 512): a workgroup's histograms are flushed
// with one global atomic per non-empty bin - 512 x 2 048 of them per pass and image were 8 us of a 33 us pass
// (`tools/rs_probe.py`: 214 -> 177 us per median + MAD of two frames).  (2) The validity of a pixel (mask == 0,
// not NaN) is the same in all six passes: the FIRST pass writes it as one bit per pixel - the 64-lane ballot of each
// of a lane's four pixels, four 64-bit words per wave and iteration - and the other five read 0.125 B per pixel
// instead of the 4 B mask (- 10 us: the passes are bound by their histogram work and their launches, not by bytes).
// Measured and not kept: the next iteration's pixels requested ahead (no change); the scan run by the last
// workgroup of a pass to finish instead of a launch of its own (a ticket per image; + 50 us: every thread waits for
// the acknowledgement of its atomics, and the scan reads the histogram past the caches).
// vbits: [image][n / 256 rounded up][4] words, or NULL (masks read every time).  vmode: 1 = masks, and write the
// bits, 2 = bits.
__global__ __launch_bounds__(1024) void k_rsel_hist(const rs_batch B, int vec_ok, int mode, int shift,
                                                    int nbins, const rs_state* __restrict__ st,
                                                    unsigned int* __restrict__ hist,
                                                    unsigned long long* __restrict__ vbits, int vmode) {
    __shared__ unsigned int lh[2][RS_BINS];
    const int im = blockIdx.y;
    const float* __restrict__ img = B.im[im].img;
    const int32_t* __restrict__ mask = B.im[im].mask;
    const rs_state S = st[im];
    const bool two = S.prefix[1] != S.prefix[0];
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) { lh[0][k] = 0; lh[1][k] = 0; }
    __syncthreads();
    int cur = -1;
    unsigned int run = 0;              // run-length aggregation: sky pixels share a bin
    auto put = [&](float v, int32_t m) {
        if (m != 0 || !(v == v)) return;
        if (mode == 1) v = fabsf(v - S.centre);
        const uint32_t key = f2key(v);
        const uint32_t hi = key & S.pmask;
        const int b = (int)((key >> shift) & (uint32_t)(nbins - 1));
        if (hi == S.prefix[0]) {
            if (b == cur) { ++run; }
            else {
                if (run) atomicAdd(&lh[0][cur], run);
                cur = b;
                run = 1;
            }
        } else if (two && hi == S.prefix[1]) {
            atomicAdd(&lh[1][b], 1u);
        }
    };
    const int64_t n = B.n;
    const int64_t n4 = vec_ok ? n / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long* vb = vbits ? vbits + (size_t)im * (size_t)((n4 + 63) / 64) * 4 : nullptr;
    const int lane = threadIdx.x & 63;
    // (with the bit plane whole waves walk the loop together: the bound is rounded up to the wave, lanes beyond n4
    // carry invalid pixels)
    const int64_t n4w = vb ? ((n4 + 63) / 64) * 64 : n4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4w; q += stride) {
        const bool in = q < n4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = reinterpret_cast<const float4*>(img)[q];
        int4 m = make_int4(0, 0, 0, 0);
        if (vb && vmode == 2) {
            // (four words per wave: the same address for every lane - one transaction)
            const unsigned long long* w = vb + (q >> 6) * 4;
            m.x = ((w[0] >> lane) & 1ull) ? 0 : 1;
            m.y = ((w[1] >> lane) & 1ull) ? 0 : 1;
            m.z = ((w[2] >> lane) & 1ull) ? 0 : 1;
            m.w = ((w[3] >> lane) & 1ull) ? 0 : 1;
        } else {
            if (mask && in) m = reinterpret_cast<const int4*>(mask)[q];
            if (!in) m = make_int4(1, 1, 1, 1);
            if (vb) {
                const unsigned long long b0 = __ballot(m.x == 0 && v.x == v.x), b1 = __ballot(m.y == 0 && v.y == v.y);
                const unsigned long long b2 = __ballot(m.z == 0 && v.z == v.z), b3 = __ballot(m.w == 0 && v.w == v.w);
                if (lane == 0) {
                    unsigned long long* w = vb + (q >> 6) * 4;
                    w[0] = b0; w[1] = b1; w[2] = b2; w[3] = b3;
                }
            }
        }
        put(v.x, m.x); put(v.y, m.y); put(v.z, m.z); put(v.w, m.w);
    }
    for (int64_t p = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride)
        put(img[p], mask ? mask[p] : 0);
    if (run) atomicAdd(&lh[0][cur], run);
    __syncthreads();
    unsigned int* h = hist + (size_t)im * 2 * RS_BINS;
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
        if (lh[0][k]) atomicAdd(&h[k], lh[0][k]);
        if (two && lh[1][k]) atomicAdd(&h[RS_BINS + k], lh[1][k]);
    }
}

// One workgroup per image: locate the bins of the two ranks, extend the prefixes,
// clear the histograms for the next pass.  `pass` 0 also fixes the count and the
// ranks; `pass` 2 finishes a select: mode 0 -> the median becomes the MAD centre and
// the state restarts, mode 1 -> the outputs are written.
__global__ __launch_bounds__(256) void k_rsel_scan(int pass, int mode, int shift, int nbins,
                                                   rs_state* __restrict__ st,
                                                   unsigned int* __restrict__ hist) {
    __shared__ unsigned long long wtot[4];
    __shared__ rs_state S;
    const int im = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    unsigned int* h = hist + (size_t)im * 2 * RS_BINS;
    if (tid == 0) S = st[im];
    __syncthreads();
    const bool two = S.prefix[1] != S.prefix[0];
    const int per = nbins / 256;                         // 8 or 4 bins per thread
    // rank k falls into the first run of `per` bins whose cumulative count exceeds it (the last
    // run if none does), then into the first such bin of that run: a block-wide prefix sum
    // finds the run, its owner walks its own bins
    for (int t = 0; t < 2; ++t) {
        const unsigned int* ht = h + (two && t == 1 ? RS_BINS : 0);
        unsigned int hl[8];
        unsigned long long loc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hl[j] = j < per ? ht[tid * per + j] : 0u;
            loc += hl[j];
        }
        unsigned long long inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned long long off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) off += wtot[w];
            tot += wtot[w];
        }
        const unsigned long long excl = off + inc - loc;
        if (pass == 0 && t == 0 && tid == 0) {
            S.count = tot;
            S.k[0] = tot ? (tot - 1) / 2 : 0;
            S.k[1] = tot / 2 < tot ? tot / 2 : (tot ? tot - 1 : 0);
        }
        __syncthreads();
        const unsigned long long k = S.k[t];
        __syncthreads();
        const bool hit = (tid == 255) ? (k >= excl) : (k >= excl && k < excl + loc);
        if (hit) {
            unsigned long long runsum = excl;
            int jsel = per - 1;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (!found && j < per - 1) {
                    if (runsum + hl[j] > k) { jsel = j; found = true; }
                    else runsum += hl[j];
                }
            S.k[t] = k - runsum;
            S.prefix[t] |= (uint32_t)(tid * per + jsel) << shift;
        }
        __syncthreads();
    }
    for (int k = tid; k < 2 * RS_BINS; k += 256) h[k] = 0;
    if (tid == 0) {
        S.pmask |= (uint32_t)(nbins - 1) << shift;
        if (pass == 2) {
            const float a = key2f_dev(S.prefix[0]), b = key2f_dev(S.prefix[1]);
            const float med = S.count ? 0.5f * (a + b) : 0.f;
            if (mode == 0) {
                S.median = med;
                S.centre = med;
                S.out[0] = med;
                S.out[2] = (double)S.count;
            } else {
                S.out[1] = 1.4826 * (double)med;
            }
            S.prefix[0] = S.prefix[1] = 0;
            S.pmask = 0;
            S.k[0] = S.k[1] = 0;
        }
        st[im] = S;
    }
}

__global__ void k_rsel_out(const rs_state* __restrict__ st, int nimg, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t < 3 * nimg) out[t] = st[t / 3].out[t % 3];
}

// median, 1.4826 MAD and count of each image -> out3[3 * nimg] (host), or - out3 == nullptr - left on the
// device in out_dev[3 * nimg] without a copy back or a wait
static int median_mad_batch(zm_ctx* ctx, int nimg, const rs_image* ims, int64_t n, double* out3, double* out_dev = nullptr) {
    ZM_CHECK(nimg >= 1 && nimg <= ZM_RS_MAXIMG, "median_mad_batch: 1..%d images", ZM_RS_MAXIMG);
    rs_state* d_st = nullptr;
    unsigned int* d_hist = nullptr;
    ZM_TRY(ctx->get("rs_state", sizeof(rs_state) * ZM_RS_MAXIMG, (void**)&d_st));
    ZM_TRY(ctx->get("rs_hist", sizeof(unsigned int) * 2 * RS_BINS * ZM_RS_MAXIMG, (void**)&d_hist));
    unsigned long long* d_vbits = nullptr;
    rs_batch B;
    memset(&B, 0, sizeof(B));
    B.n = n;
    int vec_ok = 1;
    for (int i = 0; i < nimg; ++i) {
        B.im[i] = ims[i];
        if (((uintptr_t)ims[i].img & 15) || ((uintptr_t)ims[i].mask & 15)) vec_ok = 0;
    }
    // one workgroup of 1 024 threads per CU (round 5; rounds 1 - 4: two of 256 per CU; ZM_RS_THREADS / ZM_RS_GRID for A / B:
    // 256 x 512 214 us, 512 x 256 196, 512 x 512 191, 1024 x 128 182, 1024 x 256 177, 1024 x 64 286 per median + MAD of two frames)
    static const int rs_threads = ZM_DEVENV("ZM_RS_THREADS") ? atoi(ZM_DEVENV("ZM_RS_THREADS")) : 1024;
    static const int rs_grid = ZM_DEVENV("ZM_RS_GRID") ? atoi(ZM_DEVENV("ZM_RS_GRID")) : 256;
    int grid = (int)std::min<int64_t>((n / 4 + rs_threads - 1) / rs_threads, rs_grid);
    if (grid < 1) grid = 1;
    // the validity bit plane: where the vector path runs and there is a mask to save (ZM_RS_BITS=0: masks every pass)
    bool any_mask = false;
    for (int i = 0; i < nimg; ++i) any_mask = any_mask || ims[i].mask != nullptr;
    static const bool bits_off = ZM_DEVENV("ZM_RS_BITS") && ZM_DEVENV("ZM_RS_BITS")[0] == '0';
    if (vec_ok && any_mask && n >= 4096 && !bits_off)
        ZM_TRY(ctx->get("rs_vbits", sizeof(unsigned long long) * 4 * (size_t)((n / 4 + 63) / 64) * ZM_RS_MAXIMG, (void**)&d_vbits));
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    hipStream_t s = ctx->stream;
    // Round 6: frames of a megapixel and more go through the sample-bracketed select (select_bracket.hip: two
    // streaming passes instead of six, the same bits); small or unaligned inputs keep the three-pass form below.
    static const bool classic = ZM_DEVENV("ZM_RS_CLASSIC") && ZM_DEVENV("ZM_RS_CLASSIC")[0] == '1';
    if (vec_ok && n >= ZM_RS2_MIN_N && !classic) {
        double* d_out = out_dev;
        if (!d_out) ZM_TRY(ctx->get("rs2_out", sizeof(double) * 3 * ZM_RS_MAXIMG, (void**)&d_out));
        {
            zm_scope_timer t(ctx, "median_mad");
            ZM_TRY(zm_rs2_median_mad(ctx, nimg, B, d_vbits, d_out));
        }
        if (!out3) return 0;
        double* h_out = nullptr;
        ZM_TRY(ctx->get_pinned("rs2_out_h", sizeof(double) * 3 * ZM_RS_MAXIMG, (void**)&h_out));
        ZM_HIP(hipMemcpyAsync(h_out, d_out, sizeof(double) * 3 * nimg, hipMemcpyDeviceToHost, s));
        ZM_HIP(hipStreamSynchronize(s));
        for (int k = 0; k < 3 * nimg; ++k) out3[k] = h_out[k];
        return 0;
    }
    {
        zm_scope_timer t(ctx, "median_mad");
        hipLaunchKernelGGL(k_rsel_init, dim3(nimg), dim3(256), 0, s, d_st, d_hist);
        for (int mode = 0; mode < 2; ++mode)
            for (int pass = 0; pass < 3; ++pass) {
                const int nb = 1 << bits[pass];
                hipLaunchKernelGGL(k_rsel_hist, dim3(grid, nimg), dim3(rs_threads), 0, s, B, vec_ok, mode,
                                   shifts[pass], nb, d_st, d_hist, d_vbits, (mode == 0 && pass == 0) ? 1 : 2);
                hipLaunchKernelGGL(k_rsel_scan, dim3(nimg), dim3(256), 0, s, pass, mode, shifts[pass], nb,
                                   d_st, d_hist);
            }
        ZM_HIP(hipGetLastError());
    }
    if (!out3) {
        hipLaunchKernelGGL(k_rsel_out, dim3(1), dim3(64), 0, s, d_st, nimg, out_dev);
        ZM_HIP(hipGetLastError());
        return 0;
    }
    rs_state* h_st = nullptr;
    ZM_TRY(ctx->get_pinned("rs_state_h", sizeof(rs_state) * ZM_RS_MAXIMG, (void**)&h_st));
    ZM_HIP(hipMemcpyAsync(h_st, d_st, sizeof(rs_state) * nimg, hipMemcpyDeviceToHost, s));
    ZM_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < nimg; ++i)
        for (int k = 0; k < 3; ++k) out3[3 * i + k] = h_st[i].out[k];
    return 0;
}

extern "C" int zm_median_mad_dev(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n,
                                 double* out_median, double* out_mad_sigma) {
    ZM_CHECK(ctx && img && out_median && out_mad_sigma, "zm_median_mad_dev: null argument");
    ZM_CHECK(n > 0, "zm_median_mad_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    rs_image im = {img, mask};
    double o[3];
    ZM_TRY(median_mad_batch(ctx, 1, &im, n, o));
    ZM_CHECK(o[2] > 0, "zm_median_mad: every pixel is masked");
    *out_median = o[0];
    *out_mad_sigma = o[1];
    return 0;
}

// Two images of the same size in one set of launches: the pair of
// quick_background_estimate calls of prepare_hotpants (zuds/hotpants.py:65-67).
extern "C" int zm_median_mad2_dev(zm_ctx* ctx, const float* img_a, const int32_t* mask_a,
                                  const float* img_b, const int32_t* mask_b, int64_t n,
                                  double* out4) {
    ZM_CHECK(ctx && img_a && img_b && out4, "zm_median_mad2_dev: null argument");
    ZM_CHECK(n > 0, "zm_median_mad2_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    rs_image im[2] = {{img_a, mask_a}, {img_b, mask_b}};
    double o[6];
    ZM_TRY(median_mad_batch(ctx, 2, im, n, o));
    ZM_CHECK(o[2] > 0 && o[5] > 0, "zm_median_mad2: every pixel is masked");
    out4[0] = o[0]; out4[1] = o[1]; out4[2] = o[3]; out4[3] = o[4];
    return 0;
}

extern "C" int zm_median_mad2_async_dev(zm_ctx* ctx, const float* img_a, const int32_t* mask_a,
                                        const float* img_b, const int32_t* mask_b, int64_t n,
                                        double* out6_dev) {
    ZM_CHECK(ctx && img_a && img_b && out6_dev, "zm_median_mad2_async_dev: null argument");
    ZM_CHECK(n > 0, "zm_median_mad2_async_dev: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    rs_image im[2] = {{img_a, mask_a}, {img_b, mask_b}};
    return median_mad_batch(ctx, 2, im, n, nullptr, out6_dev);
}

extern "C" int zm_median_mad(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n,
                             double* out_median, double* out_mad_sigma) {
    ZM_CHECK(ctx && img && out_median && out_mad_sigma, "zm_median_mad: null argument");
    ZM_CHECK(n > 0, "zm_median_mad: empty image");
    ZM_HIP(hipSetDevice(ctx->device));
    float* d_img = nullptr;
    int32_t* d_mask = nullptr;
    ZM_TRY(ctx->get("h_img", (size_t)n * 4, (void**)&d_img));
    ZM_HIP(hipMemcpyAsync(d_img, img, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (mask) {
        ZM_TRY(ctx->get("h_mask", (size_t)n * 4, (void**)&d_mask));
        ZM_HIP(hipMemcpyAsync(d_mask, mask, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    return zm_median_mad_dev(ctx, d_img, d_mask, n, out_median, out_mad_sigma);
}
