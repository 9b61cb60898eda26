// The box-OR pre-pass of the mask coadd: per input mask the OR over the 6 x 6 footprint of every pixel as a 16-bit
// plane, read once with 16-byte loads, so that the fused kernel gathers ONE entry per output pixel and frame
// (zuds/astromatic/makecoadd/mask.swarp:25 resamples integer masks with LANCZOS3; the convention here - OR under
// the non-zero taps - is stated in DESIGN.md section 2).
#include "resample_dev.h"

// The planes from a streaming kernel (round 4): the tiled form (k_mask_box in resample.hip, still what a single
// zm_resample call uses) reads a 69 x 21 halo box per
// 64 x 16 tile (1.41 x the mask) through LDS; here a WAVE owns a strip of columns and walks down a band of
// rows, every lane holding CPL consecutive columns: the horizontal OR of a row comes from the next lane(s)
// (cross-lane moves, no LDS tile, no barrier), the vertical OR from a ring of the last NT row results in
// registers - the mask is read once (+ NT - 1 rows per band, + one or two lanes per strip: 1.06 x) with
// 16-byte loads, NT rows in flight.  T = int16_t: a ZTF mask as it lies on disk (ZM_MASKTYPE_I16), half the
// bytes; a negative word stands for its sign extension, i.e. bits above 15: ZM_BOX_RAW.
#define MB_ROWS 121                   // output rows per band: 121 + NT - 1 = 126 = 21 x NT input rows
// CPL columns per lane, one 16-byte load per lane and row: 8 for int16, 4 for int32.  The horizontal OR
// reaches NT - 1 columns ahead: into the next lane (CPL = 8), into the next two (CPL = 4); the last one /
// two lanes of a wave only supply that halo, the next strip owns their columns.
template <typename T> struct mb_row;
template <> struct mb_row<int32_t> {
    enum { CPL = 4, HALO_LANES = 2 };
    static __device__ inline void load(const int32_t* p, bool vec, int nvalid, int32_t v[4]) {
        if (vec && nvalid == 4) {
            const int4 q = *reinterpret_cast<const int4*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = k < nvalid ? p[k] : 0;
        }
    }
};
template <> struct mb_row<int16_t> {
    enum { CPL = 8, HALO_LANES = 1 };
    static __device__ inline void load(const int16_t* p, bool vec, int nvalid, int32_t v[8]) {
        if (vec && nvalid == 8) {
            const int4 q = *reinterpret_cast<const int4*>(p);          // eight words; sign extension below
            const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = (int32_t)(int16_t)(w[k] & 0xffff);
                v[2 * k + 1] = w[k] >> 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = k < nvalid ? (int32_t)p[k] : 0;
        }
    }
};
template <typename T, int NT>
__global__ __launch_bounds__(256) void k_mask_box_rows(const zm_boxjob* __restrict__ jobs) {
    constexpr int CPL = mb_row<T>::CPL, HL = mb_row<T>::HALO_LANES, OWN = 64 - HL;
    static_assert(NT >= 2 && NT - 1 <= CPL * HL, "the horizontal OR reaches into HALO_LANES lanes");
    const zm_boxjob J = jobs[blockIdx.z];
    const int nx = J.nx, ny = J.ny;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * (CPL * OWN) + CPL * lane;             // this lane's columns
    const int y0 = (blockIdx.y * 4 + wv) * MB_ROWS;                  // first output row of this wave's band
    if (blockIdx.x * (CPL * OWN) >= nx || y0 >= ny) return;          // (wave-uniform: the grid covers the largest frame)
    const T* __restrict__ m = reinterpret_cast<const T*>(J.m);
    const bool vec = (nx % CPL) == 0 && (reinterpret_cast<uintptr_t>(m) & 15) == 0;
    const int nvalid = min(max(nx - x, 0), CPL);
    const bool owner = lane < OWN;
    int32_t ring[NT][CPL];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int e = 0; e < CPL; ++e) ring[k][e] = 0;
    bool raw_seen = false;
#pragma unroll 1
    for (int r0 = 0; r0 < MB_ROWS + NT - 1; r0 += NT) {
        if (y0 + r0 >= ny) break;                                    // nothing below the frame contributes
        int32_t a[NT][CPL];
#pragma unroll
        for (int k = 0; k < NT; ++k) {                               // NT rows requested together
            const int y = y0 + r0 + k;
            if (y < ny && nvalid > 0) {
                mb_row<T>::load(m + (size_t)y * nx + x, vec, nvalid, a[k]);
            } else {
#pragma unroll
                for (int e = 0; e < CPL; ++e) a[k][e] = 0;
            }
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int y = y0 + r0 + k;                               // input row; it completes output row y - NT + 1
            int32_t win[CPL + NT - 1];
#pragma unroll
            for (int e = 0; e < CPL; ++e) win[e] = a[k][e];
#pragma unroll
            for (int e = 0; e < NT - 1; ++e)                         // columns x + CPL + e: lane + 1 (+ 2 beyond its CPL)
                win[CPL + e] = __shfl_down(a[k][e % CPL], 1 + e / CPL);
            // h[e] = OR of columns x + e .. x + e + NT - 1 of this row
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                int32_t o = 0;
#pragma unroll
                for (int t = 0; t < NT; ++t) o |= win[e + t];
                ring[k][e] = o;
            }
            const int yo = y - (NT - 1);
            if (yo >= y0 && yo < y0 + MB_ROWS && y < ny && owner) {
                uint16_t en[CPL];
                bool all_in = true;
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    int32_t o = 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) o |= ring[t][e];
                    en[e] = box_entry(o);
                    const bool in = x + e + NT <= nx;
                    all_in = all_in && in;
                    raw_seen = raw_seen || (in && en[e] == ZM_BOX_RAW);
                }
                uint16_t* dst = J.B + (size_t)yo * J.pitch + x;
                if (all_in) {
                    unsigned pk[CPL / 2];
#pragma unroll
                    for (int e = 0; e < CPL / 2; ++e) pk[e] = (unsigned)en[2 * e] | ((unsigned)en[2 * e + 1] << 16);
                    if constexpr (CPL == 8) *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    else *reinterpret_cast<uint2*>(dst) = make_uint2(pk[0], pk[1]);
                } else {
#pragma unroll
                    for (int e = 0; e < CPL; ++e)
                        if (x + e + NT <= nx) dst[e] = en[e];
                }
            }
        }
    }
    if (J.rawflag && __any(raw_seen) && lane == 0) atomicOr(J.rawflag, 1);
}

// int16 masks whose rows are whole 16-byte pieces (nx a multiple of 8, aligned planes - every ZTF mask): the same
// walk on PACKED words.  A lane keeps its eight columns as the four dwords it loaded (two mask words each), a
// strip is 64 lanes x 8 columns = 512 columns = whole 128-byte lines (the kernel above gives a lane to the halo:
// 504-column strips, a seventh strip of 48 columns at 3072, every row load straddling two lines).  The columns
// right of the lane come from the next lane by a DPP wave shift (lane 63: from the first piece of the next strip,
// loaded once per NT rows by NT lanes and handed over by v_readlane as the shift's fill value).  A negative
// int16 word is the sign extension box_entry() turns into ZM_BOX_RAW, i.e. bit 15 of the 16-bit OR: the sliding
// OR works on halves of dwords (f = lo | hi of a pair; even columns: OR of whole pairs; odd columns: hi of the
// first, whole pairs, lo of the last - v_or3_b32 / v_and_or_b32 / v_lshl_or_b32), the vertical OR on the packed
// results, and the entry is o | 0xffff per half whose bit 15 is set.  ~75 vector instructions per row of eight
// columns instead of ~150, no LDS cross-lane traffic.  Same plane, bit for bit (tests/test_mask_i16_gpu.py).
__device__ __forceinline__ uint32_t mb_shl1(uint32_t v, uint32_t fill) {
    // lane i <- lane i + 1 (DPP wave_shl:1); lane 63 keeps `fill`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}
template <int NT>
__global__ __launch_bounds__(256) void k_mask_box_rows16(const zm_boxjob* __restrict__ jobs) {
    static_assert(NT % 2 == 0 && NT >= 2 && NT <= 6, "pairs of columns; the halo is at most three dwords");
    constexpr int HP = NT / 2;                                        // whole pairs in a window
    const zm_boxjob J = jobs[blockIdx.z];
    const int nx = J.nx, ny = J.ny;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xs = blockIdx.x * 512, x = xs + 8 * lane;              // this lane's columns
    const int y0 = (blockIdx.y * 4 + wv) * MB_ROWS;                  // first output row of this wave's band
    if (xs >= nx || y0 >= ny) return;                                // (wave-uniform: the grid covers the largest frame)
    const int16_t* __restrict__ m = reinterpret_cast<const int16_t*>(J.m);
    const bool mine = x < nx;                                        // (nx % 8 == 0: a lane's piece is whole or absent)
    const int xh = xs + 512;                                         // the piece right of the strip
    const bool halo = xh < nx;
    const bool all_in = x + 7 + NT <= nx;                            // every window of this lane lies on the frame
    uint32_t ring[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[k][i] = 0u;
    uint32_t rawm = 0u;
#pragma unroll 1
    for (int r0 = 0; r0 < MB_ROWS + NT - 1; r0 += NT) {
        if (y0 + r0 >= ny) break;                                    // nothing below the frame contributes
        uint4 a[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {                               // NT rows requested together
            const int y = y0 + r0 + k;
            a[k] = make_uint4(0u, 0u, 0u, 0u);
            if (y < ny && mine) a[k] = *reinterpret_cast<const uint4*>(m + (size_t)y * nx + x);
        }
        uint4 hp = make_uint4(0u, 0u, 0u, 0u);                       // lane k: the halo piece of row r0 + k
        if (lane < NT && y0 + r0 + lane < ny && halo)
            hp = *reinterpret_cast<const uint4*>(m + (size_t)(y0 + r0 + lane) * nx + xh);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int y = y0 + r0 + k;                               // input row; it completes output row y - NT + 1
            uint32_t P[7] = {a[k].x, a[k].y, a[k].z, a[k].w, 0u, 0u, 0u};
            const uint32_t hs[3] = {(uint32_t)__builtin_amdgcn_readlane((int)hp.x, k),
                                    (uint32_t)__builtin_amdgcn_readlane((int)hp.y, k),
                                    (uint32_t)__builtin_amdgcn_readlane((int)hp.z, k)};
#pragma unroll
            for (int i = 0; i < HP; ++i) P[4 + i] = mb_shl1(P[i], hs[i]);
            uint32_t hi[7], f[7];
#pragma unroll
            for (int i = 0; i < 4 + HP; ++i) {
                hi[i] = P[i] >> 16;
                f[i] = (P[i] & 0xffffu) | hi[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t he = f[i], ho = hi[i];
#pragma unroll
                for (int t = 1; t < HP; ++t) { he |= f[i + t]; ho |= f[i + t]; }
                ho |= P[i + HP] & 0xffffu;
                ring[k][i] = he | (ho << 16);
            }
            const int yo = y - (NT - 1);
            if (yo >= y0 && yo < y0 + MB_ROWS && y < ny && mine) {
                uint32_t en[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t o = ring[0][i];
#pragma unroll
                    for (int t = 1; t < NT; ++t) o |= ring[t][i];
                    const uint32_t neg = (o >> 15) & 0x00010001u;    // halves with bit 15: a negative word in the window
                    en[i] = o | (neg * 0xffffu);
                    if (all_in) rawm |= neg;
                }
                uint16_t* dst = J.B + (size_t)yo * J.pitch + x;
                if (all_in) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(en[0], en[1], en[2], en[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (x + e + NT <= nx) {
                            const uint16_t v = (uint16_t)(en[e >> 1] >> (16 * (e & 1)));
                            dst[e] = v;
                            rawm |= v == ZM_BOX_RAW ? 1u : 0u;
                        }
                }
            }
        }
    }
    if (J.rawflag && __any(rawm != 0u) && lane == 0) atomicOr(J.rawflag, 1);
}

// 2: int16, rows and planes in whole 16-byte pieces (k_mask_box_rows16; ZM_MASK_BOX=lanes keeps such planes on
// the unpacked kernel: developer A / B); 1: other int16 planes; 0: int32
static int box_kind(const zm_boxjob& b) {
    static const bool unpacked = ZM_DEVENV("ZM_MASK_BOX") && !strcmp(ZM_DEVENV("ZM_MASK_BOX"), "lanes");
    if (!b.is16) return 0;
    const bool pieces = b.nx % 8 == 0 && b.pitch % 8 == 0 && (reinterpret_cast<uintptr_t>(b.m) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(b.B) & 15) == 0;
    return pieces && !unpacked ? 2 : 1;
}

// The box-OR planes depend on the masks only.  after != NULL: the launch goes to the second stream, ordered
// after `after` (an event recorded on the main stream before the caller enqueues the mesh statistics: nothing
// older may still read the planes), and runs BESIDE those statistics - they are bound by their own moment /
// histogram work at 2.5 TB/s, this kernel streams.  *joined receives the event the main stream has to wait
// for before the planes are read (NULL: same stream, nothing to wait for).
int zm_launch_mask_boxes(zm_ctx* ctx, const zm_boxjob* boxes, int nboxes, hipEvent_t after, hipEvent_t* joined,
                         hipEvent_t after2) {
    if (joined) *joined = nullptr;
    if (nboxes == 0) return 0;
    hipEvent_t* ev = nullptr;
    ZM_TRY(zm_get_sync_events(ctx, 9, &ev));
    ZM_HIP(hipEventSynchronize(ev[7]));
    const size_t bb = sizeof(zm_boxjob) * (size_t)nboxes;
    char *pin = nullptr, *dev = nullptr;
    ZM_TRY(ctx->get_pinned("ff_box_h", bb, (void**)&pin));
    ZM_TRY(ctx->get("ff_box", bb, (void**)&dev));
    {
        // int16 jobs of whole 16-byte pieces first (the packed kernel), then the other int16 ones, then int32
        zm_boxjob* pj = reinterpret_cast<zm_boxjob*>(pin);
        int k = 0;
        for (int pass = 2; pass >= 0; --pass)
            for (int i = 0; i < nboxes; ++i)
                if (box_kind(boxes[i]) == pass) pj[k++] = boxes[i];
    }
    // (the scope timers record on the main stream: when this scope is being timed the kernel stays there)
    const bool timed = ctx->timing && (ctx->timing_only.empty() || ctx->timing_only == "mask_box");
    static const bool fork_off = ZM_DEVENV("ZM_FF_FORK") && ZM_DEVENV("ZM_FF_FORK")[0] == '0';
    const bool side = after != nullptr && joined != nullptr && zm_ctx_aux(ctx) != nullptr && !timed && !fork_off;
    hipStream_t s = side ? zm_ctx_aux(ctx) : ctx->stream;
    if (side) ZM_HIP(hipStreamWaitEvent(s, after, 0));
    if (side && after2) ZM_HIP(hipStreamWaitEvent(s, after2, 0));      // (round 6: behind the mesh statistics, see api_coadd.hip)
    ZM_HIP(hipMemcpyAsync(dev, pin, bb, hipMemcpyHostToDevice, s));
    ZM_HIP(hipEventRecord(ev[7], s));
    int mx = 1, my = 1;
    bool any16 = false, any32 = false;
    for (int i = 0; i < nboxes; ++i) {
        mx = std::max(mx, boxes[i].nx);
        my = std::max(my, boxes[i].ny);
        (boxes[i].is16 ? any16 : any32) = true;
    }
    {
        zm_scope_timer t(ctx, "mask_box");
        {
            // one launch per mask type over the jobs of that type (a stack normally has one): the jobs are
            // sorted by type in the staging copy, a launch covers a contiguous range of them
            const unsigned gy = zm_div_up(zm_div_up(my, MB_ROWS), 4);
            int n16 = 0, npk = 0;
            for (int i = 0; i < nboxes; ++i) {
                n16 += boxes[i].is16 ? 1 : 0;
                npk += box_kind(boxes[i]) == 2 ? 1 : 0;
            }
            const zm_boxjob* d = (const zm_boxjob*)dev;
            if (npk)
                hipLaunchKernelGGL((k_mask_box_rows16<6>), dim3(zm_div_up(mx, 512), gy, npk), dim3(256), 0, s, d);
            if (n16 - npk)
                hipLaunchKernelGGL((k_mask_box_rows<int16_t, 6>), dim3(zm_div_up(mx, 8 * 63), gy, n16 - npk), dim3(256), 0, s,
                                   d + npk);
            if (nboxes - n16)
                hipLaunchKernelGGL((k_mask_box_rows<int32_t, 6>), dim3(zm_div_up(mx, 4 * 62), gy, nboxes - n16), dim3(256), 0,
                                   s, d + n16);
        }
    }
    ZM_HIP(hipGetLastError());
    if (side) {
        ZM_HIP(hipEventRecord(ev[8], s));
        *joined = ev[8];
    }
    return 0;
}

// LDS row reads, software-pipelined by hand: the six ds_read_b64 of tap row r + 1 are issued
// before the packed FMAs of row r; `lds_wait` then waits until at most N reads are outstanding
