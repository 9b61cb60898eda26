/*
 * zudsmi.h - C-ABI of libzudsmi.so, the MI355X (gfx950) engine behind the
 * ZUDS coadd / subtraction object API.
 *
 * The reference pipeline has no FFI on this path: it builds command strings
 * and runs `subprocess.check_call` on SWarp, SExtractor and hotpants,
 * exchanging FITS files.  Every entry point below replaces one of those
 * process boundaries (cited per function, paths relative to the reference
 * tree).  INTEGRATION.md shows the ctypes stubs a maintainer adds on the
 * reference side.
 *
 * Conventions
 *  - all functions return 0 on success, non-zero on failure; the message is
 *    available per thread from zm_last_error();
 *  - images are C-contiguous, native-endian, row-major [ny][nx]; pixel (1,1)
 *    of FITS is element [0][0];
 *  - pointers passed to the host entry points are borrowed for the call;
 *    outputs are caller-allocated;
 *  - a zm_ctx owns one HIP stream and all device scratch; one ctx per
 *    process/GPU, not thread-safe;
 *  - `*_dev` entry points take device pointers (hipMalloc / torch storage)
 *    and only enqueue work on the ctx stream.
 */
#ifndef ZUDSMI_H
#define ZUDSMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zm_ctx zm_ctx;

#define ZM_NPV 40

/* TAN (flags = 0) or TPV (flags & 1) WCS; FITS 1-based pixel convention.
 * Replaces astropy.wcs.WCS(header) (zuds/fitsfile.py:233-238) and the
 * `.head` files handed to SWarp (zuds/swarp.py:114-133). */
typedef struct zm_wcs {
    double crpix[2];
    double crval[2];      /* degrees */
    double cd[4];         /* CD1_1 CD1_2 CD2_1 CD2_2, degrees / pixel */
    double pv1[ZM_NPV];   /* PV1_k; only read when flags & 1 */
    double pv2[ZM_NPV];
    int32_t naxis[2];     /* NAXIS1 (x), NAXIS2 (y) */
    int32_t flags;        /* bit 0: TPV distortion present */
    int32_t pad_;
} zm_wcs;

enum { ZM_RESAMPLE_NEAREST = 0, ZM_RESAMPLE_BILINEAR = 1, ZM_RESAMPLE_LANCZOS3 = 3 };
enum { ZM_COMBINE_WEIGHTED = 0, ZM_COMBINE_MEDIAN = 1, ZM_COMBINE_CLIPPED = 2,
       ZM_COMBINE_AVERAGE = 3 };
enum { ZM_MASK_AND = 0, ZM_MASK_OR = 1 };

/* ---- context ---------------------------------------------------------- */
int zm_ctx_create(int device, zm_ctx** out);
int zm_ctx_destroy(zm_ctx* ctx);
/* A context bound to the caller's hipStream_t from the start: it never creates a stream of its own (its second stream
 * is made on first use).  For processes that run many contexts side by side on one GPU (scripts/donightly.py:21-40
 * runs one process per job; here: one context per chain). */
int zm_ctx_create_on_stream(int device, void* hip_stream, zm_ctx** out);
/* Use an external hipStream_t (e.g. torch's current stream); NULL = own stream.  Binding the stream already bound is
 * free; a change orders the new stream behind the work enqueued on the old one with an event (the context's scratch
 * is shared) - the host never waits (round 6; rounds 1 - 5 synchronised the old stream here). */
int zm_ctx_set_stream(zm_ctx* ctx, void* hip_stream);
int zm_ctx_synchronize(zm_ctx* ctx);
/* Several contexts (one host thread each) may subtract on one GPU at the same time - the
 * reference runs 64 independent `hotpants` processes per node (nersc/controller.py:101,
 * scripts/donightly.py:21-40).  The kernel-fit solver of a context that has the GPU to itself
 * (nctx = 1, the default) keeps ~230 workgroups resident behind in-kernel barriers - the short
 * form; declaring nctx >= 2 (1 .. 64) switches the context to the one-workgroup-per-region
 * form, which claims nothing (k_chol_tp).  Results do not depend on the share: same bits. */
int zm_ctx_set_share(zm_ctx* ctx, int nctx);
const char* zm_last_error(void);
const char* zm_version(void);
/* What the context last did / how the library was built, by name (-> *out; unknown name: error):
 *   "fused_form"  the fused resample -> coadd kernel the last coadd of this context launched: 0 none yet,
 *                 1 k_coadd_fused_dma, 2 k_coadd_fused_own (bench.py labels its roofline with it)
 *   "dev_build"   1 when the library was compiled with -DZM_DEV (developer switches are read), else 0
 * No counterpart in the reference (its tools report through their logs, zuds/astromatic/makecoadd/default.swarp:111). */
int zm_ctx_query(zm_ctx* ctx, const char* what, int64_t* out);

/* SWarp's own edge and mask conventions as options of every resample / coadd call of this context (defaults: the
 * conventions of DESIGN.md section 2; both options are for pixel-for-pixel comparisons with real SWarp products and
 * make a stack take the materialised path - INTEGRATION.md says when to pick them):
 *   edge           ZM_EDGE_ZERO      an output pixel with a non-zero tap off the input frame gets value 0 / weight 0
 *                  ZM_EDGE_TRUNCATE  SWarp's truncated kernel: a pixel whose position lies on the frame is computed
 *                                    from the taps that are on it (zuds/astromatic/makecoadd/default.swarp:42-67)
 *   mask_resample  ZM_MASKRES_OR             OR of the mask words under the non-zero taps
 *                  ZM_MASKRES_LANCZOS_ROUND  the mask interpolated like an image and rounded, as SWarp does with
 *                                            mask.swarp (zuds/astromatic/makecoadd/mask.swarp:25, zuds/swarp.py:141-152) */
#define ZM_EDGE_ZERO 0
#define ZM_EDGE_TRUNCATE 1
#define ZM_MASKRES_OR 0
#define ZM_MASKRES_LANCZOS_ROUND 1
int zm_ctx_set_conventions(zm_ctx* ctx, int edge, int mask_resample);

/* ---- WCS helpers (host, fp64) ------------------------------------------ */
/* Output grid of a coadd: SWarp CENTER_TYPE ALL / PIXELSCALE_TYPE MEDIAN /
 * IMAGE_SIZE 0 (zuds/astromatic/makecoadd/default.swarp:40-49). */
int zm_autogrid(int nframes, const zm_wcs* wcs, zm_wcs* out);
/* pixel -> sky and sky -> pixel for n points (degrees, 1-based pixels). */
int zm_wcs_pix2sky(const zm_wcs* w, int n, const double* x, const double* y,
                   double* ra, double* dec);
int zm_wcs_sky2pix(const zm_wcs* w, int n, const double* ra, const double* dec,
                   double* x, double* y);
/* input-frame pixel of output pixels (the inverse map SWarp evaluates). */
int zm_wcs_map(const zm_wcs* wout, const zm_wcs* win, int n, const double* xo,
               const double* yo, double* xi, double* yi);
/* FLXSCALE x fixed pixel-area ratio (FSCALASTRO_TYPE FIXED, default.swarp:58). */
int zm_flux_scale(const zm_wcs* win, const zm_wcs* wout, double flxscale,
                  double* out);

/* ---- resample one image onto another grid ------------------------------ */
/* Replaces the SWarp run of run_align (zuds/swarp.py:157-204; HasWCS.aligned_to
 * zuds/fitsfile.py:290-314): Lanczos-3 by default, no background subtraction,
 * `wgt` NULL = WEIGHT_TYPE NONE.  out_wgt == 0 marks no-data pixels (bit 16,
 * zuds/mask.py:26-33).  mask/out_mask may be NULL. */
int zm_resample(zm_ctx* ctx, const float* img, const float* wgt,
                const int32_t* mask, const zm_wcs* win, const zm_wcs* wout,
                int kernel, double fscale, float* out_img, float* out_wgt,
                int32_t* out_mask);

/* ---- mesh background ----------------------------------------------------- */
/* Replaces `sex ... -CHECKIMAGE_TYPE BACKGROUND,BACKGROUND_RMS,-BACKGROUND`
 * (zuds/sextractor.py:110-150; used by zuds/hotpants.py:28 and
 * zuds/image.py:206) and SWarp's SUBTRACT_BACK (default.swarp:77-88).
 * wgt NULL = no weighting; any of out_bkg/out_rms/out_sub may be NULL.
 * out_stats[0..1] = global (backmean, backsig) = medians of the mesh maps. */
int zm_background(zm_ctx* ctx, const float* img, const float* wgt, int nx,
                  int ny, int mesh, int filtersize, float* out_bkg,
                  float* out_rms, float* out_sub, double* out_stats);

/* ---- coadd ---------------------------------------------------------------- */
/* Mask planes are integer bit masks (zuds/mask.py:26-72).  ZTF mask files are BITPIX 16: such a plane is
 * handed over as it is (mask_type = ZM_MASKTYPE_I16, `mask` points at int16_t) and is read as int16 by the
 * kernels - half the bytes over PCIe and from HBM; its values mean what numpy's astype(int32) would make
 * of them (sign extension).  mask_type = 0 (a zero-initialised struct): int32_t, as before. */
#define ZM_MASKTYPE_I32 0
#define ZM_MASKTYPE_I16 1
typedef struct zm_frame {
    const float* img;       /* [ny][nx] */
    const float* wgt;       /* inverse variance, NULL = WEIGHT_TYPE NONE */
    const void* mask;       /* int32_t (or int16_t, see mask_type) [ny][nx]; NULL = no mask */
    zm_wcs wcs;
    double flxscale;        /* FLXSCALE = 10^(-0.4 (MAGZP - 25)), zuds/swarp.py:31 */
    int32_t mask_type;      /* ZM_MASKTYPE_* */
    int32_t pad_;
} zm_frame;

typedef struct zm_coadd_params {
    int32_t combine;          /* ZM_COMBINE_*  (COMBINE_TYPE, default CLIPPED) */
    int32_t mask_combine;     /* ZM_MASK_*     (mask.swarp:25 AND; swarp.py:141 OR) */
    int32_t resample;         /* ZM_RESAMPLE_* (RESAMPLING_TYPE LANCZOS3) */
    int32_t subtract_back;    /* SUBTRACT_BACK Y */
    int32_t back_size;        /* -BACK_SIZE 128 (zuds/swarp.py:69) */
    int32_t back_filtersize;  /* BACK_FILTERSIZE 3 */
    int32_t rescale_weights;  /* RESCALE_WEIGHTS Y */
    int32_t pad_;
    double clip_sigma;        /* CLIP_SIGMA 4.0 */
    double clip_ampfrac;      /* CLIP_AMPFRAC 0.3 */
    double weight_thresh;     /* WEIGHT_THRESH 1e-30 */
} zm_coadd_params;

void zm_coadd_params_default(zm_coadd_params* p);

/* Replaces both SWarp runs of _coadd_from_images (zuds/coadd.py:126-163):
 * science coadd (prepare_swarp_sci, zuds/swarp.py:20-80) and mask coadd
 * (prepare_swarp_mask, zuds/swarp.py:83-104) on the grid `wout`.
 * out_mask may be NULL when no frame carries a mask; out_mask_wgt (may be NULL)
 * is the coverage map of the mask coadd (mskoutweightname, zuds/coadd.py:146-147)
 * from which the caller sets bit 16 (zuds/mask.py:26-33). */
int zm_coadd(zm_ctx* ctx, int nframes, const zm_frame* frames,
             const zm_wcs* wout, const zm_coadd_params* params, float* out_img,
             float* out_wgt, int32_t* out_mask, float* out_mask_wgt);

/* ---- subtraction ------------------------------------------------------------ */
typedef struct zm_hp_params {
    /* flags emitted by prepare_hotpants (zuds/hotpants.py:77-93) */
    double tu, tl, iu, il;    /* -tu -tl -iu -il valid data range */
    double r;                 /* -r   kernel half width (2.5 SEEING) */
    double rss;               /* -rss substamp half width (6 SEEING) */
    double fin;               /* -fin noise fill value (BIG_RMS) */
    double fi;                /* diff fill value, hotpants default 1e-30 */
    int32_t nsx, nsy;         /* -nsx -nsy stamps per region */
    int32_t nrx, nry;         /* -nrx -nry regions */
    int32_t ko, bgo;          /* -ko -bgo spatial orders (reference: 4, 0) */
    int32_t nss;              /* substamps per stamp (hotpants default 3) */
    int32_t normalize;        /* 0: -n i (reference), 1: -n t */
    double ft;                /* -ft  substamp threshold in sigma (20) */
    double ks;                /* -ks  stamp rejection sigma (2.0) */
    int32_t ngauss;           /* 3 */
    int32_t deg[4];           /* 6 4 2 */
    int32_t pad_[3];
    double sigma[4];          /* 0.7 1.5 3.0 */
    /* Round 4: lower data limits taken on the device.  When limits_dev is not NULL it points at the six doubles
     * zm_median_mad2_async_dev left in device memory ({median, 1.4826 MAD, count} of the science frame, then of
     * the template), and the subtraction uses il = median_a - limits_nsigma * sigma_a, tl = median_b -
     * limits_nsigma * sigma_b (zuds/hotpants.py:65-72: nsigma = 10) instead of the `il` / `tl` fields above:
     * the two quick_background_estimate calls and the fit are enqueued back to back, without the host reading
     * the estimates in between.  Device entry points only (zm_subtract_dev); NULL (zm_hp_params_default): the
     * fields above. */
    const double* limits_dev;
    double limits_nsigma;
    /* Round 4: the step behind hotpants folded into the call.  The reference sets bit 17 of the subtraction's mask
     * where hotpants left its fill value (zuds/subtraction.py:167-177: `submask[sub.data == 1e-30] |= 2**17`); when
     * flag_mask_dev is not NULL (device entry points only) the call enqueues exactly that - mask |= flag_bit where
     * out_diff == 1e-30f - behind the convolution, before it waits for the fit summary, instead of the caller
     * launching zm_mask_flag_dev after the call has returned (a launch with the GPU idle).  NULL: nothing. */
    int32_t* flag_mask_dev;
    int32_t flag_bit;
    /* Round 6: 1 = zm_subtract_dev returns when the LAST REJECTION ROUND has been seen by the host - the convolution,
     * the bit-17 pass and the copy of the fit summary are enqueued behind it, not waited for.  out_info then carries
     * status = ZM_HP_PENDING (niter, ncoeff, retries are final); zm_subtract_info(ctx, &info) waits for the rest and
     * fills it in.  The caller's next launches on the stream (the next coadd, the next job's alignment) are enqueued
     * while the convolution runs.  A barrier time-out of the solver makes the call fall back to the synchronous
     * form (the fit is repeated, as ever).  0 (zm_hp_params_default): wait, as rounds 1 - 5 did.  Device entry
     * point only; zm_subtract ignores it. */
    int32_t async_info;
} zm_hp_params;

void zm_hp_params_default(zm_hp_params* p);

typedef struct zm_hp_info {
    int32_t nstamps_total, nstamps_used;
    int32_t niter, ncoeff;
    double kernel_sum;        /* mean kernel sum over regions */
    double chi2;              /* mean figure of merit of the used stamps */
    int32_t nmasked;          /* output pixels filled with `fi` */
    int32_t status;           /* 0, or ZM_HP_* bits */
    int32_t nunsolved;        /* regions without a usable fit (filled with `fi`) */
    int32_t retries;          /* fits repeated after a solver barrier timed out (see below) */
} zm_hp_info;
/* status bits.  The reference sees a failed hotpants as a non-zero exit (CalledProcessError,
 * zuds/subtraction.py:162); here: ZM_HP_UNSOLVED = a region of the fit had no usable stamps or a
 * normal matrix that is not positive definite - a property of the data, the call still returns 0
 * and the region carries the fill value.  ZM_HP_TIMEOUT is reserved (never set): a barrier of the
 * many-workgroup factorisation that times out makes the fit repeat on the one-workgroup form, which
 * has no barrier that could time out - `retries` counts such repeats. */
#define ZM_HP_UNSOLVED 1
#define ZM_HP_TIMEOUT 2
#define ZM_HP_PENDING 4       /* zm_hp_params.async_info: the summary is still on its way - zm_subtract_info */

/* Replaces the hotpants run of Subtraction.from_images
 * (zuds/subtraction.py:144-162; flags zuds/hotpants.py:77-93): convolve the
 * template (-c t), D = I - (T (x) K + bg), noise = sqrt(sI^2 + sT^2 (x) K^2),
 * masked pixels filled with fi / fin (zuds/subtraction.py:170-171).
 * sci/ref and their rms maps are on the same grid; bpm is the boolean
 * bad-pixel map (zuds/subtraction.py:141-142), may be NULL. */
int zm_subtract(zm_ctx* ctx, const float* sci, const float* sci_rms,
                const float* ref, const float* ref_rms, const uint8_t* bpm,
                int nx, int ny, const zm_hp_params* params, float* out_diff,
                float* out_rms, zm_hp_info* out_info);

/* ---- robust statistics ---------------------------------------------------- */
/* Replaces quick_background_estimate (zuds/utils.py:32-53): median and
 * 1.4826 MAD of the pixels whose mask is 0 (mask NULL = all). */
int zm_median_mad(zm_ctx* ctx, const float* img, const int32_t* mask, int64_t n,
                  double* out_median, double* out_mad_sigma);
int zm_median_mad_dev(zm_ctx* ctx, const float* img, const int32_t* mask,
                      int64_t n, double* out_median, double* out_mad_sigma);
/* The two estimates prepare_hotpants needs (science and aligned reference,
 * zuds/hotpants.py:65-67) in one set of launches: out4 = {median_a, mad_a,
 * median_b, mad_b}.  Both images have n pixels. */
int zm_median_mad2_dev(zm_ctx* ctx, const float* img_a, const int32_t* mask_a,
                       const float* img_b, const int32_t* mask_b, int64_t n,
                       double* out4);
/* The same two estimates left on the device (round 4): out6_dev receives {median_a, sigma_a, count_a, median_b,
 * sigma_b, count_b} as doubles when the stream reaches that point; nothing is copied back, nothing is waited
 * for.  A frame without a single valid pixel has count 0 (the host entry points raise; here the caller looks
 * at the counts when it reads the estimates).  Consumer: zm_hp_params.limits_dev. */
int zm_median_mad2_async_dev(zm_ctx* ctx, const float* img_a, const int32_t* mask_a,
                             const float* img_b, const int32_t* mask_b, int64_t n, double* out6_dev);

/* ---- forced aperture photometry ------------------------------------------- */
/* Replaces photutils.aperture_photometry(method='exact') + the bounding-box flag
 * OR of raw_aperture_photometry / aperture_photometry (zuds/photometry.py:61-113,
 * 116-249): x, y are 0-based pixel positions, radius in pixels (APERTURE_RADIUS = 3,
 * zuds/constants.py:14); flux = sum(img frac), err = sqrt(sum(rms^2 frac)),
 * flags = OR of mask over the aperture bounding box.  rms / mask may be NULL. */
int zm_aperture_photometry(zm_ctx* ctx, const float* img, const float* rms,
                           const int32_t* mask, int nx, int ny, int npos,
                           const double* x, const double* y, double radius,
                           double* out_flux, double* out_err, int32_t* out_flags);
int zm_aperture_photometry_dev(zm_ctx* ctx, const float* img, const float* rms,
                               const int32_t* mask, int nx, int ny, int npos,
                               const double* x, const double* y, double radius,
                               double* out_flux, double* out_err, int32_t* out_flags);

/* ---- device-pointer entry points (bench, multi-GPU, pipelines) ----------- */
/* Same arithmetic as above on device-resident buffers; enqueue only. */
typedef struct zm_dframe {
    const float* img;       /* device */
    const float* wgt;       /* device or NULL */
    const void* mask;       /* device int32_t / int16_t plane (mask_type) or NULL */
    zm_wcs wcs;
    double flxscale;
    int32_t mask_type;      /* ZM_MASKTYPE_* */
    int32_t pad_;
} zm_dframe;

/* Resample + combine nframes device frames; outputs are device planes.
 * If `partial` != 0 the outputs are the partial sums S1 = sum(w v) (out_img)
 * and S0 = sum(w) (out_wgt) of a WEIGHTED coadd, ready for an RCCL
 * all-reduce across ranks followed by zm_coadd_finalize_dev. */
int zm_coadd_dev(zm_ctx* ctx, int nframes, const zm_dframe* frames,
                 const zm_wcs* wout, const zm_coadd_params* params,
                 int partial, float* out_img, float* out_wgt, int32_t* out_mask,
                 float* out_mask_wgt);
int zm_coadd_finalize_dev(zm_ctx* ctx, float* s1_to_img, const float* s0,
                          int64_t npix);
/* Mask coadd across ranks: zm_coadd_dev(partial = 1) leaves -1 ("no frame of this rank
 * covers the pixel") in out_mask instead of finalising it.  zm_mask_accum_dev folds
 * another partial mask in (AND / OR over the covering frames, -1 = identity; first != 0
 * initialises acc from m), zm_mask_finalize_dev turns the marker into 0 and writes the
 * coverage plane (cov may be NULL) - the mask SWarp run of zuds/swarp.py:83-104 sharded
 * by frame. */
int zm_mask_accum_dev(zm_ctx* ctx, int32_t* acc, const int32_t* m, int64_t npix,
                      int kind, int first);
int zm_mask_finalize_dev(zm_ctx* ctx, int32_t* acc, float* cov, int64_t npix);
/* ---- multi-GPU: the exchange step of a frame-sharded coadd on RCCL (one process per GPU) ----
 * The reference has no collective: it shards by job (zuds/mpi.py:36-64, nersc/controller.py:101);
 * BASELINE config 4 shards the FRAMES of one stack, and these calls are the reduce that adds.
 * librccl is opened on first use.  zm_comm_unique_id on rank 0, the 128 bytes handed to every rank
 * by the launcher (MPI broadcast, torch.distributed, a file), zm_comm_init on every rank.
 * zm_coadd_reduce_dev: sum-reduce of S1 / S0, the two planes of ONE buffer of 2 npix floats as
 * zm_coadd_dev(partial = 1) fills them when out_wgt == out_img + npix; then zm_coadd_finalize_dev.
 * zm_mask_reduce_dev: the partial masks (-1 marker) of all ranks folded with AND / OR by row
 * bands, every rank ends with the finalised mask (cov may be NULL).  All enqueue on the stream.
 * A communicator holds at most ZM_COMM_MAX_RANKS (64) ranks: zm_comm_init refuses more. */
#define ZM_COMM_ID_BYTES 128
typedef struct zm_comm zm_comm;
int zm_comm_unique_id(void* id128);
int zm_comm_init(zm_ctx* ctx, int nranks, int rank, const void* id128, zm_comm** out);
int zm_comm_destroy(zm_comm* comm);
int zm_coadd_reduce_dev(zm_ctx* ctx, zm_comm* comm, float* s1s0, int64_t npix);
int zm_mask_reduce_dev(zm_ctx* ctx, zm_comm* comm, int32_t* mask, int nx, int ny, int kind,
                       float* cov);
/* Host arithmetic of the banded schedule, callable without a GPU (tests replay it for every rank):
 * zm_comm_band_bounds: rows [bounds[g], bounds[g + 1]) belong to rank g, g < world - the split of
 * np.array_split / zuds/mpi.py:36-64 (bounds: world + 1 entries).
 * zm_comm_mask_plan: what zm_mask_reduce_dev sends / receives / gathers for one rank, in elements. */
#define ZM_COMM_MAX_RANKS 64
typedef struct {
    int32_t world, rank;
    int64_t band_px;                          /* slot size: rows of the largest band x nx */
    int64_t my_px;                            /* this rank's band */
    int64_t send_off[ZM_COMM_MAX_RANKS];      /* band g of this rank's plane (offset, count) -> rank g */
    int64_t send_cnt[ZM_COMM_MAX_RANKS];
    int64_t recv_off[ZM_COMM_MAX_RANKS];      /* slot of rank g in the receive buffer, my_px elements each */
    int64_t recv_cnt[ZM_COMM_MAX_RANKS];
    int64_t gather_off[ZM_COMM_MAX_RANKS];    /* folded band of rank g in the all-gather buffer */
} zm_mask_plan;
int zm_comm_band_bounds(int nrows, int world, int32_t* bounds);
int zm_comm_mask_plan(int nx, int ny, int world, int rank, zm_mask_plan* plan);
/* Resample the frames to `wout` into a resident stack [nframes][ony][onx][2]
 * of (value, weight) pairs (the CLIPPED multi-GPU exchange operates on it).
 * out_mask_partial (may be NULL): the mask coadd of these frames with the -1 marker
 * left in, as zm_coadd_dev(partial = 1) leaves it. */
int zm_resample_stack_dev(zm_ctx* ctx, int nframes, const zm_dframe* frames,
                          const zm_wcs* wout, const zm_coadd_params* params,
                          float* stack, int32_t* out_mask_partial);
/* Combine a resident stack; rows [row0, row0+nrows) of every frame. */
int zm_combine_stack_dev(zm_ctx* ctx, int nframes, const float* stack,
                         int64_t frame_stride_px, int64_t npix,
                         const zm_coadd_params* params, float* out_img,
                         float* out_wgt);
int zm_subtract_dev(zm_ctx* ctx, const float* sci, const float* sci_rms,
                    const float* ref, const float* ref_rms, const uint8_t* bpm,
                    int nx, int ny, const zm_hp_params* params, float* out_diff,
                    float* out_rms, zm_hp_info* out_info_host);
/* The summary of the subtraction this context last ran with zm_hp_params.async_info = 1: waits for its convolution
 * and the copy of the summary, then fills out_info as the synchronous call would have.  An error when nothing is
 * pending.  (What `hotpants` prints at its end and Subtraction.from_images keeps in the header:
 * zuds/subtraction.py:144-177.) */
int zm_subtract_info(zm_ctx* ctx, zm_hp_info* out_info);
/* Many subtractions, one call (round 4; the reference runs one hotpants process per job, 64 per node:
 * nersc/controller.py:101, scripts/donightly.py:21-40 -> zuds/subtraction.py:144-162).  Every job is what
 * zm_subtract_dev takes - planes of nx x ny pixels in device memory, its own flags (`params`: the data limits
 * and fill values may differ from job to job; everything that shapes the fit - half widths, basis, orders,
 * regions, stamps, thresholds - has to agree, else the call returns non-zero) - and gets the same bits.  The
 * kernel fit of all jobs runs as ONE chain of launches with the job as a grid dimension (a job that has
 * converged rides along as empty workgroups), on the one-workgroup-per-region factorisation; validity masks,
 * stamp search and the final convolution are enqueued job after job.  out_info_host: njobs entries.  A
 * configuration the batched kernels do not cover (more than 15 spatial kernel terms, more than 960 unknowns or
 * 256 stamps per region) runs job by job through zm_subtract_dev.  1 <= njobs <= ZM_SUB_BATCH_MAX. */
#define ZM_SUB_BATCH_MAX 64
typedef struct zm_sub_job {
    const float* sci;          /* background-subtracted science frame (+ pedestal) */
    const float* sci_rms;
    const float* ref;          /* template on the science grid */
    const float* ref_rms;
    const uint8_t* bpm;        /* boolean bad-pixel map or NULL */
    const zm_hp_params* params;
    float* out_diff;
    float* out_rms;
} zm_sub_job;
int zm_subtract_batch_dev(zm_ctx* ctx, int njobs, const zm_sub_job* jobs, int nx, int ny,
                          zm_hp_info* out_info_host);
/* The same on host planes (every pointer of a job in host memory; `params->limits_dev` must be NULL): staged to
 * the device, subtracted as one batch, products copied back - zm_subtract for many frames of one configuration. */
int zm_subtract_batch(zm_ctx* ctx, int njobs, const zm_sub_job* jobs, int nx, int ny,
                      zm_hp_info* out_info);
int zm_background_dev(zm_ctx* ctx, const float* img, const float* wgt, int nx,
                      int ny, int mesh, int filtersize, float* out_bkg,
                      float* out_rms, float* out_sub, double* out_stats_host);
int zm_resample_dev(zm_ctx* ctx, const float* img, const float* wgt,
                    const int32_t* mask, const zm_wcs* win, const zm_wcs* wout,
                    int kernel, double fscale, float* out_img, float* out_wgt,
                    int32_t* out_mask);

/* Two images that share a geometry through one resampling launch (device pointers): what the reference does in two
 * SWarp runs when it aligns a reference image and that image's rms map to a science grid (zuds/subtraction.py:109 ->
 * zuds/fitsfile.py:290-314, zuds/hotpants.py:51).  img_a (optionally with its int32 mask: OR under the footprint,
 * into out_mask) and img_b; no weights (WEIGHT_TYPE NONE); out_b is scaled by fscale_b.  Values equal those of two
 * zm_resample_dev calls bit for bit; uncovered pixels are 0 in both outputs; no weight planes. */
int zm_align_pair_dev(zm_ctx* ctx, const float* img_a, const float* img_b, const int32_t* mask,
                      const zm_wcs* win, const zm_wcs* wout, int kernel, double fscale_a, double fscale_b,
                      float* out_a, float* out_b, int32_t* out_mask);
/* zm_resample / zm_resample_dev for a BITPIX 16 mask plane (run_align on a ZTF mask, zuds/swarp.py:157-204):
 * the int16 plane crosses PCIe as it is and is widened on the device; out_mask stays int32 (bit 16 / 17 of
 * the products, zuds/mask.py:26-33, zuds/subtraction.py:170-171). */
int zm_resample_i16(zm_ctx* ctx, const float* img, const float* wgt,
                    const int16_t* mask, const zm_wcs* win, const zm_wcs* wout,
                    int kernel, double fscale, float* out_img, float* out_wgt,
                    int32_t* out_mask);
int zm_resample_i16_dev(zm_ctx* ctx, const float* img, const float* wgt,
                        const int16_t* mask, const zm_wcs* win, const zm_wcs* wout,
                        int kernel, double fscale, float* out_img, float* out_wgt,
                        int32_t* out_mask);
/* out[i] = (int32_t) in[i] (sign extension), n elements, device planes. */
int zm_mask_widen_dev(zm_ctx* ctx, const int16_t* in, int64_t n, int32_t* out);

/* ---- per-pixel bookkeeping on device planes ---------------------------------- */
/* rms = 1/sqrt(w), big_rms where bad or w <= 0 (zuds/image.py:173-208; the reference's numpy gives inf for a
 * zero weight on a pixel its mask does not flag - the object layer restores that, subtraction.py). */
int zm_rms_from_weight_dev(zm_ctx* ctx, const float* wgt, const uint8_t* bad,
                           int64_t n, float big_rms, float* out);
/* w = 1/rms^2, 0 where bad or img >= satur (0.9 SATURATE; zuds/image.py:136-171). */
int zm_weight_from_rms_dev(zm_ctx* ctx, const float* rms, const uint8_t* bad,
                           const float* img, float satur, int64_t n, float* out);
/* The false weight map of a background run without weights (zuds/sextractor.py:80-96: 1, 0 where the mask has a
 * bad bit, 0 in a `border`-pixel frame for raw science images) and / or the boolean bad-pixel map of the mask
 * (zuds/mask.py:42-72); mask_type ZM_MASKTYPE_I32 / _I16.  What `rms_image` / `weight_image` of a frame without
 * .rms.fits / .weight.fits need in front of zm_background_dev and zm_weight_from_rms_dev (zuds/image.py:136-208). */
int zm_false_weight_dev(zm_ctx* ctx, const void* mask, int mask_type, int32_t badsum, int border,
                        int nx, int ny, float* out_wgt, uint8_t* out_bpm);
/* out_or = a | b (b may be NULL); out_bpm = (out_or & badsum) != 0
 * (zuds/subtraction.py:135-142, zuds/mask.py:42-72). */
int zm_mask_bad_dev(zm_ctx* ctx, const int32_t* a, const int32_t* b,
                    int32_t badsum, int64_t n, int32_t* out_or, uint8_t* out_bpm);
/* mask |= bit where img == value: bit 16 from weight == 0 (zuds/mask.py:26-33),
 * bit 17 from diff == 1e-30 (zuds/subtraction.py:170-171). */
int zm_mask_flag_dev(zm_ctx* ctx, int32_t* mask, const float* img, float value,
                     int32_t bit, int64_t n);
/* img += v: the 150-count pedestal (zuds/coadd.py:205-206, zuds/hotpants.py:29). */
int zm_add_scalar_dev(zm_ctx* ctx, float* img, float v, int64_t n);
/* Measurement aid ("an achieved-copy ceiling with a float4 copy kernel"): dst = src,
 * nbytes a multiple of 16, 16-byte aligned device pointers, on the context's stream.  bench.py times it
 * and reports the roofline kernel against the rate it reaches as well as against the nominal 8 TB/s. */
int zm_copy_probe_dev(zm_ctx* ctx, const void* src, void* dst, int64_t nbytes);

/* ---- seeing estimate and detection cuts from pixels -------------------------- */
/* Pixel-only stand-ins for the parts of estimate_seeing (zuds/seeing.py:10-118) and
 * filter_sexcat (zuds/filterobjects.py:57-195) that the reference feeds from SExtractor
 * catalogs and Gaia queries.
 * zm_find_stars: isolated local maxima (strict over the (2 isolation + 1)^2 box, ties to the
 *   first pixel in raster order) with thresh_lo < peak < thresh_hi, no bad (bad[] != 0) or
 *   NaN pixel in the box, at least `border` pixels from the edges.  Unordered; *out_n is the
 *   number found (may exceed max_out: only max_out are returned).
 * zm_star_fwhm: FWHM (pixels) and centroid of each star from adaptive Gaussian-weighted
 *   second moments in a (2 half + 1)^2 window; NaN when the iteration fails.
 * zm_negpix_test: out_bad[k] = 1 when the 11 x 11 cutout around (X_IMAGE, Y_IMAGE) (1-based)
 *   holds a pixel below median - 5 sigma with a 3 x 3 neighbour above median + 5 sigma. */
int zm_find_stars(zm_ctx* ctx, const float* img, const uint8_t* bad, int nx, int ny,
                  float thresh_lo, float thresh_hi, int isolation, int border,
                  int max_out, int* out_x, int* out_y, float* out_peak, int* out_n);
int zm_star_fwhm(zm_ctx* ctx, const float* img, int nx, int ny, int nstar,
                 const int* x, const int* y, int half, double* out_fwhm,
                 double* out_cx, double* out_cy);
int zm_negpix_test(zm_ctx* ctx, const float* img, int nx, int ny, int npos,
                   const double* x, const double* y, double median, double sigma,
                   int32_t* out_bad);
/* The two star measurements on planes that are already in HBM (img / bad: device pointers; positions, widths
 * and lists: host arrays): the seeing of a coadd is estimated before it leaves the device. */
int zm_find_stars_dev(zm_ctx* ctx, const float* img, const uint8_t* bad, int nx, int ny,
                      float thresh_lo, float thresh_hi, int isolation, int border,
                      int max_out, int* out_x, int* out_y, float* out_peak, int* out_n);
int zm_star_fwhm_dev(zm_ctx* ctx, const float* img, int nx, int ny, int nstar,
                     const int* x, const int* y, int half, double* out_fwhm,
                     double* out_cx, double* out_cy);

/* ---- FITS data blocks on the device ------------------------------------------ */
/* Replaces the host-side decode / encode astropy does inside FITSFile.load_data / save
 * (zuds/fitsfile.py:69-94,146-206): raw_dev holds the big-endian data block of a primary
 * HDU as it lies on disk (n pixels of BITPIX 8 / 16 / 32 / -32 / -64); the decoded plane
 * is float32 (out_kind 0), int32 (1), uint8 (2) or int16 (3: a BITPIX 16 mask kept at 16 bits for
 * zm_dframe.mask_type = ZM_MASKTYPE_I16) with physical = bzero + bscale * stored.
 * Encode: float32 -> BITPIX -32 (in_kind 0), int32 -> 32 (1), uint8 -> 8 (2),
 * int32 -> BITPIX 16 (3). */
int zm_fits_decode_dev(zm_ctx* ctx, const void* raw_dev, int bitpix, double bscale,
                       double bzero, int64_t n, int out_kind, void* out_dev);
int zm_fits_encode_dev(zm_ctx* ctx, const void* in_dev, int in_kind, int64_t n,
                       void* raw_dev);

/* ---- timing hooks (bench.py reads per-kernel HIP-event times) ------------- */
/* Enable recording of HIP events around the dominant kernels on the ctx
 * stream; zm_timing_read returns accumulated milliseconds and launch counts. */
int zm_timing_enable(zm_ctx* ctx, int on);
/* Time only the scope `only` (e.g. "resample"); NULL = every scope.  The two event records
 * around a launch cost a few microseconds of dispatch latency each, so a throughput
 * measurement times just the kernel it needs. */
int zm_timing_filter(zm_ctx* ctx, const char* only);
int zm_timing_reset(zm_ctx* ctx);
int zm_timing_read(zm_ctx* ctx, const char* kernel_name, double* total_ms,
                   int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ZUDSMI_H */
