/*
 * nquant_abi.h -- C ABI of libnquant_hip.so: the MI355X (gfx950) implementation of the reference's
 * PnnQuantizer / PnnLABQuantizer hot path (mcychan/nQuant.android).
 *
 * NQ/ = nQuant.master/src/main/java/com/android/nQuant/ in the reference.
 * Every entry point names the reference interface it replaces.  Plain pointers and sizes only; "host"
 * entry points take host memory (what a JNI shim gets from GetPrimitiveArrayCritical on the Java int[]),
 * "_device" entry points take HIP device pointers (what bench.py / a resident pipeline hands over).
 *
 * Pixel format: 32-bit ARGB_8888, non-premultiplied, a = c>>>24, r = (c>>16)&255, g = (c>>8)&255, b = c&255,
 * row-major, index = x + y*width (NQ/PnnQuantizer.java:413-417, NQ/GilbertCurve.java:126).
 *
 * All functions return NQ_OK (0) or a negative nq_status; nq_last_error() gives the text.  A JNI shim maps a
 * non-zero status to the RuntimeException the reference app raises (app/.../MainActivity.java:205-208).
 * A handle mirrors ONE reference quantizer object: stateful, not re-entrant (NQ/PnnQuantizer.java:17-33);
 * distinct handles are independent.  There is no CPU fallback: every compute entry point fails with
 * NQ_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef NQUANT_ABI_H
#define NQUANT_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NQ_ABI_VERSION 1

typedef struct nq_handle nq_handle;

enum nq_kind { NQ_KIND_RGB = 0,   /* PnnQuantizer      (NQ/PnnQuantizer.java)    */
               NQ_KIND_LAB = 1 }; /* PnnLABQuantizer   (NQ/PnnLABQuantizer.java) */

enum nq_mode {
    /* One error-diffusion chain over the whole image with the reference's bin-keyed first-come nearest cache
     * and ONE java.util.Random(seed) stream: bit-exact against the sequential oracle.  Runs on one GPU lane
     * (debug / small images).                                                                              */
    NQ_MODE_REFERENCE_SEQUENTIAL = 0,
    /* Production mode: independent gilbert curve + error queue + Random stream per tile, lookups with the
     * reference's cache-miss semantics; bit-exact against the oracle's tiled restatement.                   */
    NQ_MODE_PARALLEL_TILED = 1,
    /* No diffusion: out_index[i] = nearestColorIndex(palette, pixel[i]) evaluated per pixel (cache-miss
     * semantics): BASELINE.json's "dither off, bit-exact index" check.                                      */
    NQ_MODE_LOOKUP_ONLY = 2
};

enum nq_status {
    NQ_OK = 0,
    NQ_ERR_INVALID = -1,          /* bad argument */
    NQ_ERR_HIP = -2,              /* HIP runtime error (text in nq_last_error) */
    NQ_ERR_UNSUPPORTED = -3,      /* a reference branch this build does not run on the GPU yet */
    NQ_ERR_REFERENCE_THROWS = -4, /* the Java code would throw here (e.g. setAlphaComponent range) */
    NQ_ERR_NO_DEVICE = -5,
    NQ_ERR_TIME_LIMIT = -6        /* a merge loop ran into its wall-clock limit (slow / shared / time-sliced device): nothing was
                                   * wrong with its state -- call again, or raise NQ_OPT_MERGE_WALL_SECONDS.  Distinct from
                                   * NQ_ERR_UNSUPPORTED, which the loop's find_nn budget (maxbins^2/2 calls) or an empty heap report */
};

/* Scalars convert() derives and the later stages consume.  Same layout as the
 * CPU oracle's parameter struct (tests compare the two field by field). */
typedef struct nq_params {
    int32_t kind;
    int32_t nMaxColors;
    int32_t hasSemiTransparency;   /* NQ/PnnQuantizer.java:431 */
    int32_t transparentPixelIndex; /* m_transparentPixelIndex (:420), -1 = none */
    int32_t transparentColor;      /* m_transparentColor (:22,:422) */
    int32_t isNano;                /* NQ/PnnLABQuantizer.java:180 */
    int32_t texicab;               /* NQ/PnnLABQuantizer.java:219 */
    int32_t quan_rt;
    int32_t maxbins;
    int32_t paletteLength;
    double PR, PG, PB, PA;         /* NQ/PnnQuantizer.java:24,432-436,176-180 */
    double ratio;
    double weight;                 /* signed (negated for semi-transparent images, :396-397) */
    int64_t distinctColors;        /* LAB: pixelMap.size() after the histogram (0 when not needed) */
} nq_params;

/* ---- lifetime: replaces `new PnnQuantizer(fname)` / `new PnnLABQuantizer(fname)` (NQ/PnnQuantizer.java:35,
 *      NQ/PnnLABQuantizer.java:24) and garbage collection.  device = HIP device ordinal. ---- */
/* Threads and devices: a handle is NOT thread-safe (like the reference object); distinct handles are independent and
 * may be driven from different threads and live on different devices of one process -- all per-device state (constant tables,
 * kernel attributes) is set up per handle on the handle's device, nothing is cached per process. */
int nq_create(int kind, int device, nq_handle** out);
void nq_destroy(nq_handle* h);
const char* nq_last_error(const nq_handle* h);   /* h may be NULL: last error of nq_create on this thread */
int nq_abi_version(void);
/* All work of the handle is enqueued on this hipStream_t (NULL = the default stream). */
int nq_set_stream(nq_handle* h, void* hip_stream);
/* Tile of the PARALLEL_TILED decomposition; <= 0 (default) = automatic: 8x8 when that gives the GPU at least 131072
 * independent chains (images from about 2900^2 pixels), 4x4 below that -- for every form of the error queue (a tile chain of the
 * sorted-by-yDiff mode, K > 128 && weight >= .02, starts with its queue in the steady state instead of re-growing it per tile). */
int nq_set_tile(nq_handle* h, int tile_w, int tile_h);
/* One image tiled over GPUs: this handle's following nq_dither[_device] calls treat their pixel buffer as the rows
 * [y0, y0 + height) of an image of image_height rows -- tile random streams, the blue-noise phase and the position-dependent
 * gates (`bidx & 4095`, `pos % 2`) are those of the whole image, so the bands of an image equal the same rows of the single-GPU
 * PARALLEL_TILED result.  y0 (and every band's row count but the last) must be a multiple of the tile height; the automatic tile
 * follows the whole image.  (0, 0) = a whole image again.  Not for REFERENCE_SEQUENTIAL. */
int nq_set_band(nq_handle* h, int y0, int image_height);

/* Tuning switches that never change results.  NQ_OPT_CELL_LISTS (default 1): scan only the per-colour-cell candidate
 * lists in nearest/closestColorIndex (exact, csrc/nq_lists.inc); 0 = scan the whole palette like the reference. */
#define NQ_OPT_CELL_LISTS 1
/* NQ_OPT_FAST_DITHER (default 1): run the specialised dither kernel (csrc/nq_dither_fast.inc) where the configuration allows
 * it (LAB, 32 < K <= 256, no semi-transparency, DITHER_MAX 25, PARALLEL_TILED); 0 = the generic kernel everywhere.  Same results. */
#define NQ_OPT_FAST_DITHER 2
/* NQ_OPT_MERGE_WALL_SECONDS (default 0 = automatic: 60 s + 1 ms per histogram bin and per merge loop sharing a compute unit with it):
 * seconds of residency after which a merge loop of this handle's calls gives up with NQ_ERR_TIME_LIMIT.  Every loop of the persistent
 * merge kernel is bounded; this bound only exists so that a corrupt heap cannot keep the GPU for hours. */
#define NQ_OPT_MERGE_WALL_SECONDS 3
int nq_set_option(nq_handle* h, int option, int value);
/* Diagnostics of the last dither pass: out_fast = 1 if the specialised kernel ran; out_failed_tiles = tiles it handed back to
 * the generic kernel (synchronises the handle's stream). */
int nq_get_dither_path(nq_handle* h, int32_t* out_fast, int32_t* out_failed_tiles);
/* Diagnostics: length of every cell's candidate list of the last dither/lookup call (255 = full scan), 65536 bytes each. */
int nq_get_list_counts(nq_handle* h, uint8_t* closest_counts, uint8_t* nearest_counts);
/* Self-test hook of the branch-free CIEDE2000 evaluation inside find_nn (csrc/nq_device.h: ciede_terms_fast): for n pairs
 * {L1, A1, B1, L2, A2, B2} returns, as float bit patterns, out9[9 i + 0..3] = deltaL', deltaC', deltaH', R_T of the fast pass,
 * out9[9 i + 4..7] = the same from the literal functions, out9[9 i + 8] = 1 when the fast pass decided (else find_nn uses the
 * literal values).  Wherever it decided the two quadruples must be identical. */
int nq_selftest_ciede(nq_handle* h, const float* lab_pairs, int64_t n, uint32_t* out9);
int nq_get_params(const nq_handle* h, nq_params* out);
int nq_set_params(nq_handle* h, const nq_params* in);

/* ---- the ditherers' own static entry points: the Ditherable is the handle (its kind and params answer
 *      getColorIndex / nearestColorIndex exactly as in nq_dither) ----
 * static int[] GilbertCurve.dither(width, height, pixels, palette, ditherable, saliencies, weight, dither)
 *      (NQ/GilbertCurve.java:367-373): saliencies may be NULL, weight is the SIGNED constructor argument (negative = the image has
 *      semi-transparent pixels, :60-61).  out_qpixels follows the reference (:278-279): ARGB when dither || K <= 32, palette
 *      indices otherwise; out_index (nullable) always receives the indices.
 * static int[] BlueNoise.dither(width, height, pixels, palette, ditherable, qPixels, weight) (NQ/BlueNoise.java:207-222):
 *      io_qpixels holds palette indices on entry (what GilbertCurve.dither returned for !dither && K > 32) and ARGB on return.
 * REFERENCE_SEQUENTIAL: nq_gilbert_dither starts from empty lookup caches and Random(rng_seed); nq_bluenoise_dither continues with
 * the caches and the random stream the previous call on the handle left behind, as the two calls inside dither() do. */
int nq_gilbert_dither(nq_handle* h, int width, int height, const uint32_t* pixels, const uint32_t* palette, int K,
                      const float* saliencies, double weight, int dither, int64_t rng_seed, int mode,
                      int32_t* out_qpixels, uint16_t* out_index);
int nq_bluenoise_dither(nq_handle* h, int width, int height, const uint32_t* pixels, const uint32_t* palette, int K,
                        int32_t* io_qpixels, float weight, int64_t rng_seed, int mode, uint16_t* out_index);

/* ---- Bitmap convert(int nMaxColors, boolean dither)  (NQ/PnnQuantizer.java:409-456) ----
 * out_argb  [w*h]  : the pixels of the returned Bitmap (always ARGB)
 * out_index [w*h]  : palette index chosen per pixel (nullable)
 * out_palette      : room for max(nMaxColors,2) entries;  *out_K = palette length
 * The input is never modified (the n<=2 rewrite of :424 is applied internally). */
int nq_convert(nq_handle* h, const uint32_t* argb, int width, int height, int nMaxColors, int dither,
               int64_t rng_seed, int mode,
               uint32_t* out_argb, uint16_t* out_index, uint32_t* out_palette, int32_t* out_K);
/* same, all pixel buffers in device memory (out_palette/out_K stay host); asynchronous on the handle's stream
 * except for the small palette-parameter readbacks. */
int nq_convert_device(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors, int dither,
                      int64_t rng_seed, int mode,
                      uint32_t* d_out_argb, uint16_t* d_out_index, uint32_t* out_palette, int32_t* out_K);

/* ---- convert() of n quantizer objects in one call (the reference app converts one file per executor task,
 *      app/src/main/java/nQuant/android/MainActivity.java:190-214; a service converting many images hands them over
 *      together).  Results are identical to n separate nq_convert_device calls.  The merge loop of one image is a
 *      sequential chain that occupies one CU; here the n merge loops run side by side in ONE launch (one workgroup
 *      each), the other stages run image after image on the first handle's stream and share its per-pixel scratch.
 *      hs[i] are distinct handles on one device (mixing kinds is allowed); all pointer arrays are host arrays of n
 *      device pointers; out_palettes[i * palette_stride ...] / out_K[i] receive palette i
 *      (palette_stride >= max(nMaxColors, 2)); d_out_index may be NULL. ---- */
int nq_convert_batch_device(nq_handle* const* hs, int n, const uint32_t* const* d_argb, const int32_t* widths,
                            const int32_t* heights, int nMaxColors, int dither, const int64_t* rng_seeds, int mode,
                            uint32_t* const* d_out_argb, uint16_t* const* d_out_index,
                            uint32_t* out_palettes, int32_t palette_stride, int32_t* out_K);

/* same with HOST buffers (what a JNI shim holds): host arrays of n host pointers.  Uploads run ahead of the per-image stages and
 * results are copied back while the next image is dithered, on a second stream; only the inputs (4 B/pixel) stay resident for
 * the batch.  Page-locked buffers (direct ByteBuffers registered with hipHostRegister, hipHostMalloc) make the copies
 * asynchronous; pageable memory works, the copies then block the calling thread.  out_index / out_index[i] may be NULL. */
int nq_convert_batch(nq_handle* const* hs, int n, const uint32_t* const* argb, const int32_t* widths,
                     const int32_t* heights, int nMaxColors, int dither, const int64_t* rng_seeds, int mode,
                     uint32_t* const* out_argb, uint16_t* const* out_index,
                     uint32_t* out_palettes, int32_t palette_stride, int32_t* out_K);

/* ---- Integer[] pnnquan(int[] pixels, int nMaxColors) incl. the alpha pre-scan of convert()
 *      (NQ/PnnQuantizer.java:410-436,134-267; NQ/PnnLABQuantizer.java:131-327) ---- */
int nq_pnnquan(nq_handle* h, const uint32_t* argb, int width, int height, int nMaxColors,
               uint32_t* out_palette, int32_t* out_K);
int nq_pnnquan_device(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors,
                      uint32_t* out_palette, int32_t* out_K);

/* ---- int[] dither(cPixels, palette, width, height, dither) of the quantizer object
 *      (RGB NQ/PnnQuantizer.java:393-407, LAB NQ/PnnLABQuantizer.java:493-522): GilbertCurve.dither
 *      (NQ/GilbertCurve.java:367-373) followed, for !dither && K>32, by BlueNoise.dither
 *      (NQ/BlueNoise.java:207-222).  Uses the handle's params (from nq_pnnquan or nq_set_params). ---- */
int nq_dither(nq_handle* h, const uint32_t* argb, int width, int height, const uint32_t* palette, int K,
              int dither, int64_t rng_seed, int mode, uint32_t* out_argb, uint16_t* out_index);
int nq_dither_device(nq_handle* h, const uint32_t* d_argb, int width, int height, const uint32_t* palette, int K,
                     int dither, int64_t rng_seed, int mode, uint32_t* d_out_argb, uint16_t* d_out_index);

/* ---- Ditherable.nearestColorIndex on a cache miss (NQ/Ditherable.java:3-7;
 *      RGB NQ/PnnQuantizer.java:269-311, LAB NQ/PnnLABQuantizer.java:330-404): pure per colour. ---- */
int nq_nearest_index(nq_handle* h, const uint32_t* palette, int K, const uint32_t* colors, int64_t M,
                     int16_t* out_index);
/* ---- the closest[4] = {idx1, idx2, (int)err1, (int)err2} tuple of closestColorIndex
 *      (RGB NQ/PnnQuantizer.java:320-363, LAB NQ/PnnLABQuantizer.java:413-464); {-1,-1,-1,-1} where the
 *      reference returns through nearestColorIndex first (alpha <= alphaThreshold). ---- */
int nq_closest_tuple(nq_handle* h, const uint32_t* palette, int K, const uint32_t* colors, int64_t M,
                     int32_t* out_closest4);

/* ---- split pipeline for an image tiled over several GPUs: each rank scans its band, the
 *      caller reduces the partial histograms between ranks (RCCL via torch.distributed), every rank then
 *      builds the same palette and dithers its own band.  Buffers are device memory. ---- */
/* pass 1 over a band: alpha pre-scan partials.  d_scan3 = int64[3]: {max global index of an alpha==0 pixel or -1,
 * its colour, count of pixels with 0xF < alpha < 0xE0}; index_offset = global index of the band's first pixel.
 * Reduce across ranks: [0] max (carry [1] of the winner), [2] sum; then nq_set_scan(). */
int nq_band_scan_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, int64_t index_offset,
                        int nMaxColors, int64_t* d_scan3);
int nq_set_scan(nq_handle* h, int nMaxColors, int64_t transparent_index, uint32_t transparent_color,
                int64_t semi_count);
/* pass 2 over a band: partial histogram, NQ_HIST_STRIDE doubles per bin x 65536 bins:
 * {count, sum0, sum1, sum2, sum3} (RGB: a,r,g,b integer sums; LAB: float32 running sums of alpha,L,A,B of
 * the band in pixel order, widened). */
#define NQ_HIST_BINS 65536
#define NQ_HIST_STRIDE 5
int nq_band_histogram_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, double* d_hist);
/* palette from per-band histograms laid out [n_bands][65536][5] (already gathered on this rank); band partials
 * are added in band order (float32 for LAB, exactly as a sequential pass over band-ordered partial sums). */
int nq_palette_from_histograms_device(nq_handle* h, const double* d_hists, int n_bands, int nMaxColors,
                                      uint32_t* out_palette, int32_t* out_K);

/* few-colours early return of the LAB class (NQ/PnnLABQuantizer.java:193-206) in the split pipeline: when the gathered histograms
 * hold <= nMaxColors occupied bins, every rank lists the distinct colours of its band as the histogram sees them (alpha <= 15 ->
 * transparent colour; needs nq_set_scan first) in first-occurrence order -- *out_count = how many there are, out_colors filled
 * when *out_count <= cap -- the caller concatenates the lists in band order, drops repeats, and hands the image-wide list to
 * every rank with nq_set_distinct (count < 0 = "more than nMaxColors": no early return).  nq_palette_from_histograms_device then takes the
 * same early return as a single GPU would. */
int nq_band_distinct_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, int cap, int64_t* out_count,
                            uint32_t* out_colors);
int nq_set_distinct(nq_handle* h, int64_t count, const uint32_t* colors);

/* Image-wide distinct-colour count of a banded LAB run (it sets the BlueNoise weight of convert(n, false),
 * NQ/PnnLABQuantizer.java:511-515, and lives in nq_params.distinctColors): every rank marks the opaque colours of its band in
 * d_presence (2^24 bytes, one per RGB value; the caller zeroes it once and may pass the same table for several bands) and gets
 * the band's non-opaque colours (after the alpha <= 15 substitution; needs nq_set_scan first) as a list -- *out_other_count = -1
 * when there are more than cap_other (<= 131072).  The caller combines the tables by byte-wise MAX (an all-reduce), counts their
 * non-zero bytes, adds the size of the union of the lists and stores the sum with nq_set_params. */
int nq_band_color_presence_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, uint8_t* d_presence, int cap_other,
                                  int64_t* out_other_count, uint32_t* out_other);

/* Wall-clock of the stages of the last nq_convert*_ call on this handle, milliseconds, measured with HIP
 * events on the handle's stream: {prescan, histogram, nn_init, merge, palette_fill, dither, bluenoise, total}.
 * Batch entry points: an event record costs the GPU ~3 us of queue time, so only the first SIXTEEN handles of a batch record all
 * eight boundaries; for the others {prescan, histogram, nn_init, merge, palette_fill, bluenoise} are reported as -1 (not recorded),
 * `dither` is the per-pixel pass as for every handle, and `total` spans from the start of the image's pre-scan to the END OF THE
 * DITHER PASS -- it leaves out the BlueNoise post-pass of convert(n, false).  nq_get_batch_phase_ms gives the amortised phases. */
#define NQ_N_STAGES 8
int nq_get_stage_ms(const nq_handle* h, float* out8);
/* Counters of the last merge loop (diagnostics), 16 values: {find_nn calls, merges, 100 MHz ticks inside find_nn, ticks in
 * the sequential heap/merge section, live-list rebuilds, find_nn list overflows, candidates evaluated exactly, ticks in
 * the bound pass, ticks in the exact pass, ticks in the replay, 64-candidate chunks visited, chunks that ran the level-1
 * bound, chunks that ran the tight bound, chunks that listed a candidate, aborted flag, ticks of the seed round inside the bound pass}. */
int nq_get_merge_stats(const nq_handle* h, int64_t* out16);
/* Counters of the last merge loop's TEAM (csrc/nq_merge.inc: a merge loop that has CUs to spare -- a single image, a batch of up to 128 -- runs as a master
 * workgroup plus 1 - 7 helper workgroups that evaluate find_nn speculatively for the bins that will surface next), 16 values:
 * {work records published, helper results used in place of an own find_nn, waits for a result that timed out, 100 MHz ticks spent
 * waiting, helpers per loop, 1 if the loop was still speculating at its end, bin-info cache hits for the heap top, results used that were
 * computed for a merge before it happened ("virtual merge"), then the control thread's 100 MHz ticks in heap sifts / in merges / fetching the heap top's bin, the deleted nodes
 * popped, ticks in the find_nn epilogues, ticks spent choosing work records, results a helper declined (RGB), times the loop gave up
 * on its helpers for a while}.  Diagnostics. */
int nq_get_team_stats(const nq_handle* h, int64_t* out16);
/* Phases of the last nq_convert_batch[_device] call, as seen by its FIRST handle: HIP-event spans on the launch stream, ms:
 * {every image's pre-scan + histogram + initial find_nn pass, the merge launch (all merge loops side by side + palette fill),
 *  every image's palette read-back + candidate lists + dither pass, whole call}.  Divided by the batch size these are the
 * amortised per-image times (the per-handle stage times of nq_get_stage_ms are event spans that include queueing behind the
 * other images of the batch). */
int nq_get_batch_phase_ms(const nq_handle* h0, float* out4);

#ifdef __cplusplus
}
#endif
#endif /* NQUANT_ABI_H */
