// nq_kernels.h -- host-callable launchers of the gfx950 kernels (defined in nq_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nq {

// scalars of the quantizer object that the per-pixel code needs
struct DevParams {
    int kind;              // 0 RGB, 1 LAB
    int K;                 // palette.length
    int hasSemi;           // hasSemiTransparency
    int hasAlpha;          // m_transparentPixelIndex > -1
    int transparentColor;  // m_transparentColor
    int isNano;            // NQ/PnnLABQuantizer.java:180
    int binKeyed;          // nearest cache keyed by histogram bin (LAB: isNano; RGB: !(weight > .015))
    int nMaxColors;
    int rewriteA0;         // nMaxColors <= 2: pixels with alpha == 0 read as m_transparentColor (NQ/PnnQuantizer.java:424)
    int pad;
    double PR, PG, PB, PA, ratio, weight;   // weight signed
};

// scalars the GilbertCurve constructor derives (NQ/GilbertCurve.java:50-112) + initWeights tables (:336-354),
// evaluated once on the host (control plane) and handed to the kernel by value
struct GilbertConsts {
    int margin, thresold, DITHER_MAX, ditherMax, sortedByYDiff, hasAlphaW, dither;
    int hasSaliencies;      // saliencies != null
    int salSubst;           // saliency computed from the alpha-substituted pixel (pnnquan path, nMaxColors < 128)
    float beta;
    double weightAbs;       // the field `weight` (abs value, :61)
    float weights[25];      // initWeights(DITHER_MAX) (non-sorted mode)
    float w1[1], w3[3], w7[7], w15[15]; // initWeights(1|3|7|15) (sorted mode growth 1 -> 3 -> 7 -> 15 -> 31)
};

struct TileGeom {
    int width, height;
    int tile_w, tile_h, tiles_x, tiles_y;
    // visiting order of each tile shape: 0 interior, 1 right edge, 2 bottom edge, 3 corner; entries (dx | dy << 16)
    const uint32_t* path[4];
    int path_len[4];
    int shape_w[4], shape_h[4];
    // a row band of a larger image (one image tiled over GPUs): the band starts at image row y_origin (a multiple of
    // tile_h) and its first tile is tile number tile_base of the whole image.  Pixel buffers are band-local; the tile's random
    // stream, the blue-noise phase (x, y), the `bidx & 4095` gates and the pixel position handed to the lookups are those of the
    // whole image, so that the bands of an image equal the same rows of the single-GPU result.  0 / 0 for a whole image.
    int y_origin, tile_base;
};

void upload_tables(const double gamma[256], double exp1_5, double exp1_75, hipStream_t s);        // nq_kernels.hip's copy
void upload_tables_fast(const double gamma[256], double exp1_5, double exp1_75, hipStream_t s);   // nq_dither_fast.hip's copy

// candidate lists per 5-6-5 colour cell (nq_lists.inc); null pointers = full palette scans
struct ListsView {
    const unsigned char* closest; const unsigned char* closestCount;
    const unsigned char* nearest; const unsigned char* nearestCount;
};
// wA..wB: total weight of da^2, dr^2, dg^2, db^2 in closestColorIndex's err; nearest: also build the nearestColorIndex lists
void launch_cell_lab_box(float* d_box /* [65536][6] */, hipStream_t s);
// a saliency map to build beside the LAB candidate lists (the three passes are independent; one launch, nq_dither.inc build_lab_lists_kernel)
struct SalJob { const int* pixels; long long N; float* out; long long vec4; int salSubst; int blocks; };
// d_sal_pixels != null: the saliency map of these N pixels is wanted too -- returns true when it went into the same launch (LAB with
// nearest lists), false when the caller has to launch_saliency itself
bool launch_build_lists(const DevParams& P, const int* d_palette, double wA, double wR, double wG, double wB, bool nearest,
                        const float* d_box, unsigned char* d_closest, unsigned char* d_closestCount, unsigned char* d_nearest,
                        unsigned char* d_nearestCount, hipStream_t s, const int* d_sal_pixels = nullptr, int64_t N = 0, float* d_sal_out = nullptr,
                        int salSubst = 0);
void launch_saliency(const DevParams& P, int salSubst, const int* d_pixels, int64_t N, float* d_out, hipStream_t s);

void launch_nearest_index(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_colors, int64_t M, short* d_out, hipStream_t s);
void launch_closest_tuple(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_colors, int64_t M, int* d_out4, hipStream_t s);
// LOOKUP_ONLY: index (+ARGB) of nearestColorIndex(pixel) for every pixel
void launch_lookup_only(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_pixels, int64_t N,
                        unsigned short* d_index, int* d_argb, hipStream_t s);

// GilbertCurve.dither over every tile; writes indices (always) and ARGB (when d_argb != nullptr)
void launch_gilbert(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv, const int* d_pixels,
                    const float* d_saliency, const int* d_palette, short* d_binCache, long long seed, int sequential,
                    long long* d_rng_state, unsigned short* d_index, int* d_argb,
                    // REFERENCE_SEQUENTIAL + LAB only (else null): log of the colours / palette entries getLab() sees during the pass
                    int* d_log, int* d_log_count, unsigned char* d_seen, int log_cap,
                    // nullable: {count, tile indices...} -- walk only these tiles (the ones gilbert_fast_kernel handed back)
                    const int* d_tile_list, hipStream_t s);
// nq_dither_fast.hip: the specialised kernel for 32 < K <= 256, no semi-transparency, DITHER_MAX 25, tiled: PnnLABQuantizer, and
// PnnQuantizer with dither = true on images without transparency
bool gilbert_fast_eligible(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv);
hipError_t launch_gilbert_fast(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv, const int* d_pixels,
                               const float* d_saliency, const int* d_palette, long long seed, unsigned short* d_index, int* d_argb,
                               int* d_failed /* int[1 + tiles] */, void* d_packed /* 65536 x 64 bytes */, hipStream_t s);
bool fast_lookup_eligible(const DevParams& P, const ListsView& lv);
bool fast_pack_wanted(const DevParams& P, const ListsView& lv);     // the packed records are needed (LAB lookups, or the RGB dither kernel)
// packs the two lists of every colour cell into the 32-byte records (+ continuations) the specialised kernels read; must follow
// launch_build_lists on the same stream whenever fast_lookup_eligible() holds
void launch_pack_lists(const ListsView& lv, void* d_packed, hipStream_t s);
void launch_fast_nearest_index(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_colors, int64_t M,
                               short* d_out, hipStream_t s);
void launch_fast_closest_tuple(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_colors, int64_t M,
                               int* d_out4, hipStream_t s);
// d_todo: unsigned[N + 1] scratch (the pixels the float32 pass leaves to the exact pass)
void launch_fast_lookup_only(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_pixels, int64_t N,
                             unsigned short* d_index, int* d_argb, unsigned* d_todo, hipStream_t s);
void launch_fast_bluenoise(const DevParams& P, const ListsView& lv, const int* d_palette, void* d_packed, const int* d_pixels, int width, int height,
                           int y_origin, float weight, long long seed, unsigned short* d_index, int* d_argb, hipStream_t s);
// BlueNoise.dither post-pass (NQ/BlueNoise.java:207-222); in-place on d_index, writes d_argb
void launch_bluenoise(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_pixels, int width, int height,
                      int y_origin /* band start row in the whole image, 0 otherwise */,
                      float weight, long long seed, int sequential, short* d_binCache, long long* d_rng_state,
                      unsigned short* d_index, int* d_argb, hipStream_t s);

// ---- palette build (nq_palette.inc) ----
struct Bins {
    float* f[4];
    double* d[4];
    float* cnt;
    float* err;
    int* nn;
    int* tm;
    int* mtm;
};
struct HistParams { int hasSemi, hasTransp, transparentColor, rewriteTransparent; };
struct NNParams {
    int kind, hasSemi, texicab;
    double ratio, PR, PG, PB, PA;
    int pgLessThanCoeff;
    double rgbTheta;            // RGB scans: assumed cap of the running error = rgbTheta x error after the seed blocks (>= 1; checked, see nq_merge.inc)
};
struct SortWorkspace {
    unsigned short *keys_a, *keys_b; // (unused since the histogram sorts packed {bin, rest-of-pixel} words; may be null)
    int *vals_a, *vals_b;            // [n] each: the packed words before / after the sort
    void* tmp; size_t tmp_bytes;
    unsigned *seg_start, *seg_end;   // [65536] each, contiguous (seg_end = seg_start + 65536), followed by the occupied-bin counter (seg_end[65536])
                                     // and, 64 words further, the list of the occupied bins [65536] and that of the fat bins [1024]
};
size_t sort_temp_bytes(int64_t n);
size_t sort32_temp_bytes(int64_t n, bool pairs);
void launch_distinct(const int* d_pixels, int64_t n, int transparentColor, unsigned* keys_a, unsigned* keys_b, unsigned* idx_a,
                     unsigned* idx_b, void* tmp, size_t tmp_bytes, unsigned long long* d_out, void* d_heads, unsigned cap, hipStream_t s);
// colours present in a band: opaque ones mark d_bytes[rgb] (2^24 bytes, NOT cleared here: bands accumulate), the others enter the set
void launch_color_presence(const int* d_pixels, int64_t n, int transparentColor, unsigned char* d_bytes, unsigned* d_set, unsigned slots,
                           unsigned* d_counters, hipStream_t s);
void launch_ciede_selftest(const float* d_pairs /* n x 6 */, int64_t n, unsigned* d_out /* n x 9 */, hipStream_t s);
void launch_prescan(const int* d_pixels, int64_t n, int64_t index_offset, long long* d_scan3, hipStream_t s);
// pre-scan + the histogram's packed sort words (5-6-5 keys, default transparent colour) in one read of the image; false: not
// applicable (n not a multiple of 4 / unaligned buffers) and nothing was launched.  The words are valid only if the scan then
// reports no alpha == 0 pixel and no semi-transparency and nMaxColors >= 64 (launch_histogram(..., words_ready = true)).
bool launch_front(const int* d_pixels, int64_t n, long long* d_scan3, int* d_words /* SortWorkspace::vals_a */, int defaultTransparent,
                  hipStream_t s);
void launch_histogram(int kind, const int* d_pixels, int64_t n, const HistParams& hp, const SortWorkspace& ws,
                      double* d_hist, hipStream_t s, bool words_ready = false);
// d_blockcnt: int[64] scratch (occupied bins per 1024-bin slice)
void launch_compact(int kind, const double* d_hists, int n_bands, const Bins& B, int* d_maxbins, int* d_blockcnt, hipStream_t s);
void launch_quanfn(float* d_cnt, int maxbins, int fn, hipStream_t s);
// d_box: float[1024 * 8] scratch (LAB: bounding boxes of the blocks of 64 consecutive bins)
// d_init_cand: int[65536 * 128 * 2 + 65536] scratch of the LAB kind (candidate lists between the bound and the exact kernel); may be null for RGB
void launch_find_nn_init(const NNParams& np, const Bins& B, int maxbins, float* d_box, int* d_init_cand, hipStream_t s);
// One merge loop (P9).  heap: int[2*(65536+2)] (ids, then float keys); live3: int[3*65536] (two live lists + position index);
// scan_f: float[2*10*65536 + 256] (LAB uses 2 x 6 x 65536, RGB 2 x 10 x 65536; a scan may read 63 records past the live list), scan_i: int[2*65536] (LAB scan arrays, two generations); stats: long long[16] (see merge_kernel)
struct MergeJob {
    NNParams np;
    Bins B;
    int maxbins, extbins;
    int* heap; int* live3; float* scan_f; int* scan_i;
    float* scan_box;            // float[1024 * 8]: bounding boxes of the 64-position blocks of the LAB scan arrays
    long long* stats;
    int plen; int* palette; int* status;   // P10 runs at the end of the merge workgroup: palette[plen], status |= 1 where Java throws
    // merge teams (launch_merge decides): 256 u64 of zeroed device memory per job for the work records / results of the helpers
    unsigned long long* team; int helpers;
    long long wall_ticks;       // watchdog: the loop stops (stats[14] = 2) after this many 100 MHz ticks of residency
};
// d_jobs: n jobs of one kind in device memory, one workgroup per job.  n_in_flight = merge loops expected to run at the same
// time on the device (the whole batch), n_cus = compute units of the handle's device (hipDeviceAttributeMultiprocessorCount: 256 on an
// unpartitioned MI355X): <= n_cus -> 512-thread workgroups, one per CU; <= 2 n_cus -> 256 threads, two per CU; more -> 128 threads, four
// per CU.  helpers > 0 (either kind, 512-thread variant only; every job's `team` area zeroed and `helpers` set to
// the same number): the grid holds 1 + helpers workgroups per job (merge teams, nq_merge.inc).  Returns the first HIP error of the
// attribute call / launch.
hipError_t launch_merge(int kind, const MergeJob* d_jobs, int n, int n_in_flight, int n_cus, int helpers, hipStream_t s);
// helpers launch_merge would use for n jobs of one kind when n_in_flight loops share a device of n_cus compute units (0..7;
// NQ_MERGE_HELPERS overrides).  Sized on THIS call's jobs: merge launches of other handles / threads on the same device are not
// counted -- correctness does not depend on it (every wait of a team is bounded, nq_merge.inc), only the speed-up does.
int merge_team_helpers(int n_jobs, int n_in_flight, int n_cus);
// first error of a hipFuncSetAttribute issued by a launch_* function on this thread since the last call (hipSuccess: none); cleared
hipError_t take_launch_error();

} // namespace nq
