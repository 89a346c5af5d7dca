// nq_kernels.hip -- the single device translation unit of libnquant_hip.so (gfx950 only).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math (see build.py): Java arithmetic has no
// fused multiply-add and no fast-math, and parity with the oracle is bit for bit.
#include "nq_device.h"
#include "nq_kernels.h"
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>

#include "nq_lists.inc"
#include "nq_dither.inc"
#include "nq_palette.inc"

namespace nq {

static const int8_t h_blue[4096] = {
#include "../../include/nq_blue_noise_64x64.inc"
};

void upload_tables(const double gamma[256], double exp1_5, double exp1_75, hipStream_t s) {
    ConstTables t;            // (the copy below is waited for; g_tab is per device: every handle uploads to its own device)
    for (int i = 0; i < 256; ++i) t.gamma[i] = gamma[i];
    t.exp1_5 = exp1_5; t.exp1_75 = exp1_75;
    for (int i = 0; i < 4096; ++i) t.blue[i] = h_blue[i];
    (void) hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tab), &t, sizeof t, 0, hipMemcpyHostToDevice, s);
    (void) hipStreamSynchronize(s);
}

static inline int grid_for(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int) g;
}

// kernels that stage a K-entry palette (+ its Lab) in dynamic LDS: above 64 KB (LAB palettes beyond ~3700 entries, up to the
// documented 8192) the launch needs the raised per-kernel limit
static thread_local hipError_t t_launch_error = hipSuccess;
static inline void note_error(hipError_t e) { if (e != hipSuccess && t_launch_error == hipSuccess) t_launch_error = e; }
hipError_t take_launch_error() { const hipError_t e = t_launch_error; t_launch_error = hipSuccess; return e; }
template <typename Kern>
static inline void allow_big_lds(Kern kernel, size_t bytes) {
    if (bytes > 48 * 1024) note_error(hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
}

static inline CellLists to_lists(const ListsView& v) {
    CellLists l; l.closest = v.closest; l.closestCount = v.closestCount; l.nearest = v.nearest; l.nearestCount = v.nearestCount;
    return l;
}

void launch_cell_lab_box(float* d_box, hipStream_t s) {
    hipLaunchKernelGGL(cell_lab_box_kernel, dim3(65536 / 256), dim3(256), 0, s, d_box);
}

bool launch_build_lists(const DevParams& P, const int* d_palette, double wA, double wR, double wG, double wB, bool nearest,
                        const float* d_box, unsigned char* d_closest, unsigned char* d_closestCount, unsigned char* d_nearest,
                        unsigned char* d_nearestCount, hipStream_t s, const int* d_sal_pixels, int64_t N, float* d_sal_out, int salSubst) {
    if (nearest && P.kind == 1) {          // LAB: both builders (and the saliency map, if one is wanted) side by side in one launch
        SalJob sal;
        std::memset(&sal, 0, sizeof sal);
        if (d_sal_pixels && d_sal_out && N > 0) {
            sal.pixels = d_sal_pixels; sal.N = (long long) N; sal.out = d_sal_out; sal.salSubst = salSubst;
            sal.vec4 = (((uintptr_t) d_sal_pixels | (uintptr_t) d_sal_out) & 15) ? 0 : (long long) (N / 4);
            sal.blocks = grid_for(N / 4 + 1, 256, 256 * 8);
        }
        const size_t smem = std::max(palette_smem_bytes(P.kind, P.K), (size_t) 256 * sizeof(double));
        allow_big_lds(build_lab_lists_kernel, smem);
        hipLaunchKernelGGL(build_lab_lists_kernel, dim3(2 * 65536 / 256 + sal.blocks), dim3(256), smem, s, P, d_palette,
                           wA, wR, wG, wB, P.hasAlpha ? 1 : 0, d_box, d_closest, d_closestCount, d_nearest, d_nearestCount, sal);
        return sal.blocks > 0;
    }
    hipLaunchKernelGGL(build_closest_lists_kernel, dim3(65536 / 256), dim3(256), 0, s, P, d_palette,
                       wA, wR, wG, wB, d_closest, d_closestCount);
    if (nearest)           // RGB: the nearestColorIndex weights are pa, pr, pg, pb themselves
        hipLaunchKernelGGL(build_nearest_rgb_lists_kernel, dim3(65536 / 256), dim3(256), (size_t) P.K * sizeof(int), s, P, d_palette,
                           P.K < 3 ? 1.0 : P.PA, P.K < 3 ? 1.0 : P.PR, P.K < 3 ? 1.0 : P.PG, P.K < 3 ? 1.0 : P.PB, d_nearest, d_nearestCount);
    return false;
}

void launch_saliency(const DevParams& P, int salSubst, const int* d_pixels, int64_t N, float* d_out, hipStream_t s) {
    const long long vec4 = (((uintptr_t) d_pixels | (uintptr_t) d_out) & 15) ? 0 : N / 4;
    hipLaunchKernelGGL(saliency_kernel, dim3(grid_for(N / 4 + 1, 256, 256 * 8)), dim3(256), 0, s, P, salSubst, d_pixels, (long long) N, d_out, vec4);
}

void launch_nearest_index(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_colors, int64_t M, short* d_out, hipStream_t s) {
    allow_big_lds(nearest_index_kernel, palette_smem_bytes(P.kind, P.K));
    hipLaunchKernelGGL(nearest_index_kernel, dim3(grid_for(M, 256)), dim3(256), palette_smem_bytes(P.kind, P.K), s,
                       P, d_palette, to_lists(lv), d_colors, (long long) M, d_out);
}
void launch_closest_tuple(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_colors, int64_t M, int* d_out4, hipStream_t s) {
    allow_big_lds(closest_tuple_kernel, palette_smem_bytes(P.kind, P.K));
    hipLaunchKernelGGL(closest_tuple_kernel, dim3(grid_for(M, 256)), dim3(256), palette_smem_bytes(P.kind, P.K), s,
                       P, d_palette, to_lists(lv), d_colors, (long long) M, d_out4);
}
void launch_lookup_only(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_pixels, int64_t N,
                        unsigned short* d_index, int* d_argb, hipStream_t s) {
    allow_big_lds(lookup_only_kernel, palette_smem_bytes(P.kind, P.K));
    hipLaunchKernelGGL(lookup_only_kernel, dim3(grid_for(N, 256, 256 * 16)), dim3(256), palette_smem_bytes(P.kind, P.K), s,
                       P, d_palette, to_lists(lv), d_pixels, (long long) N, d_index, d_argb);
}

template <bool SORTED, int DM>
static void launch_gilbert_t(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const CellLists& L, const int* d_pixels,
                             const float* d_saliency, const int* d_palette, short* d_binCache, long long seed, int sequential,
                             long long* d_rng_state, unsigned short* d_index, int* d_argb, const SeqLog& slog, const int* d_tile_list,
                             hipStream_t s) {
    const int ntiles = T.tiles_x * T.tiles_y;
    const int block = 64;
    int grid = (ntiles + block - 1) / block;
    if (d_tile_list && grid > 256) grid = 256;       // the tiles the fast kernel handed back: few (usually none), walked grid-stride
    const size_t base = palette_tables_smem_bytes(P.kind, P.K);
    const int tilepx = T.tile_w * T.tile_h;
    const bool stage = P.K <= 256 && tilepx <= 256 && !sequential;
    const size_t front = (base + (stage ? (size_t) 64 * tilepx : 0) + 15) & ~(size_t) 15;
    // sorted-by-yDiff queue: in LDS whenever it fits next to the staged palette, else per-lane arrays in scratch.  The queue holds at
    // most 15 boxes with DITHER_MAX 9 and 31 with DITHER_MAX 25 (the growth 1 -> 3 -> 7 -> 15 -> 31 stops at the first size >= DITHER_MAX):
    // 24 or 48 KB per 64-lane workgroup -- with 24 KB four of these one-wavefront workgroups still share a CU (one per SIMD)
    const size_t qbytes = (size_t) 64 * (G.DITHER_MAX <= 15 ? 16 : NQ_QCAP) * 6 * 4;
    if (SORTED && front + qbytes <= 160 * 1024 - 1024) {
        if (stage) {
            allow_big_lds(gilbert_kernel<SORTED, DM, true, SORTED>, front + qbytes);
            hipLaunchKernelGGL((gilbert_kernel<SORTED, DM, true, SORTED>), dim3(grid), dim3(block), front + qbytes, s, P, G, T, L, d_pixels,
                               d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list);
        } else {
            allow_big_lds(gilbert_kernel<SORTED, DM, false, SORTED>, front + qbytes);
            hipLaunchKernelGGL((gilbert_kernel<SORTED, DM, false, SORTED>), dim3(grid), dim3(block), front + qbytes, s, P, G, T, L, d_pixels,
                               d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list);
        }
        return;
    }
    allow_big_lds(gilbert_kernel<SORTED, DM, false>, base);
    if (stage)
        hipLaunchKernelGGL((gilbert_kernel<SORTED, DM, true>), dim3(grid), dim3(block), base + (size_t) 64 * tilepx, s, P, G, T, L, d_pixels,
                           d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list);
    else
        hipLaunchKernelGGL((gilbert_kernel<SORTED, DM, false>), dim3(grid), dim3(block), base, s, P, G, T, L, d_pixels,
                           d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list);
}

void launch_gilbert(const DevParams& P, const GilbertConsts& G, const TileGeom& T, const ListsView& lv, const int* d_pixels,
                    const float* d_saliency, const int* d_palette, short* d_binCache, long long seed, int sequential,
                    long long* d_rng_state, unsigned short* d_index, int* d_argb, int* d_log, int* d_log_count, unsigned char* d_seen,
                    int log_cap, const int* d_tile_list, hipStream_t s) {
    const CellLists L = to_lists(lv);
    SeqLog slog; slog.colors = d_log; slog.count = d_log_count; slog.seen = d_seen; slog.cap = log_cap;
    if (G.sortedByYDiff)
        launch_gilbert_t<true, 1>(P, G, T, L, d_pixels, d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list, s);
    else if (G.DITHER_MAX == 25)
        launch_gilbert_t<false, 25>(P, G, T, L, d_pixels, d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list, s);
    else if (G.DITHER_MAX == 16)
        launch_gilbert_t<false, 16>(P, G, T, L, d_pixels, d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list, s);
    else
        launch_gilbert_t<false, 9>(P, G, T, L, d_pixels, d_saliency, d_palette, d_binCache, seed, sequential, d_rng_state, d_index, d_argb, slog, d_tile_list, s);
}

void launch_bluenoise(const DevParams& P, const int* d_palette, const ListsView& lv, const int* d_pixels, int width, int height,
                      int y_origin, float weight, long long seed, int sequential, short* d_binCache, long long* d_rng_state,
                      unsigned short* d_index, int* d_argb, hipStream_t s) {
    const size_t smem = palette_smem_bytes(P.kind, P.K);
    allow_big_lds(bluenoise_seq_kernel, smem);
    allow_big_lds(bluenoise_kernel, palette_tables_smem_bytes(P.kind, P.K));
    if (sequential)
        hipLaunchKernelGGL(bluenoise_seq_kernel, dim3(1), dim3(64), smem, s, P, d_palette, to_lists(lv), d_pixels, width, height, weight,
                           d_binCache, d_rng_state, d_index, d_argb);
    else
        hipLaunchKernelGGL(bluenoise_kernel, dim3(grid_for((int64_t) width * height, 256, 256 * 16)), dim3(256), palette_tables_smem_bytes(P.kind, P.K), s,
                           P, d_palette, to_lists(lv), d_pixels, width, height, y_origin, weight, seed, d_index, d_argb);
}

// ---- palette build launchers ----
// The histogram sort orders packed words by their upper half only (bits [16, 32)).  rocPRIM 4.2's merge-sort path (default below 2^20
// keys) does not honour a partial bit range -- measured on gfx950: output not even ordered by those bits -- so the path is switched
// off (MergeSortLimit 0): single-block sort up to 1024 keys, Onesweep above, both verified stable on the range.
using HistSortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void) rocprim::radix_sort_keys<HistSortConfig>(nullptr, bytes, (const unsigned*) nullptr, (unsigned*) nullptr, (size_t) n, 16, 32, (hipStream_t) 0);
    return bytes;
}
size_t sort32_temp_bytes(int64_t n, bool pairs) {
    size_t bytes = 0;
    if (pairs)
        (void) rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*) nullptr, (unsigned*) nullptr, (const unsigned*) nullptr,
                                         (unsigned*) nullptr, (size_t) n, 0, 32, (hipStream_t) 0);
    else
        (void) rocprim::radix_sort_keys(nullptr, bytes, (const unsigned*) nullptr, (unsigned*) nullptr, (size_t) n, 0, 32, (hipStream_t) 0);
    return bytes;
}
// distinct colours (after the alpha substitution of the histogram).  keys_a/b, idx_a/b: unsigned[n] scratch (idx_* may be null:
// count only); d_out: unsigned long long[2] {runs, heads written}; d_heads: uint2[cap] {colour, first pixel index} or null
void launch_distinct(const int* d_pixels, int64_t n, int transparentColor, unsigned* keys_a, unsigned* keys_b, unsigned* idx_a,
                     unsigned* idx_b, void* tmp, size_t tmp_bytes, unsigned long long* d_out, void* d_heads, unsigned cap, hipStream_t s) {
    (void) hipMemsetAsync(d_out, 0, 2 * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(subst_colors_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_pixels, (long long) n, transparentColor,
                       keys_a, idx_a);
    size_t tb = tmp_bytes;
    if (idx_a)
        (void) rocprim::radix_sort_pairs(tmp, tb, (const unsigned*) keys_a, keys_b, (const unsigned*) idx_a, idx_b, (size_t) n, 0, 32, s);
    else
        (void) rocprim::radix_sort_keys(tmp, tb, (const unsigned*) keys_a, keys_b, (size_t) n, 0, 32, s);
    hipLaunchKernelGGL(count_runs_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, keys_b, idx_a ? idx_b : nullptr, (long long) n,
                       d_out, (uint2*) d_heads, cap);
}

void launch_color_presence(const int* d_pixels, int64_t n, int transparentColor, unsigned char* d_bytes, unsigned* d_set, unsigned slots,
                           unsigned* d_counters, hipStream_t s) {
    (void) hipMemsetAsync(d_set, 0xFF, (size_t) slots * sizeof(unsigned), s);
    (void) hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned), s);
    hipLaunchKernelGGL(color_presence_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_pixels, (long long) n, transparentColor,
                       d_bytes, d_set, slots, d_counters);
}
void launch_ciede_selftest(const float* d_pairs, int64_t n, unsigned* d_out, hipStream_t s) {
    hipLaunchKernelGGL(ciede_selftest_kernel, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, s, d_pairs, (long long) n, d_out);
}
void launch_prescan(const int* d_pixels, int64_t n, int64_t index_offset, long long* d_scan3, hipStream_t s) {
    (void) hipMemsetAsync(d_scan3, 0xFF, 3 * sizeof(long long), s);      // {-1, -1, -1}: prescan_color_kernel adds the 1 to the count
    hipLaunchKernelGGL(prescan_kernel, dim3(grid_for(n, 256, 256 * 8)), dim3(256), 0, s, d_pixels, (long long) n,
                       (long long) index_offset, d_scan3);
    hipLaunchKernelGGL(prescan_color_kernel, dim3(1), dim3(1), 0, s, d_pixels, (long long) n, (long long) index_offset, d_scan3);
}
bool launch_front(const int* d_pixels, int64_t n, long long* d_scan3, int* d_words, int defaultTransparent, hipStream_t s) {
    if (n < 4 || (n & 3) || ((uintptr_t) d_pixels & 15) || ((uintptr_t) d_words & 15)) return false;
    (void) hipMemsetAsync(d_scan3, 0xFF, 3 * sizeof(long long), s);      // {-1, -1, -1}: prescan_color_kernel adds the 1 to the count
    hipLaunchKernelGGL(front_kernel, dim3(grid_for(n / 4, 256, 256 * 8)), dim3(256), 0, s, (const int4*) d_pixels, (long long) (n / 4), 0LL,
                       d_scan3, (uint4*) d_words, defaultTransparent);
    hipLaunchKernelGGL(prescan_color_kernel, dim3(1), dim3(1), 0, s, d_pixels, (long long) n, 0LL, d_scan3);
    return true;
}
void launch_histogram(int kind, const int* d_pixels, int64_t n, const HistParams& hp, const SortWorkspace& ws,
                      double* d_hist, hipStream_t s, bool words_ready) {
    unsigned* const pk_a = reinterpret_cast<unsigned*>(ws.vals_a);
    unsigned* const pk_b = reinterpret_cast<unsigned*>(ws.vals_b);
    if (!words_ready)
        hipLaunchKernelGGL(bin_keys_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, d_pixels, (long long) n, hp, pk_a);
    size_t tmp = ws.tmp_bytes;
    (void) rocprim::radix_sort_keys<HistSortConfig>(ws.tmp, tmp, (const unsigned*) pk_a, pk_b, (size_t) n, 16, 32, s);
    unsigned* const occ_count = ws.seg_end + 65536;             // [0] occupied bins, [1] fat bins
    unsigned* const occ_list = occ_count + 64;
    unsigned* const fat_list = occ_list + 65536;                // [NQ_FAT_CAP]
    // RGB kind: a bin is "fat" (a workgroup of its own, hist_fat_rgb_kernel) from 16 384 pixels up; never more than NQ_FAT_CAP - 1 of them
    unsigned fat_min = 0xFFFFFFFFu;
    if (kind != 1) {
        fat_min = 16384u;
        if (const char* f = std::getenv("NQ_HIST_FAT_MIN")) { const long t = std::atol(f); if (t >= 1) fat_min = (unsigned) std::min<long>(t, 0x7FFFFFFFL); }
        fat_min = std::max(fat_min, (unsigned) (n / NQ_FAT_CAP) + 1u);
    }
    (void) hipMemsetAsync(occ_count, 0, 2 * sizeof(unsigned), s);
    hipLaunchKernelGGL(occupied_bins_kernel, dim3(64), dim3(1024), 0, s, (const unsigned*) pk_b, (unsigned) n, ws.seg_start, ws.seg_end, occ_count, occ_list, d_hist,
                       fat_min, fat_list);
    const int keyfmt = hp.hasSemi ? 2 : hp.hasTransp ? 1 : 0;          // getColorIndex: 4-4-4-4 / 1-5-5-5 / 5-6-5
    if (kind == 1)
        hipLaunchKernelGGL(hist_segments_kernel<1>, dim3(65536 / 4), dim3(256), keyfmt == 2 ? 0 : (size_t) 4 * (keyfmt == 0 ? 256 : 512) * 16, s,
                           (const unsigned*) pk_b, ws.seg_start, ws.seg_end, d_hist, keyfmt, (const unsigned*) occ_count, (const unsigned*) occ_list, fat_min);
    else {
        // the fat bins first (usually none: the workgroups leave at once), then one wavefront per ordinary bin
        hipLaunchKernelGGL(hist_fat_rgb_kernel, dim3(NQ_FAT_WGS), dim3(256), 0, s,
                           (const unsigned*) pk_b, ws.seg_start, ws.seg_end, d_hist, keyfmt, (const unsigned*) occ_count, (const unsigned*) fat_list);
        hipLaunchKernelGGL(hist_segments_kernel<0>, dim3(65536 / 4), dim3(256), 0, s, (const unsigned*) pk_b, ws.seg_start, ws.seg_end, d_hist, keyfmt,
                           (const unsigned*) occ_count, (const unsigned*) occ_list, fat_min);
    }
}
void launch_compact(int kind, const double* d_hists, int n_bands, const Bins& B, int* d_maxbins, int* d_blockcnt, hipStream_t s) {
    hipLaunchKernelGGL(compact_count_kernel, dim3(64), dim3(1024), 0, s, d_hists, n_bands, d_blockcnt);
    if (kind == 1) hipLaunchKernelGGL(compact_means_kernel<1>, dim3(64), dim3(1024), 0, s, d_hists, n_bands, B, d_blockcnt, d_maxbins);
    else hipLaunchKernelGGL(compact_means_kernel<0>, dim3(64), dim3(1024), 0, s, d_hists, n_bands, B, d_blockcnt, d_maxbins);
}
void launch_quanfn(float* d_cnt, int maxbins, int fn, hipStream_t s) {
    if (fn == 0 || maxbins <= 0) return;
    hipLaunchKernelGGL(quanfn_kernel, dim3((maxbins + 255) / 256), dim3(256), 0, s, d_cnt, maxbins, fn);
}
void launch_find_nn_init(const NNParams& np, const Bins& B, int maxbins, float* d_box, int* d_init_cand, hipStream_t s) {
    if (maxbins <= 0) return;
    if (np.kind == 1) {
        const int nblk = (maxbins + 63) / 64;
        int2* cand = reinterpret_cast<int2*>(d_init_cand);
        int* ncand = d_init_cand + (size_t) 65536 * 128 * 2;
        if (np.ratio >= 0.0 && np.ratio <= 1.0) {
            hipLaunchKernelGGL(init_boxes_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, B, maxbins, d_box);
            hipLaunchKernelGGL(find_nn_init_lab_bounds_kernel, dim3((maxbins + 3) / 4), dim3(256), 0, s, np, B, maxbins, (const float*) d_box, cand, ncand);
        }
        hipLaunchKernelGGL(find_nn_init_lab_exact_kernel, dim3((maxbins + 3) / 4), dim3(256), 0, s, np, B, maxbins, (const int2*) cand, (const int*) ncand);
    }
    else {
        const int nblk = (maxbins + 63) / 64;
        hipLaunchKernelGGL(init_boxes_rgb_kernel, dim3((nblk + 255) / 256), dim3(256), 0, s, B, maxbins, d_box);
        hipLaunchKernelGGL(find_nn_init_rgb_kernel, dim3((maxbins + 3) / 4), dim3(256), 0, s, np, B, maxbins, (const float*) d_box);
    }
}
// one workgroup per job; the workgroup size follows the number of jobs in flight (nq_merge.inc)
template <typename K>
static hipError_t launch_merge_variant(K kernel, size_t dyn, int threads, const MergeJob* d_jobs, int n, hipStream_t s, int n_pad = 0, int roles = 1) {
    const hipError_t e = hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(roles > 1 ? n_pad * roles : n), dim3(threads), dyn, s, d_jobs, n, n_pad);
    return hipGetLastError();
}
// XCDs of the device: workgroups are dealt out to them round-robin by blockIdx, so a team whose members share blockIdx % NQ_XCDS shares
// an L2.  There is no HIP attribute for it (gfx950: 8); a different count only costs the teams their shared L2, never a result.
#define NQ_XCDS 8
// 512 / 256 / 128 threads, or 127 = the dense 128-thread variant (six workgroups per CU: more than four loops per CU in flight)
static int merge_threads_for(int n_in_flight, int n_cus) {
    if (const char* f = std::getenv("NQ_MERGE_THREADS")) {       // tests: force one variant (512, 256, 128 or 127)
        const int t = std::atoi(f);
        if (t == 127 || t == 128 || t == 256 || t == 512) return t;
    }
    if (n_cus < 1) n_cus = 1;
    return n_in_flight <= n_cus ? 512 : n_in_flight <= 2 * n_cus ? 256 : n_in_flight <= 4 * n_cus ? 128 : 127;
}
int merge_team_helpers(int n_jobs, int n_in_flight, int n_cus) {
    if (n_jobs <= 0 || merge_threads_for(n_in_flight, n_cus) != 512) return 0;
    const int n_pad = (n_jobs + NQ_XCDS - 1) / NQ_XCDS * NQ_XCDS;
    int h = n_cus / n_pad - 1;                     // one 512-thread workgroup per CU
    if (h > 7) h = 7;
    if (h < 0) h = 0;
    if (const char* f = std::getenv("NQ_MERGE_HELPERS")) {       // tests / measurements: 0 = the single-workgroup loop
        const int t = std::atoi(f);
        if (t >= 0 && t <= 7) h = t;
    }
    return h;
}
hipError_t launch_merge(int kind, const MergeJob* d_jobs, int n, int n_in_flight, int n_cus, int helpers, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int threads = merge_threads_for(n_in_flight, n_cus);
    // Phase stamps (nq_merge.inc, STATS) cost ~20 scalar-cache round trips per find_nn event: every variant runs without them unless
    // NQ_MERGE_STATS=1 asks for the stamped build (tools/latency.py, tools/batch_rate.py: the tick fields of nq_get_merge_stats /
    // nq_get_team_stats are 0 otherwise; the event counters are always there)
    const char* fs = std::getenv("NQ_MERGE_STATS");
    const bool stats = fs && std::atoi(fs) == 1;
    if (threads == 512) {
        if (helpers > 0) {
            const int n_pad = (n + NQ_XCDS - 1) / NQ_XCDS * NQ_XCDS;
            if (kind == 1) return stats ? launch_merge_variant(m512::merge_kernel<1, true, true>, sizeof(m512::MergeLds), 512, d_jobs, n, s, n_pad, 1 + helpers)
                                        : launch_merge_variant(m512::merge_kernel<1, true, false>, sizeof(m512::MergeLds), 512, d_jobs, n, s, n_pad, 1 + helpers);
            return stats ? launch_merge_variant(m512::merge_kernel<0, true, true>, sizeof(m512::MergeLds), 512, d_jobs, n, s, n_pad, 1 + helpers)
                         : launch_merge_variant(m512::merge_kernel<0, true, false>, sizeof(m512::MergeLds), 512, d_jobs, n, s, n_pad, 1 + helpers);
        }
        if (kind == 1) return stats ? launch_merge_variant(m512::merge_kernel<1, false, true>, sizeof(m512::MergeLds), 512, d_jobs, n, s)
                                    : launch_merge_variant(m512::merge_kernel<1, false, false>, sizeof(m512::MergeLds), 512, d_jobs, n, s);
        return launch_merge_variant(m512::merge_kernel<0, false, false>, sizeof(m512::MergeLds), 512, d_jobs, n, s);
    }
    if (threads == 256) {
        if (kind == 1) return launch_merge_variant(m256::merge_kernel<1, false, false>, sizeof(m256::MergeLds), 256, d_jobs, n, s);
        return launch_merge_variant(m256::merge_kernel<0, false, false>, sizeof(m256::MergeLds), 256, d_jobs, n, s);
    }
    if (threads == 127) {
        if (kind == 1) return launch_merge_variant(m128d::merge_kernel<1, false, false>, sizeof(m128d::MergeLds), 128, d_jobs, n, s);
        return launch_merge_variant(m128d::merge_kernel<0, false, false>, sizeof(m128d::MergeLds), 128, d_jobs, n, s);
    }
    if (kind == 1) {
        if (stats) return launch_merge_variant(m128::merge_kernel<1, false, true>, sizeof(m128::MergeLds), 128, d_jobs, n, s);
        return launch_merge_variant(m128::merge_kernel<1, false, false>, sizeof(m128::MergeLds), 128, d_jobs, n, s);
    }
    return launch_merge_variant(m128::merge_kernel<0, false, false>, sizeof(m128::MergeLds), 128, d_jobs, n, s);
}
} // namespace nq
