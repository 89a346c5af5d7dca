// nq_abi.cpp -- C ABI of libnquant_hip.so (include/nquant_abi.h): handle, device workspace, and the control plane of
// the reference's convert() (scalar heuristics, GilbertCurve constructor ladder, curve tables).  All per-pixel and
// per-bin work is launched on the GPU (nq_kernels.hip); there is no CPU fallback.
//
// NQ/ = nQuant.master/src/main/java/com/android/nQuant/ in the reference.
#include "../../include/nquant_abi.h"
#include "nq_kernels.h"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include <thread>
#include <exception>

using namespace nq;

static thread_local std::string g_create_error;

namespace {

const float kCoeffs[3][3] = {{0.299f, 0.587f, 0.114f}, {-0.14713f, -0.28886f, 0.436f}, {0.615f, -0.51499f, -0.10001f}};

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t reserve(size_t count) {
        if (count <= n) return hipSuccess;
        if (p) { (void) hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void**) &p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
};

// Java narrowing (control-plane use)
inline int j_d2i(double d) {
    if (d != d) return 0;
    if (d >= 2147483647.0) return 2147483647;
    if (d <= -2147483648.0) return (int) 0x80000000;
    return (int) d;
}
inline signed char j_d2b(double d) { return (signed char) (unsigned char) (j_d2i(d) & 0xFF); }
inline double sqr(double v) { return v * v; }

// GilbertCurve.initWeights (NQ/GilbertCurve.java:336-354): weights only (the zero boxes are enqueued by the kernel)
void init_weights(float* weights, int size) {
    const float weightRatio = (float) std::pow((double) (343.0f + 1.0f), (double) (1.0f / (size - 1.0f)));
    float weight = 1.0f, sumweight = 0.0f;
    for (int c = 0; c < size; ++c) {
        sumweight += (weights[size - c - 1] = weight);
        weight /= weightRatio;
    }
    weight = 0.0f;
    for (int c = 0; c < size; ++c) weight += (weights[c] /= sumweight);
    weights[0] += 1.0f - weight;
}

// GilbertCurve constructor (NQ/GilbertCurve.java:50-112); `weight` is the signed constructor parameter
GilbertConsts gilbert_consts(int K, double weight, bool hasSaliencies, bool dither) {
    GilbertConsts g;
    std::memset(&g, 0, sizeof g);
    const bool hasAlpha = weight < 0;
    g.hasAlphaW = hasAlpha; g.hasSaliencies = hasSaliencies; g.dither = dither;
    g.weightAbs = std::fabs(weight);
    g.margin = weight < .0025 ? 12 : weight < .004 ? 8 : 6;
    g.sortedByYDiff = K > 128 && weight >= .02 && (!hasAlpha || weight < .18);
    float beta = K > 4 ? (float) (.6f - .00625f * K) : 1;
    if (K > 4) {
        double boundary = .005 - .0000625 * K;
        beta = (float) (weight > boundary ? .25 : std::fmin(1.5, beta + K * weight));
        if (K > 16 && K <= 32 && weight < .003) beta += .075f;
        else if (weight < .0015 || (K > 32 && K < 256)) beta += .1f;
        if ((K >= 64 && (weight > .012 && weight < .0125)) || (weight > .025 && weight < .03)) beta += .05f;
        else if (K > 32 && K < 64 && weight < .015) beta = .55f;
        else if (K > 16 && K <= 32 && weight <= .005) beta += (float) (.05 + weight * K);
    }
    else beta *= .95f;
    if (K > 64 || (K > 4 && weight > .02)) beta *= .4f;
    if (K > 64 && weight < .02) beta = .18f;
    signed char DITHER_MAX = weight < .015 ? ((weight > .0025) ? (signed char) 25 : (signed char) 16) : (signed char) 9;
    if (weight > .99) { beta = (float) weight; DITHER_MAX = 25; }
    const double edge = hasAlpha ? 1 : std::exp(weight) - .25;
    const double deviation = weight > .002 ? -.25 : 1;
    signed char ditherMax = (hasAlpha || DITHER_MAX > 9) ? j_d2b(sqr(std::sqrt((double) DITHER_MAX) + edge * deviation))
                                                         : j_d2b(DITHER_MAX * (hasSaliencies ? 2 : 2.718281828459045));
    const int density = K > 16 ? 3200 : 1500;
    if (K / weight > 5000 && (weight > .045 || (weight > .01 && K < 64))) ditherMax = j_d2b(sqr(5 + edge));
    else if (weight < .03 && K / weight < density && K >= 16 && K < 256) ditherMax = j_d2b(sqr(5 + edge));
    g.thresold = DITHER_MAX > 9 ? -112 : -64;
    g.beta = beta; g.DITHER_MAX = DITHER_MAX; g.ditherMax = ditherMax;
    init_weights(g.weights, DITHER_MAX);
    init_weights(g.w1, 1); init_weights(g.w3, 3); init_weights(g.w7, 7); init_weights(g.w15, 15);
    return g;
}

// generalized Hilbert ("gilbert") curve, NQ/GilbertCurve.java:282-334 + run() :361-364: entries dx | dy << 16
inline int sgn(int v) { return (v > 0) - (v < 0); }
void gilbert_rec(std::vector<uint32_t>& out, int x, int y, int ax, int ay, int bx, int by) {
    const int w = std::abs(ax + ay), h = std::abs(bx + by);
    const int dax = sgn(ax), day = sgn(ay), dbx = sgn(bx), dby = sgn(by);
    if (h == 1) { for (int i = 0; i < w; ++i) { out.push_back((uint32_t) x | ((uint32_t) y << 16)); x += dax; y += day; } return; }
    if (w == 1) { for (int i = 0; i < h; ++i) { out.push_back((uint32_t) x | ((uint32_t) y << 16)); x += dbx; y += dby; } return; }
    int ax2 = ax / 2, ay2 = ay / 2, bx2 = bx / 2, by2 = by / 2;
    const int w2 = std::abs(ax2 + ay2), h2 = std::abs(bx2 + by2);
    if (2 * w > 3 * h) {
        if ((w2 % 2) != 0 && w > 2) { ax2 += dax; ay2 += day; }
        gilbert_rec(out, x, y, ax2, ay2, bx, by);
        gilbert_rec(out, x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by);
        return;
    }
    if ((h2 % 2) != 0 && h > 2) { bx2 += dbx; by2 += dby; }
    gilbert_rec(out, x, y, bx2, by2, ax2, ay2);
    gilbert_rec(out, x + bx2, y + by2, ax, ay, bx - bx2, by - by2);
    gilbert_rec(out, x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), -bx2, -by2, -(ax - ax2), -(ay - ay2));
}
std::vector<uint32_t> gilbert_path(int w, int h) {
    std::vector<uint32_t> out;
    out.reserve((size_t) w * h);
    if (w <= 0 || h <= 0) return out;
    if (w >= h) gilbert_rec(out, 0, 0, w, 0, 0, h);
    else gilbert_rec(out, 0, 0, 0, h, w, 0);
    return out;
}

} // namespace

// per-pixel scratch that lives only inside one stage of one image: the handles of a batch share the first handle's
struct Scratch {
    DevBuf<unsigned short> keys_a, keys_b;
    DevBuf<int> vals_a, vals_b;
    DevBuf<unsigned char> sort_tmp;
    DevBuf<unsigned> seg;             // start[65536], end[65536], counters (+ padding), list of the occupied bins[65536], of the fat bins[1024]
    DevBuf<double> hist;              // [65536][5]
    DevBuf<int> init_cand;            // initial LAB pass: candidate lists {bin, bound}[65536][128], then the counts [65536]
    DevBuf<unsigned char> cell_lists; // closest lists, nearest lists (65536 x 32 each), then their counts (65536 each)
    DevBuf<float> saliency;           // saliency map of the image being dithered
    DevBuf<unsigned> lookup_todo;     // LOOKUP_ONLY: {count, pixel indices the float32 pass leaves to the exact pass}
    DevBuf<unsigned> dk_a, dk_b, di_a, di_b;   // distinct-colour sort scratch
    DevBuf<unsigned char> dtmp;
    DevBuf<unsigned long long> dheads;         // {colour, first index} pairs (uint2)
    DevBuf<float> cell_box;           // Lab bounding box of every 5-6-5 cell (palette independent, built once)
    bool cell_box_ready = false;
};

struct nq_handle {
    int kind = 0, device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    nq_params params;
    int tile_w = 0, tile_h = 0;       // 0 = automatic (pick_tile)
    float stage_ms[NQ_N_STAGES] = {0};
    Scratch own;
    Scratch* sc = &own;
    // device workspace
    DevBuf<int> d_palette, d_in, d_out_argb, d_colors, d_tuple;
    DevBuf<unsigned short> d_out_index;
    DevBuf<short> d_bincache, d_short;
    DevBuf<int> d_seqlog;             // REFERENCE_SEQUENTIAL + LAB + dither=false: colours getLab() saw during the pass (+ 1 counter)
    DevBuf<unsigned char> d_seqseen;  // ... and the palette entries it touched
    DevBuf<long long> d_scalars;      // [0] rng state, [1..3] scan3, [4..19] merge stats, [20..21] distinct-colour result, [24..39] team counters, [40] palette status
    DevBuf<int> live3;                // merge loop: two live lists + position index
    int use_lists = 1;
    int n_cus = 256;                  // compute units of the device (hipDeviceAttributeMultiprocessorCount): sizes merge workgroups / teams
    int merge_wall_s = 0;             // NQ_OPT_MERGE_WALL_SECONDS: watchdog of a merge loop, seconds of residency (0 = automatic)
    hipStream_t palette_stream = nullptr;   // stream the upload behind dev_palette was enqueued on ...
    bool palette_synced = false;            // ... unless the host has waited for d_palette's content since (then any stream may read it)
    // nq_gilbert_dither / nq_bluenoise_dither: the static entry points of the reference run the same stages with caller-supplied
    // saliencies / weight instead of the ones dither() derives
    struct StageOverride { bool gilbert_only = false, blue_only = false; const float* d_sal = nullptr; bool hasSal = false; double weight = 0; float blueWeight = 1.f; } ov;
    DevBuf<float> d_user_sal;
    int band_y0 = 0, band_image_h = 0; // nq_set_band: this handle dithers a row band of a larger image (0, 0 = a whole image)
    int use_fast_dither = 1;          // NQ_OPT_FAST_DITHER: the specialised dither kernel where the configuration allows it
    bool dither_events_fresh = false; // the last call on the handle was a stand-alone dither (events 5..7 are newer than stage_ms)
    int last_dither_fast = 0;         // diagnostics: 1 if the last dither pass ran gilbert_fast_kernel
    int last_dither_failed_tiles = 0; // ... and how many tiles it handed back to the generic kernel (read lazily)
    DevBuf<int> d_failed;             // {count, tile indices...} of those tiles
    DevBuf<float> scan_f;             // merge loop: position-indexed scan arrays (two generations)
    DevBuf<int> scan_i;
    DevBuf<float> scan_box;           // merge loop: bounding boxes of the 64-position blocks (1024 x 8 floats)
    DevBuf<nq::MergeJob> d_jobs;      // merge jobs of the current call (1, or the whole batch on the first handle)
    DevBuf<int> ring_argb[3];         // nq_convert_batch: device output ring (results leave for the host while the next image runs)
    DevBuf<unsigned short> ring_index[3];
    hipStream_t copy_stream = nullptr;
    hipStream_t lane_stream = nullptr;  // second lane of the batch entry points
    hipStream_t more_lanes[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // ... third to eighth
    long long merge_stats[16] = {0};
    std::vector<uint32_t> dev_palette;  // what d_palette holds, as far as the host knows (empty: unknown): an upload of the same entries is skipped
    uint32_t* fetched_palette = nullptr; int fetched_len = 0;      // palette_fetch -> palette_check
    long long merge_readback[37] = {0}; // d_scalars[4..41) as the merge kernel left it: one copy per image ([36] = status word)
    long long team_stats[16] = {0};    // merge teams: {work records published, results used, timed-out waits, ticks waited, helpers, still speculating}
    DevBuf<unsigned long long> team;  // 256 u64 of hand-off words of this handle's merge team
    bool light_events = false;        // batch entry points: this image records only the stage events 0, 5, 6 (see rec)
    hipEvent_t bev[4] = {nullptr};    // batch entry points: phase boundaries on the launch stream (first handle of the batch)
    float batch_phase_ms[4] = {0};
    bool ext_distinct_valid = false, ext_distinct_many = false;  // nq_set_distinct: image-wide distinct colours (first-occurrence order) of the split pipeline
    std::vector<int32_t> ext_distinct;
    DevBuf<int> d_ints;               // [0] maxbins, [1] status, [8..71] occupied slots per 1024-slot slice
    DevBuf<int> heap;
    DevBuf<float> binf;               // f[4], cnt, err : 6 x 65536
    DevBuf<double> bind;              // d[4] : 4 x 65536
    DevBuf<int> bini;                 // nn, tm, mtm : 3 x 65536
    std::map<std::pair<int, int>, uint32_t*> paths;   // curve tables on the device, by shape
    hipEvent_t ev[NQ_N_STAGES + 1] = {nullptr};
    bool tables_ready = false;
    ~nq_handle() {
        for (auto& kv : paths) (void) hipFree(kv.second);
        for (auto& e : ev) if (e) (void) hipEventDestroy(e);
        for (auto& e : bev) if (e) (void) hipEventDestroy(e);
        if (copy_stream) (void) hipStreamDestroy(copy_stream);
        if (lane_stream) (void) hipStreamDestroy(lane_stream);
        for (auto& ls : more_lanes) if (ls) (void) hipStreamDestroy(ls);
    }
};

// what the launches since the last check left behind: a failed hipFuncSetAttribute of a launch_* function (kept per thread by
// nq_kernels.hip) or the runtime's own last error
static inline hipError_t launch_status() {
    const hipError_t a = nq::take_launch_error(), b = hipGetLastError();
    return a != hipSuccess ? a : b;
}

#define NQ_FAIL(h, code, ...) do { char _b[512]; std::snprintf(_b, sizeof _b, __VA_ARGS__); (h)->err = _b; return (code); } while (0)
#define NQ_HIP(h, call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
    NQ_FAIL(h, NQ_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

namespace {

int use_device(nq_handle* h) {
    NQ_HIP(h, hipSetDevice(h->device));
    if (!h->tables_ready) {
        double gamma[256];
        for (int ch = 0; ch < 256; ++ch) {         // CIELABConvertor.gammaToLinear (NQ/CIELABConvertor.java:71-75)
            const double c = ch / 255.0;
            gamma[ch] = c < 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
        }
        upload_tables(gamma, std::exp(1.5), std::exp(1.75), h->stream);
        upload_tables_fast(gamma, std::exp(1.5), std::exp(1.75), h->stream);
        NQ_HIP(h, launch_status());
        int cus = 0;
        NQ_HIP(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        if (cus > 0) h->n_cus = cus;
        NQ_HIP(h, h->d_scalars.reserve(64));
        NQ_HIP(h, h->d_ints.reserve(8 + 64));
        NQ_HIP(h, h->d_bincache.reserve(65536));
        for (auto& e : h->ev) NQ_HIP(h, hipEventCreate(&e));
        for (auto& e : h->bev) NQ_HIP(h, hipEventCreate(&e));
        h->tables_ready = true;
    }
    return NQ_OK;
}

DevParams dev_params(const nq_handle* h, int K) {
    const nq_params& p = h->params;
    DevParams d;
    d.kind = h->kind; d.K = K; d.hasSemi = p.hasSemiTransparency; d.hasAlpha = p.transparentPixelIndex > -1;
    d.transparentColor = p.transparentColor; d.isNano = p.isNano;
    d.binKeyed = h->kind == NQ_KIND_LAB ? (p.isNano != 0) : !(p.weight > .015);
    d.nMaxColors = p.nMaxColors; d.rewriteA0 = p.nMaxColors <= 2 && p.nMaxColors > 0;
    d.pad = 0;
    d.PR = p.PR; d.PG = p.PG; d.PB = p.PB; d.PA = p.PA; d.ratio = p.ratio; d.weight = p.weight;
    return d;
}

// the packed list records of the specialised kernels (nq_dither_fast.hip) live behind the lists and their counts
// the palette of a per-pixel pass on the device; skipped when d_palette already holds exactly these entries (convert(): the merge
// workgroup wrote them there and the host read them back -- one small copy per image less)
int upload_palette(nq_handle* h, const uint32_t* palette, int K) {
    const size_t cap = h->d_palette.n;
    NQ_HIP(h, h->d_palette.reserve((size_t) std::max(K, 2)));
    if (h->d_palette.n != cap) h->dev_palette.clear();          // (a new allocation)
    // (an earlier upload is ordered only against work on the stream it was enqueued on: after nq_set_stream the copy is repeated)
    const bool ordered = h->palette_synced || h->palette_stream == h->stream;
    if (ordered && (int) h->dev_palette.size() == K && std::memcmp(h->dev_palette.data(), palette, (size_t) K * sizeof(uint32_t)) == 0) return NQ_OK;
    h->dev_palette.clear();
    NQ_HIP(h, hipMemcpyAsync(h->d_palette.p, palette, K * sizeof(int), hipMemcpyHostToDevice, h->stream));
    h->dev_palette.assign(palette, palette + K);
    h->palette_stream = h->stream; h->palette_synced = false;
    return NQ_OK;
}
void* packed_lists(nq_handle* h) { return h->sc->cell_lists.p + 2 * (size_t) 65536 * 32 + 2 * 65536; }

// candidate lists per colour cell for this palette (nq_lists.inc); empty view = full scans
// sal_pixels != null: the saliency map of these n pixels is wanted in h->sc->saliency as well; *sal_done says whether it was built here
// (beside the LAB list builders, one launch) or is left to the caller
int prepare_lists(nq_handle* h, const DevParams& P, nq::ListsView* out, const int* sal_pixels = nullptr, int64_t sal_n = 0, int sal_subst = 0,
                  bool* sal_done = nullptr) {
    if (sal_done) *sal_done = false;
    out->closest = out->closestCount = out->nearest = out->nearestCount = nullptr;
    if (!h->use_lists || P.K > 256 || P.K < 8) return NQ_OK;
    const size_t LB = (size_t) 65536 * 32;
    NQ_HIP(h, h->sc->cell_lists.reserve(2 * LB + 2 * 65536 + 2 * LB));     // + the packed records of the specialised dither kernel
    unsigned char* base = h->sc->cell_lists.p;
    double wA, wR, wG, wB;
    if (h->kind == NQ_KIND_LAB) {
        // err of NQ/PnnLABQuantizer.java:421-445: PR(1-ratio) dr^2 + ... + ratio * sum_i (coeffs[i][c] d)^2
        double s[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) for (int c = 0; c < 3; ++c) s[c] += (double) kCoeffs[i][c] * (double) kCoeffs[i][c];
        wR = P.PR * (1 - P.ratio) + P.ratio * s[0]; wG = P.PG * (1 - P.ratio) + P.ratio * s[1]; wB = P.PB * (1 - P.ratio) + P.ratio * s[2];
        wA = P.hasSemi ? P.PA : 0.0;
    } else {
        // NQ/PnnQuantizer.java:325-345
        double pr = P.PR, pg = P.PG, pb = P.PB, pa = P.PA;
        if (P.K < 3) pr = pg = pb = pa = 1;
        wR = pr; wG = pg; wB = pb; wA = P.hasSemi ? pa : 0.0;
    }
    // nearestColorIndex lists: LAB for the K > 32 metric; RGB for opaque images (no transparent colour: the scan starts at 0)
    const bool nearest = h->kind == NQ_KIND_LAB ? (P.K > 32 && !P.hasSemi) : (!P.hasSemi && !P.hasAlpha);
    if (nearest && h->kind == NQ_KIND_LAB && !h->sc->cell_box_ready) {
        NQ_HIP(h, h->sc->cell_box.reserve((size_t) 65536 * 6));
        launch_cell_lab_box(h->sc->cell_box.p, h->stream);
        h->sc->cell_box_ready = true;
    }
    if (sal_pixels) NQ_HIP(h, h->sc->saliency.reserve((size_t) sal_n));
    const bool sal_built = launch_build_lists(P, h->d_palette.p, wA, wR, wG, wB, nearest, h->sc->cell_box.p, base, base + 2 * LB, base + LB,
                                              base + 2 * LB + 65536, h->stream, sal_pixels, sal_n, sal_pixels ? h->sc->saliency.p : nullptr, sal_subst);
    if (sal_done) *sal_done = sal_built;
    // (a negative ratio makes the closest error non-monotone in its terms: the list argument does not hold, full scans)
    if (!(h->kind == NQ_KIND_LAB && P.ratio < 0)) { out->closest = base; out->closestCount = base + 2 * LB; }
    if (nearest) { out->nearest = base + LB; out->nearestCount = base + 2 * LB + 65536; }
    if (h->use_fast_dither && fast_pack_wanted(P, *out)) launch_pack_lists(*out, packed_lists(h), h->stream);
    return NQ_OK;
}


int get_path(nq_handle* h, int w, int hgt, const uint32_t** out) {
    auto key = std::make_pair(w, hgt);
    auto it = h->paths.find(key);
    if (it == h->paths.end()) {
        std::vector<uint32_t> p = gilbert_path(w, hgt);
        uint32_t* d = nullptr;
        NQ_HIP(h, hipMalloc((void**) &d, (p.size() ? p.size() : 1) * sizeof(uint32_t)));
        hipError_t e = hipMemcpyAsync(d, p.data(), p.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);   // p goes out of scope
        if (e != hipSuccess) { (void) hipFree(d); NQ_FAIL(h, NQ_ERR_HIP, "curve table upload failed: %s", hipGetErrorString(e)); }
        it = h->paths.emplace(key, d).first;
    }
    *out = it->second;
    return NQ_OK;
}

nq::Bins bins_of(nq_handle* h) {
    nq::Bins B;
    for (int c = 0; c < 4; ++c) { B.f[c] = h->binf.p + (size_t) c * 65536; B.d[c] = h->bind.p + (size_t) c * 65536; }
    B.cnt = h->binf.p + (size_t) 4 * 65536; B.err = h->binf.p + (size_t) 5 * 65536;
    B.nn = h->bini.p; B.tm = h->bini.p + 65536; B.mtm = h->bini.p + 2 * 65536;
    return B;
}

int reserve_palette_ws(nq_handle* h, int64_t n) {
    NQ_HIP(h, h->sc->vals_a.reserve((size_t) n)); NQ_HIP(h, h->sc->vals_b.reserve((size_t) n));
    NQ_HIP(h, h->sc->sort_tmp.reserve(sort_temp_bytes(n) + 256));
    NQ_HIP(h, h->sc->seg.reserve(3 * 65536 + 64 + 1024));     // start[65536], end[65536], counters, occupied-bin list[65536], fat-bin list[1024]
    NQ_HIP(h, h->sc->hist.reserve((size_t) 65536 * 5));
    if (h->kind == 1) NQ_HIP(h, h->sc->init_cand.reserve((size_t) 65536 * 128 * 2 + 65536));
    NQ_HIP(h, h->binf.reserve((size_t) 6 * 65536)); NQ_HIP(h, h->bind.reserve((size_t) 4 * 65536));
    NQ_HIP(h, h->bini.reserve((size_t) 3 * 65536)); NQ_HIP(h, h->heap.reserve(2 * (65536 + 2)));
    NQ_HIP(h, h->live3.reserve((size_t) 3 * 65536));
    NQ_HIP(h, h->scan_f.reserve((size_t) 2 * 10 * 65536 + 256)); NQ_HIP(h, h->scan_i.reserve((size_t) 2 * 65536));
    NQ_HIP(h, h->scan_box.reserve((size_t) 1024 * 8));
    NQ_HIP(h, h->team.reserve(256));
    return NQ_OK;
}

// number of distinct colours of the image as the histogram sees it (= pixelMap.size() after the histogram); when it is
// <= cap the colours are returned in first-occurrence order (the insertion order of the reference's HashMap)
int distinct_colors(nq_handle* h, const uint32_t* d_argb, int64_t n, int64_t cap, int64_t* out_count, std::vector<int32_t>* out_colors) {
    const bool want = out_colors != nullptr && cap > 0;
    NQ_HIP(h, h->sc->dk_a.reserve((size_t) n)); NQ_HIP(h, h->sc->dk_b.reserve((size_t) n));
    if (want) { NQ_HIP(h, h->sc->di_a.reserve((size_t) n)); NQ_HIP(h, h->sc->di_b.reserve((size_t) n)); NQ_HIP(h, h->sc->dheads.reserve((size_t) cap + 1)); }
    const size_t tb = sort32_temp_bytes(n, want) + 256;
    NQ_HIP(h, h->sc->dtmp.reserve(tb));
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(h->d_scalars.p + 20);
    launch_distinct((const int*) d_argb, n, h->params.transparentColor, h->sc->dk_a.p, h->sc->dk_b.p, want ? h->sc->di_a.p : nullptr,
                    want ? h->sc->di_b.p : nullptr, h->sc->dtmp.p, tb, d_out, want ? (void*) h->sc->dheads.p : nullptr, (unsigned) cap, h->stream);
    unsigned long long res[2] = {0, 0};
    NQ_HIP(h, hipMemcpyAsync(res, d_out, sizeof res, hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    NQ_HIP(h, launch_status());
    *out_count = (int64_t) res[0];
    if (want && (int64_t) res[0] <= cap) {
        std::vector<unsigned long long> heads(res[0]);
        NQ_HIP(h, hipMemcpy(heads.data(), h->sc->dheads.p, res[0] * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        // uint2 {colour, index} little endian: low word = colour, high word = first index
        std::vector<std::pair<uint32_t, uint32_t>> byIndex;
        for (unsigned long long v : heads) byIndex.emplace_back((uint32_t) (v >> 32), (uint32_t) (v & 0xFFFFFFFFu));
        std::sort(byIndex.begin(), byIndex.end());
        out_colors->clear();
        for (auto& pr : byIndex) out_colors->push_back((int32_t) pr.second);
    }
    return NQ_OK;
}

// keySet() order of java.util.HashMap<Integer, ?> for keys inserted in the given order (OpenJDK 8+; treeified buckets ignored):
// capacity 16 doubling while size > 0.75 capacity; bucket = (h ^ h >>> 16) & (capacity - 1), h = key; insertion order inside
std::vector<int32_t> java_hashmap_keyset(const std::vector<int32_t>& inserted) {
    size_t cap = 16;
    while ((double) inserted.size() > 0.75 * (double) cap) cap <<= 1;
    std::vector<std::vector<int32_t>> buckets(cap);
    for (int32_t k : inserted) { uint32_t hh = (uint32_t) k; hh ^= hh >> 16; buckets[hh & (cap - 1)].push_back(k); }
    std::vector<int32_t> out;
    for (auto& b : buckets) for (int32_t k : b) out.push_back(k);
    return out;
}

// the scalar part of convert() after the pre-scan (NQ/PnnQuantizer.java:431-436)
void apply_scan(nq_handle* h, int nMaxColors, int64_t transparent_index, uint32_t transparent_color, int64_t semi_count) {
    nq_params& p = h->params;
    p.kind = h->kind; p.nMaxColors = nMaxColors;
    p.transparentPixelIndex = transparent_index >= 0 ? (int32_t) transparent_index : -1;
    p.transparentColor = (int32_t) 0x00FFFFFFu;                       // Color.argb(0,255,255,255) (:22)
    if (transparent_index >= 0 && nMaxColors > 2) p.transparentColor = (int32_t) transparent_color;   // :421-424
    p.hasSemiTransparency = semi_count > 0;
    if (nMaxColors <= 32) p.PR = p.PG = p.PB = p.PA = 1;
    else { p.PR = kCoeffs[0][0]; p.PG = kCoeffs[0][1]; p.PB = kCoeffs[0][2]; p.PA = .3333; }
    p.ratio = .5; p.weight = 1; p.isNano = 0; p.texicab = 0; p.quan_rt = 1; p.maxbins = 0; p.paletteLength = 0;
    p.distinctColors = 0;
}

// stage boundary i on the handle's stream.  An event record costs the GPU ~3 us of queue time, eight of them per image of a big batch
// 0.9 % of the batch: the batch entry points keep all eight for their first images only, the rest record the start and the two ends of
// the per-pixel pass (what bench.py's roofline needs)
void rec(nq_handle* h, int i) {
    if (h->light_events && ((i >= 1 && i <= 4) || i == 7)) return;
    (void) hipEventRecord(h->ev[i], h->stream);
}

// what palette_prepare leaves for the merge launch and palette_finish; merge == false: the palette is already final
struct PaletteJob {
    bool merge = false;
    nq::MergeJob mj;
    int plen = 0;
};

// pnnquan after the histogram(s) exist on the device, up to the initial find_nn pass (P4..P8)
int palette_prepare(nq_handle* h, const double* d_hists, int n_bands, int nMaxColors, uint32_t* out_palette, int32_t* out_K,
                    const uint32_t* d_argb, int64_t n_pixels, PaletteJob* job) {
    job->merge = false;
    nq_params& p = h->params;
    const int kind = h->kind;
    nq::Bins B = bins_of(h);
    int* d_maxbins = h->d_ints.p;
    launch_compact(kind, d_hists, n_bands, B, d_maxbins, h->d_ints.p + 8, h->stream);
    int maxbins = 0;
    NQ_HIP(h, hipMemcpyAsync(&maxbins, d_maxbins, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    if (maxbins <= 0) NQ_FAIL(h, NQ_ERR_INVALID, "empty image");
    p.maxbins = maxbins;
    short quan_rt = 1;
    int fn = 0;
    bool texicab = false;
    double proportional = 0;
    if (kind == NQ_KIND_RGB) {
        // NQ/PnnQuantizer.java:172-182
        if (nMaxColors < 16) quan_rt = -1;
        p.weight = std::fmin(0.9, nMaxColors * 1.0 / maxbins);
        if (p.weight < .04 && p.PG >= kCoeffs[0][1]) {
            p.PR = p.PG = p.PB = p.PA = 1;
            if (nMaxColors >= 64) quan_rt = 0;
        }
        if (quan_rt > 0) fn = nMaxColors < 64 ? 1 : 2;
        else if (quan_rt < 0) fn = 3;
    } else {
        // NQ/PnnLABQuantizer.java:175-241
        proportional = sqr(nMaxColors) / maxbins;
        if ((p.transparentPixelIndex >= 0 || p.hasSemiTransparency) && nMaxColors < 32) quan_rt = -1;
        p.weight = std::fmin(0.9, nMaxColors * 1.0 / maxbins);
        p.isNano = p.weight <= .015;
        const double weight = p.weight;
        if ((nMaxColors < 16 && weight < .0075) || weight < .001 || (weight > .0015 && weight < .0022)) quan_rt = 2;
        if (weight < .04 && p.PG < 1 && p.PG >= kCoeffs[0][1]) {
            if (nMaxColors >= 64) quan_rt = 0;
        }
        if (nMaxColors > 16 && nMaxColors < 64) {
            double weightB = nMaxColors / 8000.0;
            if (std::fabs(weightB - weight) < .001) quan_rt = 2;
        }
        if (maxbins <= nMaxColors) {
            // pixelMap.size() <= nMaxColors is only possible here (every occupied bin holds >= 1 distinct colour)
            int64_t cnt = 0;
            std::vector<int32_t> inserted;
            if (d_argb) {
                int rcd = distinct_colors(h, d_argb, n_pixels, nMaxColors, &cnt, &inserted);
                if (rcd) return rcd;
            } else if (h->ext_distinct_valid) {                 // split pipeline: the caller merged the bands' lists
                inserted = h->ext_distinct;
                cnt = h->ext_distinct_many ? (int64_t) nMaxColors + 1 : (int64_t) inserted.size();
            } else
                NQ_FAIL(h, NQ_ERR_UNSUPPORTED, "<= nMaxColors occupied bins in the multi-band path: exchange the bands' distinct colours "
                        "(nq_band_distinct_device / nq_set_distinct) before nq_palette_from_histograms_device "
                        "(NQ/PnnLABQuantizer.java:193-206)");
            p.distinctColors = (!d_argb && h->ext_distinct_many) ? 0 : cnt;     // (0 = not known: "more than nMaxColors")
            if (cnt <= nMaxColors) {
                // NQ/PnnLABQuantizer.java:193-206: palette = pixelMap.keySet() in HashMap order, a transparent colour swapped to slot 0
                std::vector<int32_t> keys = java_hashmap_keyset(inserted);
                int k = 0;
                for (int32_t pixel : keys) {
                    out_palette[k++] = (uint32_t) pixel;
                    if (k > 1 && (((uint32_t) pixel) >> 24) == 0) { out_palette[k - 1] = out_palette[0]; out_palette[0] = (uint32_t) pixel; }
                }
                p.quan_rt = quan_rt; p.texicab = 0; p.paletteLength = k;
                *out_K = k;
                rec(h, 2); rec(h, 3); rec(h, 4); rec(h, 5);
                return NQ_OK;
            }
        }
        if (quan_rt > 0) fn = quan_rt > 1 ? 4 : (nMaxColors < 64 ? 2 : 1);
        texicab = proportional > .0225 && !p.hasSemiTransparency;
        if (p.hasSemiTransparency) p.ratio = .5;
        else if (quan_rt != 0 && nMaxColors < 64) {
            if (proportional > .018 && proportional < .022) p.ratio = std::fmin(1.0, proportional + weight * std::exp(3.13));
            else if (proportional > .1) p.ratio = std::fmin(1.0, 1.0 - weight);
            else if (proportional > .04) p.ratio = std::fmin(1.0, weight * std::exp(1.56));
            else if (proportional > .025 && (weight < .002 || weight > .0022)) p.ratio = std::fmin(1.0, proportional + weight * std::exp(3.66));
            else p.ratio = std::fmin(1.0, proportional + weight * std::exp(1.718));
        }
        else if (nMaxColors > 256) p.ratio = std::fmin(1.0, 1 - 1.0 / proportional);
        else p.ratio = std::fmin(1.0, 1 - weight * .7);
        if (!p.hasSemiTransparency && quan_rt < 0) p.ratio = std::fmin(1.0, weight * std::exp(3.13));
    }
    p.quan_rt = quan_rt; p.texicab = texicab;
    launch_quanfn(B.cnt, maxbins, fn, h->stream);

    nq::NNParams np;
    np.kind = kind; np.hasSemi = p.hasSemiTransparency; np.texicab = texicab;
    np.ratio = p.ratio; np.PR = p.PR; np.PG = p.PG; np.PB = p.PB; np.PA = p.PA;
    np.pgLessThanCoeff = p.PG < kCoeffs[0][1];
    np.rgbTheta = 2.0;
    if (const char* t = std::getenv("NQ_RGB_THETA")) {           // tests: 1.0 makes the checked assumption fail often (fallback path)
        const double v = std::atof(t);
        if (v >= 1.0 && v <= 1e6) np.rgbTheta = v;
    }
    rec(h, 2);
    launch_find_nn_init(np, B, maxbins, h->scan_box.p, h->sc->init_cand.p, h->stream);
    rec(h, 3);
    if (kind == NQ_KIND_LAB) {
        // NQ/PnnLABQuantizer.java:259-264: ratio retuned AFTER the initial pass
        const double weight = p.weight;
        if (quan_rt > 0 && nMaxColors < 64 && proportional > .035 && proportional < .1) {
            const int dir = proportional > .04 ? 1 : -1;
            const double margin = dir > 0 ? .002 : .0025;
            const double delta = weight > margin && weight < .003 ? 1.872 : 1.632;
            p.ratio = std::fmin(1.0, proportional + dir * weight * std::exp(delta));
            np.ratio = p.ratio;
        }
    }
    const int extbins = maxbins - nMaxColors;
    job->merge = true;
    job->mj.np = np; job->mj.B = B; job->mj.maxbins = maxbins; job->mj.extbins = extbins;
    job->mj.heap = h->heap.p; job->mj.live3 = h->live3.p; job->mj.scan_f = h->scan_f.p; job->mj.scan_i = h->scan_i.p; job->mj.scan_box = h->scan_box.p;
    job->mj.stats = h->d_scalars.p + 4;       // [4..19] the 16 counters of nq_get_merge_stats, [24..31] the team counters
    job->mj.team = h->team.p; job->mj.helpers = 0;
    job->plen = extbins > 0 ? nMaxColors : maxbins;
    // the merge workgroup also fills the palette (P10)
    NQ_HIP(h, h->d_palette.reserve((size_t) std::max(job->plen, 2)));
    h->dev_palette.clear();                   // (the merge workgroup is about to write it)
    job->mj.plen = job->plen; job->mj.palette = h->d_palette.p; job->mj.status = reinterpret_cast<int*>(h->d_scalars.p + 40);
    return NQ_OK;
}

// the merge loops (P9) of n prepared images in one launch per kind, on the stream of `owner`
int merge_launch(nq_handle* owner, const PaletteJob* const* jobs, int n) {
    std::vector<nq::MergeJob> host;
    int n_lab = 0;
    for (int pass = 1; pass >= 0; --pass) {                 // LAB jobs first, then RGB
        for (int i = 0; i < n; ++i)
            if (jobs[i]->merge && jobs[i]->mj.np.kind == pass) host.push_back(jobs[i]->mj);
        if (pass == 1) n_lab = (int) host.size();
    }
    if (host.empty()) return NQ_OK;
    // merge teams: when the loops of one kind in this call leave CUs free (one 512-thread workgroup per CU), every loop gets helper
    // workgroups that evaluate find_nn speculatively (nq_merge.inc); their hand-off words start zeroed.  The two kinds run one after
    // the other on this stream, so each is sized on its own.
    const int n_rgb = (int) host.size() - n_lab;
    const int helpers_lab = nq::merge_team_helpers(n_lab, (int) host.size(), owner->n_cus),
              helpers_rgb = nq::merge_team_helpers(n_rgb, (int) host.size(), owner->n_cus);
    // watchdog (stats[14] = 2 -> NQ_ERR_TIME_LIMIT): seconds of RESIDENCY a loop may take.  Automatic: 60 s + 1 ms per bin and per loop that
    // shares a CU with it (the slowest legitimate loop measured -- 65 536 bins, four loops per CU -- takes < 10 s; a time-sliced or
    // shared device is slower, hence the margin and NQ_OPT_MERGE_WALL_SECONDS)
    const int per_cu = std::max(1, ((int) host.size() + owner->n_cus - 1) / owner->n_cus);
    for (int i = 0; i < (int) host.size(); ++i) {
        const long long secs = owner->merge_wall_s > 0 ? owner->merge_wall_s : 60LL + ((long long) host[i].maxbins * per_cu) / 1000;
        host[i].wall_ticks = secs * 100000000LL;
        host[i].helpers = i < n_lab ? helpers_lab : helpers_rgb;
        if (host[i].helpers > 0) NQ_HIP(owner, hipMemsetAsync(host[i].team, 0, 256 * sizeof(unsigned long long), owner->stream));
    }
    NQ_HIP(owner, owner->d_jobs.reserve(host.size()));
    NQ_HIP(owner, hipMemcpyAsync(owner->d_jobs.p, host.data(), host.size() * sizeof(nq::MergeJob), hipMemcpyHostToDevice, owner->stream));
    NQ_HIP(owner, hipStreamSynchronize(owner->stream));    // `host` goes out of scope
    NQ_HIP(owner, launch_merge(1, owner->d_jobs.p, n_lab, (int) host.size(), owner->n_cus, helpers_lab, owner->stream));
    NQ_HIP(owner, launch_merge(0, owner->d_jobs.p + n_lab, n_rgb, (int) host.size(), owner->n_cus, helpers_rgb, owner->stream));
    NQ_HIP(owner, launch_status());
    return NQ_OK;
}

// read-back of the palette the merge workgroup wrote (P10): the copies are enqueued by palette_fetch and looked at by palette_check
// once the stream has been waited for (a batch waits ONCE for all its images, not once per image)
int palette_fetch(nq_handle* h, const PaletteJob& job, uint32_t* out_palette, int* status) {
    rec(h, 5);
    NQ_HIP(h, hipMemcpyAsync(out_palette, h->d_palette.p, job.plen * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    h->fetched_palette = out_palette; h->fetched_len = job.plen;
    NQ_HIP(h, hipMemcpyAsync(h->merge_readback, h->d_scalars.p + 4, sizeof h->merge_readback, hipMemcpyDeviceToHost, h->stream));
    (void) status;
    return NQ_OK;
}
int palette_check(nq_handle* h, const PaletteJob& job, int status, int32_t* out_K) {
    std::memcpy(h->merge_stats, h->merge_readback, sizeof h->merge_stats);
    std::memcpy(h->team_stats, h->merge_readback + 20, sizeof h->team_stats);
    status = (int) (h->merge_readback[36] & 0xFFFFFFFFLL);
    if (h->fetched_palette) { h->dev_palette.assign(h->fetched_palette, h->fetched_palette + h->fetched_len); h->palette_synced = true; }   // (read back and waited for)
    h->fetched_palette = nullptr;
    if (h->merge_stats[14] == 2)
        NQ_FAIL(h, NQ_ERR_TIME_LIMIT, "merge loop stopped at its time limit (a slow, shared or time-sliced device: the state was sound; call again, or "
                                      "raise NQ_OPT_MERGE_WALL_SECONDS)");
    if (h->merge_stats[14]) NQ_FAIL(h, NQ_ERR_UNSUPPORTED, "merge loop stopped by its watchdog (more than maxbins^2/2 find_nn calls, or an empty heap)");
    if (status) NQ_FAIL(h, NQ_ERR_REFERENCE_THROWS, "ColorUtils.setAlphaComponent: alpha outside 0..255 (the reference throws IllegalArgumentException)");
    h->params.paletteLength = job.plen;
    *out_K = job.plen;
    return NQ_OK;
}
int palette_finish(nq_handle* h, const PaletteJob& job, uint32_t* out_palette, int32_t* out_K) {
    int status = 0;
    int rc = palette_fetch(h, job, out_palette, &status);
    if (rc) return rc;
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    NQ_HIP(h, launch_status());
    return palette_check(h, job, status, out_K);
}

int palette_from_hist(nq_handle* h, const double* d_hists, int n_bands, int nMaxColors, uint32_t* out_palette, int32_t* out_K,
                      const uint32_t* d_argb = nullptr, int64_t n_pixels = 0) {
    PaletteJob job;
    int rc = palette_prepare(h, d_hists, n_bands, nMaxColors, out_palette, out_K, d_argb, n_pixels, &job);
    if (rc || !job.merge) return rc;
    const PaletteJob* jp = &job;
    rc = merge_launch(h, &jp, 1);
    if (rc) return rc;
    rec(h, 4);
    return palette_finish(h, job, out_palette, out_K);
}

// alpha pre-scan + histogram + palette_prepare
int pnnquan_prepare(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors, uint32_t* out_palette, int32_t* out_K,
                    PaletteJob* job) {
    job->merge = false;
    if (!d_argb || width <= 0 || height <= 0 || !out_palette || !out_K) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (nMaxColors < 1 || nMaxColors > 32767) NQ_FAIL(h, NQ_ERR_INVALID, "nMaxColors out of range");
    const int64_t n = (int64_t) width * height;
    if (n > 2147483647LL) NQ_FAIL(h, NQ_ERR_INVALID, "image larger than a Java int[]");
    long long* d_scan3 = h->d_scalars.p + 1;
    rec(h, 0);
    // the common case (nMaxColors >= 64: 5-6-5 keys unless the scan finds transparency) gets the sort words with the scan's read
    bool words = false;
    if (nMaxColors >= 64) {
        int rcw = reserve_palette_ws(h, n);
        if (rcw) return rcw;
        words = launch_front((const int*) d_argb, n, d_scan3, h->sc->vals_a.p, (int) 0x00FFFFFFu, h->stream);
    }
    if (!words) launch_prescan((const int*) d_argb, n, 0, d_scan3, h->stream);
    long long scan3[3];
    NQ_HIP(h, hipMemcpyAsync(scan3, d_scan3, sizeof scan3, hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    apply_scan(h, nMaxColors, scan3[0], (uint32_t) scan3[1], scan3[2]);
    rec(h, 1);
    nq_params& p = h->params;
    if (nMaxColors <= 2) {
        // NQ/PnnQuantizer.java:441-452
        p.weight = 1;
        if (p.transparentPixelIndex >= 0) { out_palette[0] = (uint32_t) p.transparentColor; out_palette[1] = 0xFF000000u; }
        else { out_palette[0] = 0xFF000000u; out_palette[1] = 0xFFFFFFFFu; }
        p.paletteLength = nMaxColors; *out_K = nMaxColors;
        rec(h, 2); rec(h, 3); rec(h, 4); rec(h, 5);
        return NQ_OK;
    }
    int rc = reserve_palette_ws(h, n);
    if (rc) return rc;
    nq::HistParams hp;
    hp.hasSemi = p.hasSemiTransparency; hp.hasTransp = nMaxColors < 64 || p.transparentPixelIndex >= 0;
    hp.transparentColor = p.transparentColor; hp.rewriteTransparent = 0;
    nq::SortWorkspace ws;
    ws.keys_a = h->sc->keys_a.p; ws.keys_b = h->sc->keys_b.p; ws.vals_a = h->sc->vals_a.p; ws.vals_b = h->sc->vals_b.p;
    ws.tmp = h->sc->sort_tmp.p; ws.tmp_bytes = h->sc->sort_tmp.n; ws.seg_start = h->sc->seg.p; ws.seg_end = h->sc->seg.p + 65536;
    // (the speculative words hold 5-6-5 keys and the default transparent colour: right exactly for an image without transparency)
    const bool words_ready = words && !hp.hasSemi && !hp.hasTransp;
    launch_histogram(h->kind, (const int*) d_argb, n, hp, ws, h->sc->hist.p, h->stream, words_ready);
    return palette_prepare(h, h->sc->hist.p, 1, nMaxColors, out_palette, out_K, d_argb, n, job);
}

int pnnquan_device(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors, uint32_t* out_palette, int32_t* out_K) {
    PaletteJob job;
    int rc = pnnquan_prepare(h, d_argb, width, height, nMaxColors, out_palette, out_K, &job);
    if (rc || !job.merge) return rc;
    const PaletteJob* jp = &job;
    rc = merge_launch(h, &jp, 1);
    if (rc) return rc;
    rec(h, 4);
    return palette_finish(h, job, out_palette, out_K);
}

int dither_device(nq_handle* h, const uint32_t* d_argb, int width, int height, const uint32_t* palette, int K, int dither,
                  int64_t seed, int mode, uint32_t* d_out_argb, uint16_t* d_out_index) {
    if (!d_argb || width <= 0 || height <= 0 || !palette || K < 1 || !d_out_argb) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (width > 65535 || height > 65535) NQ_FAIL(h, NQ_ERR_INVALID, "image side > 65535");
    if (K > 8192) NQ_FAIL(h, NQ_ERR_UNSUPPORTED, "palettes above 8192 entries do not fit the LDS staging");
    const int64_t n = (int64_t) width * height;
    nq_params& p = h->params;
    { int rcp = upload_palette(h, palette, K); if (rcp) return rcp; }
    if (!d_out_index) { NQ_HIP(h, h->d_out_index.reserve((size_t) n)); d_out_index = h->d_out_index.p; }
    // stage events 5..7 belong to THIS call only once it has recorded all of them (set at the successful exits below); until then
    // nq_get_stage_ms reports what the last finished convert left
    h->dither_events_fresh = false;

    if (mode == NQ_MODE_LOOKUP_ONLY) {
        DevParams P = dev_params(h, K);
        nq::ListsView lv;
        int rcl = prepare_lists(h, P, &lv);
        if (rcl) return rcl;
        const bool fast_lookup = h->use_fast_dither && fast_lookup_eligible(P, lv);
        if (fast_lookup) NQ_HIP(h, h->sc->lookup_todo.reserve((size_t) n + 1));
        rec(h, 5);       // stage "dither" = the lookup kernels alone (the candidate lists are built in front of them), "bluenoise" = 0
        if (fast_lookup)
            launch_fast_lookup_only(P, lv, h->d_palette.p, packed_lists(h), (const int*) d_argb, n, d_out_index, (int*) d_out_argb,
                                    h->sc->lookup_todo.p, h->stream);
        else
            launch_lookup_only(P, h->d_palette.p, lv, (const int*) d_argb, n, d_out_index, (int*) d_out_argb, h->stream);
        rec(h, 6); rec(h, 7);
        NQ_HIP(h, launch_status());
        h->dither_events_fresh = true;
        return NQ_OK;
    }
    if (mode != NQ_MODE_PARALLEL_TILED && mode != NQ_MODE_REFERENCE_SEQUENTIAL) NQ_FAIL(h, NQ_ERR_INVALID, "unknown mode %d", mode);
    const bool sequential = mode == NQ_MODE_REFERENCE_SEQUENTIAL;

    // dither(): RGB NQ/PnnQuantizer.java:393-407, LAB NQ/PnnLABQuantizer.java:493-522
    // The reference negates the field once per convert() (dither() runs once per object); here dither may be called repeatedly on
    // one handle, so the negation holds for this call only and the handle keeps the value pnnquan left.
    struct WeightGuard { double& w; double saved; ~WeightGuard() { w = saved; } } weight_guard{p.weight, p.weight};
    const nq_handle::StageOverride ov = h->ov;       // (the static entry points GilbertCurve.dither / BlueNoise.dither: nq_gilbert_dither ...)
    const bool staged = ov.gilbert_only || ov.blue_only;
    if (p.hasSemiTransparency && !staged) p.weight = -std::fabs(p.weight);
    bool hasSal = false, salSubst = false;
    if (staged) hasSal = ov.hasSal;
    else if (h->kind == NQ_KIND_LAB) {
        if (p.nMaxColors > 2 && p.nMaxColors < 128) { hasSal = true; salSubst = true; }       // pnnquan :135,:155-156
        else if (dither && (K <= 256 || p.weight > .99)) { hasSal = true; }                   // :499-508
    }
    const bool post = ov.blue_only || (!ov.gilbert_only && !dither && K > 32);
    float blueWeight = staged ? ov.blueWeight : 1.0f;
    const bool seq_lab_post = !staged && post && h->kind == NQ_KIND_LAB && sequential;   // needs pixelMap.size() AFTER the gilbert pass
    if (!staged && post && h->kind == NQ_KIND_LAB && !sequential) {
        if (p.distinctColors <= 0) {
            int64_t cnt = 0;
            int rcd = distinct_colors(h, d_argb, n, 0, &cnt, nullptr);
            if (rcd) return rcd;
            p.distinctColors = cnt;
        }
        const double delta = sqr(K) / (double) p.distinctColors;
        blueWeight = delta > 0.023 ? 1.0f : (float) (37.013 * delta + 0.906);
    }
    DevParams P = dev_params(h, K);
    GilbertConsts G = gilbert_consts(K, staged ? ov.weight : p.weight, hasSal, dither != 0);
    G.salSubst = salSubst;

    TileGeom T;
    std::memset(&T, 0, sizeof T);
    T.width = width; T.height = height;
    const bool banded = h->band_image_h > 0;
    if (banded && (sequential || h->band_y0 + height > h->band_image_h))
        NQ_FAIL(h, NQ_ERR_INVALID, "nq_set_band: the band [%d, %d) does not fit the image height %d (or REFERENCE_SEQUENTIAL mode)", h->band_y0, h->band_y0 + height, h->band_image_h);
    const int rule_h = banded ? h->band_image_h : height;          // the automatic tile follows the WHOLE image
    if (sequential) { T.tile_w = width; T.tile_h = height; }
    else if (h->tile_w > 0 && h->tile_h > 0) { T.tile_w = std::min(h->tile_w, width); T.tile_h = std::min(h->tile_h, height); }
    else {
        // automatic: 8x8 when that yields >= 2 wavefronts of chains per SIMD (131072 chains), else 4x4.  The tile size does not change
        // the measured dither quality, it sets how many chains run in parallel and how much LDS a chain's staged indices
        // take: 16x16 (the round-1 choice for images beyond 5793^2) ran the 16384^2 pass in 22.7 ms, 8x8 in 12.5 ms
        int tsz = 4;
        {
            const int cand = 8;
            const int64_t tiles = (int64_t) ((width + cand - 1) / cand) * ((rule_h + cand - 1) / cand);
            if (tiles >= 131072) tsz = cand;
        }
        // (the sorted-by-yDiff queue takes the same rule since round 4: its tile chains start in the queue's steady state, nq_dither.inc)
        T.tile_w = std::min(tsz, width); T.tile_h = std::min(tsz, rule_h);
    }
    if (banded) {
        T.tile_h = std::min(T.tile_h, h->band_image_h);
        if (h->band_y0 % T.tile_h != 0 || (h->band_y0 + height != h->band_image_h && height % T.tile_h != 0))
            NQ_FAIL(h, NQ_ERR_INVALID, "nq_set_band: band origin %d / rows %d must be multiples of the tile height %d", h->band_y0, height, T.tile_h);
        T.y_origin = h->band_y0;
        T.tile_base = (h->band_y0 / T.tile_h) * ((width + T.tile_w - 1) / T.tile_w);
        T.tile_h = std::min(T.tile_h, height);
    }
    T.tiles_x = (width + T.tile_w - 1) / T.tile_w; T.tiles_y = (height + T.tile_h - 1) / T.tile_h;
    const int rw = width - (T.tiles_x - 1) * T.tile_w, rh = height - (T.tiles_y - 1) * T.tile_h;
    const int sw[4] = {T.tile_w, rw, T.tile_w, rw}, shh[4] = {T.tile_h, T.tile_h, rh, rh};
    for (int s = 0; s < 4; ++s) {
        int rc = get_path(h, sw[s], shh[s], &T.path[s]);
        if (rc) return rc;
        T.path_len[s] = sw[s] * shh[s]; T.shape_w[s] = sw[s]; T.shape_h[s] = shh[s];
    }
    if (sequential && !ov.blue_only) NQ_HIP(h, hipMemsetAsync(h->d_bincache.p, 0xFF, 65536 * sizeof(short), h->stream));   // (BlueNoise.dither alone continues the caches)
    nq::ListsView lv;
    bool sal_done = false;
    const bool want_sal = !staged && hasSal;            // (the map of THIS call's pixels; the builders and the map share one launch where they can)
    { int rcl = prepare_lists(h, P, &lv, want_sal ? (const int*) d_argb : nullptr, n, salSubst ? 1 : 0, &sal_done); if (rcl) return rcl; }
    const float* d_sal = nullptr;
    if (staged) d_sal = hasSal ? ov.d_sal : nullptr;
    else if (hasSal) {
        NQ_HIP(h, h->sc->saliency.reserve((size_t) n));
        if (!sal_done) launch_saliency(P, salSubst ? 1 : 0, (const int*) d_argb, n, h->sc->saliency.p, h->stream);
        d_sal = h->sc->saliency.p;
    }
    rec(h, 5);       // stage "palette_fill" ends here: it includes the candidate-list build and the saliency map
    int log_cap = 0;
    if (seq_lab_post) {
        // every colour handed to nearestColorIndex on a cache miss (<= 3 lookups per pixel) + a flag per palette entry
        if (3 * n + 16 > 2147483647LL) NQ_FAIL(h, NQ_ERR_INVALID, "image too large for the sequential pixelMap log");
        log_cap = (int) (3 * n + 16);
        NQ_HIP(h, h->d_seqlog.reserve((size_t) log_cap + 4));
        NQ_HIP(h, h->d_seqseen.reserve((size_t) K));
        NQ_HIP(h, hipMemsetAsync(h->d_seqlog.p + log_cap, 0, sizeof(int), h->stream));
        NQ_HIP(h, hipMemsetAsync(h->d_seqseen.p, 0, (size_t) K, h->stream));
    }
    const int* d_tile_list = nullptr;
    h->last_dither_fast = 0;
    if (ov.blue_only) { /* BlueNoise.dither alone: the indices are already in d_out_index */ }
    else {
    if (!sequential && h->use_fast_dither && gilbert_fast_eligible(P, G, T, lv)) {
        // production path (nq_dither_fast.hip); the tiles it cannot finish come back as a list for the generic kernel below
        NQ_HIP(h, h->d_failed.reserve((size_t) T.tiles_x * T.tiles_y + 1));
        NQ_HIP(h, launch_gilbert_fast(P, G, T, lv, (const int*) d_argb, d_sal, h->d_palette.p, (long long) seed, d_out_index,
                                      post ? nullptr : (int*) d_out_argb, h->d_failed.p, packed_lists(h), h->stream));
        d_tile_list = h->d_failed.p;
        h->last_dither_fast = 1;
    }
    launch_gilbert(P, G, T, lv, (const int*) d_argb, d_sal, h->d_palette.p, h->d_bincache.p, (long long) seed, sequential ? 1 : 0,
                   h->d_scalars.p, d_out_index, post ? nullptr : (int*) d_out_argb,
                   seq_lab_post ? h->d_seqlog.p : nullptr, seq_lab_post ? h->d_seqlog.p + log_cap : nullptr,
                   seq_lab_post ? h->d_seqseen.p : nullptr, log_cap, d_tile_list, h->stream);
    }
    rec(h, 6);
    if (seq_lab_post) {
        // pixelMap.size() at NQ/PnnLABQuantizer.java:512 = |{image colours as the histogram saw them} U {colours looked up on a
        // nearest-cache miss} U {palette entries those lookups touched}| (getLab memoises all three).  Debugging mode: on the host.
        int count = 0;
        NQ_HIP(h, hipMemcpyAsync(&count, h->d_seqlog.p + log_cap, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        NQ_HIP(h, hipStreamSynchronize(h->stream));
        if (count > log_cap) NQ_FAIL(h, NQ_ERR_HIP, "sequential pixelMap log overflow (internal error)");
        std::vector<int32_t> all((size_t) n + count);
        std::vector<unsigned char> seen(K);
        NQ_HIP(h, hipMemcpy(all.data(), d_argb, (size_t) n * sizeof(int), hipMemcpyDeviceToHost));
        if (count) NQ_HIP(h, hipMemcpy(all.data() + n, h->d_seqlog.p, (size_t) count * sizeof(int), hipMemcpyDeviceToHost));
        NQ_HIP(h, hipMemcpy(seen.data(), h->d_seqseen.p, (size_t) K, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i)
            if ((((uint32_t) all[i]) >> 24) <= 0xF) all[i] = p.transparentColor;      // the histogram's substitution (:141-142)
        for (int i = 0; i < K; ++i) if (seen[i]) all.push_back((int32_t) palette[i]);
        std::sort(all.begin(), all.end());
        const int64_t distinct = (int64_t) (std::unique(all.begin(), all.end()) - all.begin());
        const double delta = sqr(K) / (double) distinct;
        blueWeight = delta > 0.023 ? 1.0f : (float) (37.013 * delta + 0.906);
    }
    if (post && !sequential && h->use_fast_dither && fast_lookup_eligible(P, lv))
        launch_fast_bluenoise(P, lv, h->d_palette.p, packed_lists(h), (const int*) d_argb, width, height, T.y_origin, blueWeight, (long long) seed,
                              d_out_index, (int*) d_out_argb, h->stream);
    else if (post)
        launch_bluenoise(P, h->d_palette.p, lv, (const int*) d_argb, width, height, T.y_origin, blueWeight, (long long) seed, sequential ? 1 : 0,
                         h->d_bincache.p, h->d_scalars.p, d_out_index, (int*) d_out_argb, h->stream);
    rec(h, 7);
    NQ_HIP(h, launch_status());
    h->dither_events_fresh = true;
    return NQ_OK;
}

void finish_timing(nq_handle* h) {
    // {prescan, histogram, nn_init, merge, palette_fill, dither, bluenoise, total}
    for (int i = 0; i < 7; ++i) {
        float ms = 0;
        if (h->light_events && i != 5) { h->stage_ms[i] = -1.0f; continue; }      // (not recorded: see rec)
        if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) != hipSuccess) ms = 0;
        h->stage_ms[i] = ms;
    }
    float tot = 0;
    if (hipEventElapsedTime(&tot, h->ev[0], h->ev[h->light_events ? 6 : 7]) != hipSuccess) tot = 0;
    h->stage_ms[7] = tot;
    h->dither_events_fresh = false;
}

void finish_batch_timing(nq_handle* h0) {
    for (int i = 0; i < 3; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h0->bev[i], h0->bev[i + 1]) != hipSuccess) ms = 0;
        h0->batch_phase_ms[i] = ms;
    }
    float tot = 0;
    if (hipEventElapsedTime(&tot, h0->bev[0], h0->bev[3]) != hipSuccess) tot = 0;
    h0->batch_phase_ms[3] = tot;
}

} // namespace

extern "C" {

int nq_abi_version(void) { return NQ_ABI_VERSION; }

int nq_create(int kind, int device, nq_handle** out) {
    if (!out) { g_create_error = "out is NULL"; return NQ_ERR_INVALID; }
    *out = nullptr;
    if (kind != NQ_KIND_RGB && kind != NQ_KIND_LAB) { g_create_error = "unknown kind"; return NQ_ERR_INVALID; }
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = "no HIP device available (libnquant_hip has no CPU fallback)";
        return NQ_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) { g_create_error = "device ordinal out of range"; return NQ_ERR_INVALID; }
    nq_handle* h = new nq_handle();
    h->kind = kind; h->device = device;
    std::memset(&h->params, 0, sizeof h->params);
    h->params.kind = kind; h->params.transparentPixelIndex = -1; h->params.transparentColor = (int32_t) 0x00FFFFFFu;
    h->params.PR = 0.299; h->params.PG = 0.587; h->params.PB = 0.114; h->params.PA = .3333; h->params.ratio = .5; h->params.weight = 1;
    int rc = use_device(h);
    if (rc) { g_create_error = h->err; delete h; return rc; }
    *out = h;
    return NQ_OK;
}

void nq_destroy(nq_handle* h) {
    if (!h) return;
    (void) hipSetDevice(h->device);
    (void) hipStreamSynchronize(h->stream);
    delete h;
}

const char* nq_last_error(const nq_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nq_set_stream(nq_handle* h, void* hip_stream) {
    if (!h) return NQ_ERR_INVALID;
    h->stream = (hipStream_t) hip_stream;
    return NQ_OK;
}
int nq_set_tile(nq_handle* h, int tile_w, int tile_h) {
    if (!h) return NQ_ERR_INVALID;
    if (tile_w <= 0 || tile_h <= 0) { tile_w = 0; tile_h = 0; }
    h->tile_w = tile_w; h->tile_h = tile_h;
    return NQ_OK;
}
int nq_set_band(nq_handle* h, int y0, int image_height) {
    if (!h) return NQ_ERR_INVALID;
    if (y0 < 0 || image_height < 0 || (image_height == 0 && y0 != 0) || (image_height > 0 && y0 >= image_height))
        NQ_FAIL(h, NQ_ERR_INVALID, "nq_set_band: bad origin %d / image height %d", y0, image_height);
    h->band_y0 = y0; h->band_image_h = image_height;
    return NQ_OK;
}
int nq_get_list_counts(nq_handle* h, uint8_t* closest_counts, uint8_t* nearest_counts) {
    if (!h || !closest_counts || !nearest_counts) return NQ_ERR_INVALID;
    if (!h->sc->cell_lists.p) NQ_FAIL(h, NQ_ERR_INVALID, "no lists built yet");
    const size_t LB = (size_t) 65536 * 32;
    NQ_HIP(h, hipMemcpy(closest_counts, h->sc->cell_lists.p + 2 * LB, 65536, hipMemcpyDeviceToHost));
    NQ_HIP(h, hipMemcpy(nearest_counts, h->sc->cell_lists.p + 2 * LB + 65536, 65536, hipMemcpyDeviceToHost));
    return NQ_OK;
}
int nq_set_option(nq_handle* h, int option, int value) {
    if (!h) return NQ_ERR_INVALID;
    if (option == NQ_OPT_CELL_LISTS) { h->use_lists = value != 0; return NQ_OK; }
    if (option == NQ_OPT_FAST_DITHER) { h->use_fast_dither = value != 0; return NQ_OK; }
    if (option == NQ_OPT_MERGE_WALL_SECONDS) { h->merge_wall_s = value > 0 ? value : 0; return NQ_OK; }
    NQ_FAIL(h, NQ_ERR_INVALID, "unknown option %d", option);
}
int nq_selftest_ciede(nq_handle* h, const float* lab_pairs, int64_t n, uint32_t* out9) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!lab_pairs || n <= 0 || !out9) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    float* d_in = nullptr; unsigned* d_out = nullptr;
    NQ_HIP(h, hipMalloc((void**) &d_in, (size_t) n * 6 * sizeof(float)));
    hipError_t e = hipMalloc((void**) &d_out, (size_t) n * 9 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, lab_pairs, (size_t) n * 6 * sizeof(float), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) { launch_ciede_selftest(d_in, n, d_out, h->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(out9, d_out, (size_t) n * 9 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void) hipFree(d_in); if (d_out) (void) hipFree(d_out);
    if (e != hipSuccess) NQ_FAIL(h, NQ_ERR_HIP, "nq_selftest_ciede: %s", hipGetErrorString(e));
    return NQ_OK;
}
int nq_get_dither_path(nq_handle* h, int32_t* out_fast, int32_t* out_failed_tiles) {
    if (!h) return NQ_ERR_INVALID;
    if (out_fast) *out_fast = h->last_dither_fast;
    if (out_failed_tiles) {
        *out_failed_tiles = 0;
        if (h->last_dither_fast && h->d_failed.p) {
            NQ_HIP(h, hipStreamSynchronize(h->stream));
            int cnt = 0;
            NQ_HIP(h, hipMemcpy(&cnt, h->d_failed.p, sizeof(int), hipMemcpyDeviceToHost));
            *out_failed_tiles = cnt;
        }
    }
    return NQ_OK;
}
int nq_get_params(const nq_handle* h, nq_params* out) {
    if (!h || !out) return NQ_ERR_INVALID;
    *out = h->params;
    return NQ_OK;
}
int nq_set_params(nq_handle* h, const nq_params* in) {
    if (!h || !in) return NQ_ERR_INVALID;
    h->params = *in; h->params.kind = h->kind;
    return NQ_OK;
}
int nq_get_merge_stats(const nq_handle* h, int64_t* out8) {
    if (!h || !out8) return NQ_ERR_INVALID;
    std::memcpy(out8, h->merge_stats, 16 * sizeof(long long));
    return NQ_OK;
}
int nq_get_team_stats(const nq_handle* h, int64_t* out16) {
    if (!h || !out16) return NQ_ERR_INVALID;
    std::memcpy(out16, h->team_stats, sizeof h->team_stats);
    return NQ_OK;
}
int nq_get_batch_phase_ms(const nq_handle* h0, float* out4) {
    if (!h0 || !out4) return NQ_ERR_INVALID;
    std::memcpy(out4, h0->batch_phase_ms, sizeof h0->batch_phase_ms);
    return NQ_OK;
}
int nq_get_stage_ms(const nq_handle* h, float* out8) {
    if (!h || !out8) return NQ_ERR_INVALID;
    std::memcpy(out8, h->stage_ms, sizeof h->stage_ms);
    if (h->dither_events_fresh) {
        // a stand-alone nq_dither[_device] call (asynchronous: no finish_timing there): its two stages, once their events have completed
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev[5], h->ev[6]) == hipSuccess) out8[5] = ms;
        if (hipEventElapsedTime(&ms, h->ev[6], h->ev[7]) == hipSuccess) out8[6] = ms;
    }
    return NQ_OK;
}

int nq_pnnquan_device(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors,
                      uint32_t* out_palette, int32_t* out_K) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    return pnnquan_device(h, d_argb, width, height, nMaxColors, out_palette, out_K);
}

int nq_pnnquan(nq_handle* h, const uint32_t* argb, int width, int height, int nMaxColors, uint32_t* out_palette, int32_t* out_K) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!argb || width <= 0 || height <= 0) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    const size_t n = (size_t) width * height;
    NQ_HIP(h, h->d_in.reserve(n));
    NQ_HIP(h, hipMemcpyAsync(h->d_in.p, argb, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    return pnnquan_device(h, (const uint32_t*) h->d_in.p, width, height, nMaxColors, out_palette, out_K);
}

int nq_dither_device(nq_handle* h, const uint32_t* d_argb, int width, int height, const uint32_t* palette, int K,
                     int dither, int64_t rng_seed, int mode, uint32_t* d_out_argb, uint16_t* d_out_index) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    return dither_device(h, d_argb, width, height, palette, K, dither, rng_seed, mode, d_out_argb, d_out_index);
}

int nq_dither(nq_handle* h, const uint32_t* argb, int width, int height, const uint32_t* palette, int K,
              int dither, int64_t rng_seed, int mode, uint32_t* out_argb, uint16_t* out_index) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!argb || !out_argb || width <= 0 || height <= 0) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    const size_t n = (size_t) width * height;
    NQ_HIP(h, h->d_in.reserve(n)); NQ_HIP(h, h->d_out_argb.reserve(n)); NQ_HIP(h, h->d_out_index.reserve(n));
    NQ_HIP(h, hipMemcpyAsync(h->d_in.p, argb, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    rc = dither_device(h, (const uint32_t*) h->d_in.p, width, height, palette, K, dither, rng_seed, mode,
                       (uint32_t*) h->d_out_argb.p, h->d_out_index.p);
    if (rc) return rc;
    NQ_HIP(h, hipMemcpyAsync(out_argb, h->d_out_argb.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (out_index) NQ_HIP(h, hipMemcpyAsync(out_index, h->d_out_index.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    return NQ_OK;
}

// GilbertCurve.dither(width, height, pixels, palette, ditherable, saliencies, weight, dither) (NQ/GilbertCurve.java:367-373)
int nq_gilbert_dither(nq_handle* h, int width, int height, const uint32_t* pixels, const uint32_t* palette, int K, const float* saliencies,
                      double weight, int dither, int64_t rng_seed, int mode, int32_t* out_qpixels, uint16_t* out_index) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!pixels || !out_qpixels || width <= 0 || height <= 0) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (mode != NQ_MODE_PARALLEL_TILED && mode != NQ_MODE_REFERENCE_SEQUENTIAL) NQ_FAIL(h, NQ_ERR_INVALID, "unknown mode %d", mode);
    const size_t n = (size_t) width * height;
    NQ_HIP(h, h->d_in.reserve(n)); NQ_HIP(h, h->d_out_argb.reserve(n)); NQ_HIP(h, h->d_out_index.reserve(n));
    NQ_HIP(h, hipMemcpyAsync(h->d_in.p, pixels, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (saliencies) {
        NQ_HIP(h, h->d_user_sal.reserve(n));
        NQ_HIP(h, hipMemcpyAsync(h->d_user_sal.p, saliencies, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    h->ov = nq_handle::StageOverride();
    h->ov.gilbert_only = true; h->ov.hasSal = saliencies != nullptr; h->ov.d_sal = h->d_user_sal.p; h->ov.weight = weight;
    rc = dither_device(h, (const uint32_t*) h->d_in.p, width, height, palette, K, dither, rng_seed, mode, (uint32_t*) h->d_out_argb.p, h->d_out_index.p);
    h->ov = nq_handle::StageOverride();
    if (rc) return rc;
    std::vector<uint16_t> idx(n);
    NQ_HIP(h, hipMemcpyAsync(idx.data(), h->d_out_index.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    const bool argb_out = dither || K <= 32;          // :278-279: qPixels holds ARGB only then, palette indices otherwise
    if (argb_out) NQ_HIP(h, hipMemcpyAsync(out_qpixels, h->d_out_argb.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    if (!argb_out) for (size_t i = 0; i < n; ++i) out_qpixels[i] = idx[i];
    if (out_index) std::memcpy(out_index, idx.data(), n * sizeof(uint16_t));
    return NQ_OK;
}

// BlueNoise.dither(width, height, pixels, palette, ditherable, qPixels, weight) (NQ/BlueNoise.java:207-222): qPixels holds palette
// indices on entry and ARGB on return
int nq_bluenoise_dither(nq_handle* h, int width, int height, const uint32_t* pixels, const uint32_t* palette, int K, int32_t* io_qpixels,
                        float weight, int64_t rng_seed, int mode, uint16_t* out_index) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!pixels || !io_qpixels || width <= 0 || height <= 0 || !palette || K < 1) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (mode != NQ_MODE_PARALLEL_TILED && mode != NQ_MODE_REFERENCE_SEQUENTIAL) NQ_FAIL(h, NQ_ERR_INVALID, "unknown mode %d", mode);
    const size_t n = (size_t) width * height;
    std::vector<uint16_t> idx(n);
    for (size_t i = 0; i < n; ++i) {
        if (io_qpixels[i] < 0 || io_qpixels[i] >= K) NQ_FAIL(h, NQ_ERR_INVALID, "qPixels[%zu] = %d is not a palette index", i, io_qpixels[i]);
        idx[i] = (uint16_t) io_qpixels[i];
    }
    NQ_HIP(h, h->d_in.reserve(n)); NQ_HIP(h, h->d_out_argb.reserve(n)); NQ_HIP(h, h->d_out_index.reserve(n));
    NQ_HIP(h, hipMemcpyAsync(h->d_in.p, pixels, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    NQ_HIP(h, hipMemcpyAsync(h->d_out_index.p, idx.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));        // idx goes out of scope
    h->ov = nq_handle::StageOverride();
    h->ov.blue_only = true; h->ov.blueWeight = weight; h->ov.weight = h->params.weight;
    rc = dither_device(h, (const uint32_t*) h->d_in.p, width, height, palette, K, 0, rng_seed, mode, (uint32_t*) h->d_out_argb.p, h->d_out_index.p);
    h->ov = nq_handle::StageOverride();
    if (rc) return rc;
    NQ_HIP(h, hipMemcpyAsync(io_qpixels, h->d_out_argb.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (out_index) NQ_HIP(h, hipMemcpyAsync(out_index, h->d_out_index.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    return NQ_OK;
}

int nq_convert_device(nq_handle* h, const uint32_t* d_argb, int width, int height, int nMaxColors, int dither,
                      int64_t rng_seed, int mode, uint32_t* d_out_argb, uint16_t* d_out_index,
                      uint32_t* out_palette, int32_t* out_K) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    rc = pnnquan_device(h, d_argb, width, height, nMaxColors, out_palette, out_K);
    if (rc) return rc;
    rc = dither_device(h, d_argb, width, height, out_palette, *out_K, dither, rng_seed, mode, d_out_argb, d_out_index);
    if (rc) return rc;
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    finish_timing(h);
    return NQ_OK;
}

int nq_convert_batch_device(nq_handle* const* hs, int n, const uint32_t* const* d_argb, const int32_t* widths, const int32_t* heights,
                            int nMaxColors, int dither, const int64_t* rng_seeds, int mode,
                            uint32_t* const* d_out_argb, uint16_t* const* d_out_index,
                            uint32_t* out_palettes, int32_t palette_stride, int32_t* out_K) {
    if (!hs || n <= 0 || !hs[0]) return NQ_ERR_INVALID;
    nq_handle* h0 = hs[0];
    if (!d_argb || !widths || !heights || !rng_seeds || !d_out_argb || !out_palettes || !out_K)
        NQ_FAIL(h0, NQ_ERR_INVALID, "bad argument");
    if (palette_stride < std::max(nMaxColors, 2)) NQ_FAIL(h0, NQ_ERR_INVALID, "palette_stride < max(nMaxColors, 2)");
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) NQ_FAIL(h0, NQ_ERR_INVALID, "null handle in batch");
        if (hs[i]->device != h0->device) NQ_FAIL(h0, NQ_ERR_INVALID, "handles of a batch must share one device");
        for (int j = 0; j < i; ++j) if (hs[j] == hs[i]) NQ_FAIL(h0, NQ_ERR_INVALID, "a handle appears twice in the batch");
    }
    // The per-image stages in front of the merge loops run on FOUR lanes by default, eight at most (image i: stream and per-pixel scratch of
    // lane i % L; the scratch is that of the first L handles): while the host waits for a read-back of one image, and while a kernel with a
    // long tail or a small grid runs (the fullest bin's chain of the histogram, the list compactions), the queued kernels of the
    // other lanes keep the GPU busy (NQ_BATCH_LANES = 1..8 overrides).  The merge launch joins the lanes.
    struct Restore {
        nq_handle* const* hs; int n; std::vector<hipStream_t> streams;
        ~Restore() { for (int i = 0; i < n; ++i) { hs[i]->stream = streams[i]; hs[i]->sc = &hs[i]->own; hs[i]->light_events = false; } }
    } restore{hs, n, {}};
    for (int i = 0; i < n; ++i) hs[i]->light_events = i >= 16;
    for (int i = 0; i < n; ++i) restore.streams.push_back(hs[i]->stream);
    auto fail_from = [&](nq_handle* h, int rc) { if (h != h0) h0->err = h->err; return rc; };
    for (int i = 0; i < n; ++i) {
        int rc = use_device(hs[i]);            // tables / events on the handle's own stream, before it is redirected
        if (rc) return fail_from(hs[i], rc);
        if (i) NQ_HIP(h0, hipStreamSynchronize(hs[i]->stream));
    }
    if (n > 1 && !h0->lane_stream) NQ_HIP(h0, hipStreamCreateWithFlags(&h0->lane_stream, hipStreamNonBlocking));
    int L = std::min(n, 4);          // (measured on 1024 images of 4096^2, ONE issuing thread: prepare phase 640 / 557 / 552 / 543 us per image with 1 / 2 / 3 / 4 lanes)
    if (const char* f = std::getenv("NQ_BATCH_LANES")) { const int t = std::atoi(f); if (t >= 1 && t <= 8) L = std::min(t, n); }
    hipStream_t lane_s[8] = {h0->stream, n > 1 ? h0->lane_stream : h0->stream, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Scratch* lane_sc[8] = {&h0->own, n > 1 ? &hs[1]->own : &h0->own, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int k = 2; k < L; ++k) {
        if (!h0->more_lanes[k - 2]) NQ_HIP(h0, hipStreamCreateWithFlags(&h0->more_lanes[k - 2], hipStreamNonBlocking));
        lane_s[k] = h0->more_lanes[k - 2]; lane_sc[k] = &hs[k]->own;
    }
    for (int i = 0; i < n; ++i) { hs[i]->stream = lane_s[i % L]; hs[i]->sc = lane_sc[i % L]; }
    std::vector<PaletteJob> jobs(n);
    std::vector<const PaletteJob*> jp(n);
    (void) hipEventRecord(h0->bev[0], lane_s[0]);
    {
        // One host thread per lane (round 4): pnnquan_prepare waits twice for the device per image (the pre-scan's scalars, the bin count),
        // and a single issuing thread that blocks there leaves the other lanes without new work -- only two images ever overlapped.  Lane k's
        // thread walks the images k, k + L, ... in order (they share lane k's scratch); handles, streams and scratch sets of different lanes
        // are disjoint, the launch-error slot is per thread.
        std::vector<int> lane_rc(L, NQ_OK), lane_bad(L, -1);
        auto lane_work = [&](int k) {
            int at = k;
            try {                                        // (nothing may leave a lane's thread: an escaping exception would end the process)
                if (hipSetDevice(h0->device) != hipSuccess) { lane_rc[k] = NQ_ERR_HIP; lane_bad[k] = k; hs[k]->err = "hipSetDevice failed in a batch lane"; return; }
                for (int i = k; i < n; i += L) {
                    at = i;
                    const int rc = pnnquan_prepare(hs[i], d_argb[i], widths[i], heights[i], nMaxColors, out_palettes + (size_t) i * palette_stride,
                                                   out_K + i, &jobs[i]);
                    if (rc) { lane_rc[k] = rc; lane_bad[k] = i; return; }
                    jp[i] = &jobs[i];
                }
            } catch (const std::exception& e) {
                lane_rc[k] = NQ_ERR_HIP; lane_bad[k] = at;
                try { hs[at]->err = std::string("exception in a batch lane: ") + e.what(); } catch (...) {}
            } catch (...) { lane_rc[k] = NQ_ERR_HIP; lane_bad[k] = at; }
        };
        std::vector<std::thread> workers;
        std::vector<int> inline_lanes;                   // (a lane whose thread could not be created is walked by the caller's thread)
        for (int k = 1; k < L; ++k) {
            try { workers.emplace_back(lane_work, k); } catch (const std::exception&) { inline_lanes.push_back(k); }
        }
        lane_work(0);
        for (int k : inline_lanes) lane_work(k);
        for (auto& t : workers) t.join();
        for (int k = 0; k < L; ++k)
            if (lane_rc[k]) return fail_from(hs[lane_bad[k] >= 0 && lane_bad[k] < n ? lane_bad[k] : 0], lane_rc[k]);
    }
    for (int k = 1; k < L; ++k) NQ_HIP(h0, hipStreamSynchronize(lane_s[k]));          // every prepare has been issued: join before the merge launch
    (void) hipEventRecord(h0->bev[1], lane_s[0]);
    int rc = merge_launch(h0, jp.data(), n);
    if (rc) return rc;
    (void) hipEventRecord(h0->bev[2], lane_s[0]);
    // behind the merge launch everything runs on lane 0 again: two dither kernels side by side would only slow each other down
    for (int i = 0; i < n; ++i) { hs[i]->stream = lane_s[0]; hs[i]->sc = lane_sc[0]; }
    for (int i = 0; i < n; ++i) if (jobs[i].merge) rec(hs[i], 4);
    // every palette of the batch comes back behind ONE wait; the per-image passes then follow each other on the stream with no
    // host round trip in between (a wait per image left the GPU idle for ~0.1 ms of every image's ~1 ms)
    std::vector<int> pal_status(n, 0);
    for (int i = 0; i < n; ++i)
        if (jobs[i].merge) {
            rc = palette_fetch(hs[i], jobs[i], out_palettes + (size_t) i * palette_stride, &pal_status[i]);
            if (rc) return fail_from(hs[i], rc);
        }
    NQ_HIP(h0, hipStreamSynchronize(lane_s[0]));
    NQ_HIP(h0, launch_status());
    for (int i = 0; i < n; ++i)
        if (jobs[i].merge) {
            rc = palette_check(hs[i], jobs[i], pal_status[i], out_K + i);
            if (rc) return fail_from(hs[i], rc);
        }
    for (int i = 0; i < n; ++i) {
        uint32_t* pal = out_palettes + (size_t) i * palette_stride;
        rc = dither_device(hs[i], d_argb[i], widths[i], heights[i], pal, out_K[i], dither, rng_seeds[i], mode, d_out_argb[i],
                           d_out_index ? d_out_index[i] : nullptr);
        if (rc) return fail_from(hs[i], rc);
    }
    (void) hipEventRecord(h0->bev[3], lane_s[0]);
    NQ_HIP(h0, hipStreamSynchronize(lane_s[0]));
    for (int i = 0; i < n; ++i) finish_timing(hs[i]);
    finish_batch_timing(h0);
    return NQ_OK;
}

// Host-buffer form of the batch: uploads run ahead of the per-image stages on a copy stream, every image's result is copied
// back while the next image is dithered (a ring of three device output buffers), so only the inputs (4 B/pixel) and the
// quantizer state stay resident for the whole batch.
int nq_convert_batch(nq_handle* const* hs, int n, const uint32_t* const* argb, const int32_t* widths, const int32_t* heights,
                     int nMaxColors, int dither, const int64_t* rng_seeds, int mode,
                     uint32_t* const* out_argb, uint16_t* const* out_index,
                     uint32_t* out_palettes, int32_t palette_stride, int32_t* out_K) {
    if (!hs || n <= 0 || !hs[0]) return NQ_ERR_INVALID;
    nq_handle* h0 = hs[0];
    if (!argb || !widths || !heights || !rng_seeds || !out_argb || !out_palettes || !out_K)
        NQ_FAIL(h0, NQ_ERR_INVALID, "bad argument");
    if (palette_stride < std::max(nMaxColors, 2)) NQ_FAIL(h0, NQ_ERR_INVALID, "palette_stride < max(nMaxColors, 2)");
    size_t max_px = 0;
    for (int i = 0; i < n; ++i) {
        if (!hs[i] || !argb[i] || !out_argb[i] || widths[i] <= 0 || heights[i] <= 0) NQ_FAIL(h0, NQ_ERR_INVALID, "bad argument for image %d", i);
        if (hs[i]->device != h0->device) NQ_FAIL(h0, NQ_ERR_INVALID, "handles of a batch must share one device");
        for (int j = 0; j < i; ++j) if (hs[j] == hs[i]) NQ_FAIL(h0, NQ_ERR_INVALID, "a handle appears twice in the batch");
        max_px = std::max(max_px, (size_t) widths[i] * heights[i]);
    }
    struct Restore {
        nq_handle* const* hs; int n; std::vector<hipStream_t> streams; std::vector<hipEvent_t> events;
        ~Restore() {
            for (int i = 0; i < n; ++i) { hs[i]->stream = streams[i]; hs[i]->sc = &hs[i]->own; }
            for (hipEvent_t e : events) (void) hipEventDestroy(e);
        }
    } restore{hs, n, {}, {}};
    for (int i = 0; i < n; ++i) restore.streams.push_back(hs[i]->stream);
    auto fail_from = [&](nq_handle* h, int rc) { if (h != h0) h0->err = h->err; return rc; };
    for (int i = 0; i < n; ++i) {
        int rc = use_device(hs[i]);
        if (rc) return fail_from(hs[i], rc);
        if (i) NQ_HIP(h0, hipStreamSynchronize(hs[i]->stream));
        NQ_HIP(h0, hs[i]->d_in.reserve((size_t) widths[i] * heights[i]));
        hs[i]->stream = h0->stream; hs[i]->sc = &h0->own;
    }
    constexpr int RING = 3;
    for (int r = 0; r < RING; ++r) { NQ_HIP(h0, h0->ring_argb[r].reserve(max_px)); NQ_HIP(h0, h0->ring_index[r].reserve(max_px)); }
    if (!h0->copy_stream) NQ_HIP(h0, hipStreamCreateWithFlags(&h0->copy_stream, hipStreamNonBlocking));
    hipStream_t cs = h0->copy_stream;
    auto new_event = [&](hipEvent_t* e) -> hipError_t {
        hipError_t rc = hipEventCreateWithFlags(e, hipEventDisableTiming);
        if (rc == hipSuccess) restore.events.push_back(*e);
        return rc;
    };
    std::vector<hipEvent_t> ev_up(n), ev_done(n), ev_dl(n);
    for (int i = 0; i < n; ++i) { NQ_HIP(h0, new_event(&ev_up[i])); NQ_HIP(h0, new_event(&ev_done[i])); NQ_HIP(h0, new_event(&ev_dl[i])); }
    int uploaded = 0;
    auto upload_until = [&](int last) -> int {          // asynchronous when the caller's buffers are page-locked
        for (; uploaded <= last && uploaded < n; ++uploaded) {
            const int i = uploaded;
            NQ_HIP(h0, hipMemcpyAsync(hs[i]->d_in.p, argb[i], (size_t) widths[i] * heights[i] * sizeof(int), hipMemcpyHostToDevice, cs));
            NQ_HIP(h0, hipEventRecord(ev_up[i], cs));
        }
        return NQ_OK;
    };
    std::vector<PaletteJob> jobs(n);
    std::vector<const PaletteJob*> jp(n);
    (void) hipEventRecord(h0->bev[0], h0->stream);
    for (int i = 0; i < n; ++i) {
        int rc = upload_until(i + 2);
        if (rc) return rc;
        NQ_HIP(h0, hipStreamWaitEvent(h0->stream, ev_up[i], 0));
        rc = pnnquan_prepare(hs[i], (const uint32_t*) hs[i]->d_in.p, widths[i], heights[i], nMaxColors,
                             out_palettes + (size_t) i * palette_stride, out_K + i, &jobs[i]);
        if (rc) return fail_from(hs[i], rc);
        jp[i] = &jobs[i];
    }
    (void) hipEventRecord(h0->bev[1], h0->stream);
    int rc = merge_launch(h0, jp.data(), n);
    if (rc) return rc;
    (void) hipEventRecord(h0->bev[2], h0->stream);
    for (int i = 0; i < n; ++i) if (jobs[i].merge) rec(hs[i], 4);
    for (int i = 0; i < n; ++i) {
        uint32_t* pal = out_palettes + (size_t) i * palette_stride;
        if (jobs[i].merge) {
            rc = palette_finish(hs[i], jobs[i], pal, out_K + i);
            if (rc) return fail_from(hs[i], rc);
        }
        const int r = i % RING;
        if (i >= RING) NQ_HIP(h0, hipStreamWaitEvent(h0->stream, ev_dl[i - RING], 0));     // the slot's previous result has left
        rc = dither_device(hs[i], (const uint32_t*) hs[i]->d_in.p, widths[i], heights[i], pal, out_K[i], dither, rng_seeds[i], mode,
                           (uint32_t*) h0->ring_argb[r].p, h0->ring_index[r].p);
        if (rc) return fail_from(hs[i], rc);
        NQ_HIP(h0, hipEventRecord(ev_done[i], h0->stream));
        NQ_HIP(h0, hipStreamWaitEvent(cs, ev_done[i], 0));
        const size_t px = (size_t) widths[i] * heights[i];
        NQ_HIP(h0, hipMemcpyAsync(out_argb[i], h0->ring_argb[r].p, px * sizeof(int), hipMemcpyDeviceToHost, cs));
        if (out_index && out_index[i])
            NQ_HIP(h0, hipMemcpyAsync(out_index[i], h0->ring_index[r].p, px * sizeof(uint16_t), hipMemcpyDeviceToHost, cs));
        NQ_HIP(h0, hipEventRecord(ev_dl[i], cs));
    }
    (void) hipEventRecord(h0->bev[3], h0->stream);
    NQ_HIP(h0, hipStreamSynchronize(h0->stream));
    NQ_HIP(h0, hipStreamSynchronize(cs));
    for (int i = 0; i < n; ++i) finish_timing(hs[i]);
    finish_batch_timing(h0);
    return NQ_OK;
}

int nq_convert(nq_handle* h, const uint32_t* argb, int width, int height, int nMaxColors, int dither,
               int64_t rng_seed, int mode, uint32_t* out_argb, uint16_t* out_index, uint32_t* out_palette, int32_t* out_K) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!argb || !out_argb || width <= 0 || height <= 0) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    const size_t n = (size_t) width * height;
    NQ_HIP(h, h->d_in.reserve(n)); NQ_HIP(h, h->d_out_argb.reserve(n)); NQ_HIP(h, h->d_out_index.reserve(n));
    NQ_HIP(h, hipMemcpyAsync(h->d_in.p, argb, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    rc = nq_convert_device(h, (const uint32_t*) h->d_in.p, width, height, nMaxColors, dither, rng_seed, mode,
                           (uint32_t*) h->d_out_argb.p, h->d_out_index.p, out_palette, out_K);
    if (rc) return rc;
    NQ_HIP(h, hipMemcpyAsync(out_argb, h->d_out_argb.p, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (out_index) NQ_HIP(h, hipMemcpyAsync(out_index, h->d_out_index.p, n * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    return NQ_OK;
}

int nq_nearest_index(nq_handle* h, const uint32_t* palette, int K, const uint32_t* colors, int64_t M, int16_t* out_index) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!palette || K < 1 || K > 8192 || !colors || M < 0 || !out_index) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (M == 0) return NQ_OK;
    { int rcp = upload_palette(h, palette, K); if (rcp) return rcp; }
    NQ_HIP(h, h->d_colors.reserve((size_t) M)); NQ_HIP(h, h->d_short.reserve((size_t) M));
    NQ_HIP(h, hipMemcpyAsync(h->d_colors.p, colors, (size_t) M * sizeof(int), hipMemcpyHostToDevice, h->stream));
    DevParams P = dev_params(h, K);
    nq::ListsView lv;
    rc = prepare_lists(h, P, &lv);
    if (rc) return rc;
    if (h->use_fast_dither && fast_lookup_eligible(P, lv))
        launch_fast_nearest_index(P, lv, h->d_palette.p, packed_lists(h), h->d_colors.p, M, h->d_short.p, h->stream);
    else
        launch_nearest_index(P, h->d_palette.p, lv, h->d_colors.p, M, h->d_short.p, h->stream);
    NQ_HIP(h, launch_status());
    NQ_HIP(h, hipMemcpyAsync(out_index, h->d_short.p, (size_t) M * sizeof(short), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    return NQ_OK;
}

int nq_closest_tuple(nq_handle* h, const uint32_t* palette, int K, const uint32_t* colors, int64_t M, int32_t* out_closest4) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!palette || K < 1 || K > 8192 || !colors || M < 0 || !out_closest4) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    if (M == 0) return NQ_OK;
    { int rcp = upload_palette(h, palette, K); if (rcp) return rcp; }
    NQ_HIP(h, h->d_colors.reserve((size_t) M)); NQ_HIP(h, h->d_tuple.reserve((size_t) 4 * M));
    NQ_HIP(h, hipMemcpyAsync(h->d_colors.p, colors, (size_t) M * sizeof(int), hipMemcpyHostToDevice, h->stream));
    DevParams P = dev_params(h, K);
    nq::ListsView lv;
    rc = prepare_lists(h, P, &lv);
    if (rc) return rc;
    if (h->use_fast_dither && fast_lookup_eligible(P, lv))
        launch_fast_closest_tuple(P, lv, h->d_palette.p, packed_lists(h), h->d_colors.p, M, h->d_tuple.p, h->stream);
    else
        launch_closest_tuple(P, h->d_palette.p, lv, h->d_colors.p, M, h->d_tuple.p, h->stream);
    NQ_HIP(h, launch_status());
    NQ_HIP(h, hipMemcpyAsync(out_closest4, h->d_tuple.p, (size_t) 4 * M * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    return NQ_OK;
}

int nq_band_scan_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, int64_t index_offset, int nMaxColors, int64_t* d_scan3) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!d_argb || n_pixels <= 0 || !d_scan3) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    (void) nMaxColors;
    launch_prescan((const int*) d_argb, n_pixels, index_offset, (long long*) d_scan3, h->stream);
    NQ_HIP(h, launch_status());
    return NQ_OK;
}

int nq_set_scan(nq_handle* h, int nMaxColors, int64_t transparent_index, uint32_t transparent_color, int64_t semi_count) {
    if (!h) return NQ_ERR_INVALID;
    apply_scan(h, nMaxColors, transparent_index, transparent_color, semi_count);
    return NQ_OK;
}

int nq_band_distinct_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, int cap, int64_t* out_count, uint32_t* out_colors) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!d_argb || n_pixels <= 0 || cap < 1 || !out_count || !out_colors) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    std::vector<int32_t> cols;
    rc = distinct_colors(h, d_argb, n_pixels, cap, out_count, &cols);
    if (rc) return rc;
    if (*out_count <= cap) std::memcpy(out_colors, cols.data(), cols.size() * sizeof(int32_t));
    return NQ_OK;
}

int nq_band_color_presence_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, uint8_t* d_presence, int cap_other,
                                  int64_t* out_other_count, uint32_t* out_other) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!d_argb || n_pixels <= 0 || !d_presence || cap_other < 1 || !out_other_count || !out_other) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    const unsigned slots = 1u << 18;
    if ((unsigned) cap_other > slots / 2) NQ_FAIL(h, NQ_ERR_INVALID, "cap_other above %u", slots / 2);
    NQ_HIP(h, h->sc->dk_a.reserve((size_t) slots + 2));
    unsigned* d_set = h->sc->dk_a.p;
    unsigned* d_cnt = d_set + slots;
    launch_color_presence((const int*) d_argb, n_pixels, h->params.transparentColor, d_presence, d_set, slots, d_cnt, h->stream);
    NQ_HIP(h, launch_status());
    unsigned cnt[2] = {0, 0};
    NQ_HIP(h, hipMemcpyAsync(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
    NQ_HIP(h, hipStreamSynchronize(h->stream));
    if (cnt[1] || cnt[0] > (unsigned) cap_other) { *out_other_count = -1; return NQ_OK; }      // too many non-opaque colours: caller decides
    std::vector<unsigned> set(slots);
    NQ_HIP(h, hipMemcpy(set.data(), d_set, (size_t) slots * sizeof(unsigned), hipMemcpyDeviceToHost));
    int64_t k = 0;
    for (unsigned v : set) if (v != 0xFFFFFFFFu && k < cap_other) out_other[k++] = v;
    *out_other_count = k;
    return NQ_OK;
}

int nq_set_distinct(nq_handle* h, int64_t count, const uint32_t* colors) {
    if (!h) return NQ_ERR_INVALID;
    h->ext_distinct_valid = true;
    h->ext_distinct_many = count < 0;
    h->ext_distinct.clear();
    if (count > 0) {
        if (!colors) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
        h->ext_distinct.assign((const int32_t*) colors, (const int32_t*) colors + count);
    }
    return NQ_OK;
}

int nq_band_histogram_device(nq_handle* h, const uint32_t* d_argb, int64_t n_pixels, double* d_hist) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!d_argb || n_pixels <= 0 || !d_hist) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    rc = reserve_palette_ws(h, n_pixels);
    if (rc) return rc;
    const nq_params& p = h->params;
    nq::HistParams hp;
    hp.hasSemi = p.hasSemiTransparency; hp.hasTransp = p.nMaxColors < 64 || p.transparentPixelIndex >= 0;
    hp.transparentColor = p.transparentColor; hp.rewriteTransparent = 0;
    nq::SortWorkspace ws;
    ws.keys_a = h->sc->keys_a.p; ws.keys_b = h->sc->keys_b.p; ws.vals_a = h->sc->vals_a.p; ws.vals_b = h->sc->vals_b.p;
    ws.tmp = h->sc->sort_tmp.p; ws.tmp_bytes = h->sc->sort_tmp.n; ws.seg_start = h->sc->seg.p; ws.seg_end = h->sc->seg.p + 65536;
    launch_histogram(h->kind, (const int*) d_argb, n_pixels, hp, ws, d_hist, h->stream);
    NQ_HIP(h, launch_status());
    return NQ_OK;
}

int nq_palette_from_histograms_device(nq_handle* h, const double* d_hists, int n_bands, int nMaxColors,
                                      uint32_t* out_palette, int32_t* out_K) {
    if (!h) return NQ_ERR_INVALID;
    int rc = use_device(h);
    if (rc) return rc;
    if (!d_hists || n_bands < 1 || !out_palette || !out_K || nMaxColors < 3) NQ_FAIL(h, NQ_ERR_INVALID, "bad argument");
    rc = reserve_palette_ws(h, 1);
    if (rc) return rc;
    return palette_from_hist(h, d_hists, n_bands, nMaxColors, out_palette, out_K);
}

} // extern "C"
