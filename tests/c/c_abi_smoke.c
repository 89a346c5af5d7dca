/* Plain-C caller of the C ABI (include/nquant_abi.h): what a cgo / JNI / FFI binding does, without Python in between.
 * Built by tests/test_gpu_boundary.py with gcc against libnquant_hip.so and run on the GPU box.
 *   nq_create -> nq_convert (LAB, 256 colours, dither) on a synthetic 96x64 ARGB image -> checks every output pixel against
 *   palette[index], writes the palette and the index map to the file named on the command line, then exercises the error path
 *   (nq_convert with a NULL pixel pointer must fail with a message from nq_last_error) and nq_destroy. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nquant_abi.h"

static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int W = 96, H = 64, K = 256;
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 2; }
    if (nq_abi_version() != NQ_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    uint32_t* px = malloc(sizeof(uint32_t) * W * H);
    for (int i = 0; i < W * H; ++i) px[i] = 0xFF000000u | (uint32_t) (mix64(41 + (uint64_t) i) & 0xFFFFFF);
    nq_handle* h = NULL;
    int rc = nq_create(NQ_KIND_LAB, 0, &h);
    if (rc != NQ_OK) { fprintf(stderr, "nq_create: %d %s\n", rc, nq_last_error(NULL)); return 1; }
    uint32_t* out = malloc(sizeof(uint32_t) * W * H);
    uint16_t* idx = malloc(sizeof(uint16_t) * W * H);
    uint32_t pal[256];
    int32_t k = 0;
    rc = nq_set_tile(h, 8, 8);
    if (rc == NQ_OK) rc = nq_convert(h, px, W, H, K, 1, 7, NQ_MODE_PARALLEL_TILED, out, idx, pal, &k);
    if (rc != NQ_OK) { fprintf(stderr, "nq_convert: %d %s\n", rc, nq_last_error(h)); return 1; }
    if (k < 2 || k > K) { fprintf(stderr, "palette length %d\n", k); return 1; }
    for (int i = 0; i < W * H; ++i)
        if (idx[i] >= k || out[i] != pal[idx[i]]) { fprintf(stderr, "pixel %d: out != palette[index]\n", i); return 1; }
    nq_params p;
    if (nq_get_params(h, &p) != NQ_OK || p.paletteLength != k || p.kind != NQ_KIND_LAB) { fprintf(stderr, "nq_get_params\n"); return 1; }
    FILE* f = fopen(argv[1], "wb");
    if (!f) return 1;
    fwrite(&k, sizeof k, 1, f); fwrite(pal, sizeof(uint32_t), (size_t) k, f); fwrite(idx, sizeof(uint16_t), (size_t) W * H, f); fwrite(px, sizeof(uint32_t), (size_t) W * H, f);
    fclose(f);
    rc = nq_convert(h, NULL, W, H, K, 1, 7, NQ_MODE_PARALLEL_TILED, out, idx, pal, &k);
    if (rc == NQ_OK || strlen(nq_last_error(h)) == 0) { fprintf(stderr, "the error path did not report\n"); return 1; }
    nq_destroy(h);
    printf("c_abi_smoke ok: K=%d\n", k);
    free(px); free(out); free(idx);
    return 0;
}
