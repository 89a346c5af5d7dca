/*
 * nquant_jni.c -- JNI shim between the reference's Java API and libnquant_hip.so (include/nquant_abi.h).
 * NOT compiled in the build image (no JDK / jni.h there); build on a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include nquant_jni.c \
 *       -L.. -lnquant_hip -o libnquant_jni.so
 * Java side: ../java/com/android/nQuant/PnnQuantizer.java, PnnLABQuantizer.java (same class names, constructor and
 * convert()/hasAlpha() signatures as the reference: NQ/PnnQuantizer.java:35,409,458; NQ/PnnLABQuantizer.java:24).
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include "nquant_abi.h"

static void throw_rt(JNIEnv* env, const char* msg) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg ? msg : "nquant error");
}

JNIEXPORT jlong JNICALL Java_com_android_nQuant_PnnQuantizer_nqCreate(JNIEnv* env, jclass c, jint kind, jint device) {
    nq_handle* h = NULL;
    if (nq_create(kind, device, &h) != NQ_OK) { throw_rt(env, nq_last_error(NULL)); return 0; }
    return (jlong) (intptr_t) h;
}

JNIEXPORT void JNICALL Java_com_android_nQuant_PnnQuantizer_nqDestroy(JNIEnv* env, jclass c, jlong h) {
    nq_destroy((nq_handle*) (intptr_t) h);
}

/* returns the palette; fills outArgb (w*h) and, when non-null, outIndex (w*h) */
JNIEXPORT jintArray JNICALL Java_com_android_nQuant_PnnQuantizer_nqConvert(JNIEnv* env, jclass c, jlong hh, jintArray argb,
        jint w, jint hgt, jint nMaxColors, jboolean dither, jlong seed, jint mode, jintArray outArgb, jshortArray outIndex) {
    nq_handle* h = (nq_handle*) (intptr_t) hh;
    const int cap = nMaxColors > 2 ? nMaxColors : 2;
    uint32_t* palette = (uint32_t*) malloc((size_t) cap * sizeof(uint32_t));     /* per call: two threads may convert different objects at once */
    if (!palette) { throw_rt(env, "out of memory"); return NULL; }
    int32_t K = 0;
    /* The convert blocks on the GPU for up to seconds (the merge loop): plain Get/Release<Type>ArrayElements, not critical regions --
     * a critical region must be short and non-blocking and would stall the collector JVM-wide for the whole call. */
    jint* in = (*env)->GetIntArrayElements(env, argb, NULL);
    jint* out = (*env)->GetIntArrayElements(env, outArgb, NULL);
    jshort* idx = outIndex ? (*env)->GetShortArrayElements(env, outIndex, NULL) : NULL;
    if (!in || !out || (outIndex && !idx)) {
        if (idx) (*env)->ReleaseShortArrayElements(env, outIndex, idx, JNI_ABORT);
        if (out) (*env)->ReleaseIntArrayElements(env, outArgb, out, JNI_ABORT);
        if (in) (*env)->ReleaseIntArrayElements(env, argb, in, JNI_ABORT);
        free(palette);
        return NULL;                                                             /* OutOfMemoryError already pending */
    }
    int rc = nq_convert(h, (const uint32_t*) in, w, hgt, nMaxColors, dither ? 1 : 0, seed, mode,
                        (uint32_t*) out, (uint16_t*) idx, palette, &K);
    if (idx) (*env)->ReleaseShortArrayElements(env, outIndex, idx, 0);
    (*env)->ReleaseIntArrayElements(env, outArgb, out, 0);
    (*env)->ReleaseIntArrayElements(env, argb, in, JNI_ABORT);                   /* the input is never modified */
    if (rc != NQ_OK) { free(palette); throw_rt(env, nq_last_error(h)); return NULL; }        /* convert() `throws Exception` */
    jintArray pal = (*env)->NewIntArray(env, K);
    if (pal) (*env)->SetIntArrayRegion(env, pal, 0, K, (const jint*) palette);
    free(palette);
    return pal;
}

JNIEXPORT jboolean JNICALL Java_com_android_nQuant_PnnQuantizer_nqHasAlpha(JNIEnv* env, jclass c, jlong hh) {
    nq_params p;
    if (nq_get_params((nq_handle*) (intptr_t) hh, &p) != NQ_OK) return JNI_FALSE;
    return p.transparentPixelIndex > -1 ? JNI_TRUE : JNI_FALSE;               /* NQ/PnnQuantizer.java:458-460 */
}

/* convertBatch(): direct IntBuffers in, direct IntBuffers out -> nq_convert_batch (uploads / read-backs overlapped; all merge
 * loops in one launch).  Returns int[n][] palettes. */
JNIEXPORT jobjectArray JNICALL Java_com_android_nQuant_PnnQuantizer_nqConvertBatch(JNIEnv* env, jclass c, jlongArray handles,
        jobjectArray in, jintArray widths, jintArray heights, jint nMaxColors, jboolean dither, jlongArray seeds, jint mode,
        jobjectArray out) {
    const jsize n = (*env)->GetArrayLength(env, handles);
    const int stride = nMaxColors > 2 ? nMaxColors : 2;
    nq_handle** hs = malloc(sizeof(*hs) * n);
    const uint32_t** src = malloc(sizeof(*src) * n);
    uint32_t** dst = malloc(sizeof(*dst) * n);
    uint32_t* palettes = malloc(sizeof(uint32_t) * (size_t) stride * n);
    int32_t* K = malloc(sizeof(int32_t) * n);
    jlong* hh = (*env)->GetLongArrayElements(env, handles, NULL);
    jlong* sd = (*env)->GetLongArrayElements(env, seeds, NULL);
    jint* w = (*env)->GetIntArrayElements(env, widths, NULL);
    jint* hg = (*env)->GetIntArrayElements(env, heights, NULL);
    for (jsize i = 0; i < n; ++i) {
        hs[i] = (nq_handle*) (intptr_t) hh[i];
        src[i] = (const uint32_t*) (*env)->GetDirectBufferAddress(env, (*env)->GetObjectArrayElement(env, in, i));
        dst[i] = (uint32_t*) (*env)->GetDirectBufferAddress(env, (*env)->GetObjectArrayElement(env, out, i));
    }
    const int rc = nq_convert_batch(hs, n, src, (const int32_t*) w, (const int32_t*) hg, nMaxColors, dither ? 1 : 0,
                                    (const int64_t*) sd, mode, dst, NULL, palettes, stride, K);
    jobjectArray result = NULL;
    if (rc != NQ_OK) throw_rt(env, nq_last_error(hs[0]));
    else {
        result = (*env)->NewObjectArray(env, n, (*env)->FindClass(env, "[I"), NULL);
        for (jsize i = 0; i < n; ++i) {
            jintArray pal = (*env)->NewIntArray(env, K[i]);
            (*env)->SetIntArrayRegion(env, pal, 0, K[i], (const jint*) (palettes + (size_t) i * stride));
            (*env)->SetObjectArrayElement(env, result, i, pal);
        }
    }
    (*env)->ReleaseIntArrayElements(env, heights, hg, JNI_ABORT); (*env)->ReleaseIntArrayElements(env, widths, w, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, seeds, sd, JNI_ABORT); (*env)->ReleaseLongArrayElements(env, handles, hh, JNI_ABORT);
    free(K); free(palettes); free(dst); free(src); free(hs);
    return result;
}
