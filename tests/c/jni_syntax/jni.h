/* tests/c/jni_syntax/jni.h -- NOT a JDK header: the handful of JNI declarations (names and signatures as in the public JNI
 * specification) that nquant.android_amd/jni/nquant_jni.c uses, so that `gcc -fsyntax-only` can check the shim's calls into
 * include/nquant_abi.h in an image without a JDK (tests/test_abi_cpu.py).  Never linked, never shipped; a real build uses
 * $JAVA_HOME/include/jni.h (see the shim's header comment). */
#ifndef NQ_JNI_SYNTAX_STUB_H
#define NQ_JNI_SYNTAX_STUB_H
#include <stdint.h>
#define JNIEXPORT
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_ABORT 2
typedef int32_t jint;
typedef int64_t jlong;
typedef int16_t jshort;
typedef uint8_t jboolean;
typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jshortArray;
typedef jarray jobjectArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv*, const char*);
    jint (*ThrowNew)(JNIEnv*, jclass, const char*);
    jsize (*GetArrayLength)(JNIEnv*, jarray);
    jobjectArray (*NewObjectArray)(JNIEnv*, jsize, jclass, jobject);
    jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
    void (*SetObjectArrayElement)(JNIEnv*, jobjectArray, jsize, jobject);
    jintArray (*NewIntArray)(JNIEnv*, jsize);
    jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
    jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
    jshort* (*GetShortArrayElements)(JNIEnv*, jshortArray, jboolean*);
    void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
    void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
    void (*ReleaseShortArrayElements)(JNIEnv*, jshortArray, jshort*, jint);
    void (*SetIntArrayRegion)(JNIEnv*, jintArray, jsize, jsize, const jint*);
    void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
};
#endif
