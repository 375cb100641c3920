/*
 * Minimal hand-declared subset of <jni.h> — ONLY so that `gcc -fsyntax-only jni/stellar_rw_jni.c` can run in an image
 * without a JDK (tests/test_host_cpu.py::test_jni_shim_syntax).  Not a JDK header, never used to build a library: the
 * real build (`make -C jni JAVA_HOME=...`) uses the JDK's own jni.h.  The used entries sit at their JNI 1.6 function
 * table indices (the slot order is fixed by the JNI specification); everything else is padding.
 */
#ifndef SRW_TEST_JNI_STUB_H
#define SRW_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
  void *pad0[6];                                                                       /* 0 .. 5 */
  jclass (*FindClass)(JNIEnv *, const char *);                                         /* 6 */
  void *pad7[7];                                                                       /* 7 .. 13 */
  jint (*ThrowNew)(JNIEnv *, jclass, const char *);                                    /* 14 */
  void *pad15[2];                                                                      /* 15 .. 16 */
  void (*ExceptionClear)(JNIEnv *);                                                    /* 17 */
  void *pad18[149];                                                                    /* 18 .. 166 */
  jstring (*NewStringUTF)(JNIEnv *, const char *);                                     /* 167 */
  void *pad168[1];                                                                     /* 168 */
  const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);                     /* 169 */
  void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);                      /* 170 */
  jsize (*GetArrayLength)(JNIEnv *, jarray);                                           /* 171 */
  void *pad172[7];                                                                     /* 172 .. 178 */
  jintArray (*NewIntArray)(JNIEnv *, jsize);                                           /* 179 */
  jlongArray (*NewLongArray)(JNIEnv *, jsize);                                         /* 180 */
  void *pad181[22];                                                                    /* 181 .. 202 */
  void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);                /* 203 */
  void *pad204[7];                                                                     /* 204 .. 210 */
  void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);          /* 211 */
  void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);       /* 212 */
  void *pad213[9];                                                                     /* 213 .. 221 */
  void *(*GetPrimitiveArrayCritical)(JNIEnv *, jarray, jboolean *);                    /* 222 */
  void (*ReleasePrimitiveArrayCritical)(JNIEnv *, jarray, void *, jint);               /* 223 */
};
#endif
