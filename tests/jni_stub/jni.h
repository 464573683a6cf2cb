/* Minimal stand-in for <jni.h>: just enough declarations to COMPILE-CHECK the JNI shim cli/java/kao_jni.c
 * (tests/test_host.py::test_jni_shim_compiles).  No JDK exists in the build image; this is test
 * scaffolding, not a JNI implementation -- types and the few JNIEnv members the shim uses, with their real signatures. */
#ifndef KAO_TEST_JNI_STUB_H
#define KAO_TEST_JNI_STUB_H
#include <stdint.h>
#include <stdio.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef int16_t jshort;
typedef double jdouble;
typedef jint jsize;
typedef uint8_t jboolean;
typedef void *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray, jlongArray, jbyteArray, jshortArray;
typedef jobject jstring;
#define JNIEXPORT
#define JNICALL
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jboolean (*ExceptionCheck)(JNIEnv *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jintArray (*NewIntArray)(JNIEnv *, jsize);
    jlongArray (*NewLongArray)(JNIEnv *, jsize);
    jstring (*NewStringUTF)(JNIEnv *, const char *);
    void (*GetByteArrayRegion)(JNIEnv *, jbyteArray, jsize, jsize, jbyte *);
    void (*GetShortArrayRegion)(JNIEnv *, jshortArray, jsize, jsize, jshort *);
    void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);
    void (*SetShortArrayRegion)(JNIEnv *, jshortArray, jsize, jsize, const jshort *);
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
};
#endif
