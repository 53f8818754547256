/* TEST INFRASTRUCTURE ONLY — NOT the JDK's jni.h.
 * The build image has no JDK, so jni/wittgpu_jni.c cannot be compiled into a library here. This file declares, with the
 * signatures the JNI specification (chapter 4, "JNI Functions") gives them, exactly the types and the JNINativeInterface
 * members that wittgpu_jni.c uses, so that tests/test_jni_sources.py can run `gcc -fsyntax-only -Wall -Werror` over the
 * shim: it proves the shim is well-formed C against the JNI call shapes and against include/wittgpu.h — nothing more.
 * (In the C binding of JNI every reference type is the same pointer type, as below.) */
#ifndef WG_TEST_JNI_MIN_H
#define WG_TEST_JNI_MIN_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jbyteArray;
typedef jarray jfloatArray;
typedef jarray jdoubleArray;
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define JNI_ERR (-1)
#define JNI_ABORT 2
#define JNI_VERSION_1_8 0x00010008
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNIInvokeInterface_;
typedef const struct JNIInvokeInterface_* JavaVM;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv* env, const char* name);
  jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* msg);
  jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
  const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* isCopy);
  void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
  void (*SetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index, jobject val);
  jbyteArray (*NewByteArray)(JNIEnv* env, jsize len);
  jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* isCopy);
  jlong* (*GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* isCopy);
  jbyte* (*GetByteArrayElements)(JNIEnv* env, jbyteArray array, jboolean* isCopy);
  jfloat* (*GetFloatArrayElements)(JNIEnv* env, jfloatArray array, jboolean* isCopy);
  jdouble* (*GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* isCopy);
  void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (*ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
  void (*ReleaseByteArrayElements)(JNIEnv* env, jbyteArray array, jbyte* elems, jint mode);
  void (*ReleaseFloatArrayElements)(JNIEnv* env, jfloatArray array, jfloat* elems, jint mode);
  void (*ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
  void (*GetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, jint* buf);
  void (*GetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, jlong* buf);
  void (*GetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, jbyte* buf);
  void (*SetIntArrayRegion)(JNIEnv* env, jintArray array, jsize start, jsize len, const jint* buf);
  void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
  void (*SetByteArrayRegion)(JNIEnv* env, jbyteArray array, jsize start, jsize len, const jbyte* buf);
  void (*SetDoubleArrayRegion)(JNIEnv* env, jdoubleArray array, jsize start, jsize len, const jdouble* buf);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buf);
};
#endif
