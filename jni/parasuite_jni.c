/*
 * parasuite_jni.c -- JNI stub between mapping.NativePARAsuiteMapping and the C ABI of libparasuite_hip.so
 * (include/parasuite_hip.h).  NOT compiled in this repository's image (no JDK / jni.h); see jni/README.md.
 * Replaces the three Mapping.executeCommand spawns of /root/reference/src/src/mapping/PARAsuiteMapping.java:45-92.
 */
#include <jni.h>
#include "parasuite_hip.h"

static const char *str(JNIEnv *e, jstring s) { return s ? (*e)->GetStringUTFChars(e, s, 0) : NULL; }
static void rel(JNIEnv *e, jstring s, const char *c) { if (s) (*e)->ReleaseStringUTFChars(e, s, c); }

JNIEXPORT jint JNICALL Java_mapping_NativePARAsuiteMapping_nativeIndex(JNIEnv *e, jclass k, jstring ref)
{
    const char *r = str(e, ref);
    jint rc = ps_index(r);
    rel(e, ref, r);
    return rc;
}

JNIEXPORT jint JNICALL Java_mapping_NativePARAsuiteMapping_nativeMap(JNIEnv *e, jclass k, jint threads, jstring mm,
        jstring ep, jstring ip, jstring ref, jstring in, jstring out)
{
    const char *a = str(e, mm), *b = str(e, ep), *c = str(e, ip), *d = str(e, ref), *f = str(e, in), *g = str(e, out);
    jint rc = ps_map(threads, a, b, c, d, f, g);      /* b == NULL (no profile) => stock `aln -n mm` */
    rel(e, mm, a); rel(e, ep, b); rel(e, ip, c); rel(e, ref, d); rel(e, in, f); rel(e, out, g);
    return rc;
}

/* the first pass together with the ErrorProfiling step behind it (Main.java:288-334): ps_map_profiled */
JNIEXPORT jint JNICALL Java_mapping_NativePARAsuiteMapping_nativeMapProfiled(JNIEnv *e, jclass k, jint threads, jstring mm,
        jstring ref, jstring in, jstring out, jint minMapq, jint maxReadLength, jstring profilePrefix)
{
    const char *a = str(e, mm), *d = str(e, ref), *f = str(e, in), *g = str(e, out), *p = str(e, profilePrefix);
    jint rc = ps_map_profiled(threads, a, NULL, NULL, d, f, g, minMapq, maxReadLength, p);
    rel(e, mm, a); rel(e, ref, d); rel(e, in, f); rel(e, out, g); rel(e, profilePrefix, p);
    return rc;
}

/* ErrorProfiling.inferErrorProfile on an existing SAM / BAM file (Main.java:327-334): ps_error_profile */
JNIEXPORT jint JNICALL Java_mapping_NativePARAsuiteMapping_nativeErrorProfile(JNIEnv *e, jclass k, jstring mapping, jstring ref,
        jint maxReadLength, jstring prefix)
{
    const char *a = str(e, mapping), *b = str(e, ref), *c = str(e, prefix);
    jint rc = ps_error_profile(a, b, maxReadLength, c);
    rel(e, mapping, a); rel(e, ref, b); rel(e, prefix, c);
    return rc;
}

JNIEXPORT jstring JNICALL Java_mapping_NativePARAsuiteMapping_nativeLastError(JNIEnv *e, jclass k)
{
    return (*e)->NewStringUTF(e, ps_last_error());
}
