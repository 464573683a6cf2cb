/* kao_jni.c -- JNI shim between io.sqooba.kao.Kao (cli/java/io/sqooba/kao/Kao.java) and the C ABI of libkao.so
 * (include/kao.h).  A pure forwarding layer: every decision stays behind the C ABI.
 *
 * What it stands in for: the reference hands its generated 0-1 model to lp_solve 5.5 "behind the scene"
 * (README.md:135-136) and reads one 0/1 value per variable back; a Java host of the reference would reach a native
 * solver exactly here.  Arrays are copied in and out with Get/Set<Type>ArrayRegion, so no JVM array stays pinned during a
 * seconds-long solve; errors surface as java.lang.RuntimeException built from the negative return code
 * (kao_strerror + kao_last_error) -- nothing throws or aborts across the ABI.
 *
 * Build (needs a JDK; none exists in the build image, where this file is compile-checked against tests/jni_stub/jni.h):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include kao_jni.c \
 *       -L../../kafka_assignment_optimizer_amd -lkao -o libkao_jni.so          (see cli/java/Makefile)
 * Java `short` is signed: broker indices >= 32768 and KAO_NONE (0xFFFF = -1) round-trip bit-exactly through (uint16_t). */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kao.h"

static void throw_kao(JNIEnv *env, int rc) {
    char msg[512];
    snprintf(msg, sizeof msg, "%s: %s", kao_strerror(rc), kao_last_error());
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/RuntimeException"), msg);
}
static void throw_arg(JNIEnv *env, const char *what) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), what);
}
static void throw_oom(JNIEnv *env) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "kao_jni: malloc failed");
}

/* The shim trusts nothing it is handed: sizes are checked against the Java arrays' real lengths BEFORE any native buffer is
 * sized from them (a short `current` / `outAssignment` was a heap overflow), every Get*ArrayRegion is followed by
 * ExceptionCheck, every malloc by a NULL test.  One topic's inputs, copied out of the JVM: */
typedef struct {
    jbyte *rack;
    jshort *cur;
    jint w[4];
} topic_in;

static void topic_in_free(topic_in *in) { free(in->rack); free(in->cur); in->rack = NULL; in->cur = NULL; }

/* rackOf[nBrokers], weights[4], current[>= needCur]; returns 0 with a Java exception pending on any failure */
static int topic_in_load(JNIEnv *env, topic_in *in, jint nBrokers, jint nRacks, jbyteArray rackOf, jintArray weights,
                         jshortArray current, size_t needCur) {
    in->rack = NULL; in->cur = NULL;
    if (nBrokers <= 0 || nBrokers > 65534 || nRacks <= 0 || nRacks > KAO_MAX_RACKS) { throw_arg(env, "nBrokers / nRacks out of range"); return 0; }
    if (!rackOf || !weights || !current) { throw_arg(env, "null array"); return 0; }
    if ((*env)->GetArrayLength(env, rackOf) < nBrokers) { throw_arg(env, "rackOf shorter than nBrokers"); return 0; }
    if ((*env)->GetArrayLength(env, weights) < 4) { throw_arg(env, "weights needs 4 entries {LL, LF, FL, FF}"); return 0; }
    if ((size_t)(*env)->GetArrayLength(env, current) < needCur) { throw_arg(env, "current shorter than sum(P * rfCur)"); return 0; }
    in->rack = malloc((size_t)nBrokers);
    in->cur = malloc(2 * needCur + 2);
    if (!in->rack || !in->cur) { topic_in_free(in); throw_oom(env); return 0; }
    (*env)->GetByteArrayRegion(env, rackOf, 0, nBrokers, in->rack);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, weights, 0, 4, in->w);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetShortArrayRegion(env, current, 0, (jsize)needCur, in->cur);
    if ((*env)->ExceptionCheck(env)) { topic_in_free(in); return 0; }
    return 1;
}
/* P, rf, rfCur of one topic: positive, rf within the kernels' limit, slot counts that fit a Java array */
static int dims_ok(jint P, jint rf, jint rfCur) {
    return P > 0 && rf > 0 && rf <= KAO_MAX_RF && rfCur > 0 && rfCur <= KAO_MAX_RF && (int64_t)P * KAO_MAX_RF < 0x7FFFFFFF;
}

static void fill_topic(kao_topic *t, jint nBrokers, jint nRacks, const jbyte *rack, jint P, jint rf, jint rfCur,
                       const jshort *cur, const jint w[4]) {
    memset(t, 0, sizeof *t);
    t->n_brokers = nBrokers; t->n_racks = nRacks; t->n_partitions = P; t->rf = rf; t->rf_cur = rfCur;
    t->rack_of = (const uint8_t *)rack; t->current = (const uint16_t *)cur;
    t->w[0][0] = w[0]; t->w[0][1] = w[1]; t->w[1][0] = w[2]; t->w[1][1] = w[3];
    t->rep_lo = t->rep_hi = t->lead_lo = t->lead_hi = -1;   /* bands derived as floor/ceil of the averages (README.md:158-180) */
    t->rack_lo = t->rack_hi = t->prack_lo = t->prack_hi = -1;
}

/* void init(int device) */
JNIEXPORT void JNICALL Java_io_sqooba_kao_Kao_init(JNIEnv *env, jclass cls, jint device) {
    (void)cls;
    const int rc = kao_init(device);
    if (rc) throw_kao(env, rc);
}

/* int[] solve(...): status per topic; assignment / objective / bound written to the out arrays (README.md:135-136) */
JNIEXPORT jintArray JNICALL Java_io_sqooba_kao_Kao_solve(JNIEnv *env, jclass cls, jint nTopics, jint nBrokers,
        jint nRacks, jbyteArray rackOf, jintArray nPartitions, jintArray rf, jintArray rfCur, jshortArray current,
        jintArray weights, jlong seed, jdouble timeLimit, jintArray devices, jshortArray outAssignment, jlongArray outObjective,
        jlongArray outUpperBound) {
    (void)cls;
    jint dev[64];
    jsize nDev = devices ? (*env)->GetArrayLength(env, devices) : 0;
    if (nDev > 64) { throw_arg(env, "more than 64 devices"); return NULL; }
    if (nDev > 0) {
        (*env)->GetIntArrayRegion(env, devices, 0, nDev, dev);
        if ((*env)->ExceptionCheck(env)) return NULL;
    }
    if (nTopics <= 0 || nTopics > (1 << 20)) { throw_arg(env, "nTopics out of range"); return NULL; }
    if (!nPartitions || !rf || !rfCur || !outAssignment || !outObjective || !outUpperBound) { throw_arg(env, "null array"); return NULL; }
    if ((*env)->GetArrayLength(env, nPartitions) < nTopics || (*env)->GetArrayLength(env, rf) < nTopics ||
        (*env)->GetArrayLength(env, rfCur) < nTopics || (*env)->GetArrayLength(env, outObjective) < nTopics ||
        (*env)->GetArrayLength(env, outUpperBound) < nTopics) { throw_arg(env, "per-topic array shorter than nTopics"); return NULL; }
    jint *P = malloc(4 * (size_t)nTopics), *RF = malloc(4 * (size_t)nTopics), *RC = malloc(4 * (size_t)nTopics);
    kao_topic *t = calloc((size_t)nTopics, sizeof *t);
    kao_result *r = calloc((size_t)nTopics, sizeof *r);
    jshort *out = NULL;
    topic_in in = {NULL, NULL, {0, 0, 0, 0}};
    jintArray status = NULL;
    if (!P || !RF || !RC || !t || !r) { throw_oom(env); goto done; }
    (*env)->GetIntArrayRegion(env, nPartitions, 0, nTopics, P);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, rf, 0, nTopics, RF);
    if (!(*env)->ExceptionCheck(env)) (*env)->GetIntArrayRegion(env, rfCur, 0, nTopics, RC);
    if ((*env)->ExceptionCheck(env)) goto done;
    size_t needCur = 0, needOut = 0;
    for (int i = 0; i < nTopics; ++i) {
        if (!dims_ok(P[i], RF[i], RC[i])) { throw_arg(env, "nPartitions / rf / rfCur out of range"); goto done; }
        needCur += (size_t)P[i] * (size_t)RC[i]; needOut += (size_t)P[i] * (size_t)RF[i];
        if (needCur > 0x7FFFFFFF || needOut > 0x7FFFFFFF) { throw_arg(env, "more slots than a Java array holds"); goto done; }
    }
    if ((size_t)(*env)->GetArrayLength(env, outAssignment) < needOut) { throw_arg(env, "outAssignment shorter than sum(P * rf)"); goto done; }
    if (!topic_in_load(env, &in, nBrokers, nRacks, rackOf, weights, current, needCur)) goto done;
    out = malloc(2 * needOut + 2);
    if (!out) { throw_oom(env); goto done; }
    memset(out, 0xFF, 2 * needOut + 2);   /* KAO_NONE: topics without a feasible answer hand back "no broker", never heap bytes */
    size_t co = 0, oo = 0;
    for (int i = 0; i < nTopics; ++i) {
        fill_topic(&t[i], nBrokers, nRacks, in.rack, P[i], RF[i], RC[i], in.cur + co, in.w);
        r[i].assignment = (uint16_t *)out + oo;
        co += (size_t)P[i] * (size_t)RC[i]; oo += (size_t)P[i] * (size_t)RF[i];
    }
    kao_opts o;
    memset(&o, 0, sizeof o);
    o.seed = (uint64_t)seed; o.time_limit_s = timeLimit; o.stop_at_bound = 1;
    const int rc = nDev >= 1 ? kao_solve_multi(t, nTopics, (const int32_t *)dev, (int32_t)nDev, &o, r) : kao_solve(t, nTopics, &o, r);
    if (rc) { throw_kao(env, rc); goto done; }
    status = (*env)->NewIntArray(env, nTopics);
    if (!status) goto done;   /* OutOfMemoryError pending */
    for (int i = 0; i < nTopics && !(*env)->ExceptionCheck(env); ++i) {
        const jint s = r[i].status; const jlong ob = r[i].objective, ub = r[i].upper_bound;
        (*env)->SetIntArrayRegion(env, status, i, 1, &s);
        (*env)->SetLongArrayRegion(env, outObjective, i, 1, &ob);
        (*env)->SetLongArrayRegion(env, outUpperBound, i, 1, &ub);
    }
    if (!(*env)->ExceptionCheck(env)) (*env)->SetShortArrayRegion(env, outAssignment, 0, (jsize)needOut, out);
    if ((*env)->ExceptionCheck(env)) status = NULL;
done:
    topic_in_free(&in);
    free(P); free(RF); free(RC); free(out); free(t); free(r);
    return status;
}

/* one topic + one complete assignment [P*rf], copied out of the JVM; returns NULL with an exception pending on failure */
static jshort *assignment_load(JNIEnv *env, jshortArray assignment, jint P, jint rf) {
    if (!assignment) { throw_arg(env, "null assignment"); return NULL; }
    const size_t need = (size_t)P * (size_t)rf;
    if ((size_t)(*env)->GetArrayLength(env, assignment) != need) { throw_arg(env, "assignment length != nPartitions * rf"); return NULL; }
    jshort *a = malloc(2 * need + 2);
    if (!a) { throw_oom(env); return NULL; }
    (*env)->GetShortArrayRegion(env, assignment, 0, (jsize)need, a);
    if ((*env)->ExceptionCheck(env)) { free(a); return NULL; }
    return a;
}

/* long[9] evaluate(...): {objective, viol[0..7]} of one complete assignment -- every row of the model (README.md:145-180) */
JNIEXPORT jlongArray JNICALL Java_io_sqooba_kao_Kao_evaluate(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights,
        jshortArray assignment) {
    (void)cls;
    if (!dims_ok(nPartitions, rf, rfCur)) { throw_arg(env, "nPartitions / rf / rfCur out of range"); return NULL; }
    topic_in in;
    if (!topic_in_load(env, &in, nBrokers, nRacks, rackOf, weights, current, (size_t)nPartitions * (size_t)rfCur)) return NULL;
    jshort *a = assignment_load(env, assignment, nPartitions, rf);
    jlongArray res = NULL;
    if (a) {
        kao_topic t;
        fill_topic(&t, nBrokers, nRacks, in.rack, nPartitions, rf, rfCur, in.cur, in.w);
        int64_t obj = 0;
        int32_t viol[8];
        const int rc = kao_evaluate(&t, (const uint16_t *)a, &obj, viol);
        if (rc) throw_kao(env, rc);
        else {
            jlong v[9];
            v[0] = obj;
            for (int i = 0; i < 8; ++i) v[1 + i] = viol[i];
            res = (*env)->NewLongArray(env, 9);
            if (res) (*env)->SetLongArrayRegion(env, res, 0, 9, v);
        }
    }
    topic_in_free(&in); free(a);
    return res;
}

/* void canonicalize(..., short[] assignment): the tie-break that reproduces README.md:88 `[8,1]`, in place */
JNIEXPORT void JNICALL Java_io_sqooba_kao_Kao_canonicalize(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights,
        jshortArray assignment) {
    (void)cls;
    if (!dims_ok(nPartitions, rf, rfCur)) { throw_arg(env, "nPartitions / rf / rfCur out of range"); return; }
    topic_in in;
    if (!topic_in_load(env, &in, nBrokers, nRacks, rackOf, weights, current, (size_t)nPartitions * (size_t)rfCur)) return;
    jshort *a = assignment_load(env, assignment, nPartitions, rf);
    if (a) {
        kao_topic t;
        fill_topic(&t, nBrokers, nRacks, in.rack, nPartitions, rf, rfCur, in.cur, in.w);
        const int rc = kao_canonicalize(&t, (uint16_t *)a);
        if (rc) throw_kao(env, rc);
        else (*env)->SetShortArrayRegion(env, assignment, 0, (jsize)((size_t)nPartitions * (size_t)rf), a);
    }
    topic_in_free(&in); free(a);
}

/* String checkInfeasible(...): "" or the counting argument that proves the topic infeasible (lp_solve: "This problem is infeasible") */
JNIEXPORT jstring JNICALL Java_io_sqooba_kao_Kao_checkInfeasible(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights) {
    (void)cls;
    if (!dims_ok(nPartitions, rf, rfCur)) { throw_arg(env, "nPartitions / rf / rfCur out of range"); return NULL; }
    topic_in in;
    if (!topic_in_load(env, &in, nBrokers, nRacks, rackOf, weights, current, (size_t)nPartitions * (size_t)rfCur)) return NULL;
    kao_topic t;
    fill_topic(&t, nBrokers, nRacks, in.rack, nPartitions, rf, rfCur, in.cur, in.w);
    char why[256] = "";
    const int rc = kao_check_infeasible(&t, why, (int)sizeof why);
    jstring res = NULL;
    if (rc < 0) throw_kao(env, rc);
    else res = (*env)->NewStringUTF(env, rc == 1 ? why : "");
    topic_in_free(&in);
    return res;
}
