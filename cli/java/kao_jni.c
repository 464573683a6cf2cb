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
        jintArray weights, jlong seed, jdouble timeLimit, jshortArray outAssignment, jlongArray outObjective,
        jlongArray outUpperBound) {
    (void)cls;
    jbyte *rack = malloc((size_t)nBrokers);
    (*env)->GetByteArrayRegion(env, rackOf, 0, nBrokers, rack);
    jint *P = malloc(4 * (size_t)nTopics), *RF = malloc(4 * (size_t)nTopics), *RC = malloc(4 * (size_t)nTopics), w[4];
    (*env)->GetIntArrayRegion(env, nPartitions, 0, nTopics, P);
    (*env)->GetIntArrayRegion(env, rf, 0, nTopics, RF);
    (*env)->GetIntArrayRegion(env, rfCur, 0, nTopics, RC);
    (*env)->GetIntArrayRegion(env, weights, 0, 4, w);
    const jsize curLen = (*env)->GetArrayLength(env, current), outLen = (*env)->GetArrayLength(env, outAssignment);
    jshort *cur = malloc(2 * (size_t)curLen + 2), *out = malloc(2 * (size_t)outLen + 2);
    (*env)->GetShortArrayRegion(env, current, 0, curLen, cur);
    kao_topic *t = calloc((size_t)nTopics, sizeof *t);
    kao_result *r = calloc((size_t)nTopics, sizeof *r);
    size_t co = 0, oo = 0;
    for (int i = 0; i < nTopics; ++i) {
        fill_topic(&t[i], nBrokers, nRacks, rack, P[i], RF[i], RC[i], cur + co, w);
        r[i].assignment = (uint16_t *)out + oo;
        co += (size_t)P[i] * (size_t)RC[i]; oo += (size_t)P[i] * (size_t)RF[i];
    }
    kao_opts o;
    memset(&o, 0, sizeof o);
    o.seed = (uint64_t)seed; o.time_limit_s = timeLimit; o.stop_at_bound = 1;
    const int rc = kao_solve(t, nTopics, &o, r);
    jintArray status = NULL;
    if (rc) throw_kao(env, rc);
    else {
        status = (*env)->NewIntArray(env, nTopics);
        for (int i = 0; i < nTopics; ++i) {
            const jint s = r[i].status; const jlong ob = r[i].objective, ub = r[i].upper_bound;
            (*env)->SetIntArrayRegion(env, status, i, 1, &s);
            (*env)->SetLongArrayRegion(env, outObjective, i, 1, &ob);
            (*env)->SetLongArrayRegion(env, outUpperBound, i, 1, &ub);
        }
        (*env)->SetShortArrayRegion(env, outAssignment, 0, outLen, out);
    }
    free(rack); free(P); free(RF); free(RC); free(cur); free(out); free(t); free(r);
    return status;
}

/* long[9] evaluate(...): {objective, viol[0..7]} of one complete assignment -- every row of the model (README.md:145-180) */
JNIEXPORT jlongArray JNICALL Java_io_sqooba_kao_Kao_evaluate(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights,
        jshortArray assignment) {
    (void)cls;
    jbyte *rack = malloc((size_t)nBrokers);
    (*env)->GetByteArrayRegion(env, rackOf, 0, nBrokers, rack);
    jint w[4];
    (*env)->GetIntArrayRegion(env, weights, 0, 4, w);
    const jsize curLen = (*env)->GetArrayLength(env, current), aLen = (*env)->GetArrayLength(env, assignment);
    jshort *cur = malloc(2 * (size_t)curLen + 2), *a = malloc(2 * (size_t)aLen + 2);
    (*env)->GetShortArrayRegion(env, current, 0, curLen, cur);
    (*env)->GetShortArrayRegion(env, assignment, 0, aLen, a);
    kao_topic t;
    fill_topic(&t, nBrokers, nRacks, rack, nPartitions, rf, rfCur, cur, w);
    int64_t obj = 0;
    int32_t viol[8];
    const int rc = kao_evaluate(&t, (const uint16_t *)a, &obj, viol);
    jlongArray res = NULL;
    if (rc) throw_kao(env, rc);
    else {
        jlong v[9];
        v[0] = obj;
        for (int i = 0; i < 8; ++i) v[1 + i] = viol[i];
        res = (*env)->NewLongArray(env, 9);
        (*env)->SetLongArrayRegion(env, res, 0, 9, v);
    }
    free(rack); free(cur); free(a);
    return res;
}

/* void canonicalize(..., short[] assignment): the tie-break that reproduces README.md:88 `[8,1]`, in place */
JNIEXPORT void JNICALL Java_io_sqooba_kao_Kao_canonicalize(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights,
        jshortArray assignment) {
    (void)cls;
    jbyte *rack = malloc((size_t)nBrokers);
    (*env)->GetByteArrayRegion(env, rackOf, 0, nBrokers, rack);
    jint w[4];
    (*env)->GetIntArrayRegion(env, weights, 0, 4, w);
    const jsize curLen = (*env)->GetArrayLength(env, current), aLen = (*env)->GetArrayLength(env, assignment);
    jshort *cur = malloc(2 * (size_t)curLen + 2), *a = malloc(2 * (size_t)aLen + 2);
    (*env)->GetShortArrayRegion(env, current, 0, curLen, cur);
    (*env)->GetShortArrayRegion(env, assignment, 0, aLen, a);
    kao_topic t;
    fill_topic(&t, nBrokers, nRacks, rack, nPartitions, rf, rfCur, cur, w);
    const int rc = kao_canonicalize(&t, (uint16_t *)a);
    if (rc) throw_kao(env, rc);
    else (*env)->SetShortArrayRegion(env, assignment, 0, aLen, a);
    free(rack); free(cur); free(a);
}

/* String checkInfeasible(...): "" or the counting argument that proves the topic infeasible (lp_solve: "This problem is infeasible") */
JNIEXPORT jstring JNICALL Java_io_sqooba_kao_Kao_checkInfeasible(JNIEnv *env, jclass cls, jint nBrokers, jint nRacks,
        jbyteArray rackOf, jint nPartitions, jint rf, jint rfCur, jshortArray current, jintArray weights) {
    (void)cls;
    jbyte *rack = malloc((size_t)nBrokers);
    (*env)->GetByteArrayRegion(env, rackOf, 0, nBrokers, rack);
    jint w[4];
    (*env)->GetIntArrayRegion(env, weights, 0, 4, w);
    const jsize curLen = (*env)->GetArrayLength(env, current);
    jshort *cur = malloc(2 * (size_t)curLen + 2);
    (*env)->GetShortArrayRegion(env, current, 0, curLen, cur);
    kao_topic t;
    fill_topic(&t, nBrokers, nRacks, rack, nPartitions, rf, rfCur, cur, w);
    char why[256] = "";
    const int rc = kao_check_infeasible(&t, why, (int)sizeof why);
    jstring res = NULL;
    if (rc < 0) throw_kao(env, rc);
    else res = (*env)->NewStringUTF(env, rc == 1 ? why : "");
    free(rack); free(cur);
    return res;
}
