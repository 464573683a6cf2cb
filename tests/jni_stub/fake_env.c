/* A fake JNIEnv, just enough to RUN cli/java/kao_jni.c without a JVM (there is no JDK in the build image): Java arrays are
 * {length, element size, data} blocks, exceptions are recorded (class name + message) instead of thrown, strings are C
 * strings.  Test scaffolding only (tests/test_host.py, tests/test_gpu_parity.py); the member signatures are the real JNI ones
 * declared in tests/jni_stub/jni.h.
 *
 *   jni_harness validate          -- CPU: every argument check of the shim throws IllegalArgumentException, nothing crashes
 *   jni_harness solve             -- GPU: the README example (README.md:52-63) through Kao.solve / evaluate / canonicalize
 */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { jsize len; size_t esz; unsigned char *data; } fake_array;
static char g_exc_class[128], g_exc_msg[600];
static int g_exc = 0;

static jclass f_FindClass(JNIEnv *e, const char *name) { (void)e; return (jclass)name; }
static jint f_ThrowNew(JNIEnv *e, jclass c, const char *msg) {
    (void)e;
    if (!g_exc) { snprintf(g_exc_class, sizeof g_exc_class, "%s", (const char *)c); snprintf(g_exc_msg, sizeof g_exc_msg, "%s", msg); }
    g_exc = 1;
    return 0;
}
static jboolean f_ExceptionCheck(JNIEnv *e) { (void)e; return (jboolean)g_exc; }
static jsize f_GetArrayLength(JNIEnv *e, jarray a) { (void)e; return ((fake_array *)a)->len; }
static fake_array *new_array(jsize n, size_t esz) {
    fake_array *a = calloc(1, sizeof *a);
    a->len = n; a->esz = esz; a->data = calloc((size_t)n + 1, esz);
    return a;
}
static jintArray f_NewIntArray(JNIEnv *e, jsize n) { (void)e; return new_array(n, 4); }
static jlongArray f_NewLongArray(JNIEnv *e, jsize n) { (void)e; return new_array(n, 8); }
static jstring f_NewStringUTF(JNIEnv *e, const char *s) { (void)e; char *c = malloc(strlen(s) + 1); strcpy(c, s); return (jstring)c; }
static void region(jarray arr, jsize start, jsize n, void *buf, int get) {
    fake_array *a = arr;
    if (start < 0 || n < 0 || start + n > a->len) {   /* the JVM would throw ArrayIndexOutOfBoundsException */
        if (!g_exc) { snprintf(g_exc_class, sizeof g_exc_class, "java/lang/ArrayIndexOutOfBoundsException"); g_exc_msg[0] = 0; }
        g_exc = 1;
        return;
    }
    if (get) memcpy(buf, a->data + (size_t)start * a->esz, (size_t)n * a->esz);
    else memcpy(a->data + (size_t)start * a->esz, buf, (size_t)n * a->esz);
}
static void f_GetByte(JNIEnv *e, jbyteArray a, jsize s, jsize n, jbyte *b) { (void)e; region(a, s, n, b, 1); }
static void f_GetShort(JNIEnv *e, jshortArray a, jsize s, jsize n, jshort *b) { (void)e; region(a, s, n, b, 1); }
static void f_GetInt(JNIEnv *e, jintArray a, jsize s, jsize n, jint *b) { (void)e; region(a, s, n, b, 1); }
static void f_SetShort(JNIEnv *e, jshortArray a, jsize s, jsize n, const jshort *b) { (void)e; region(a, s, n, (void *)b, 0); }
static void f_SetInt(JNIEnv *e, jintArray a, jsize s, jsize n, const jint *b) { (void)e; region(a, s, n, (void *)b, 0); }
static void f_SetLong(JNIEnv *e, jlongArray a, jsize s, jsize n, const jlong *b) { (void)e; region(a, s, n, (void *)b, 0); }

static const struct JNINativeInterface_ g_iface = {
    f_FindClass, f_ThrowNew, f_ExceptionCheck, f_GetArrayLength, f_NewIntArray, f_NewLongArray, f_NewStringUTF,
    f_GetByte, f_GetShort, f_GetInt, f_SetShort, f_SetInt, f_SetLong};
static JNIEnv g_env = &g_iface;

/* the shim's entry points (cli/java/kao_jni.c is compiled into this harness) */
void Java_io_sqooba_kao_Kao_init(JNIEnv *, jclass, jint);
jintArray Java_io_sqooba_kao_Kao_solve(JNIEnv *, jclass, jint, jint, jint, jbyteArray, jintArray, jintArray, jintArray, jshortArray, jintArray,
                                       jlong, jdouble, jintArray, jshortArray, jlongArray, jlongArray);
jlongArray Java_io_sqooba_kao_Kao_evaluate(JNIEnv *, jclass, jint, jint, jbyteArray, jint, jint, jint, jshortArray, jintArray, jshortArray);
void Java_io_sqooba_kao_Kao_canonicalize(JNIEnv *, jclass, jint, jint, jbyteArray, jint, jint, jint, jshortArray, jintArray, jshortArray);
jstring Java_io_sqooba_kao_Kao_checkInfeasible(JNIEnv *, jclass, jint, jint, jbyteArray, jint, jint, jint, jshortArray, jintArray);

static fake_array *bytes(const jbyte *v, jsize n) { fake_array *a = new_array(n, 1); memcpy(a->data, v, (size_t)n); return a; }
static fake_array *shorts(const jshort *v, jsize n) { fake_array *a = new_array(n, 2); memcpy(a->data, v, (size_t)n * 2); return a; }
static fake_array *ints(const jint *v, jsize n) { fake_array *a = new_array(n, 4); memcpy(a->data, v, (size_t)n * 4); return a; }
static int expect_exc(const char *cls, const char *what) {
    const int ok = g_exc && strcmp(g_exc_class, cls) == 0;
    if (!ok) fprintf(stderr, "jni_harness: %s: expected %s, got %s (%s)\n", what, cls, g_exc ? g_exc_class : "no exception", g_exc_msg);
    g_exc = 0;
    return ok;
}

/* README.md:27-31, 43-63: 20 brokers, odd ids in AZ b; target list 0..18; 10 partitions, RF 2 */
static const jshort kCur[20] = {7, 18, 8, -1 /* broker 19: not in the target list */, 9, 10, 0, 11, 1, 12, 2, 13, 3, 14, 4, 15, 5, 16, 6, 17};

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "validate";
    jbyte rack[19];
    for (int b = 0; b < 19; ++b) rack[b] = (jbyte)(b & 1);
    const jint w[4] = {4, 1, 2, 2}, P[1] = {10}, RF[1] = {2};
    fake_array *aRack = bytes(rack, 19), *aW = ints(w, 4), *aP = ints(P, 1), *aRF = ints(RF, 1), *aCur = shorts(kCur, 20);
    fake_array *out = new_array(20, 2), *obj = new_array(1, 8), *ub = new_array(1, 8);
    if (strcmp(mode, "validate") == 0) {
        int ok = 1;
        /* every call below must throw IllegalArgumentException BEFORE anything reaches the C ABI (no GPU is needed) */
        fake_array *shortCur = shorts(kCur, 12), *shortOut = new_array(7, 2), *w3 = ints(w, 3), *rack5 = bytes(rack, 5);
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, aRack, aP, aRF, aRF, shortCur, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "short current");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, aRack, aP, aRF, aRF, aCur, aW, 1, 1.0, NULL, shortOut, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "short outAssignment");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, -3, 19, 2, aRack, aP, aRF, aRF, aCur, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "negative nTopics");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, -19, 2, aRack, aP, aRF, aRF, aCur, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "negative nBrokers");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, rack5, aP, aRF, aRF, aCur, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "short rackOf");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, aRack, aP, aRF, aRF, aCur, w3, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "three weights");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 2, 19, 2, aRack, aP, aRF, aRF, aCur, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "per-topic arrays shorter than nTopics");
        Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, aRack, aP, aRF, aRF, NULL, aW, 1, 1.0, NULL, out, obj, ub);
        ok &= expect_exc("java/lang/IllegalArgumentException", "null current");
        Java_io_sqooba_kao_Kao_evaluate(&g_env, NULL, 19, 2, aRack, 10, 2, 2, aCur, aW, shortOut);
        ok &= expect_exc("java/lang/IllegalArgumentException", "evaluate: assignment length");
        Java_io_sqooba_kao_Kao_canonicalize(&g_env, NULL, 19, 2, aRack, 10, 9, 2, aCur, aW, out);
        ok &= expect_exc("java/lang/IllegalArgumentException", "canonicalize: rf 9");
        Java_io_sqooba_kao_Kao_checkInfeasible(&g_env, NULL, 19, 2, aRack, 0, 2, 2, aCur, aW);
        ok &= expect_exc("java/lang/IllegalArgumentException", "checkInfeasible: no partitions");
        /* host-only entry point with good arguments: the README topic is not provably infeasible */
        jstring why = Java_io_sqooba_kao_Kao_checkInfeasible(&g_env, NULL, 19, 2, aRack, 10, 2, 2, aCur, aW);
        ok &= !g_exc && why && ((const char *)why)[0] == 0;
        puts(ok ? "jni_harness validate: ok" : "jni_harness validate: FAILED");
        return ok ? 0 : 1;
    }
    /* ---- GPU: README.md:52-63 in, README.md:85-91 out (only partition 1 changes, to [8,1]) ---- */
    Java_io_sqooba_kao_Kao_init(&g_env, NULL, 0);
    if (g_exc) { fprintf(stderr, "jni_harness: init threw %s: %s\n", g_exc_class, g_exc_msg); return 1; }
    const jint two[2] = {0, 0};
    fake_array *devs = strcmp(mode, "solve-multi") == 0 ? ints(two, 2) : NULL;   /* two logical shards on device 0 */
    fake_array *status = Java_io_sqooba_kao_Kao_solve(&g_env, NULL, 1, 19, 2, aRack, aP, aRF, aRF, aCur, aW, 1, 10.0, devs, out, obj, ub);
    if (g_exc || !status) { fprintf(stderr, "jni_harness: solve threw %s: %s\n", g_exc_class, g_exc_msg); return 1; }
    const jint st = *(jint *)status->data;
    const jlong o = *(jlong *)obj->data, u = *(jlong *)ub->data;
    Java_io_sqooba_kao_Kao_canonicalize(&g_env, NULL, 19, 2, aRack, 10, 2, 2, aCur, aW, out);
    if (g_exc) { fprintf(stderr, "jni_harness: canonicalize threw %s: %s\n", g_exc_class, g_exc_msg); return 1; }
    fake_array *ev = Java_io_sqooba_kao_Kao_evaluate(&g_env, NULL, 19, 2, aRack, 10, 2, 2, aCur, aW, out);
    if (g_exc || !ev) { fprintf(stderr, "jni_harness: evaluate threw %s: %s\n", g_exc_class, g_exc_msg); return 1; }
    const jlong *e9 = (const jlong *)ev->data;
    const jshort *a = (const jshort *)out->data;
    printf("status=%d objective=%lld bound=%lld eval_objective=%lld eval_violation=%lld p1=[%d,%d]\n", (int)st, (long long)o, (long long)u,
           (long long)e9[0], (long long)e9[1], (int)a[2], (int)a[3]);
    int moved = 0;
    for (int i = 0; i < 20; ++i) moved += a[i] != kCur[i];
    const int ok = st == 0 && o == 58 && u == 58 && e9[0] == 58 && e9[1] == 0 && a[2] == 8 && a[3] == 1 && moved == 1;
    puts(ok ? "jni_harness solve: ok" : "jni_harness solve: FAILED");
    return ok ? 0 : 1;
}
