// io/sqooba/kao/Kao.java -- Java face of libkao.so (the MI355X solver), bound through cli/java/kao_jni.c.
// What it replaces: the reference solves its generated 0-1 model with lp_solve 5.5 "behind the scene"
// (README.md:135-136); a Java host reaches the native solver here instead.  Index conventions are those of
// include/kao.h: dense broker index = position in the --broker-list (README.md:48), slot 0 of a partition = its
// preferred leader (README.md:52-63), NONE = broker outside the target list.
package io.sqooba.kao;

public final class Kao {
    static { System.loadLibrary("kao_jni"); }          // libkao_jni.so, which links libkao.so

    public static final int NONE = 0xFFFF;
    public static final int OPTIMAL_PROVEN = 0, FEASIBLE_BOUND_GAP = 1, NO_FEASIBLE = 2, TIME_LIMIT = 3, INFEASIBLE_PROVEN = 4;

    private Kao() {}

    /** Selects the HIP device of this process (kao_init). */
    public static native void init(int device);

    /** Solves nTopics topics sharing one broker set (kao_solve; kao_solve_multi when `devices` lists more than one HIP device
     *  ordinal -- topics sharded over them, or replicated with an RCCL min-allreduce when there are fewer topics than
     *  devices; null or one entry = the device selected by init()).  Arrays are flattened per topic, in order.
     *  @return status per topic; assignment (dense broker index, leader first) is written to outAssignment,
     *          objective and its certified upper bound to outObjective / outUpperBound. */
    public static native int[] solve(int nTopics, int nBrokers, int nRacks, byte[] rackOf,
                                     int[] nPartitions, int[] rf, int[] rfCur,
                                     short[] current,          // concatenated [P*rfCur] per topic
                                     int[] weights,            // {LL, LF, FL, FF}  (README.md:145-146)
                                     long seed, double timeLimitSeconds,
                                     int[] devices,            // null, or the HIP device ordinals to shard over
                                     short[] outAssignment,    // concatenated [P*rf] per topic
                                     long[] outObjective, long[] outUpperBound);

    /** Full evaluation of one complete assignment (kao_evaluate): {objective, viol0 (total), viol1..7 (C1..C7)}. */
    public static native long[] evaluate(int nBrokers, int nRacks, byte[] rackOf, int nPartitions,
                                         int rf, int rfCur, short[] current, int[] weights, short[] assignment);

    /** Canonical tie-break among equal-objective optima (kao_canonicalize), in place: reproduces README.md:88. */
    public static native void canonicalize(int nBrokers, int nRacks, byte[] rackOf, int nPartitions,
                                           int rf, int rfCur, short[] current, int[] weights, short[] assignment);

    /** "" or the counting argument that proves the topic infeasible (kao_check_infeasible). */
    public static native String checkInfeasible(int nBrokers, int nRacks, byte[] rackOf, int nPartitions,
                                                int rf, int rfCur, short[] current, int[] weights);
}
