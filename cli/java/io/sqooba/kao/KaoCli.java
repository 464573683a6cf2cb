// io/sqooba/kao/KaoCli.java -- the thin Java CLI: same reassignment JSON in and out as the reference and as
// kafka-reassign-partitions (README.md:52-63 in, README.md:67-78 out), same flags as cli/kao-cli (the C++ CLI over the
// same C ABI), solving on the GPU through io.sqooba.kao.Kao (JNI -> libkao.so).  No dependencies beyond the JDK: the
// reassignment format is small enough for the hand-written JSON reader below.
//
//   java -Djava.library.path=cli/java -cp cli/java io.sqooba.kao.KaoCli --current current.json \
//        --broker-list 0,1,...,18 --racks racks.json [--rf N] [--weights LL,LF,FL,FF] [--seed S] [--time-limit SEC]
//        [--device D] [--gpus N | d0,d1,...] [--no-canonical] [--out FILE] [--report] [--require-optimal]
//
// Exit status as kao-cli: 0 = every topic solved (a warning on stderr marks a plan that is feasible but not PROVEN
// optimal), 3 = a topic is infeasible / no feasible plan found, 4 = --require-optimal and a plan was withheld, 1 = error,
// 2 = usage.  (Model export as lp_solve LP text, --emit-lp, is host-only and lives in cli/kao-cli.)
package io.sqooba.kao;

import java.io.IOException;
import java.io.PrintStream;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.ArrayList;
import java.util.LinkedHashMap;
import java.util.List;
import java.util.Map;
import java.util.TreeMap;
import java.util.TreeSet;

public final class KaoCli {
    private KaoCli() {}

    // ---- minimal JSON reader: objects -> LinkedHashMap, arrays -> ArrayList, numbers -> Double, strings -> String
    private static final class Json {
        private final String s;
        private int i = 0;
        Json(String s) { this.s = s; }
        private void ws() { while (i < s.length() && Character.isWhitespace(s.charAt(i))) i++; }
        Object parse() {
            ws();
            if (i >= s.length()) throw new IllegalArgumentException("unexpected end of JSON");
            char c = s.charAt(i);
            if (c == '{') {
                Map<String, Object> m = new LinkedHashMap<>();
                i++; ws();
                if (s.charAt(i) == '}') { i++; return m; }
                while (true) {
                    ws();
                    String k = (String) parse();
                    ws();
                    if (s.charAt(i++) != ':') throw new IllegalArgumentException("':' expected at " + i);
                    m.put(k, parse());
                    ws();
                    char d = s.charAt(i++);
                    if (d == '}') return m;
                    if (d != ',') throw new IllegalArgumentException("',' or '}' expected at " + i);
                }
            }
            if (c == '[') {
                List<Object> a = new ArrayList<>();
                i++; ws();
                if (s.charAt(i) == ']') { i++; return a; }
                while (true) {
                    a.add(parse());
                    ws();
                    char d = s.charAt(i++);
                    if (d == ']') return a;
                    if (d != ',') throw new IllegalArgumentException("',' or ']' expected at " + i);
                }
            }
            if (c == '"') {
                StringBuilder b = new StringBuilder();
                i++;
                while (s.charAt(i) != '"') {
                    char d = s.charAt(i++);
                    if (d == '\\') {
                        char e = s.charAt(i++);
                        switch (e) {
                            case 'n': b.append('\n'); break;
                            case 't': b.append('\t'); break;
                            case 'u': b.append((char) Integer.parseInt(s.substring(i, i + 4), 16)); i += 4; break;
                            default: b.append(e);
                        }
                    } else b.append(d);
                }
                i++;
                return b.toString();
            }
            int j = i;
            while (j < s.length() && "+-0123456789.eE".indexOf(s.charAt(j)) >= 0) j++;
            if (j == i) {  // true / false / null
                for (String lit : new String[] {"true", "false", "null"})
                    if (s.startsWith(lit, i)) { i += lit.length(); return lit.equals("null") ? null : Boolean.valueOf(lit); }
                throw new IllegalArgumentException("bad JSON at " + i);
            }
            double v = Double.parseDouble(s.substring(i, j));
            i = j;
            return v;
        }
    }

    private static void usage(String msg) {
        if (msg != null) System.err.println("KaoCli: " + msg);
        System.err.println("usage: KaoCli --current <reassignment.json|-> --broker-list <id,id,...> --racks <racks.json | id:rack,...>\n"
            + "              [--rf N] [--weights LL,LF,FL,FF] [--seed S] [--time-limit SEC] [--device D] [--gpus N]\n"
            + "              [--no-canonical] [--out <file>] [--report] [--require-optimal]");
        System.exit(2);
    }

    private static String slurp(String path) throws IOException {
        if (path.equals("-")) return new String(System.in.readAllBytes(), StandardCharsets.UTF_8);
        return new String(Files.readAllBytes(Paths.get(path)), StandardCharsets.UTF_8);
    }

    private static final String[] STATUS = {"OPTIMAL_PROVEN", "FEASIBLE_BOUND_GAP", "NO_FEASIBLE", "TIME_LIMIT", "INFEASIBLE_PROVEN"};

    @SuppressWarnings("unchecked")
    public static void main(String[] argv) {
        String curPath = null, brokersCsv = null, racksArg = null, outPath = null;
        int rfOverride = 0, device = 0, gpus = 1;
        int[] gpuList = null;
        int[] w = {4, 1, 2, 2};
        long seed = 1;
        double timeLimit = 10.0;
        boolean canonical = true, report = false, requireOptimal = false;
        for (int i = 0; i < argv.length; ++i) {
            String a = argv[i];
            boolean hasNext = i + 1 < argv.length;
            switch (a) {
                case "--current": if (!hasNext) usage(a + " needs a value"); curPath = argv[++i]; break;
                case "--broker-list": if (!hasNext) usage(a + " needs a value"); brokersCsv = argv[++i]; break;
                case "--racks": if (!hasNext) usage(a + " needs a value"); racksArg = argv[++i]; break;
                case "--rf": if (!hasNext) usage(a + " needs a value"); rfOverride = Integer.parseInt(argv[++i]); break;
                case "--weights": {
                    if (!hasNext) usage(a + " needs a value");
                    String[] p = argv[++i].split(",");
                    if (p.length != 4) usage("--weights needs LL,LF,FL,FF");
                    for (int k = 0; k < 4; ++k) w[k] = Integer.parseInt(p[k].trim());
                    break;
                }
                case "--seed": if (!hasNext) usage(a + " needs a value"); seed = Long.decode(argv[++i]); break;
                case "--time-limit": if (!hasNext) usage(a + " needs a value"); timeLimit = Double.parseDouble(argv[++i]); break;
                case "--device": if (!hasNext) usage(a + " needs a value"); device = Integer.parseInt(argv[++i]); break;
                case "--gpus": {   // N = devices device .. device+N-1; a,b,... = exactly these ordinals (as cli/kao-cli)
                    if (!hasNext) usage(a + " needs a value");
                    String v = argv[++i];
                    if (v.indexOf(',') < 0) gpus = Integer.parseInt(v.trim());
                    else {
                        String[] p = v.split(",");
                        gpuList = new int[p.length];
                        for (int k = 0; k < p.length; ++k) gpuList[k] = Integer.parseInt(p[k].trim());
                        gpus = p.length;
                    }
                    if (gpus < 1) usage("--gpus needs a count >= 1 or a device list");
                    break;
                }
                case "--no-canonical": canonical = false; break;
                case "--out": if (!hasNext) usage(a + " needs a value"); outPath = argv[++i]; break;
                case "--report": report = true; break;
                case "--require-optimal": requireOptimal = true; break;
                case "-h": case "--help": usage(null); break;
                default: usage("unknown flag " + a);
            }
        }
        if (curPath == null || brokersCsv == null || racksArg == null) usage("--current, --broker-list and --racks are required");
        try {
            // ---- target brokers (README.md:48) and racks (README.md:27-29) ----
            List<Integer> brokers = new ArrayList<>();
            for (String t : brokersCsv.split(",")) if (!t.trim().isEmpty()) brokers.add(Integer.parseInt(t.trim()));
            if (brokers.isEmpty()) throw new IllegalArgumentException("empty broker list");
            Map<Integer, Integer> dense = new TreeMap<>();
            for (int i = 0; i < brokers.size(); ++i)
                if (dense.put(brokers.get(i), i) != null) throw new IllegalArgumentException("duplicate id in broker list");
            Map<Integer, String> rackName = new TreeMap<>();
            if (racksArg.indexOf(':') >= 0 && racksArg.indexOf('{') < 0 && !racksArg.endsWith(".json")) {
                for (String t : racksArg.split(",")) {
                    String[] kv = t.split(":");
                    if (kv.length != 2) throw new IllegalArgumentException("bad --racks entry " + t);
                    rackName.put(Integer.parseInt(kv[0].trim()), kv[1]);
                }
            } else {
                Object doc = new Json(slurp(racksArg)).parse();
                if (!(doc instanceof Map)) throw new IllegalArgumentException("racks file must be a JSON object {\"<brokerId>\": \"<rack>\"}");
                for (Map.Entry<String, Object> e : ((Map<String, Object>) doc).entrySet()) {
                    Object v = e.getValue();
                    rackName.put(Integer.parseInt(e.getKey().trim()), v instanceof Double ? Long.toString(((Double) v).longValue()) : String.valueOf(v));
                }
            }
            TreeSet<String> names = new TreeSet<>();
            for (int b : brokers) {
                if (!rackName.containsKey(b)) throw new IllegalArgumentException("no rack given for broker " + b);
                names.add(rackName.get(b));
            }
            List<String> rackList = new ArrayList<>(names);
            if (rackList.size() > 255) throw new IllegalArgumentException("more than 255 racks");
            byte[] rackOf = new byte[brokers.size()];
            for (int i = 0; i < brokers.size(); ++i) rackOf[i] = (byte) rackList.indexOf(rackName.get(brokers.get(i)));

            // ---- current assignment (README.md:52-63): topic -> partition -> replicas ----
            Object cdoc = new Json(slurp(curPath)).parse();
            Object parts = cdoc instanceof Map ? ((Map<String, Object>) cdoc).get("partitions") : null;
            if (!(parts instanceof List)) throw new IllegalArgumentException("missing \"partitions\" array");
            TreeMap<String, TreeMap<Integer, int[]>> byTopic = new TreeMap<>();
            for (Object eo : (List<Object>) parts) {
                Map<String, Object> e = (Map<String, Object>) eo;
                Object t = e.get("topic"), p = e.get("partition"), r = e.get("replicas");
                if (!(t instanceof String) || !(p instanceof Double) || !(r instanceof List))
                    throw new IllegalArgumentException("partition entry needs topic/partition/replicas");
                List<Object> rl = (List<Object>) r;
                int[] reps = new int[rl.size()];
                for (int k = 0; k < reps.length; ++k) reps[k] = ((Double) rl.get(k)).intValue();
                byTopic.computeIfAbsent((String) t, x -> new TreeMap<>()).put(((Double) p).intValue(), reps);
            }
            final int T = byTopic.size(), B = brokers.size(), R = rackList.size();
            String[] topicName = new String[T];
            int[][] partIds = new int[T][];
            int[] nP = new int[T], rf = new int[T], rfCur = new int[T];
            int curLen = 0, outLen = 0, ti = 0;
            for (Map.Entry<String, TreeMap<Integer, int[]>> kv : byTopic.entrySet()) {
                topicName[ti] = kv.getKey();
                nP[ti] = kv.getValue().size();
                for (int[] reps : kv.getValue().values()) rfCur[ti] = Math.max(rfCur[ti], reps.length);
                rf[ti] = rfOverride > 0 ? rfOverride : rfCur[ti];
                curLen += nP[ti] * rfCur[ti];
                outLen += nP[ti] * rf[ti];
                ti++;
            }
            short[] current = new short[curLen];
            int co = 0;
            ti = 0;
            for (TreeMap<Integer, int[]> pm : byTopic.values()) {
                partIds[ti] = new int[nP[ti]];
                int pi = 0;
                for (Map.Entry<Integer, int[]> pe : pm.entrySet()) {
                    partIds[ti][pi++] = pe.getKey();
                    int[] reps = pe.getValue();
                    for (int k = 0; k < rfCur[ti]; ++k) {
                        Integer d = k < reps.length ? dense.get(reps[k]) : null;
                        current[co++] = (short) (d == null ? Kao.NONE : d);   // NONE: broker not in the target list
                    }
                }
                ti++;
            }

            // ---- solve on the GPU (replaces lp_solve, README.md:135-136) ----
            Kao.init(device);
            short[] out = new short[outLen];
            long[] obj = new long[T], ub = new long[T];
            // several GPUs of this node: topics sharded by kao_solve_multi (fewer topics than GPUs: every GPU searches every
            // topic and the best is min-allreduced over RCCL); null = the one device selected by init()
            int[] devices = gpuList;
            if (devices == null && gpus > 1) { devices = new int[gpus]; for (int d = 0; d < gpus; ++d) devices[d] = device + d; }
            int[] status = Kao.solve(T, B, R, rackOf, nP, rf, rfCur, current, w, seed, timeLimit, devices, out, obj, ub);

            int exit = 0;
            boolean[] emit = new boolean[T];
            int[] curOff = new int[T], outOff = new int[T];
            for (int t = 1; t < T; ++t) { curOff[t] = curOff[t - 1] + nP[t - 1] * rfCur[t - 1]; outOff[t] = outOff[t - 1] + nP[t - 1] * rf[t - 1]; }
            for (int t = 0; t < T; ++t) {
                short[] cur = java.util.Arrays.copyOfRange(current, curOff[t], curOff[t] + nP[t] * rfCur[t]);
                if (status[t] == Kao.INFEASIBLE_PROVEN) {
                    System.err.println("KaoCli: topic " + topicName[t] + ": This problem is infeasible ("
                        + Kao.checkInfeasible(B, R, rackOf, nP[t], rf[t], rfCur[t], cur, w) + ")");
                    exit = 3;
                    continue;
                }
                if (status[t] == Kao.NO_FEASIBLE) {
                    System.err.println("KaoCli: topic " + topicName[t] + ": no feasible assignment found within the time limit (not a proof of infeasibility)");
                    exit = 3;
                    continue;
                }
                if (status[t] != Kao.OPTIMAL_PROVEN) {
                    // lp_solve only ever returns the exact optimum (README.md:135-136): never emit a possibly suboptimal plan silently
                    System.err.println("KaoCli: warning: topic " + topicName[t] + ": plan is feasible but NOT proven optimal ("
                        + (status[t] == Kao.TIME_LIMIT ? "time limit" : "bound gap") + "): objective=" + obj[t] + " bound=" + ub[t]
                        + " gap=" + (ub[t] - obj[t]) + (requireOptimal ? "; withheld (--require-optimal)" : ""));
                    if (requireOptimal) { if (exit == 0) exit = 4; continue; }
                }
                if (canonical) {
                    short[] a = java.util.Arrays.copyOfRange(out, outOff[t], outOff[t] + nP[t] * rf[t]);
                    Kao.canonicalize(B, R, rackOf, nP[t], rf[t], rfCur[t], cur, w, a);
                    System.arraycopy(a, 0, out, outOff[t], a.length);
                }
                emit[t] = true;
            }

            // ---- emit (README.md:67-78 shape, directly consumable by kafka-reassign-partitions --execute) ----
            StringBuilder os = new StringBuilder("{\"version\":1,\"partitions\":[");
            boolean first = true;
            for (int t = 0; t < T; ++t) {
                if (!emit[t]) continue;
                for (int p = 0; p < nP[t]; ++p) {
                    os.append(first ? "\n" : ",\n").append("    {\"topic\":\"").append(topicName[t]).append("\",\"partition\":")
                      .append(partIds[t][p]).append(",\"replicas\":[");
                    for (int k = 0; k < rf[t]; ++k)
                        os.append(k > 0 ? "," : "").append(brokers.get(out[outOff[t] + p * rf[t] + k] & 0xFFFF));
                    os.append("]}");
                    first = false;
                }
            }
            os.append("\n]}\n");
            if (outPath == null) System.out.print(os);
            else try (PrintStream f = new PrintStream(outPath, "UTF-8")) { f.print(os); }
            if (report) {
                for (int t = 0; t < T; ++t) {
                    int moves = 0, lead = 0;
                    if (emit[t])
                        for (int p = 0; p < nP[t]; ++p) {
                            for (int k = 0; k < rf[t]; ++k) {
                                boolean kept = false;
                                for (int j = 0; j < rfCur[t]; ++j) kept |= current[curOff[t] + p * rfCur[t] + j] == out[outOff[t] + p * rf[t] + k];
                                if (!kept) moves++;
                            }
                            if (current[curOff[t] + p * rfCur[t]] != out[outOff[t] + p * rf[t]]) lead++;
                        }
                    System.err.println("topic " + topicName[t] + ": status=" + STATUS[Math.max(0, Math.min(status[t], STATUS.length - 1))] + " objective=" + obj[t]
                        + " bound=" + ub[t] + " replica_moves=" + moves + " leader_changes=" + lead);
                }
            }
            System.exit(exit);
        } catch (IOException | RuntimeException e) {
            System.err.println("KaoCli: " + e.getMessage());
            System.exit(1);
        }
    }
}
