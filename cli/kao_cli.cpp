// kao-cli -- thin command-line front end over the C ABI (include/kao.h).
//
// Same reassignment JSON in and out as the reference and `kafka-reassign-partitions`
// (README.md:52-63 in, README.md:67-78 / README.md:88 out); target brokers as the
// `--broker-list` CSV (README.md:48).  The README gives the broker->rack map only in prose
// (README.md:27-29), so it is passed as JSON {"<brokerId>":"<rack>"} or CSV "id:rack,id:rack".
// north_star asks for a Java CLI over JNI; this image has no JDK, so the host side above the C ABI
// is C++ (INTEGRATION.md shows the JNI stub).  All computation happens in libkao.so on the GPU.
//
//   kao-cli --current cur.json --broker-list 0,1,2 --racks racks.json [--rf N] [--weights 4,1,2,2]
//           [--seed S] [--time-limit SEC] [--device D] [--no-canonical] [--out out.json] [--report]
//           [--emit-lp PREFIX [--lp-only]]
// --emit-lp writes the generated 0-1 model of every topic as lp_solve LP text (README.md:144-185) to
// PREFIX<topic#>.lp -- no GPU needed -- so the answer can be cross-checked with real lp_solve 5.5.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/kao.h"

namespace {

// ---- minimal JSON reader (objects, arrays, strings, integers, true/false/null) -----------------
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue *get(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const std::string &s;
    size_t i = 0;
    explicit JParser(const std::string &src) : s(src) {}
    [[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("JSON: ") + what + " at offset " + std::to_string(i)); }
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    JValue parse() { JValue v = value(); ws(); if (i != s.size()) bad("trailing characters"); return v; }
    JValue value() {
        ws();
        if (i >= s.size()) bad("unexpected end");
        char c = s[i];
        JValue v;
        if (c == '{') {
            v.kind = JValue::Obj; ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws(); JValue k = value();
                if (k.kind != JValue::Str) bad("object key must be a string");
                ws(); if (i >= s.size() || s[i] != ':') bad("expected ':'"); ++i;
                v.obj.emplace_back(k.str, value());
                ws(); if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; return v; }
                bad("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JValue::Arr; ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws(); if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; return v; }
                bad("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = JValue::Str; ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) { ++i; char e = s[i]; v.str += (e == 'n' ? '\n' : e == 't' ? '\t' : e); }
                else v.str += s[i];
                ++i;
            }
            if (i >= s.size()) bad("unterminated string");
            ++i; return v;
        }
        if (c == '-' || std::isdigit((unsigned char)c)) {
            size_t j = i; if (s[j] == '-') ++j;
            while (j < s.size() && (std::isdigit((unsigned char)s[j]) || s[j] == '.' || s[j] == 'e' || s[j] == 'E' || s[j] == '+' || s[j] == '-')) ++j;
            v.kind = JValue::Num; v.num = std::strtod(s.substr(i, j - i).c_str(), nullptr); i = j; return v;
        }
        if (s.compare(i, 4, "true") == 0) { v.kind = JValue::Bool; v.b = true; i += 4; return v; }
        if (s.compare(i, 5, "false") == 0) { v.kind = JValue::Bool; i += 5; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
        bad("unexpected character");
    }
};

std::string slurp(const std::string &path) {
    if (path == "-") { std::stringstream ss; ss << std::cin.rdbuf(); return ss.str(); }
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf(); return ss.str();
}

std::vector<std::string> split(const std::string &s, char sep) {
    std::vector<std::string> out; std::string cur;
    for (char c : s) { if (c == sep) { out.push_back(cur); cur.clear(); } else if (!std::isspace((unsigned char)c)) cur += c; }
    if (!cur.empty() || !s.empty()) out.push_back(cur);
    return out;
}

struct TopicData {
    std::string name;
    std::vector<int> partition_ids;
    std::vector<uint16_t> current;  // [P * rf_cur]
    int rf_cur = 0, rf = 0;
};

[[noreturn]] void usage(const char *msg) {
    if (msg) std::fprintf(stderr, "kao-cli: %s\n", msg);
    std::fprintf(stderr,
        "usage: kao-cli --current <reassignment.json|-> --broker-list <id,id,...> --racks <racks.json | id:rack,...>\n"
        "               [--rf N] [--weights LL,LF,FL,FF] [--seed S] [--time-limit SEC] [--device D] [--gpus N | d0,d1,...]\n"
        "               [--no-canonical] [--out <file>] [--report] [--require-optimal] [--emit-lp <prefix> [--lp-only]]\n"
        "exit status: 0 = every topic solved (a warning on stderr marks any plan that is feasible but not PROVEN optimal),\n"
        "             3 = a topic is infeasible / no feasible plan found, 4 = --require-optimal and a plan was not proven\n"
        "             optimal (that topic is withheld from the output), 1 = error, 2 = usage\n");
    std::exit(2);
}

// lp_solve LP-format text of one topic's model (README.md:144-185): `max:` objective over the current
// placements, rows C1..C7 in README order, `bin` section in the declared variable order broker-major /
// partition-minor / follower-then-leader (README.md:184).  Same text as oracle/kao_oracle.py::write_lp.
std::string lp_text(const kao_topic &t, const TopicData &td, const std::vector<int> &brokers, int t_index) {
    const int B = t.n_brokers, R = t.n_racks, P = t.n_partitions;
    int32_t bd[8];
    if (kao_derive_bounds(&t, bd) != 0) throw std::runtime_error(std::string("kao_derive_bounds: ") + kao_last_error());
    auto name = [&](int b, int p, bool leader) {
        return "t" + std::to_string(t_index) + "b" + std::to_string(brokers[(size_t)b]) + "p" + std::to_string(td.partition_ids[(size_t)p]) + (leader ? "_l" : "");
    };
    std::ostringstream os;
    os << "// Optimization function, based on current assignment\nmax: ";
    bool first = true;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            int role = -1;  // current role of b on p
            for (int k = 0; k < t.rf_cur; ++k) if (t.current[(size_t)p * t.rf_cur + k] == b) { role = k == 0 ? 0 : 1; break; }
            if (role < 0) continue;
            const int cf = t.w[role][1], cl = t.w[role][0];
            if (cf != 0) { os << (first ? "" : " + ") << cf << " " << name(b, p, false); first = false; }
            if (cl != 0) { os << (first ? "" : " + ") << cl << " " << name(b, p, true); first = false; }
        }
    if (first) os << "0";
    os << ";\n";
    auto row = [&](const std::vector<std::string> &vars, long lo, long hi, bool has_lo, bool has_hi) {
        std::string lhs;
        for (size_t i = 0; i < vars.size(); ++i) { if (i) lhs += " + "; lhs += vars[i]; }
        if (has_lo && has_hi && lo == hi) { os << lhs << " = " << lo << ";\n"; return; }
        if (has_hi) os << lhs << " <= " << hi << ";\n";
        if (has_lo) os << lhs << " >= " << lo << ";\n";
    };
    std::vector<std::string> v;
    os << "\n// Constrain on replication factor for every partition\n";
    for (int p = 0; p < P; ++p) { v.clear(); for (int b = 0; b < B; ++b) { v.push_back(name(b, p, false)); v.push_back(name(b, p, true)); } row(v, t.rf, t.rf, true, true); }
    os << "\n// Constraint on having one and only one leader per partition\n";
    for (int p = 0; p < P; ++p) { v.clear(); for (int b = 0; b < B; ++b) v.push_back(name(b, p, true)); row(v, 1, 1, true, true); }
    os << "\n// Constraint on min/max replicas per broker\n";
    for (int b = 0; b < B; ++b) { v.clear(); for (int p = 0; p < P; ++p) { v.push_back(name(b, p, false)); v.push_back(name(b, p, true)); } row(v, bd[0], bd[1], true, true); }
    os << "\n// Constraint on min/max leaders per broker\n";
    for (int b = 0; b < B; ++b) { v.clear(); for (int p = 0; p < P; ++p) v.push_back(name(b, p, true)); row(v, bd[2], bd[3], true, true); }
    os << "\n// Constraint on no leader and replicas on the same broker\n";
    for (int b = 0; b < B; ++b) for (int p = 0; p < P; ++p) { v = {name(b, p, false), name(b, p, true)}; row(v, 0, 1, false, true); }
    os << "\n// Constrain on min/max total replicas per racks\n";
    for (int r = 0; r < R; ++r) { v.clear(); for (int b = 0; b < B; ++b) if (t.rack_of[b] == r) for (int p = 0; p < P; ++p) { v.push_back(name(b, p, false)); v.push_back(name(b, p, true)); } row(v, bd[4], bd[5], true, true); }
    os << "\n// Constrain on min/max replicas per partitions per racks\n";
    for (int p = 0; p < P; ++p) for (int r = 0; r < R; ++r) { v.clear(); for (int b = 0; b < B; ++b) if (t.rack_of[b] == r) { v.push_back(name(b, p, false)); v.push_back(name(b, p, true)); } if (!v.empty()) row(v, bd[6], bd[7], true, true); }
    os << "\n// All variables are binary\nbin\n";
    first = true;
    for (int b = 0; b < B; ++b) for (int p = 0; p < P; ++p) { os << (first ? "" : ", ") << name(b, p, false) << ", " << name(b, p, true); first = false; }
    os << ";\n";
    return os.str();
}

}  // namespace

int main(int argc, char **argv) {
    std::string cur_path, brokers_csv, racks_arg, out_path;
    int rf_override = 0, device = 0, gpus = 1;
    std::vector<int32_t> gpu_list;
    int w[4] = {4, 1, 2, 2};
    unsigned long long seed = 1;
    double time_limit = 10.0;
    bool canonical = true, report = false, lp_only = false, require_optimal = false;
    std::string lp_prefix;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *flag) -> std::string { if (i + 1 >= argc) usage((std::string(flag) + " needs a value").c_str()); return argv[++i]; };
        if (a == "--current") cur_path = need("--current");
        else if (a == "--broker-list") brokers_csv = need("--broker-list");
        else if (a == "--racks") racks_arg = need("--racks");
        else if (a == "--rf") rf_override = std::atoi(need("--rf").c_str());
        else if (a == "--weights") { auto p = split(need("--weights"), ','); if (p.size() != 4) usage("--weights needs LL,LF,FL,FF"); for (int k = 0; k < 4; ++k) w[k] = std::atoi(p[k].c_str()); }
        else if (a == "--seed") seed = std::strtoull(need("--seed").c_str(), nullptr, 0);
        else if (a == "--time-limit") time_limit = std::atof(need("--time-limit").c_str());
        else if (a == "--device") device = std::atoi(need("--device").c_str());
        else if (a == "--gpus") {   // N = devices device .. device+N-1; a,b,... = exactly these ordinals (one may repeat: logical shards)
            const std::string v = need("--gpus");
            if (v.find(',') == std::string::npos) gpus = std::atoi(v.c_str());
            else { for (auto &t : split(v, ',')) if (!t.empty()) gpu_list.push_back(std::atoi(t.c_str())); gpus = (int)gpu_list.size(); }
            if (gpus < 1) usage("--gpus needs a count >= 1 or a device list");
        }
        else if (a == "--no-canonical") canonical = false;
        else if (a == "--out") out_path = need("--out");
        else if (a == "--report") report = true;
        else if (a == "--require-optimal") require_optimal = true;
        else if (a == "--emit-lp") lp_prefix = need("--emit-lp");
        else if (a == "--lp-only") lp_only = true;
        else if (a == "-h" || a == "--help") usage(nullptr);
        else usage(("unknown flag " + a).c_str());
    }
    if (cur_path.empty() || brokers_csv.empty() || racks_arg.empty()) usage("--current, --broker-list and --racks are required");
    try {
        // ---- target brokers and racks ---------------------------------------------------------
        std::vector<int> brokers;
        for (auto &t : split(brokers_csv, ',')) if (!t.empty()) brokers.push_back(std::atoi(t.c_str()));
        if (brokers.empty()) throw std::runtime_error("empty broker list");
        std::map<int, int> dense;
        for (size_t i = 0; i < brokers.size(); ++i) if (!dense.emplace(brokers[i], (int)i).second) throw std::runtime_error("duplicate id in broker list");
        std::map<int, std::string> rack_name;
        if (racks_arg.find(':') != std::string::npos && racks_arg.find('{') == std::string::npos) {
            for (auto &t : split(racks_arg, ',')) { auto kv = split(t, ':'); if (kv.size() != 2) throw std::runtime_error("bad --racks entry " + t); rack_name[std::atoi(kv[0].c_str())] = kv[1]; }
        } else {
            std::string txt = slurp(racks_arg);
            JValue doc = JParser(txt).parse();
            if (doc.kind != JValue::Obj) throw std::runtime_error("racks file must be a JSON object {\"<brokerId>\": \"<rack>\"}");
            for (auto &kv : doc.obj) rack_name[std::atoi(kv.first.c_str())] = kv.second.kind == JValue::Str ? kv.second.str : std::to_string((long long)kv.second.num);
        }
        std::set<std::string> names;
        for (int b : brokers) { auto it = rack_name.find(b); if (it == rack_name.end()) throw std::runtime_error("no rack given for broker " + std::to_string(b)); names.insert(it->second); }
        std::map<std::string, int> rack_idx; for (auto &n : names) { int k = (int)rack_idx.size(); rack_idx[n] = k; }
        std::vector<uint8_t> rack_of; for (int b : brokers) rack_of.push_back((uint8_t)rack_idx[rack_name[b]]);

        // ---- current assignment (README.md:52-63) ----------------------------------------------
        std::string cur_txt = slurp(cur_path);
        JValue doc = JParser(cur_txt).parse();
        const JValue *parts = doc.get("partitions");
        if (!parts || parts->kind != JValue::Arr) throw std::runtime_error("missing \"partitions\" array");
        std::map<std::string, std::map<int, std::vector<int>>> by_topic;
        for (auto &e : parts->arr) {
            const JValue *t = e.get("topic"), *p = e.get("partition"), *r = e.get("replicas");
            if (!t || !p || !r || r->kind != JValue::Arr) throw std::runtime_error("partition entry needs topic/partition/replicas");
            std::vector<int> reps; for (auto &x : r->arr) reps.push_back((int)x.num);
            by_topic[t->str][(int)p->num] = reps;
        }
        std::vector<TopicData> tds;
        for (auto &kv : by_topic) {
            TopicData td; td.name = kv.first;
            for (auto &pr : kv.second) td.rf_cur = std::max(td.rf_cur, (int)pr.second.size());
            td.rf = rf_override > 0 ? rf_override : td.rf_cur;
            for (auto &pr : kv.second) {
                td.partition_ids.push_back(pr.first);
                for (int k = 0; k < td.rf_cur; ++k) {
                    uint16_t v = KAO_NONE;
                    if (k < (int)pr.second.size()) { auto it = dense.find(pr.second[k]); if (it != dense.end()) v = (uint16_t)it->second; }
                    td.current.push_back(v);
                }
            }
            tds.push_back(std::move(td));
        }
        std::vector<kao_topic> topics(tds.size());
        std::vector<std::vector<uint16_t>> assigns(tds.size());
        std::vector<kao_result> results(tds.size());
        for (size_t i = 0; i < tds.size(); ++i) {
            kao_topic &t = topics[i];
            t.n_brokers = (int)brokers.size(); t.n_racks = (int)rack_idx.size(); t.n_partitions = (int)tds[i].partition_ids.size();
            t.rf = tds[i].rf; t.rf_cur = tds[i].rf_cur; t.rack_of = rack_of.data(); t.current = tds[i].current.data();
            t.w[0][0] = w[0]; t.w[0][1] = w[1]; t.w[1][0] = w[2]; t.w[1][1] = w[3];
            t.rep_lo = t.rep_hi = t.lead_lo = t.lead_hi = t.rack_lo = t.rack_hi = t.prack_lo = t.prack_hi = -1;
            assigns[i].assign((size_t)t.n_partitions * t.rf, KAO_NONE);
            results[i] = kao_result{};
            results[i].assignment = assigns[i].data();
        }
        if (!lp_prefix.empty()) {  // the generated model as lp_solve LP text; host only
            for (size_t i = 0; i < tds.size(); ++i) {
                std::ofstream f(lp_prefix + std::to_string(i + 1) + ".lp");
                if (!f) throw std::runtime_error("cannot write " + lp_prefix + std::to_string(i + 1) + ".lp");
                f << lp_text(topics[i], tds[i], brokers, (int)i + 1);
            }
            if (lp_only) return 0;
        }
        // ---- solve on the GPU ------------------------------------------------------------------
        int rc = kao_init(gpu_list.empty() ? device : gpu_list[0]);
        if (rc) throw std::runtime_error(std::string("kao_init: ") + kao_strerror(rc) + " " + kao_last_error());
        kao_opts opts{};
        opts.seed = seed; opts.time_limit_s = time_limit; opts.stop_at_bound = 0; opts.iters_per_launch = 0;  // 0 = by topic size
        // stop early when every topic is proven optimal; otherwise search until the time limit
        opts.stop_at_bound = 1;
        if (gpus > 1 || !gpu_list.empty()) {  // (a list of ONE ordinal runs kao_solve on that device, not on --device)  // devices device .. device+gpus-1 of this node: topics sharded (or, with fewer topics than GPUs, replicated with an
                         // RCCL min-allreduce of the global best between them)
            std::vector<int32_t> devs = gpu_list;
            if (devs.empty()) for (int d = 0; d < gpus; ++d) devs.push_back(device + d);
            rc = kao_solve_multi(topics.data(), (int)topics.size(), devs.data(), gpus, &opts, results.data());
        } else rc = kao_solve(topics.data(), (int)topics.size(), &opts, results.data());
        if (rc) throw std::runtime_error(std::string("kao_solve: ") + kao_strerror(rc) + " " + kao_last_error());
        int exit_code = 0;
        std::vector<char> withheld(tds.size(), 0);
        for (size_t i = 0; i < tds.size(); ++i) {
            if (results[i].status == KAO_STATUS_INFEASIBLE_PROVEN) {
                char why[256] = "";
                (void)kao_check_infeasible(&topics[i], why, (int)sizeof why);
                std::fprintf(stderr, "kao-cli: topic %s: This problem is infeasible (%s)\n", tds[i].name.c_str(), why);
                exit_code = 3;
                continue;
            }
            if (results[i].status == KAO_STATUS_NO_FEASIBLE) {
                std::fprintf(stderr, "kao-cli: topic %s: no feasible assignment found within the time limit (not a proof of infeasibility)\n", tds[i].name.c_str());
                exit_code = 3;
                continue;
            }
            if (results[i].status != KAO_STATUS_OPTIMAL_PROVEN) {
                // lp_solve only ever returns the exact optimum (README.md:135-136); a plan that is not proven optimal may
                // move more replicas than necessary, so it is never emitted silently
                const long long gap = (long long)(results[i].upper_bound - results[i].objective);
                std::fprintf(stderr, "kao-cli: warning: topic %s: plan is feasible but NOT proven optimal (%s): objective=%lld bound=%lld gap=%lld%s\n",
                             tds[i].name.c_str(), results[i].status == KAO_STATUS_TIME_LIMIT ? "time limit" : "bound gap",
                             (long long)results[i].objective, (long long)results[i].upper_bound, gap,
                             require_optimal ? "; withheld (--require-optimal)" : "");
                if (require_optimal) { exit_code = exit_code ? exit_code : 4; withheld[i] = 1; continue; }
            }
            if (canonical) {
                rc = kao_canonicalize(&topics[i], assigns[i].data());
                if (rc) throw std::runtime_error(std::string("kao_canonicalize: ") + kao_strerror(rc));
            }
        }
        // ---- emit (README.md:67-78 shape, directly consumable by kafka-reassign-partitions --execute)
        std::ostringstream os;
        os << "{\"version\":1,\"partitions\":[";
        bool first = true;
        for (size_t i = 0; i < tds.size(); ++i) {
            if (results[i].status == KAO_STATUS_NO_FEASIBLE || results[i].status == KAO_STATUS_INFEASIBLE_PROVEN || withheld[i]) continue;
            const int P = topics[i].n_partitions, RF = topics[i].rf;
            for (int p = 0; p < P; ++p) {
                os << (first ? "\n" : ",\n") << "    {\"topic\":\"" << tds[i].name << "\",\"partition\":" << tds[i].partition_ids[p] << ",\"replicas\":[";
                for (int k = 0; k < RF; ++k) os << (k ? "," : "") << brokers[assigns[i][(size_t)p * RF + k]];
                os << "]}";
                first = false;
            }
        }
        os << "\n]}\n";
        if (out_path.empty()) std::fputs(os.str().c_str(), stdout);
        else { std::ofstream f(out_path); f << os.str(); }
        if (report) {
            static const char *st[] = {"OPTIMAL_PROVEN", "FEASIBLE_BOUND_GAP", "NO_FEASIBLE", "TIME_LIMIT", "INFEASIBLE_PROVEN", "?", "?", "?"};
            for (size_t i = 0; i < tds.size(); ++i) {
                int moves = 0, lead = 0;
                const int P = topics[i].n_partitions, RF = topics[i].rf, RC = topics[i].rf_cur;
                if (results[i].status != KAO_STATUS_NO_FEASIBLE && results[i].status != KAO_STATUS_INFEASIBLE_PROVEN)
                    for (int p = 0; p < P; ++p) {
                        for (int k = 0; k < RF; ++k) {
                            bool kept = false;
                            for (int j = 0; j < RC; ++j) kept |= tds[i].current[(size_t)p * RC + j] == assigns[i][(size_t)p * RF + k];
                            moves += !kept;
                        }
                        lead += tds[i].current[(size_t)p * RC] != assigns[i][(size_t)p * RF];
                    }
                std::fprintf(stderr, "topic %s: status=%s objective=%lld bound=%lld replica_moves=%d leader_changes=%d seconds_to_best=%.4f\n",
                             tds[i].name.c_str(), st[results[i].status & 7], (long long)results[i].objective, (long long)results[i].upper_bound,
                             moves, lead, results[i].seconds_to_best);
            }
        }
        kao_shutdown();
        return exit_code;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "kao-cli: %s\n", e.what());
        return 1;
    }
}
