// Test-only door to the host evaluator (tla_rust_amd/csrc/tlaeval.cpp) for modules the PRODUCT refuses to evaluate on the host
// because they have a GPU lowering (raft, the snapshot-isolation specs, PlusCal translations): the tests run the evaluator on
// those texts too and compare its state graphs with the fixtures / the C oracle.  Never linked into libtlamc.so.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../tla_rust_amd/csrc/tlaeval.h"

extern "C" int tlaeval_door(const char *tla, const char *cfg, const char *search, uint64_t max_levels, int check_deadlock, int symmetry, const char *dump,
                            const char *order, char *out, size_t cap) {
    tlaeval::Options o;
    tlaeval::Result r;
    o.max_levels = max_levels;
    o.check_deadlock = check_deadlock != 0;
    o.symmetry = symmetry != 0;
    auto split = [](const char *s, char sep, std::vector<std::string> &dst) {
        if (!s) return;
        std::string t = s;
        size_t a = 0;
        while (a <= t.size()) { const size_t b = t.find(sep, a); std::string x = t.substr(a, b == std::string::npos ? std::string::npos : b - a); if (!x.empty()) dst.push_back(x); if (b == std::string::npos) break; a = b + 1; }
    };
    split(search, ':', o.search);
    split(order, ',', o.dump_order);
    if (dump) o.dump_path = dump;
    std::string err;
    const int rc = tlaeval::check_files(tla, cfg, o, r, err);
    std::string j;
    if (rc) { j = "{\"rc\": " + std::to_string(rc) + ", \"error\": \""; for (char c : err) { if (c == '"' || c == '\\') j += '\\'; j += c == '\n' ? ' ' : c; } j += "\"}"; }
    else {
        j = "{\"rc\": 0, \"distinct\": " + std::to_string(r.distinct) + ", \"generated\": " + std::to_string(r.generated) + ", \"depth\": " + std::to_string(r.depth) +
            ", \"verdict\": " + std::to_string(r.verdict) + ", \"violated_invariant\": " + std::to_string(r.violated_invariant) + ", \"trace_len\": " + std::to_string(r.trace.size()) +
            ", \"queue_left\": " + std::to_string(r.queue_left) + ", \"seconds\": " + std::to_string(r.seconds) + ", \"levels\": [";
        for (size_t i = 0; i < r.levels.size(); i++) j += (i ? ", " : "") + std::to_string(r.levels[i]);
        j += "], \"trace_labels\": [";
        for (size_t i = 0; i < r.trace.size(); i++) j += std::string(i ? ", " : "") + "\"" + r.trace[i].first + "\"";
        j += "]}";
    }
    if (j.size() + 1 > cap) return -1;
    memcpy(out, j.c_str(), j.size() + 1);
    return rc;
}
