// Error reporting + ABI version for the layoutdetr_amd C ABI (include/ldetr_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include "../../include/ldetr_hip.h"

namespace ldetr { void note_engine_launch(int kind, int bm, int bn, int bk, int waves, int fast, int split, int splitk, long blocks, int modes); }

namespace ldetr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// LDETR_DEBUG="KEY=value,KEY=value": the development knobs of every kernel's launch policy behind one variable (ldetr_common.hpp)
static const std::map<std::string, std::string>& knobs() {
    static std::map<std::string, std::string> m;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("LDETR_DEBUG");
        if (!e) return;
        std::string s(e);
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i); if (j == std::string::npos) j = s.size();
            const std::string kv = s.substr(i, j - i);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos && eq > 0) m[kv.substr(0, eq)] = kv.substr(eq + 1);
            i = j + 1;
        }
    });
    return m;
}
long knob(const char* key, long dflt) {
    const auto& m = knobs();
    const auto it = m.find(key);
    return it == m.end() ? dflt : atol(it->second.c_str());
}
double knob_f(const char* key, double dflt) {
    const auto& m = knobs();
    const auto it = m.find(key);
    return it == m.end() ? dflt : atof(it->second.c_str());
}
struct EngineLastLaunch { int kind, bm, bn, bk, waves, fast, split, splitk; long blocks; int modes; };
static thread_local EngineLastLaunch t_engine_last = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
void note_engine_launch(int kind, int bm, int bn, int bk, int waves, int fast, int split, int splitk, long blocks, int modes) {
    t_engine_last = {kind, bm, bn, bk, waves, fast, split, splitk, blocks, modes};
}
}  // namespace ldetr

extern "C" int ldetr_engine_last_launch(int32_t* info10) {
    if (!info10) { ldetr::set_error("engine_last_launch: null output"); return 1; }
    const ldetr::EngineLastLaunch& l = ldetr::t_engine_last;
    const int32_t v[10] = {l.kind, l.bm, l.bn, l.bk, l.waves, l.fast, l.split, l.splitk, (int32_t)(l.blocks > 0x7fffffffL ? 0x7fffffff : l.blocks), l.modes};
    for (int i = 0; i < 10; i++) info10[i] = v[i];
    return 0;
}

extern "C" const char* ldetr_last_error(void) { return ldetr::g_err; }
extern "C" int ldetr_abi_version(void) { return 24; }
// sizeof of every argument block of the group launches, in header order: the host bindings (layoutdetr_amd/_lib.py) mirror them field by field
extern "C" int ldetr_struct_sizes(int32_t* out6) {
    if (!out6) return 1;
    out6[0] = (int32_t)sizeof(ldetr_ln_args); out6[1] = (int32_t)sizeof(ldetr_ffn_args); out6[2] = (int32_t)sizeof(ldetr_mha_small_args);
    out6[3] = (int32_t)sizeof(ldetr_mha_cross_args); out6[4] = (int32_t)sizeof(ldetr_wgrad_desc); out6[5] = (int32_t)sizeof(ldetr_p3_epilogue);
    return 0;
}
