// hooks.h -- EVERY environment hook of the library in one table (INTEGRATION.md section 5 documents each; tests/test_host.py checks that
// the two lists agree and that no other translation unit calls getenv).  A hook is read from the environment AT MOST ONCE PER API CALL:
// c_api.cpp's `guarded` opens a new call (`hooks::begin_call`), the first `hooks::raw(id)` of that call reads the variable and keeps a copy,
// every later read -- per tree level, per launch -- is a thread-local array look-up.  The tests flip hooks between calls, so nothing is
// latched across calls.
#pragma once
#include <cstdlib>

namespace gbrl {
namespace hooks {

#define GBRL_HIP_HOOKS(X) \
    X(BIN_PLAIN) \
    X(CAT_CHECK) \
    X(CAT_PROF) \
    X(DEVICE_LEVELS) \
    X(EVENT_RESULTS) \
    X(FORCE_BISECTION) \
    X(FORCE_COLLECTIVE) \
    X(HIST_ALLREDUCE_MAX_KB) \
    X(HIST_GENERIC) \
    X(HIST_PIPE) \
    X(HOST_CATEGORICAL) \
    X(KEEP_LAST_DERIVED) \
    X(NEARTIE_DEBUG) \
    X(NEARTIE_MAX_ROWS) \
    X(NEARTIE_REL) \
    X(NEARTIE_SERIAL) \
    X(NO_DIRECT_HIST) \
    X(NO_IOTA_CACHE) \
    X(NO_NEARTIE_REPLAY) \
    X(NO_SMALL_GROW) \
    X(NO_SMALL_PREP) \
    X(NO_SMALL_STATS) \
    X(PREDICT_CHAIN) \
    X(PREDICT_GENERIC) \
    X(PREDICT_GRD_STREAM_MIN_ROWS) \
    X(PREDICT_GRD_STREAM_WAVES) \
    X(PREDICT_NB) \
    X(PREDICT_NOSPLIT) \
    X(PREDICT_NO_GRD_STREAM) \
    X(PREDICT_NO_PC) \
    X(PREDICT_NO_PERSIST) \
    X(PREDICT_NO_REG) \
    X(PREDICT_NO_RESIDENT) \
    X(PREDICT_OBL1) \
    X(PREDICT_REG_GROUPED) \
    X(PREDICT_REG_MIN_ROWS) \
    X(PREDICT_REG_ONLY) \
    X(PREDICT_REG_WAVES) \
    X(PREDICT_RG) \
    X(PREDICT_TT) \
    X(QUANTILE_RADIX) \
    X(QUANTILE_SAMPLE) \
    X(RADIX_PLAIN_LOADS) \
    X(ROOT_COUNTS) \
    X(SHAP_HOST) \
    X(SMALL_GROW_BLOCKS) \
    X(SMALL_GROW_PROF) \
    X(SMALL_PREP_PROF) \
    X(SORT_NO_CODES) \
    X(SPIN_SECONDS) \
    X(TEST_CAT_CLASH) \
    X(TEST_SMALL_GROW_FAIL) \
    X(TRANSPOSE_COUNT) \
    X(TRANSPOSE_PLAIN)

enum Id : int {
#define X(n) n,
    GBRL_HIP_HOOKS(X)
#undef X
    kCount
};

void begin_call();                 // entry of every exported function (c_api.cpp `guarded`)
const char *raw(Id id);            // the variable's value for THIS call, nullptr when unset (valid until the thread's next begin_call)
const char *name(Id id);           // "GBRL_HIP_..."
inline bool on(Id id) { const char *e = raw(id); return e && e[0] == '1'; }
inline int num(Id id, int dflt) { const char *e = raw(id); return e ? std::atoi(e) : dflt; }

}  // namespace hooks
}  // namespace gbrl
