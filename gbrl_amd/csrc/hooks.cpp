// hooks.cpp -- see hooks.h.  The only getenv of the library.
#include "hooks.h"

#include <cstdint>
#include <cstring>

namespace gbrl {
namespace hooks {
namespace {
const char *const kNames[kCount] = {
#define X(n) "GBRL_HIP_" #n,
    GBRL_HIP_HOOKS(X)
#undef X
};
struct Slot { uint64_t epoch; bool set; char val[48]; };
thread_local uint64_t t_epoch = 1;
thread_local Slot t_slot[kCount];
}  // namespace

void begin_call() { ++t_epoch; }

const char *name(Id id) { return kNames[id]; }

const char *raw(Id id) {
    Slot &s = t_slot[id];
    if (s.epoch != t_epoch) {
        const char *e = std::getenv(kNames[id]);   // (copied: a later setenv of another thread may move the environment block)
        s.set = e != nullptr;
        if (e) { std::strncpy(s.val, e, sizeof(s.val) - 1); s.val[sizeof(s.val) - 1] = 0; }
        s.epoch = t_epoch;
    }
    return s.set ? s.val : nullptr;
}

}  // namespace hooks
}  // namespace gbrl
