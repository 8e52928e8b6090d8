// c4_config.h — the library's C4GPU_* switches (measurement shapes, test hooks, fallbacks kept for A/B runs), read from the
// environment ONCE, in one function, and held in one table: no getenv on any call path of a
// library whose calls now run on three threads (main, device flush, SDP flight) beside HIP's own start-up setenv.
// The table is filled when the first context opens (or on the first question, whichever comes first);
// c4gpu_config_reload() (include/c4gpu.h) reads the environment again: the explicit hook of the tests, which flip a variable
// between two calls -- exonerate_amd's Python wrapper calls it in front of every library call, nothing else does.
#pragma once
#include <cstdlib>
#include <mutex>

namespace c4cfg {

#define C4CFG_KEYS(X) \
    X(TRACE) X(PK16) X(WPE) X(WIN16) X(PACK) X(SEED_KSHIFT) X(WIN_NW) X(WINDOW_HOPS) X(WINDOWED) X(STRICT) X(SPLICE_TILE) \
    X(SDP_ARENA_MB) X(REPAIR_REJOIN) X(PK16_R6) X(PK16_NW8) X(PK16_LONG) X(PK16_IO) X(PK16_C8) X(PIN_XFER) X(NESTED_REDO) X(MW) \
    X(LOOP_SHORTCUT) X(LOCAL_EXACT) X(LANES) X(FUSED) X(FREE_NOW) X(FORCE_CORNER_MISMATCH) X(DL_SYNC_FIRST) X(CONT_FREE) \
    X(CK16_TMAX) X(CK16_ROOT) X(CK16) X(CELL_STRICT) X(BYROOT) X(SS16_CHECK) X(SCORE_FIRST) X(HOST_THREADS) X(FORCE_SEQUENTIAL) \
    X(FILL_ALLOC)

enum Key {
#define X(name) name,
    C4CFG_KEYS(X)
#undef X
    N_KEYS
};

struct Entry { bool set; int value; double real; };
struct Table { Entry e[N_KEYS]; };

inline Table &table() { static Table t{}; return t; }
inline std::once_flag &loaded_flag() { static std::once_flag f; return f; }

// THE place where the library reads its environment
inline void load_from_environment() {
    static const char *const names[N_KEYS] = {
#define X(name) "C4GPU_" #name,
        C4CFG_KEYS(X)
#undef X
    };
    Table &t = table();
    for (int k = 0; k < N_KEYS; k++) {
        const char *v = getenv(names[k]);
        t.e[k].set = v != nullptr;
        t.e[k].value = v ? atoi(v) : 0;
        t.e[k].real = v ? atof(v) : 0.0;
    }
}
inline const Table &get() {
    std::call_once(loaded_flag(), load_from_environment);
    return table();
}
inline bool has(Key k) { return get().e[k].set; }                                  // the variable is there, whatever it says
inline int num(Key k, int dflt) { const Entry &x = get().e[k]; return x.set ? x.value : dflt; }
inline double real(Key k, double dflt) { const Entry &x = get().e[k]; return x.set ? x.real : dflt; }
inline bool is(Key k, int v) { const Entry &x = get().e[k]; return x.set && x.value == v; }   // set AND equal to v
inline bool nonzero(Key k) { const Entry &x = get().e[k]; return x.set && x.value != 0; }

}  // namespace c4cfg
