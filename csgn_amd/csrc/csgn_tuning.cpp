// csgn_tuning.cpp -- storage of the tuning knobs (csgn_tuning.h).  Host code only.
#include "csgn_tuning.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>

namespace csgn {

namespace {

struct Knob {
    const char *name;
    int dflt;
};

// order = enum TuneKey
const Knob kKnobs[TUNE_COUNT] = {
    {"mul_m", 0},        {"mul_ti", 4},        {"mul_nt", 1},        {"mul_flat", 0},
    {"mul_bs", 0},       {"mul_xcd", 1},       {"mul_touch", -1},    {"mul_pf_kb", -1},
    {"stream_xcd", -1},  {"ragged_c", 0},      {"ragged_flat", 0},   {"ragged_pf", 32},
    {"ragged_touch", 1}, {"ragged_m", 4},      {"perm_ballot", 0},   {"perm_narrow", 0},
    {"perm_waves", 0},   {"perm_persist", 1},  {"dec_loop", 0},
    {"enc_lds", 0},      {"enc_wave", 1},      {"enc_compact", -1},  {"shared_gpu", 0},
    {"compact_tag_bits", 0}, {"compact_nt", 1}, {"compact_grid", 0}, {"ragged_classes", 0},
    {"ragged_coop", -1}, {"ragged_coop_span", 0}, {"ragged_coop_k", 0}, {"ragged_coop_touch", 128}, {"ragged_slice_mb", 0}, {"ragged_coop_pipe", 0},
    {"ragged_xcd_group", 64},
    {"ragged_coop_xcd_group", 0},
    {"compact_stagger_us", 8},
    {"zero_memset", 0},
};

// Knob values are PER HOST THREAD: a thread that sets a knob changes the dispatch of its own later
// calls only, so the one-host-thread-per-GPU callers of the C ABI (include/csgn_hip.h, "thread
// safety") cannot alter one another's kernels in mid-stream (round 2 kept them in
// process-wide atomics).  A thread's first use copies the defaults + the environment snapshot.
struct ThreadKnobs {
    bool ready = false;
    int v[TUNE_COUNT];
};
thread_local ThreadKnobs t_knobs;
int g_env_value[TUNE_COUNT];
bool g_env_set[TUNE_COUNT];

int *values()
{
    if (!t_knobs.ready)
        tune_reset();                    // first use on this thread: defaults + the environment snapshot
    return t_knobs.v;
}

// The one place the environment is read: when the library is loaded, before any entry point can
// be called.
struct EnvSnapshot {
    EnvSnapshot()
    {
        for (int k = 0; k < TUNE_COUNT; ++k) {
            std::string env = "CSGN_";
            for (const char *p = kKnobs[k].name; *p; ++p)
                env += (char)toupper((unsigned char)*p);
            const char *v = getenv(env.c_str());
            g_env_set[k] = v && *v;
            g_env_value[k] = g_env_set[k] ? atoi(v) : 0;
        }
    }
};
EnvSnapshot g_snapshot;

int find(const char *name)
{
    if (!name)
        return -1;
    for (int k = 0; k < TUNE_COUNT; ++k)
        if (strcmp(kKnobs[k].name, name) == 0)
            return k;
    return -1;
}

} // namespace

int tune(TuneKey k) { return values()[k]; }

const char *tune_name(int k) { return (k >= 0 && k < TUNE_COUNT) ? kKnobs[k].name : nullptr; }

bool tune_set(const char *name, int value)
{
    const int k = find(name);
    if (k < 0)
        return false;
    values()[k] = value;
    return true;
}

bool tune_get(const char *name, int *value)
{
    const int k = find(name);
    if (k < 0)
        return false;
    *value = values()[k];
    return true;
}

void tune_reset()
{
    t_knobs.ready = true;
    for (int k = 0; k < TUNE_COUNT; ++k)
        t_knobs.v[k] = g_env_set[k] ? g_env_value[k] : kKnobs[k].dflt;
}

} // namespace csgn
