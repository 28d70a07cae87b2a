// bench_mul.cpp -- the headline measurement from plain C++ over the C ABI (no Python, no torch):
// batch independent 1024x1024-term products at N=1247 streamed through a fixed output arena,
// timed with HIP events on the launch stream.  Cross-checks bench.py.
//   g++ -std=c++11 -O2 -Iinclude tools/bench_mul.cpp -Lcsgn_amd/lib -lcsgn_hip -Wl,-rpath,$PWD/csgn_amd/lib -o tools/bin/bench_mul
//   tools/bin/bench_mul [batch=8192] [slots=128] [steps=3]
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "csgn_hip.h"

#define OK(x) do { int rc_ = (x); if (rc_ != CSGN_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, csgn_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint64_t n = 1247, T = 1024, dl = csgn_default_len(n);
    const uint64_t batch = argc > 1 ? strtoull(argv[1], 0, 10) : 8192;
    const uint64_t slots = argc > 2 ? strtoull(argv[2], 0, 10) : 128;
    const int steps = argc > 3 ? atoi(argv[3]) : 3;
    OK(csgn_init(0));
    void *stream = nullptr, *e0 = nullptr, *e1 = nullptr;
    OK(csgn_stream_create(&stream));
    OK(csgn_event_create(&e0));
    OK(csgn_event_create(&e1));
    void *L = nullptr, *R = nullptr, *arena = nullptr;
    const uint64_t opw = batch * T * dl, prodw = T * T * dl;
    OK(csgn_malloc(&L, opw * 8));
    OK(csgn_malloc(&R, opw * 8));
    OK(csgn_malloc(&arena, slots * prodw * 8));
    OK(csgn_synth_fill(0x43534743 + 1, n, 0, opw, (uint64_t *)L, stream));
    OK(csgn_synth_fill(0x43534743 + 2, n, 0, opw, (uint64_t *)R, stream));
    OK(csgn_mul_uniform(n, batch, T, T, (uint64_t *)L, (uint64_t *)R, (uint64_t *)arena, slots, stream));   // warm-up
    OK(csgn_stream_sync(stream));
    OK(csgn_event_record(e0, stream));
    for (int s = 0; s < steps; ++s)
        OK(csgn_mul_uniform(n, batch, T, T, (uint64_t *)L, (uint64_t *)R, (uint64_t *)arena, slots, stream));
    OK(csgn_event_record(e1, stream));
    float ms = 0;
    OK(csgn_event_elapsed_ms(e0, e1, &ms));
    const double mults = (double)batch * steps / (ms * 1e-3);
    const double bytes = 8.0 * dl * (2 * T + (double)T * T);
    printf("{\"mult_per_s\": %.1f, \"algorithmic_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"batch\": %llu, \"slots\": %llu, \"steps\": %d, \"ms\": %.3f}\n",
           mults, mults * bytes / 1e9, mults * bytes / 8e12, (unsigned long long)batch, (unsigned long long)slots, steps, ms);
    csgn_free(L); csgn_free(R); csgn_free(arena);
    csgn_event_destroy(e0); csgn_event_destroy(e1); csgn_stream_destroy(stream);
    return 0;
}
