// dev probe: which C-ABI calls disturb libc's rand() stream?
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "csgn_hip.h"
static int peek() { srand(12345); return 0; }
static void report(const char *what) { int r = rand(); srand(12345); int want = rand(); srand(12345); printf("%-28s next rand()=%d (undisturbed %d) %s\n", what, r, want, r == want ? "" : "<-- DISTURBED"); }
int main() {
    srand(12345); report("baseline");
    csgn_init(0); report("csgn_init");
    void *d = nullptr; csgn_malloc(&d, 4096); report("csgn_malloc #1");
    uint64_t h[16] = {0}; csgn_memcpy_h2d(d, h, 128, nullptr); csgn_stream_sync(nullptr); report("h2d #1");
    csgn_synth_fill(1, 1247, 0, 16, (uint64_t *)d, nullptr); csgn_stream_sync(nullptr); report("kernel #1");
    csgn_memcpy_d2h(h, d, 128, nullptr); report("d2h #1");
    void *d2 = nullptr; csgn_malloc(&d2, 1 << 20); report("csgn_malloc #2");
    csgn_memcpy_h2d(d2, h, 128, nullptr); csgn_stream_sync(nullptr); report("h2d #2");
    csgn_synth_fill(1, 1247, 0, 16, (uint64_t *)d2, nullptr); csgn_stream_sync(nullptr); report("kernel #2");
    csgn_digest((uint64_t *)d2, 16, 0, (uint64_t *)d, nullptr); csgn_stream_sync(nullptr); report("other kernel");
    csgn_memset(d2, 0, 128, nullptr); csgn_stream_sync(nullptr); report("memset");
    csgn_memcpy_d2h(h, d2, 128, nullptr); report("d2h #2");
    csgn_free(d2); report("free");
    csgn_free(d); report("free #2");
    for (int i = 0; i < 3; ++i) { csgn_malloc(&d, 160); csgn_memcpy_h2d(d, h, 128, nullptr); csgn_stream_sync(nullptr); csgn_memcpy_d2h(h, d, 128, nullptr); csgn_free(d); report("cycle"); }
    return 0;
}
