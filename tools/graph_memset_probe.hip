// graph_memset_probe.hip -- does a hipGraph with a MEMSET node start before work enqueued earlier on its launch stream?
// (round 4 saw the inputs of a circuit with a compaction node "read as zeros" and replaced every captured
// hipMemsetAsync by a kernel without finding the cause.)  Two graphs of the same shape are captured from a side stream:
//   A: hipMemsetAsync(scratch) -> k_plus_one(in -> out)        (memset node + kernel node)
//   B: k_zero(scratch)         -> k_plus_one(in -> out)        (kernel nodes only)
//   C: as A, but input and scratch are two ranges of ONE allocation (input at offset 0, scratch behind it) -- how a
//      circuit's block is laid out: does the memset node clear the range it was given, or the start of the allocation?
//   D: as C with the scratch range cleared by a 2-byte-wide memset of odd length (hipMemsetD16Async is not what the
//      library used; this is hipMemsetAsync of a length that is not a multiple of 4, the decrypt's per-batch partial words)
// and each is replayed ROUNDS times behind an upload of fresh input on the SAME stream, for every combination of launch
// stream (the legacy NULL stream, a blocking created stream, a non-blocking created stream) and upload (H2D from pinned
// memory, H2D from pageable memory, D2D).  A result that does not match the input uploaded just before the launch means
// the graph ran before the copy landed.  One short run; nothing here faults.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));     \
            exit(2);                                                                              \
        }                                                                                         \
    } while (0)

__global__ void k_plus_one(const unsigned long long *in, unsigned long long *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i] + 1ull;
}
__global__ void k_zero(unsigned long long *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0ull;
}
__global__ void k_fill(unsigned long long *p, size_t n, unsigned long long v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v + i;
}

int main(int argc, char **argv)
{
    const size_t words = (argc > 1 ? (size_t)atoll(argv[1]) : 64) << 17;      // MiB -> u64 words
    const int rounds = argc > 2 ? atoi(argv[2]) : 12;
    const size_t bytes = words * 8, scratch_words = 1 << 16;
    unsigned long long *d_in0, *d_out, *d_scratch, *d_stage, *h_pin, *h_res;
    unsigned long long *d_block;
    CHECK(hipMalloc(&d_block, bytes + scratch_words * 8 + 256));
    CHECK(hipMalloc(&d_in0, bytes));
    CHECK(hipMalloc(&d_out, bytes));
    CHECK(hipMalloc(&d_stage, bytes));
    CHECK(hipMalloc(&d_scratch, scratch_words * 8));
    CHECK(hipHostMalloc(&h_pin, bytes));
    CHECK(hipHostMalloc(&h_res, bytes));
    std::vector<unsigned long long> h_page(words);

    hipGraphExec_t exec[4];
    for (int g = 0; g < 4; ++g) {
        hipStream_t cap;
        CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        if (g == 0)
            CHECK(hipMemsetAsync(d_scratch, 0, scratch_words * 8, cap));
        else if (g == 1)
            k_zero<<<64, 256, 0, cap>>>(d_scratch, scratch_words);
        else if (g == 2)
            CHECK(hipMemsetAsync(d_block + words, 0, scratch_words * 8, cap));
        else
            CHECK(hipMemsetAsync(reinterpret_cast<char *>(d_block + words) + 2, 0, scratch_words * 8 - 3, cap));
        k_plus_one<<<4096, 256, 0, cap>>>(g >= 2 ? d_block : d_in0, d_out, words);
        hipGraph_t graph;
        CHECK(hipStreamEndCapture(cap, &graph));
        CHECK(hipGraphInstantiate(&exec[g], graph, nullptr, nullptr, 0));
        CHECK(hipStreamDestroy(cap));
    }
    hipStream_t blocking, nonblocking;
    CHECK(hipStreamCreate(&blocking));
    CHECK(hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking));
    hipStream_t streams[3] = {nullptr, blocking, nonblocking};
    const char *stream_name[3] = {"NULL stream", "created (blocking)", "created (non-blocking)"};
    const char *copy_name[3] = {"H2D pinned", "H2D pageable", "D2D"};
    const char *graph_name[4] = {"memset node + kernel", "kernel nodes only", "memset INSIDE the block", "odd memset inside block"};
    int bad_total = 0;
    unsigned long long tag = 1;
    for (int g = 0; g < 4; ++g)
        for (int si = 0; si < 3; ++si)
            for (int ci = 0; ci < 3; ++ci) {
                hipStream_t s = streams[si];
                unsigned long long *d_in = g >= 2 ? d_block : d_in0;
                int bad = 0, zeros = 0;
                for (int r = 0; r < rounds; ++r) {
                    tag += 0x100000001ull;
                    if (ci == 0) {
                        for (size_t i = 0; i < words; i += 4096)
                            h_pin[i] = tag + i;
                        h_pin[words - 1] = tag + words - 1;
                        CHECK(hipMemcpyAsync(d_in, h_pin, bytes, hipMemcpyHostToDevice, s));
                    } else if (ci == 1) {
                        for (size_t i = 0; i < words; i += 4096)
                            h_page[i] = tag + i;
                        h_page[words - 1] = tag + words - 1;
                        CHECK(hipMemcpyAsync(d_in, h_page.data(), bytes, hipMemcpyHostToDevice, s));
                    } else {
                        k_fill<<<4096, 256, 0, s>>>(d_stage, words, tag);
                        CHECK(hipMemcpyAsync(d_in, d_stage, bytes, hipMemcpyDeviceToDevice, s));
                    }
                    CHECK(hipGraphLaunch(exec[g], s));
                    CHECK(hipMemcpyAsync(h_res, d_out, bytes, hipMemcpyDeviceToHost, s));
                    CHECK(hipStreamSynchronize(s));
                    // sampled words (every 4096th and the last): want tag + i + 1
                    bool ok = true, zero = false;
                    for (size_t i = 0; i < words && ok; i += 4096)
                        if (h_res[i] != tag + i + 1) {
                            ok = false;
                            zero = h_res[i] == 1;
                        }
                    if (h_res[words - 1] != tag + words)
                        ok = false;
                    bad += ok ? 0 : 1;
                    zeros += zero ? 1 : 0;
                }
                printf("%-24s | launch on %-24s | upload %-13s | %2d of %d runs read stale input%s\n", graph_name[g],
                       stream_name[si], copy_name[ci], bad, rounds, zeros ? " (zeros)" : "");
                bad_total += bad;
            }
    printf("stale runs in all: %d\n", bad_total);
    return 0;
}
