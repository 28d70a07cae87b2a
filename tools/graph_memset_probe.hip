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

__global__ void k_add_one(unsigned long long *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] += 1ull;
}

// Second question (what the circuits really depend on): is a memset node ORDERED with the kernel nodes around it?
//   E: k_fill(buf, 7) -> hipMemsetAsync(buf, 0) -> k_add_one(buf)       every replay must leave buf[i] == 1
//   F: k_fill(buf, 7) -> k_zero(buf)            -> k_add_one(buf)       (kernel nodes only: the control)
// A memset that runs before the fill leaves 8 + i, one that runs after the add leaves 0, one that is skipped 8 + i.
// Prints the graph's nodes and edges as captured (a missing edge would be a capture problem, not an execution one).
static int order_probe(size_t words, int rounds)
{
    unsigned long long *d_buf, *h_res;
    CHECK(hipMalloc(&d_buf, words * 8));
    CHECK(hipHostMalloc(&h_res, words * 8));
    int bad_total = 0;
    for (int g = 0; g < 2; ++g) {
        hipStream_t cap;
        CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        k_fill<<<1024, 256, 0, cap>>>(d_buf, words, 7);
        if (g == 0)
            CHECK(hipMemsetAsync(d_buf, 0, words * 8, cap));
        else
            k_zero<<<1024, 256, 0, cap>>>(d_buf, words);
        k_add_one<<<1024, 256, 0, cap>>>(d_buf, words);
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamEndCapture(cap, &graph));
        size_t nn = 0, ne = 0;
        CHECK(hipGraphGetNodes(graph, nullptr, &nn));
        CHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CHECK(hipStreamDestroy(cap));
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        int bad = 0;
        unsigned long long first_bad = 0;
        for (int r = 0; r < rounds; ++r) {
            CHECK(hipGraphLaunch(exec, s));
            CHECK(hipMemcpyAsync(h_res, d_buf, words * 8, hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
            bool ok = true;
            for (size_t i = 0; i < words && ok; i += 1021)
                if (h_res[i] != 1ull) {
                    ok = false;
                    if (!bad)
                        first_bad = h_res[i] - (h_res[i] >= 8 ? i : 0);
                }
            bad += ok ? 0 : 1;
        }
        printf("fill -> %-14s -> add one, %zu MiB | %zu nodes, %zu edges | %2d of %d replays wrong%s\n",
               g == 0 ? "MEMSET node" : "zeroing kernel", words >> 17, nn, ne, bad, rounds,
               bad ? (first_bad == 0 ? " (0: the memset ran AFTER the add)" : first_bad == 8 ? " (8 + i: the memset ran before the fill, or not at all)" : " (other)") : "");
        bad_total += bad;
        CHECK(hipStreamDestroy(s));
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
    }
    CHECK(hipFree(d_buf));
    CHECK(hipHostFree(h_res));
    return bad_total;
}

// Third form, the shape of the circuit that showed a wrong bit (tests/test_gpu_parity.py::
// test_circuit_decrypts_a_long_uniform_value_uploaded_just_before_the_run[1], gpurun_out/s3): a long kernel, then a
// 16-byte zero fill of accumulators, then a kernel whose workgroups atomicXor into them, then a reader.
//   [k_fill(big)] -> zero(acc, 16 B) -> k_xor(acc) -> k_copy(acc -> out)      every replay must leave out == expected
__global__ void k_xor(unsigned int *acc, int n_acc)
{
    if (threadIdx.x == 0)
        atomicXor(acc + blockIdx.x % n_acc, 1u << (blockIdx.x / n_acc % 31));
}
__global__ void k_copy4(const unsigned int *acc, unsigned int *out)
{
    if (threadIdx.x < 4)
        out[threadIdx.x] = acc[threadIdx.x];
}
static int accum_probe(int replays, bool null_stream)
{
    unsigned long long *d_big;
    unsigned int *d_acc, *d_out, *h_out;
    const size_t big = (size_t)1 << 22;
    CHECK(hipMalloc(&d_big, big * 8 + 4096));
    d_acc = reinterpret_cast<unsigned int *>(d_big + big) + 6;          // inside the block, 8-byte aligned, not 16
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipHostMalloc(&h_out, 64));
    int bad_total = 0;
    for (int g = 0; g < 2; ++g) {
        hipStream_t cap;
        CHECK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        CHECK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        k_fill<<<1024, 256, 0, cap>>>(d_big, big, 3);
        if (g == 0)
            CHECK(hipMemsetAsync(d_acc, 0, 16, cap));
        else
            k_zero<<<1, 64, 0, cap>>>(reinterpret_cast<unsigned long long *>(d_acc), 2);
        k_xor<<<3 * 31, 256, 0, cap>>>(d_acc, 3);
        k_copy4<<<1, 64, 0, cap>>>(d_acc, d_out);
        hipGraph_t graph;
        hipGraphExec_t exec;
        CHECK(hipStreamEndCapture(cap, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CHECK(hipStreamDestroy(cap));
        hipStream_t s = nullptr;
        if (!null_stream)
            CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        int bad = 0;
        unsigned int seen = 0;
        for (int r = 0; r < replays; ++r) {
            CHECK(hipGraphLaunch(exec, s));
            CHECK(hipMemcpyAsync(h_out, d_out, 16, hipMemcpyDeviceToHost, s));
            CHECK(hipStreamSynchronize(s));
            const bool ok = h_out[0] == 0x7fffffffu && h_out[1] == 0x7fffffffu && h_out[2] == 0x7fffffffu;
            if (!ok && !bad)
                seen = h_out[0] ^ h_out[1] ^ h_out[2];
            bad += ok ? 0 : 1;
        }
        printf("long kernel -> %-14s (16 B) -> atomicXor kernel -> reader, launched on %-12s | %3d of %d replays wrong%s\n",
               g == 0 ? "MEMSET node" : "zeroing kernel", null_stream ? "NULL stream" : "a stream", bad, replays,
               bad ? " (accumulators not zero when the XORs ran, or zeroed after some)" : "");
        (void)seen;
        bad_total += bad;
        if (s)
            CHECK(hipStreamDestroy(s));
        CHECK(hipGraphExecDestroy(exec));
        CHECK(hipGraphDestroy(graph));
    }
    CHECK(hipFree(d_big));
    CHECK(hipFree(d_out));
    CHECK(hipHostFree(h_out));
    return bad_total;
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
    int order_bad = 0;
    for (size_t w : {(size_t)1 << 4, (size_t)1 << 10, (size_t)1 << 17, (size_t)1 << 23})      // 128 B, 8 KiB, 1 MiB, 64 MiB
        order_bad += order_probe(w, rounds);
    printf("mis-ordered replays in all: %d\n", order_bad);
    const int accum_bad = accum_probe(300, true) + accum_probe(300, false);
    printf("wrong accumulator replays in all: %d\n", accum_bad);
    return 0;
}
