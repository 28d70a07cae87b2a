// bench_native.cpp -- bench.py's measurement without torch and without Python: one process, one host
// thread per GPU, the C ABI only (include/csgn_hip.h, include/csgn_shard.h), ONE HIP runtime and the RCCL this
// file's libraries were built against (csgn_comm_init_all: CSGN_COMM_STRICT, no version skew accepted).
//
// One step = `batch` independent Ciphertext x Ciphertext products per GPU (N=1247, 1024 x 1024 terms) through
// csgn_mul_uniform into a `slots`-slot output arena, then -- with more than one GPU, or --force-collective 1 --
// csgn_shard_product_counts + csgn_comm_gather_counts (ncclAllGather of one uint64 per pair) on the same stream.
// W warm-up steps, then exactly K steps between two thread barriers with a stream synchronise on both sides;
// the slowest rank's time counts; rank 0's kernel time comes from HIP events on its launch stream.  Prints ONE
// JSON line with bench.py's keys.  After the clock: every arena slot of rank 0 is decrypted under random short
// keys and compared with Dec(L) & Dec(R) computed from the operands, and the gathered counts are checked.
//
//   make tools
//   tools/bin/bench_native [--gpus 1] [--steps 5] [--warmup 1] [--batch 65536] [--slots 128] [--terms 1024]
//                          [--force-collective 0]
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "csgn_hip.h"
#include "csgn_shard.h"

namespace {

const uint64_t kN = 1247, kD = 16, kSeed = 0x43534743;
const double kPeak = 8.0e12;

struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t round = 0;
    explicit Barrier(int count) : n(count) {}
    void wait()
    {
        std::unique_lock<std::mutex> g(m);
        const uint64_t r = round;
        if (++waiting == n) {
            waiting = 0;
            ++round;
            cv.notify_all();
        } else {
            cv.wait(g, [&] { return round != r; });
        }
    }
};

struct Rank {
    int rank = 0;
    double seconds = 0;
    std::vector<float> mul_ms, step_ms;
    std::string error;
    uint64_t slot_mismatches = 0, one_bits = 0, bad_counts = 0;
};

#define TRY(x)                                                                              \
    do {                                                                                    \
        const int rc_ = (x);                                                                \
        if (rc_ != CSGN_OK) {                                                               \
            me.error = std::string(#x) + ": " + csgn_last_error() + " / " + csgn_shard_last_error(); \
            goto fail;                                                                      \
        }                                                                                   \
    } while (0)

uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace

int main(int argc, char **argv)
{
    int gpus = 1, steps = 5, warmup = 1, force = 0;
    uint64_t batch = 65536, slots = 128, T = 1024;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "--gpus") gpus = atoi(v);
        else if (k == "--steps") steps = atoi(v);
        else if (k == "--warmup") warmup = atoi(v);
        else if (k == "--batch") batch = strtoull(v, 0, 10);
        else if (k == "--slots") slots = strtoull(v, 0, 10);
        else if (k == "--terms") T = strtoull(v, 0, 10);
        else if (k == "--force-collective") force = atoi(v);
        else { fprintf(stderr, "bench_native: unknown option %s\n", k.c_str()); return 2; }
    }
    int visible = 0;
    if (csgn_comm_device_count(&visible) != 0 || visible < gpus || gpus < 1) {
        fprintf(stderr, "bench_native: --gpus %d requested, %d device(s) visible; refusing to run on fewer\n", gpus, visible);
        return 2;
    }
    slots = std::min(slots, batch);
    const int world = gpus;
    const bool use_comm = world > 1 || force;
    const uint64_t dl = csgn_default_len(kN), opw = T * dl, prodw = T * T * dl;
    const uint64_t launches_per_step = (batch + slots - 1) / slots;
    std::vector<csgn_comm *> comms(world, nullptr);
    char rccl_path[512] = "";
    int rt = 0, hd = 0;
    if (use_comm) {
        if (csgn_comm_init_all(world, nullptr, comms.data()) != 0) {      // STRICT: the RCCL this was built against
            fprintf(stderr, "bench_native: csgn_comm_init_all: %s\n", csgn_shard_last_error());
            return 1;
        }
        csgn_comm_rccl_info(&rt, &hd, rccl_path, sizeof(rccl_path));
    }
    Barrier barrier(world);
    std::vector<Rank> ranks(world);
    std::vector<std::thread> threads;
    for (int r = 0; r < world; ++r) {
        threads.emplace_back([&, r] {
            Rank &me = ranks[r];
            me.rank = r;
            void *stream = nullptr, *L = nullptr, *R = nullptr, *arena = nullptr, *counts = nullptr, *gathered = nullptr;
            std::vector<void *> ev;
            bool in_barrier_protocol = true;
            uint64_t lo = 0, hi = 0;
            csgn_shard_range((uint64_t)world * batch, r, world, &lo, &hi);
            TRY(csgn_init(r));
            TRY(csgn_stream_create(&stream));
            TRY(csgn_malloc(&L, batch * opw * 8));
            TRY(csgn_malloc(&R, batch * opw * 8));
            TRY(csgn_malloc(&arena, slots * prodw * 8));
            TRY(csgn_malloc(&counts, batch * 8));
            TRY(csgn_malloc(&gathered, (uint64_t)world * batch * 8));
            // operand words are a function of the GLOBAL pair index
            TRY(csgn_synth_fill(kSeed + 1, kN, lo * opw, batch * opw, (uint64_t *)L, stream));
            TRY(csgn_synth_fill(kSeed + 2, kN, lo * opw, batch * opw, (uint64_t *)R, stream));
            for (int k = 0; k < 4 * steps; ++k) {
                void *e = nullptr;
                TRY(csgn_event_create(&e));
                ev.push_back(e);
            }
            for (int k = -warmup; k < steps; ++k) {
                if (k == 0) {
                    TRY(csgn_stream_sync(stream));
                    barrier.wait();
                    me.seconds = -std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                }
                if (k >= 0) {
                    TRY(csgn_event_record(ev[4 * k + 0], stream));
                    TRY(csgn_event_record(ev[4 * k + 1], stream));
                }
                TRY(csgn_mul_uniform(kN, batch, T, T, (const uint64_t *)L, (const uint64_t *)R, (uint64_t *)arena, slots, stream));
                if (k >= 0)
                    TRY(csgn_event_record(ev[4 * k + 2], stream));
                if (use_comm) {
                    TRY(csgn_shard_product_counts(batch, nullptr, nullptr, T, T, (uint64_t *)counts, stream));
                    TRY(csgn_comm_gather_counts(comms[r], (const uint64_t *)counts, (uint64_t)world * batch, (uint64_t *)gathered, stream));
                }
                if (k >= 0)
                    TRY(csgn_event_record(ev[4 * k + 3], stream));
            }
            TRY(csgn_stream_sync(stream));
            barrier.wait();
            me.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            in_barrier_protocol = false;
            if (use_comm)
                TRY(csgn_comm_check(comms[r]));
            for (int k = 0; k < steps; ++k) {
                float a = 0, b = 0;
                TRY(csgn_event_elapsed_ms(ev[4 * k + 1], ev[4 * k + 2], &a));
                TRY(csgn_event_elapsed_ms(ev[4 * k + 0], ev[4 * k + 3], &b));
                me.mul_ms.push_back(a);
                me.step_ms.push_back(b);
            }
            // ---- off the clock: the arena against the operands (rank 0), the gathered counts (every rank) ----
            if (use_comm) {
                std::vector<uint64_t> h((uint64_t)world * batch);
                TRY(csgn_memcpy_d2h(h.data(), gathered, h.size() * 8, stream));
                TRY(csgn_stream_sync(stream));
                for (uint64_t v : h)
                    me.bad_counts += v != T * T;
            }
            if (r == 0) {
                const uint64_t first_of_last = (launches_per_step - 1) * slots, live = batch - first_of_last;
                void *lsel = nullptr, *rsel = nullptr, *mask = nullptr, *bits = nullptr, *scratch = nullptr;
                TRY(csgn_malloc(&lsel, slots * opw * 8));
                TRY(csgn_malloc(&rsel, slots * opw * 8));
                for (uint64_t s = 0; s < slots; ++s) {                // the pair that wrote slot s last
                    const uint64_t p = s < live ? first_of_last + s : first_of_last - slots + s;
                    TRY(csgn_memcpy_d2d((uint64_t *)lsel + s * opw, (uint64_t *)L + p * opw, opw * 8, stream));
                    TRY(csgn_memcpy_d2d((uint64_t *)rsel + s * opw, (uint64_t *)R + p * opw, opw * 8, stream));
                }
                const size_t sa = csgn_decrypt_scratch_bytes(slots, slots * T * T);
                const size_t sb = csgn_decrypt_combined_scratch_bytes(slots, T, T);
                TRY(csgn_malloc(&scratch, std::max(sa, sb) + 256));
                TRY(csgn_malloc(&mask, dl * 8));
                TRY(csgn_malloc(&bits, 2 * slots));
                uint64_t seed = kSeed ^ 0x5EED;
                for (int trial = 0; trial < 6; ++trial) {
                    uint64_t key[3];
                    const uint64_t d = 2 + trial % 2;
                    for (uint64_t i = 0; i < d; ++i)
                        key[i] = splitmix(seed) % kN;
                    std::vector<uint64_t> hm(dl);
                    TRY(csgn_key_mask(kN, key, d, hm.data()));
                    TRY(csgn_memcpy_h2d(mask, hm.data(), dl * 8, stream));
                    TRY(csgn_stream_sync(stream));
                    TRY(csgn_decrypt_uniform(kN, slots, T * T, (const uint64_t *)arena, (const uint64_t *)mask, (uint8_t *)bits, scratch, stream));
                    TRY(csgn_decrypt_product_uniform(kN, slots, T, T, (const uint64_t *)lsel, (const uint64_t *)rsel, (const uint64_t *)mask,
                                                     (uint8_t *)bits + slots, scratch, stream));
                    std::vector<uint8_t> hb(2 * slots);
                    TRY(csgn_memcpy_d2h(hb.data(), bits, 2 * slots, stream));
                    TRY(csgn_stream_sync(stream));
                    for (uint64_t s = 0; s < slots; ++s) {
                        me.slot_mismatches += hb[s] != hb[slots + s];
                        me.one_bits += hb[slots + s];
                    }
                }
                csgn_free(lsel); csgn_free(rsel); csgn_free(mask); csgn_free(bits); csgn_free(scratch);
            }
            csgn_free(L); csgn_free(R); csgn_free(arena); csgn_free(counts); csgn_free(gathered);
            for (void *e : ev)
                csgn_event_destroy(e);
            csgn_stream_destroy(stream);
            return;
        fail:
            // a failing rank releases its peers (their collectives and the thread barrier) and the job ends non-zero
            if (use_comm)
                for (csgn_comm *c : comms)
                    if (c)
                        csgn_comm_abort(c);
            fprintf(stderr, "bench_native: rank %d FAILED: %s\n", r, me.error.c_str());
            fflush(stderr);
            if (in_barrier_protocol)
                _Exit(1);                                          // peers may be parked in the thread barrier
        });
    }
    for (auto &t : threads)
        t.join();
    for (csgn_comm *c : comms)
        if (c)
            csgn_comm_destroy(c);
    double elapsed = 0;
    bool ok = true;
    for (const Rank &rk : ranks) {
        ok = ok && rk.error.empty() && rk.bad_counts == 0;
        elapsed = std::max(elapsed, rk.seconds);
    }
    const Rank &r0 = ranks[0];
    ok = ok && r0.slot_mismatches == 0 && r0.one_bits > 0;
    if (!ok || r0.mul_ms.empty()) {
        fprintf(stderr, "bench_native: run or verification failed (slot mismatches %llu, one-bits %llu)\n",
                (unsigned long long)r0.slot_mismatches, (unsigned long long)r0.one_bits);
        return 1;
    }
    const double bytes_per_mul = 8.0 * dl * (2.0 * T + (double)T * T);
    double kernel_ms = 0;
    std::vector<float> mul_sorted = r0.mul_ms, step_sorted = r0.step_ms;
    for (float v : r0.mul_ms)
        kernel_ms += v;
    std::sort(mul_sorted.begin(), mul_sorted.end());
    std::sort(step_sorted.begin(), step_sorted.end());
    const double n_launches = (double)launches_per_step * steps;
    const double avg_launch_s = kernel_ms / 1e3 / n_launches;
    const double pairs_per_launch = (double)batch / launches_per_step;
    const double achieved = pairs_per_launch * bytes_per_mul / avg_launch_s;
    const double value = (double)world * batch * steps / elapsed;
    char collective[900];
    if (use_comm)
        snprintf(collective, sizeof(collective),
                 "csgn_comm_gather_counts -> ncclAllGather(result term counts) [RCCL %d.%d.%d from %s, header %d.%d.%d, "
                 "CSGN_COMM_STRICT; one process, one host thread per GPU, no torch]",
                 rt / 10000, (rt / 100) % 100, rt % 100, rccl_path, hd / 10000, (hd / 100) % 100, hd % 100);
    else
        snprintf(collective, sizeof(collective), "none");
    printf("{\"metric\": \"ciphertext-mults/sec (N=%llu, %llu-term operands)\", \"value\": %.3f, \"unit\": \"mult/s\", "
           "\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, "
           "\"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"u64\", \"data\": \"synthetic\", "
           "\"config\": {\"workload\": \"Ciphertext*Ciphertext all-pairs AND, Context(%llu,%llu), %llux%llu terms, "
           "batch=%llu pairs/GPU streamed through a %llu-slot output arena (%.1f GiB)\", \"driver\": \"tools/bench_native "
           "(C++ over the C ABI, thread per GPU)\", \"n_bits\": %llu, \"terms\": %llu, \"batch_per_gpu\": %llu, "
           "\"arena_slots\": %llu, \"pairs_per_launch\": %.1f, \"seed\": %llu, \"bytes_per_mult\": %.0f, "
           "\"collective\": \"%s\", \"verified_slots\": %llu, \"verification\": \"%llu slots x 6 random keys (d=2,3): "
           "Dec(slot) == Dec(L)&Dec(R) from the operands, %llu mismatches, %llu one-bits\", "
           "\"step_ms_rank0\": {\"median\": %.4f, \"min\": %.4f, \"max\": %.4f, \"n\": %d}}, "
           "\"roofline\": {\"bound\": \"hbm\", \"achieved\": %.2f, \"peak\": %.1f, \"unit\": \"GB/s\", \"frac\": %.5f, "
           "\"traffic\": null, \"algorithmic_bytes_per_launch\": %.0f, \"kernel\": \"%s\", \"avg_launch_ms\": %.5f, "
           "\"launch_ms\": {\"median\": %.5f, \"min\": %.5f, \"max\": %.5f}, \"launches\": %.0f}}\n",
           (unsigned long long)kN, (unsigned long long)T, value, world, steps, warmup, elapsed / steps * 1e3,
           (unsigned long long)kN, (unsigned long long)kD, (unsigned long long)T, (unsigned long long)T,
           (unsigned long long)batch, (unsigned long long)slots, slots * prodw * 8 / 1073741824.0,
           (unsigned long long)kN, (unsigned long long)T, (unsigned long long)batch, (unsigned long long)slots,
           pairs_per_launch, (unsigned long long)kSeed, bytes_per_mul, collective, (unsigned long long)slots,
           (unsigned long long)slots, (unsigned long long)r0.slot_mismatches, (unsigned long long)r0.one_bits,
           step_sorted[step_sorted.size() / 2], step_sorted.front(), step_sorted.back(), steps,
           achieved / 1e9, kPeak / 1e9, achieved / kPeak, pairs_per_launch * bytes_per_mul,
           csgn_mul_uniform_kernel(kN, batch, T, T), avg_launch_s * 1e3,
           mul_sorted[mul_sorted.size() / 2] / launches_per_step, mul_sorted.front() / launches_per_step,
           mul_sorted.back() / launches_per_step, n_launches);
    return 0;
}
