// shard_mul.cpp -- BASELINE config 4 as a user of the class API: one process, every visible MI355X,
// a global batch of independent ciphertext pairs through certFHE::ShardedBatch
// (include/certfhe/ShardedBatch.h): contiguous shards, one host thread + one RCCL communicator per
// GPU, and per step
//     a = Enc(bits_a), b = Enc(bits_b)         (fresh ciphertexts, keyed generator, on the owning GPU)
//     c = a * b                                (Ciphertext::operator*, src/Ciphertext.cpp:231-247, per element)
//     c.termCounts()                           (ncclAllGather of one uint64 per pair over xGMI)
//     c.decrypt(key)                           (second all-gather: one byte per pair)
// every gathered bit checked against b1 & b0 computed in the clear.  --fused 1 makes c in ONE kernel
// per GPU (ShardedBatch::encryptProduct -> csgn_encrypt_mul_keyed).  With --terms T > 1 the operands are
// synthetic T-term ciphertexts instead.  Element words are a function of the GLOBAL element index, so
// the digest printed at the end is the same for any GPU count.  No torch, no Python, no MPI.
//
// A failure on any GPU (try --fail-rank R --fail-step S) releases the other GPUs' collectives
// (ncclCommAbort inside ShardGroup) and ends the program with exit code 1 and the failing rank's
// message; --deadline SECONDS is a last-resort watchdog (exit code 3).
//
//   make tools      (g++ -Iinclude tools/shard_mul.cpp -Lcsgn_amd/lib -lcertFHE_shard -lcertFHE -lcsgn_shard -lcsgn_hip)
//   tools/bin/shard_mul [--pairs 1048576] [--terms 1] [--gpus 0=all] [--steps 5] [--nbits 1247] [--fused 0]
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "certfhe/ShardedBatch.h"
#include "csgn_shard.h"

using namespace certFHE;

namespace {

struct Args {
    uint64_t pairs = 1ull << 20, terms = 1, nbits = 1247;
    int gpus = 0, steps = 5, warmup = 1, fused = 0, fail_rank = -1, fail_step = -1, uneven = 0;
    double deadline = 600;
};

// plaintext bits of global pair g (any fixed functions of g will do)
inline unsigned char bit_a(uint64_t g) { return (unsigned char)(((g * 2654435761ull) >> 13) & 1u); }
inline unsigned char bit_b(uint64_t g) { return (unsigned char)(((g * 40503ull + 7u) >> 5) & 1u); }

} // namespace

int main(int argc, char **argv)
{
    Args a;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "--pairs") a.pairs = strtoull(v, 0, 10);
        else if (k == "--terms") a.terms = strtoull(v, 0, 10);
        else if (k == "--nbits") a.nbits = strtoull(v, 0, 10);
        else if (k == "--gpus") a.gpus = atoi(v);
        else if (k == "--steps") a.steps = atoi(v);
        else if (k == "--warmup") a.warmup = atoi(v);
        else if (k == "--fused") a.fused = atoi(v);
        else if (k == "--fail-rank") a.fail_rank = atoi(v);
        else if (k == "--fail-step") a.fail_step = atoi(v);
        else if (k == "--force-uneven") a.uneven = atoi(v);
        else if (k == "--deadline") a.deadline = atof(v);
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    int visible = 0;
    if (csgn_comm_device_count(&visible) != 0 || visible == 0) {
        fprintf(stderr, "shard_mul: no HIP device visible (%s); there is no CPU path\n", csgn_shard_last_error());
        return 1;
    }
    if (a.gpus > visible) {
        fprintf(stderr, "shard_mul: --gpus %d but only %d device(s) visible\n", a.gpus, visible);
        return 1;
    }
    // last resort: nothing below is supposed to block for ever, but a multi-GPU program that has never
    // met its hardware gets a watchdog anyway
    std::atomic<bool> finished(false);
    std::thread watchdog([&] {
        const auto t0 = std::chrono::steady_clock::now();
        while (!finished.load()) {
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > a.deadline) {
                fprintf(stderr, "shard_mul: deadline of %.0f s passed; giving up\n", a.deadline);
                fflush(stderr);
                _exit(3);
            }
        }
    });
    int rc = 1;
    try {
        std::vector<int> devices;
        for (int i = 0; i < a.gpus; ++i)
            devices.push_back(i);
        ShardGroup group(devices);                       // throws without a GPU: there is no CPU path
        if (a.uneven)
            group.forceGroupedBroadcast(true);
        const int world = group.size();
        const uint64_t T = a.terms;
        Context ctx(a.nbits, 16);
        const uint64_t dl = ctx.getDefaultN(), D = ctx.getD();
        SecretKey key(ctx);
        std::vector<uint64_t> k(D);
        for (uint64_t i = 0; i < D; ++i)
            k[i] = (i * (a.nbits / D) + 3) % a.nbits;    // D distinct positions, the same in every run
        key.setKey(k.data(), D);
        const bool fresh = (T == 1);
        std::vector<unsigned char> pa(a.pairs), pb(a.pairs);
        for (uint64_t g = 0; g < a.pairs; ++g) {
            pa[g] = bit_a(g);
            pb[g] = bit_b(g);
        }
        uint64_t digest = 0, counts_sum = 0, counts_bad = 0, bits_bad = 0;
        double seconds = 0;
        for (int s = -a.warmup; s < a.steps; ++s) {
            if (s == 0)
                seconds = 0;
            if (s == a.fail_step && a.fail_rank >= 0)
                group.injectFailure(a.fail_rank);
            const auto t0 = std::chrono::steady_clock::now();
            // ciphertext i draws stream position i of its generator: the same words for any GPU count
            ShardedBatch c = fresh ? (a.fused ? ShardedBatch::encryptProduct(group, key, pa, pb, 1234, 1235)
                                              : ShardedBatch::encrypt(group, key, pa, 1234) * ShardedBatch::encrypt(group, key, pb, 1235))
                                   : ShardedBatch::synthetic(group, ctx, a.pairs, T, 0x43534743 + 1) *
                                         ShardedBatch::synthetic(group, ctx, a.pairs, T, 0x43534743 + 2);
            const std::vector<uint64_t> counts = c.termCounts();
            std::vector<unsigned char> bits;
            if (fresh)
                bits = c.decrypt(key);
            else
                c.synchronize();
            seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s == a.steps - 1) {
                counts_sum = counts_bad = bits_bad = 0;
                for (uint64_t g = 0; g < a.pairs; ++g) {
                    counts_sum += counts[g];
                    counts_bad += (counts[g] != T * T);
                    if (fresh)
                        bits_bad += (bits[g] != (unsigned char)(pa[g] & pb[g]));
                }
                digest = c.digest();
            }
        }
        const double bytes = 8.0 * dl * (2.0 * T + (double)T * T);
        const double mults = (double)a.pairs * a.steps / seconds;
        printf("{\"tool\": \"shard_mul\", \"api\": \"certFHE::ShardedBatch\", \"n_gpus\": %d, \"pairs\": %llu, \"terms\": %llu, "
               "\"n_bits\": %llu, \"steps\": %d, \"fused\": %s, \"seconds\": %.6f, \"mult_per_s\": %.1f, \"algorithmic_GBps\": %.1f, "
               "\"collective\": \"ncclAllGather(term counts, %llu x u64)%s; %s\", "
               "\"gathered_counts_sum\": %llu, \"gathered_counts_wrong\": %llu, \"fresh_ciphertexts\": %s, "
               "\"decrypted_bits_wrong\": %llu, \"products_digest\": \"%016llx\", \"shards\": [",
               world, (unsigned long long)a.pairs, (unsigned long long)T, (unsigned long long)a.nbits, a.steps,
               a.fused ? "true" : "false", seconds, mults, mults * bytes / 1e9, (unsigned long long)a.pairs,
               a.uneven ? " forced to the grouped-broadcast form" : "", group.collective().c_str(),
               (unsigned long long)counts_sum, (unsigned long long)counts_bad, fresh ? "true" : "false",
               (unsigned long long)bits_bad, (unsigned long long)digest);
        {
            ShardedBatch probe = ShardedBatch::synthetic(group, ctx, a.pairs, 1, 1);
            for (int r = 0; r < world; ++r)
                printf("%s[%llu, %llu]", r ? ", " : "", (unsigned long long)probe.shardRange(r).first,
                       (unsigned long long)probe.shardRange(r).second);
        }
        printf("]}\n");
        rc = (counts_bad == 0 && bits_bad == 0 && counts_sum == a.pairs * T * T) ? 0 : 1;
    } catch (const std::exception &e) {
        fprintf(stderr, "shard_mul: FAILED: %s\n", e.what());
        rc = 1;
    }
    finished.store(true);
    watchdog.join();
    fflush(stdout);
    return rc;
}
