// shard_mul.cpp -- the native multi-GPU driver of the hot path (BASELINE
// config 4): one process, one host thread per visible MI355X, a global batch of independent
// ciphertext pairs cut into contiguous shards (csgn_shard_range), every shard multiplied on its own
// GPU through the C ABI (csgn_mul_uniform), and ONE exchange per step: the RCCL all-gather of the
// per-pair result term counts (csgn_comm_gather_counts -> ncclAllGather over xGMI).  No torch, no
// Python, no MPI.  Operand words are a function of the GLOBAL pair index, so the digests printed
// at the end are the same for any GPU count.
// With --terms 1 (the default) the operands are FRESH ciphertexts, as in config 4: every step
// encrypts both operands of the shard on its GPU with the keyed generator (csgn_encrypt_keyed,
// first_ciphertext = the shard's first global pair index), multiplies, decrypts, and gathers the term
// counts AND the decrypted bits; every gathered bit is checked against b1 & b0 computed in the clear.
// With --terms T > 1 the operands are synthetic T-term ciphertexts (csgn_synth_fill).
//
//   hipcc -O2 -std=c++17 -Iinclude tools/shard_mul.cpp -Lcsgn_amd/lib -lcsgn_hip -lcsgn_shard \
//         -Wl,-rpath,$PWD/csgn_amd/lib -lpthread -o tools/bin/shard_mul     (or: make tools)
//   tools/bin/shard_mul [--pairs 1048576] [--terms 1] [--gpus 0=all] [--steps 5] [--slots 0] [--nbits 1247]
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "csgn_hip.h"
#include "csgn_shard.h"

namespace {

struct Args {
    uint64_t pairs = 1ull << 20, terms = 1, slots = 0, nbits = 1247;
    int gpus = 0, steps = 5, warmup = 1;
};

struct RankResult {
    int rc = 0;
    std::string error;
    double seconds = 0;           // timed region of this rank
    uint64_t out_digest = 0;      // digest of the products still in this rank's arena
    uint64_t counts_sum = 0;      // sum of ALL gathered counts (every rank must see the same)
    uint64_t counts_bad = 0;      // gathered counts that are not t1*t2
    uint64_t bits_bad = 0;        // gathered decryptions that are not b1 & b0 (fresh mode)
    uint64_t lo = 0, hi = 0;
};

// plaintext bits of global pair g (any fixed functions of g will do)
inline unsigned char bit_a(uint64_t g) { return (unsigned char)(((g * 2654435761ull) >> 13) & 1u); }
inline unsigned char bit_b(uint64_t g) { return (unsigned char)(((g * 40503ull + 7u) >> 5) & 1u); }

#define TRY(x)                                                                              \
    do {                                                                                    \
        int rc_ = (x);                                                                      \
        if (rc_ != CSGN_OK) {                                                               \
            res.rc = rc_;                                                                   \
            res.error = std::string(#x) + ": " + csgn_last_error() + " / " + csgn_shard_last_error(); \
            failed.store(true);                                                             \
            return;                                                                         \
        }                                                                                   \
    } while (0)

void rank_main(const Args &a, csgn_comm *comm, RankResult &res, std::atomic<bool> &failed)
{
    const int rank = csgn_comm_rank(comm), world = csgn_comm_world(comm);
    const uint64_t n = a.nbits, T = a.terms, dl = csgn_default_len(n);
    TRY(csgn_init(csgn_comm_device(comm)));
    void *stream = csgn_comm_stream(comm);
    uint64_t lo = 0, hi = 0;
    TRY(csgn_shard_range(a.pairs, rank, world, &lo, &hi));
    res.lo = lo;
    res.hi = hi;
    const uint64_t mine = hi - lo;
    const uint64_t slots = (a.slots == 0 || a.slots > mine) ? mine : a.slots;
    const uint64_t opw = mine * T * dl, prodw = T * T * dl;
    void *L = nullptr, *R = nullptr, *arena = nullptr, *counts = nullptr, *all = nullptr, *dig = nullptr;
    TRY(csgn_malloc(&L, opw * 8));
    TRY(csgn_malloc(&R, opw * 8));
    TRY(csgn_malloc(&arena, slots * prodw * 8));
    TRY(csgn_malloc(&counts, (mine ? mine : 1) * 8));
    TRY(csgn_malloc(&all, a.pairs * 8));
    TRY(csgn_malloc(&dig, 8));
    const bool fresh = (T == 1);
    // fresh mode: key, plaintext bits of the shard, decrypt scratch, gathered bits
    const uint64_t D = 16;
    void *d_key = nullptr, *d_mask = nullptr, *d_pa = nullptr, *d_pb = nullptr, *d_bits = nullptr, *d_allbits = nullptr,
         *d_scratch = nullptr;
    csgn_rng rng_a, rng_b;
    if (fresh) {
        std::vector<uint64_t> key(D), mask(dl);
        for (uint64_t i = 0; i < D; ++i)
            key[i] = (i * (n / D) + 3) % n;                   // D distinct positions
        TRY(csgn_key_mask(n, key.data(), D, mask.data()));
        std::vector<unsigned char> pa(mine ? mine : 1), pb(mine ? mine : 1);
        for (uint64_t i = 0; i < mine; ++i) {
            pa[i] = bit_a(lo + i);
            pb[i] = bit_b(lo + i);
        }
        TRY(csgn_malloc(&d_key, D * 8));
        TRY(csgn_malloc(&d_mask, dl * 8));
        TRY(csgn_malloc(&d_pa, pa.size()));
        TRY(csgn_malloc(&d_pb, pb.size()));
        TRY(csgn_malloc(&d_bits, mine ? mine : 1));
        TRY(csgn_malloc(&d_allbits, a.pairs));
        TRY(csgn_malloc(&d_scratch, csgn_decrypt_scratch_bytes(mine, mine)));
        TRY(csgn_memcpy_h2d(d_key, key.data(), D * 8, stream));
        TRY(csgn_memcpy_h2d(d_mask, mask.data(), dl * 8, stream));
        TRY(csgn_memcpy_h2d(d_pa, pa.data(), pa.size(), stream));
        TRY(csgn_memcpy_h2d(d_pb, pb.data(), pb.size(), stream));
        TRY(csgn_stream_sync(stream));
        TRY(csgn_rng_from_seed(&rng_a, 1234, 8));             // reproducible streams: the digest is checkable
        TRY(csgn_rng_from_seed(&rng_b, 1235, 8));
    } else {
        // operands = f(global pair index): word w of the global operand stream
        TRY(csgn_synth_fill(0x43534743 + 1, n, lo * T * dl, opw, (uint64_t *)L, stream));
        TRY(csgn_synth_fill(0x43534743 + 2, n, lo * T * dl, opw, (uint64_t *)R, stream));
    }

    auto step = [&]() -> int {
        if (fresh) {
            // ciphertext i of the shard draws stream position lo + i: the same words for any GPU count
            if (int rc = csgn_encrypt_keyed(n, D, mine, lo, (const uint8_t *)d_pa, (const uint64_t *)d_key,
                                            (const uint64_t *)d_mask, &rng_a, (uint64_t *)L, stream))
                return rc;
            if (int rc = csgn_encrypt_keyed(n, D, mine, lo, (const uint8_t *)d_pb, (const uint64_t *)d_key,
                                            (const uint64_t *)d_mask, &rng_b, (uint64_t *)R, stream))
                return rc;
        }
        if (int rc = csgn_mul_uniform(n, mine, T, T, (const uint64_t *)L, (const uint64_t *)R, (uint64_t *)arena, slots, stream))
            return rc;
        if (int rc = csgn_shard_product_counts(mine, nullptr, nullptr, T, T, (uint64_t *)counts, stream))
            return rc;
        if (int rc = csgn_comm_gather_counts(comm, (const uint64_t *)counts, a.pairs, (uint64_t *)all, stream))
            return rc;
        if (fresh && slots == mine) {
            if (int rc = csgn_decrypt_uniform(n, mine, 1, (const uint64_t *)arena, (const uint64_t *)d_mask,
                                              (uint8_t *)d_bits, d_scratch, stream))
                return rc;
            return csgn_comm_gather_bytes(comm, (const uint8_t *)d_bits, a.pairs, (uint8_t *)d_allbits, stream);
        }
        return CSGN_OK;
    };
    for (int w = 0; w < a.warmup; ++w)
        TRY(step());
    TRY(csgn_comm_barrier(comm, stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < a.steps; ++s)
        TRY(step());
    TRY(csgn_comm_barrier(comm, stream));
    res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // what every rank received
    std::vector<uint64_t> h_all(a.pairs);
    TRY(csgn_memcpy_d2h(h_all.data(), all, a.pairs * 8, stream));
    for (uint64_t c : h_all) {
        res.counts_sum += c;
        res.counts_bad += (c != T * T);
    }
    if (fresh && slots == mine) {                    // every rank holds every pair's decrypted bit
        std::vector<unsigned char> h_bits(a.pairs);
        TRY(csgn_memcpy_d2h(h_bits.data(), d_allbits, a.pairs, stream));
        for (uint64_t g = 0; g < a.pairs; ++g)
            res.bits_bad += (h_bits[g] != (unsigned char)(bit_a(g) & bit_b(g)));
    }
    // digest of the products this rank still holds (the last `slots` pairs of its shard), indexed by
    // their GLOBAL word position so that the sum over ranks does not depend on the GPU count when
    // slots == 0
    if (mine) {
        const uint64_t first_kept = ((mine - 1) / slots) * slots;        // first pair of the last launch
        const uint64_t kept = mine - first_kept;
        TRY(csgn_memset(dig, 0, 8, stream));
        TRY(csgn_digest((const uint64_t *)arena, kept * prodw, (lo + first_kept) * prodw, (uint64_t *)dig, stream));
        TRY(csgn_memcpy_d2h(&res.out_digest, dig, 8, stream));
    }
    csgn_free(L);
    csgn_free(R);
    csgn_free(arena);
    csgn_free(counts);
    csgn_free(all);
    csgn_free(dig);
    csgn_free(d_key);
    csgn_free(d_mask);
    csgn_free(d_pa);
    csgn_free(d_pb);
    csgn_free(d_bits);
    csgn_free(d_allbits);
    csgn_free(d_scratch);
}

} // namespace

int main(int argc, char **argv)
{
    Args a;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const char *v = argv[i + 1];
        if (k == "--pairs") a.pairs = strtoull(v, 0, 10);
        else if (k == "--terms") a.terms = strtoull(v, 0, 10);
        else if (k == "--slots") a.slots = strtoull(v, 0, 10);
        else if (k == "--nbits") a.nbits = strtoull(v, 0, 10);
        else if (k == "--gpus") a.gpus = atoi(v);
        else if (k == "--steps") a.steps = atoi(v);
        else if (k == "--warmup") a.warmup = atoi(v);
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    int visible = 0;
    if (csgn_comm_device_count(&visible) != CSGN_OK || visible == 0) {
        fprintf(stderr, "shard_mul: no HIP device visible (%s); there is no CPU path\n", csgn_shard_last_error());
        return 1;
    }
    const int world = a.gpus > 0 ? a.gpus : visible;
    if (world > visible) {
        fprintf(stderr, "shard_mul: --gpus %d but only %d device(s) visible\n", world, visible);
        return 1;
    }
    std::vector<csgn_comm *> comms(world, nullptr);
    if (csgn_comm_init_all(world, nullptr, comms.data()) != CSGN_OK) {
        fprintf(stderr, "shard_mul: csgn_comm_init_all: %s\n", csgn_shard_last_error());
        return 1;
    }
    std::vector<RankResult> res(world);
    std::atomic<bool> failed(false);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back(rank_main, std::cref(a), comms[r], std::ref(res[r]), std::ref(failed));
    for (auto &t : th)
        t.join();
    int rc = 0;
    double tmax = 0;
    uint64_t digest = 0, bad = 0, bits_bad = 0;
    for (int r = 0; r < world; ++r) {
        if (res[r].rc) {
            fprintf(stderr, "rank %d failed [%d]: %s\n", r, res[r].rc, res[r].error.c_str());
            rc = 1;
        }
        tmax = res[r].seconds > tmax ? res[r].seconds : tmax;
        digest += res[r].out_digest;
        bad += res[r].counts_bad;
        bits_bad += res[r].bits_bad;
        if (res[r].counts_sum != res[0].counts_sum)
            rc = 1;                                   // every rank must have received the same vector
    }
    for (auto c : comms)
        csgn_comm_destroy(c);
    if (rc)
        return rc;
    const uint64_t dl = csgn_default_len(a.nbits), T = a.terms;
    const double bytes = 8.0 * dl * (2.0 * T + (double)T * T);
    const double mults = (double)a.pairs * a.steps / tmax;
    printf("{\"tool\": \"shard_mul\", \"n_gpus\": %d, \"pairs\": %llu, \"terms\": %llu, \"n_bits\": %llu, \"steps\": %d, "
           "\"seconds\": %.6f, \"mult_per_s\": %.1f, \"algorithmic_GBps\": %.1f, \"collective\": \"ncclAllGather(term counts, %llu x u64)\", "
           "\"gathered_counts_sum\": %llu, \"gathered_counts_wrong\": %llu, \"fresh_ciphertexts\": %s, "
           "\"decrypted_bits_wrong\": %llu, \"products_digest\": \"%016llx\", \"shards\": [",
           world, (unsigned long long)a.pairs, (unsigned long long)T, (unsigned long long)a.nbits, a.steps, tmax, mults,
           mults * bytes / 1e9, (unsigned long long)a.pairs, (unsigned long long)res[0].counts_sum,
           (unsigned long long)bad, T == 1 ? "true" : "false", (unsigned long long)bits_bad, (unsigned long long)digest);
    for (int r = 0; r < world; ++r)
        printf("%s[%llu, %llu]", r ? ", " : "", (unsigned long long)res[r].lo, (unsigned long long)res[r].hi);
    printf("]}\n");
    return (bad == 0 && bits_bad == 0 && res[0].counts_sum == a.pairs * T * T) ? 0 : 1;
}
