// sharded_driver.cpp -- certFHE::ShardedBatch (include/certfhe/ShardedBatch.h) against
// certFHE::CiphertextBatch (include/certfhe/Batch.h) on whatever GPUs are visible (world 1 on the
// one-GPU box): the sharded batch must hold exactly the words the one-GPU batch holds, element by
// element, for encrypt / * / + / decrypt (plain and fused) / applyPermutation / termCounts, in both forms of the gather; the fused
// Enc*Enc must equal the unfused chain; and an injected failure must surface as an exception that
// names the rank, at once, and leave the group dead instead of hanging.
//
//   sharded_driver compare <count>     prints "OK ..." lines, exit 0 iff every check holds
//   sharded_driver fail <count>        failure-injection path
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "certFHE.h"
#include "ShardedBatch.h"
#include "csgn_hip.h"

using namespace certFHE;

static int failures = 0;
#define CHECK(cond, what)                                        \
    do {                                                         \
        if (!(cond)) {                                           \
            printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); \
            ++failures;                                          \
        } else {                                                 \
            printf("OK %s\n", what);                             \
        }                                                        \
    } while (0)

static std::vector<uint64_t> hostWords(const Ciphertext &c)
{
    const uint64_t *v = c.getValues();
    return std::vector<uint64_t>(v, v + c.getLen());
}

static bool sameElements(const ShardedBatch &s, const CiphertextBatch &b, uint64_t step)
{
    if (s.size() != b.size() || s.terms() != b.terms())
        return false;
    for (uint64_t i = 0; i < s.size(); i += step) {
        if (s.values(i) != hostWords(b.at(i)))
            return false;
    }
    return s.size() == 0 || s.values(s.size() - 1) == hostWords(b.at(b.size() - 1));
}

static uint64_t digestOf(const CiphertextBatch &b)
{
    void *d = nullptr;
    uint64_t h = 0;
    if (csgn_malloc(&d, 8) || csgn_memset(d, 0, 8, nullptr) ||
        csgn_digest(b.deviceValues(), b.size() * b.terms() * b.context().getDefaultN(), 0, (uint64_t *)d, nullptr) ||
        csgn_memcpy_d2h(&h, d, 8, nullptr) || csgn_stream_sync(nullptr))
        throw std::runtime_error(csgn_last_error());
    csgn_free(d);
    return h;
}

static int cmd_compare(uint64_t count, uint64_t n_bits, uint64_t d_key)
{
    ShardGroup group;
    printf("group: %d GPU(s); %s\n", group.size(), group.collective().c_str());
    Context ctx(n_bits, d_key);
    SecretKey key(ctx);
    std::vector<unsigned char> pa(count), pb(count);
    for (uint64_t i = 0; i < count; ++i) {
        pa[i] = (unsigned char)((i * 7 + 1) % 3 == 0);
        pb[i] = (unsigned char)((i * 5 + 2) % 2);
    }
    const uint64_t step = count > 64 ? count / 37 : 1;
    for (int uneven = 0; uneven < 2; ++uneven) {
        group.forceGroupedBroadcast(uneven != 0);
        printf("-- gather form: %s\n", uneven ? "grouped ncclBroadcast (uneven-shard branch, forced)" : "ncclAllGather");
        ShardedBatch sa = ShardedBatch::encrypt(group, key, pa, 77, 1000), sb = ShardedBatch::encrypt(group, key, pb, 78, 5);
        CiphertextBatch ba = CiphertextBatch::encrypt(key, pa, 77, 1000), bb = CiphertextBatch::encrypt(key, pb, 78, 5);
        CHECK(sameElements(sa, ba, step) && sameElements(sb, bb, step), "encrypt: sharded words == one-GPU words");
        ShardedBatch sp = sa * sb, ss = sa + sb, sq = ss * sp;           // 1, 2 and 2x1 = 2 terms
        CiphertextBatch bp = ba * bb, bs = ba + bb, bq = bs * bp;
        CHECK(sameElements(sp, bp, step), "operator*: element i == a[i]*b[i]");
        CHECK(sameElements(ss, bs, step), "operator+: element i == a[i]+b[i]");
        CHECK(sameElements(sq, bq, step), "(a+b)*(a*b)");
        CHECK(sp.digest() == digestOf(bp) && sq.digest() == digestOf(bq), "digest over shards == digest on one GPU");
        const std::vector<unsigned char> dp = sp.decrypt(key), ds = ss.decrypt(key), dq = sq.decrypt(key);
        bool bits_ok = dp == bp.decrypt(key) && ds == bs.decrypt(key) && dq == bq.decrypt(key);
        for (uint64_t i = 0; i < count && bits_ok; ++i)
            bits_ok = dp[i] == (pa[i] & pb[i]) && ds[i] == (pa[i] ^ pb[i]) && dq[i] == ((pa[i] ^ pb[i]) & pa[i] & pb[i]);
        CHECK(bits_ok, "decrypt: gathered bits == one-GPU bits == the clear circuit");
        const std::vector<uint64_t> cp = sp.termCounts(), cq = sq.termCounts();
        bool counts_ok = cp.size() == count && cq.size() == count;
        for (uint64_t i = 0; i < count && counts_ok; ++i)
            counts_ok = cp[i] == 1 && cq[i] == 2;
        CHECK(counts_ok, "termCounts: gathered vector == terms per element");
        // fused fresh chain: one kernel, same words as encrypt, encrypt, *
        ShardedBatch sf = ShardedBatch::encryptProduct(group, key, pa, pb, 77, 78, 1000);
        ShardedBatch su = ShardedBatch::encrypt(group, key, pa, 77, 1000) * ShardedBatch::encrypt(group, key, pb, 78, 1000);
        bool fused_ok = sf.digest() == su.digest();
        for (uint64_t i = 0; i < count && fused_ok; i += step)
            fused_ok = sf.values(i) == su.values(i);
        CHECK(fused_ok, "encryptProduct == encrypt * encrypt (words)");
        CHECK(sf.decrypt(key) == su.decrypt(key), "encryptProduct decrypts like the unfused product");
        // fused decrypt (no product / sum materialised) and the permutation, against the one-GPU batch
        CHECK(ss.decryptProduct(sp, key) == bs.decryptProduct(bp, key) && ss.decryptProduct(sp, key) == dq,
              "decryptProduct == Dec of the materialised product");
        CHECK(sa.decryptSum(sb, key) == ba.decryptSum(bb, key) && sa.decryptSum(sb, key) == ds,
              "decryptSum == Dec of the materialised sum");
        Permutation perm(ctx);
        ShardedBatch spm = sq.applyPermutation(perm);                    // 2 terms in, ONE term out (first term permuted)
        CiphertextBatch bpm = bq.applyPermutation(perm);
        CHECK(spm.terms() == 1 && sameElements(spm, bpm, step), "applyPermutation: element i == CiphertextBatch's");
        SecretKey pkey = key.applyPermutation(perm);
        CHECK(sa.applyPermutation(perm).decrypt(pkey) == std::vector<unsigned char>(pa.begin(), pa.end()),
              "a permuted fresh batch decrypts under the permuted key");
    }
    int sum = 0;
    ShardedBatch probe = ShardedBatch::synthetic(group, ctx, count, 3, 9);
    uint64_t prev = 0;
    for (int r = 0; r < probe.shards(); ++r) {
        CHECK(probe.shardRange(r).first == prev, "shard ranges are contiguous");
        prev = probe.shardRange(r).second;
        ++sum;
    }
    CHECK(prev == count && sum == group.size(), "shards cover the batch");
    // knobs are per host thread: setTuning reaches the worker threads (a knob that selects another kernel for the
    // same words -- results must not move; an unknown knob is an exception from the worker, not silence)
    {
        group.setTuning("mul_flat", -1);
        ShardedBatch x = ShardedBatch::synthetic(group, ctx, count, 3, 11) * ShardedBatch::synthetic(group, ctx, count, 5, 12);
        group.setTuning("mul_flat", 0);
        ShardedBatch y = ShardedBatch::synthetic(group, ctx, count, 3, 11) * ShardedBatch::synthetic(group, ctx, count, 5, 12);
        CHECK(x.digest() == y.digest(), "setTuning on the workers: same words under another kernel");
        bool threw = false;
        try {
            group.setTuning("no_such_knob", 1);
        } catch (const std::exception &) {
            threw = true;
        }
        CHECK(threw, "setTuning: an unknown knob is an exception");
    }
    CHECK(group.healthy(), "group healthy at the end");
    return failures ? 1 : 0;
}

static int cmd_fail(uint64_t count)
{
    ShardGroup group;
    Context ctx(1247, 16);
    SecretKey key(ctx);
    std::vector<unsigned char> bits(count, 1);
    ShardedBatch a = ShardedBatch::encrypt(group, key, bits, 5);
    group.setTimeoutMs(20000);
    group.injectFailure(group.size() - 1);
    bool threw = false;
    try {
        (void)a.termCounts();               // a collective: the failing rank never joins it
    } catch (const std::exception &e) {
        threw = true;
        printf("caught: %s\n", e.what());
        CHECK(strstr(e.what(), "injected failure") != nullptr && strstr(e.what(), "rank") != nullptr,
              "the exception names the failing rank and its error");
    }
    CHECK(threw, "a failing rank surfaces as an exception (no hang)");
    CHECK(!group.healthy(), "the group is dead afterwards");
    threw = false;
    try {
        (void)(a * a);
    } catch (const std::exception &e) {
        threw = strstr(e.what(), "dead") != nullptr;
    }
    CHECK(threw, "later calls fail at once");
    return failures ? 1 : 0;
}

int main(int argc, char **argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    const uint64_t count = argc > 2 ? strtoull(argv[2], 0, 10) : 1000;
    try {
        if (cmd == "compare")
            return cmd_compare(count, argc > 3 ? strtoull(argv[3], 0, 10) : 1247, argc > 4 ? strtoull(argv[4], 0, 10) : 16);
        if (cmd == "fail")
            return cmd_fail(count);
    } catch (const std::exception &e) {
        fprintf(stderr, "sharded_driver: %s\n", e.what());
        return 3;
    }
    fprintf(stderr, "usage: sharded_driver compare|fail <count> [n_bits d]\n");
    return 2;
}
