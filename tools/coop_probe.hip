// Where a wave of k_mul_ragged_coop spends its time (dev tool; build: make tools/bin/coop_probe).
// Compiles csgn_mul.hip into this program with -DCSGN_COOP_STAMPS: every wave leaves
//   {cycles in all, cycles in the start-up search, cycles inside its hand-counted waits, blocks, pairs, windows, start}
// behind, and the program prints the distribution for a long-tailed batch of small pairs (the one tools/bench_ragged.py
// calls "lognormal mean~8 x262144") -- or of mean MEAN, COUNT pairs.  Operands are random words, the products are not
// checked here (tests/test_gpu_parity.py does that); three operand sets in turn, as the benchmark.
//   usage: coop_probe [mean [count [K [pipe [span_blocks]]]]]
#include "../csgn_amd/csrc/csgn_mul.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace csgn;

template <int K, bool PIPE>
static void launch(u32 wgs, const unit16 *L, const u64 *offL, const unit16 *R, const u64 *offR, unit16 *out, const u64 *offOut,
                   u32 batch, u64 v_end, u32 U, u32 span)
{
    // PROBE_LDS: bytes of (unused) dynamic LDS per workgroup, to hold fewer workgroups on a CU: 81920 -> 2 (8 waves), 54000 -> 3
    k_mul_ragged_coop<unit16, K, PIPE><<<wgs, 256, getenv("PROBE_LDS") ? (size_t)atoi(getenv("PROBE_LDS")) : 0>>>(L, offL, R, offR, out, offOut, batch, 0, v_end, U, csgn_fastdiv_make(U), span,
                                                    kWave, nullptr, (getenv("PROBE_TOUCH") ? (u32)atoi(getenv("PROBE_TOUCH")) : 128u) * 1024u / (U * 16u));
}

int main(int argc, char **argv)
{
    const double mean = argc > 1 ? atof(argv[1]) : 8.0;
    const u32 batch = argc > 2 ? (u32)atoi(argv[2]) : 1u << 18;
    const int K = argc > 3 ? atoi(argv[3]) : 4, pipe = argc > 4 ? atoi(argv[4]) : 1;
    const u32 U = getenv("PROBE_U") ? (u32)atoi(getenv("PROBE_U")) : 10;   // N = 1247: 20 words = 10 units of 16 bytes
#if defined(COOP_PROBE_NO_STORE)
    if (argc <= 4 || atoi(argv[4]) != 0) {
        fprintf(stderr, "this build drops instructions the hand-counted waits of the pipelined form rely on: run it with pipe = 0\n");
        return 2;
    }
#endif
    std::mt19937_64 rng(0);
    std::lognormal_distribution<double> ln(std::log(mean) - 0.5, 1.0);
    std::vector<u64> offL(batch + 1, 0), offR(batch + 1, 0), offOut(batch + 1, 0);
    for (u32 b = 0; b < batch; ++b) {
        const u64 t1 = (u64)std::min(600.0, std::max(1.0, ln(rng))), t2 = (u64)std::min(600.0, std::max(1.0, ln(rng)));
        offL[b + 1] = offL[b] + t1;
        offR[b + 1] = offR[b] + t2;
        offOut[b + 1] = offOut[b] + t1 * t2;
    }
    const u64 total_units = offOut[batch] * U;
    const u64 v_end = total_units + (u64)kWave * batch;
    u32 span = (u32)std::min<u64>(4096, std::max<u64>(512, (v_end / 16384u) & ~63ull));
    if (argc > 5)
        span = kWave * (u32)atoi(argv[5]);
    const u64 wgs = (v_end + 4ull * span - 1) / (4ull * span);
    const u64 waves = wgs * 4;
    printf("mean %.0f, %u pairs: %.1f MB of operands, %.1f MB of products; span %u units, %llu waves, K %d, pipe %d\n", mean, batch,
           (offL[batch] + offR[batch]) * U * 16 / 1e6, total_units * 16 / 1e6, span, (unsigned long long)waves, K, pipe);
    u64 *dOL, *dOR, *dOO, *dStamps;
    unit16 *dL[3], *dR[3], *dOut;
    CK(hipMalloc(&dOL, (batch + 1) * 8));
    CK(hipMalloc(&dOR, (batch + 1) * 8));
    CK(hipMalloc(&dOO, (batch + 1) * 8));
    CK(hipMemcpy(dOL, offL.data(), (batch + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dOR, offR.data(), (batch + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dOO, offOut.data(), (batch + 1) * 8, hipMemcpyHostToDevice));
    for (int k = 0; k < 3; ++k) {
        CK(hipMalloc(&dL[k], offL[batch] * U * 16));
        CK(hipMalloc(&dR[k], offR[batch] * U * 16));
        CK(hipMemset(dL[k], 0x5a + k, offL[batch] * U * 16));
        CK(hipMemset(dR[k], 0xa5 + k, offR[batch] * U * 16));
    }
    CK(hipMalloc(&dOut, total_units * 16));
    CK(hipMalloc(&dStamps, waves * 8 * 8));
    CK(hipMemset(dStamps, 0, waves * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(csgn::g_coop_stamps), &dStamps, sizeof(dStamps)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 12; ++it) {
        const int k = it % (getenv("NSETS") ? atoi(getenv("NSETS")) : 3);
        if (it == 11)
            CK(hipEventRecord(e0));
#define GO(KK, PP) launch<KK, PP>((u32)wgs, dL[k], dOL, dR[k], dOR, dOut, dOO, batch, v_end, U, span)
        if (K == 2 && pipe) GO(2, true); else if (K == 2) GO(2, false); else if (pipe) GO(4, true); else GO(4, false);
        CK(hipGetLastError());
        if (it == 11)
            CK(hipEventRecord(e1));
    }
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)(offL[batch] + offR[batch] + offOut[batch]) * U * 16;
    printf("last launch: %.3f ms = %.0f GB/s algorithmic\n", ms, bytes / ms / 1e6);
    std::vector<u64> st(waves * 8);
    CK(hipMemcpy(st.data(), dStamps, waves * 64, hipMemcpyDeviceToHost));
    // per wave: cycles, search, waits, blocks, pairs, windows
    std::vector<double> all, search, waits, perblock;
    double sum_all = 0, sum_search = 0, sum_wait = 0, sum_blocks = 0, sum_pairs = 0, sum_windows = 0;
    u64 t_min = ~0ull, t_max = 0;
    for (u64 w = 0; w < waves; ++w) {
        const u64 *s = &st[w * 8];
        if (s[0] == 0)
            continue;
        all.push_back((double)s[0]);
        search.push_back((double)s[1]);
        waits.push_back((double)s[2]);
        if (s[3])
            perblock.push_back((double)(s[0] - s[1]) / (double)s[3]);
        sum_all += s[0]; sum_search += s[1]; sum_wait += s[2]; sum_blocks += s[3]; sum_pairs += s[4]; sum_windows += s[5];
        t_min = std::min(t_min, s[6]);
        t_max = std::max(t_max, s[6] + s[0]);
    }
    auto pct = [](std::vector<double> &v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(q * (v.size() - 1))]; };
    printf("waves that ran: %zu; blocks %.0f (%.1f per wave), pairs %.0f (%.1f per wave), windows %.0f\n", all.size(), sum_blocks,
           sum_blocks / all.size(), sum_pairs, sum_pairs / all.size(), sum_windows);
    printf("cycles per wave      : median %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f\n", pct(all, .5), pct(all, .1), pct(all, .9), pct(all, 1));
    printf("  start-up search    : median %8.0f  p10 %8.0f  p90 %8.0f   (%.1f %% of all wave cycles)\n", pct(search, .5), pct(search, .1), pct(search, .9),
           100 * sum_search / sum_all);
    printf("  inside the waits   : median %8.0f  p10 %8.0f  p90 %8.0f   (%.1f %% of all wave cycles)\n", pct(waits, .5), pct(waits, .1), pct(waits, .9),
           100 * sum_wait / sum_all);
    printf("  per block, search excluded: median %6.0f  p10 %6.0f  p90 %6.0f cycles\n", pct(perblock, .5), pct(perblock, .1), pct(perblock, .9));
    printf("  first start to last end: %.0f cycles (s_memtime)\n", (double)(t_max - t_min));
    return 0;
}
