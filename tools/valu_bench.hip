// valu_bench.hip -- what does the CDNA4 (gfx950) VALU issue per SIMD for the integer instructions the
// keyed encrypt (ChaCha: v_add_u32 / v_xor_b32 / v_alignbit_b32) and the permutation are made of?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_bench tools/valu_bench.hip
//   tools/bin/valu_bench > profiles/r03/valu_issue.txt
//
// For each instruction (and for the real ChaCha double round) a kernel runs a long register-only loop
// of `ILP` independent dependency chains (ILP = 1: every instruction waits for the one before it)
// on every CU, at 1, 2, 4 and 8 waves per SIMD.  It reports wave-instructions per second for the
// whole chip and, with the in-kernel clock (s_memtime / s_memrealtime, 100 MHz real-time counter),
// cycles per wave-instruction per SIMD.  2.0 = the SIMD-32 rate for wave64, 4.0 = the cost of one
// wave issuing alone (guide: MI355X_MICROARCH.md, per-instruction cycle constants).
// Register-only: no memory traffic inside the timed loop, so the HBM plays no part.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

enum Op { OP_ADD = 0, OP_XOR = 1, OP_ALIGNBIT = 2, OP_CHACHA = 3, OP_CHACHA2 = 4, OP_MIX = 5 };

struct Stamp {
    unsigned long long cyc, rt;
};

__device__ inline unsigned rotl32(unsigned x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

#define QR(a, b, c, d)                 \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7)

#define DOUBLE_ROUND(x)                 \
    QR(x[0], x[4], x[8], x[12]);        \
    QR(x[1], x[5], x[9], x[13]);        \
    QR(x[2], x[6], x[10], x[14]);       \
    QR(x[3], x[7], x[11], x[15]);       \
    QR(x[0], x[5], x[10], x[15]);       \
    QR(x[1], x[6], x[11], x[12]);       \
    QR(x[2], x[7], x[8], x[13]);        \
    QR(x[3], x[4], x[9], x[14])

// ILP independent chains of ONE instruction; 64 instructions per loop trip (64/ILP per chain), all in
// ONE asm statement (between separate asm statements hipcc pads with s_nop, which costs issue slots).
#define I1(OPS, a) OPS(a)
#define REP16_1(OPS) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0) OPS(0)
#define REP16_4(OPS) OPS(0) OPS(1) OPS(2) OPS(3) OPS(0) OPS(1) OPS(2) OPS(3) OPS(0) OPS(1) OPS(2) OPS(3) OPS(0) OPS(1) OPS(2) OPS(3)
#define REP16_8(OPS) OPS(0) OPS(1) OPS(2) OPS(3) OPS(4) OPS(5) OPS(6) OPS(7) OPS(0) OPS(1) OPS(2) OPS(3) OPS(4) OPS(5) OPS(6) OPS(7)
#define S_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define S_ROT(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 25\n"
#define BODY64(REP, OPS)                                                                                     \
    asm volatile(REP(OPS) REP(OPS) REP(OPS) REP(OPS)                                                         \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(k))

template <int OP, int ILP>
__global__ void __launch_bounds__(1024) k_chain(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];      // only there to pin the number of workgroups per CU
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r[i] = seed * (threadIdx.x + 1u) + (unsigned)i * 0x9E3779B9u;
    unsigned k = seed | 1u;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
        if (OP == OP_ADD) {
            if (ILP == 1) BODY64(REP16_1, S_ADD); else if (ILP == 4) BODY64(REP16_4, S_ADD); else BODY64(REP16_8, S_ADD);
        } else if (OP == OP_XOR) {
            if (ILP == 1) BODY64(REP16_1, S_XOR); else if (ILP == 4) BODY64(REP16_4, S_XOR); else BODY64(REP16_8, S_XOR);
        } else {
            if (ILP == 1) BODY64(REP16_1, S_ROT); else if (ILP == 4) BODY64(REP16_4, S_ROT); else BODY64(REP16_8, S_ROT);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc ^= r[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

// The ChaCha double round as the compiler schedules it (BLOCKS = 1: four independent quarter rounds
// at a time, ILP 4; BLOCKS = 2: two blocks per lane interleaved, ILP 8).  96 instructions per
// double round and block.
template <int BLOCKS>
__global__ void __launch_bounds__(1024) k_chacha(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];
    unsigned x[BLOCKS][16];
#pragma unroll
    for (int b = 0; b < BLOCKS; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[b][i] = seed * (threadIdx.x + 1u) + (unsigned)(b * 16 + i) * 0x9E3779B9u;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int b = 0; b < BLOCKS; ++b) {
            DOUBLE_ROUND(x[b]);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
#pragma unroll
    for (int b = 0; b < BLOCKS; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc ^= x[b][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}


// Other instructions and mixes, 8 independent chains (registers %0..%7, %8 = a constant VGPR): what is
// the rate of the 64-bit ENCODINGS (VOP3, VOP2 + literal, SDWA, DPP) against the 32-bit ones, and what
// does a stream that alternates them issue at?  PAT(i) expands to the instruction(s) for register i.
#define REP8(PAT) PAT(0) PAT(1) PAT(2) PAT(3) PAT(4) PAT(5) PAT(6) PAT(7)
#define P_ADD_E64(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define P_ADD_LIT(i) "v_add_u32 %" #i ", 0x12345678, %" #i "\n"
#define P_PERM(i) "v_perm_b32 %" #i ", %" #i ", %" #i ", %8\n"
#define P_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %" #i ", 1\n"
#define P_LSHL_OR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define P_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %8\n"
#define P_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define P_BFI(i) "v_bfi_b32 %" #i ", %8, %" #i ", %" #i "\n"
#define P_LSHL(i) "v_lshlrev_b32 %" #i ", 7, %" #i "\n"
#define P_XOR_SDWA(i) "v_xor_b32_sdwa %" #i ", %" #i ", %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define P_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define P_PK_ADD16(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define P_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define P_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define P_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define P_AXR(i) "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %" #i ", 25\n"
#define P_AX(i) "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %8\n"
#define P_AR(i) "v_add_u32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %" #i ", 25\n"
#define P_AAR(i) "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %8\nv_add_u32 %" #i ", %" #i ", %8\nv_alignbit_b32 %" #i ", %" #i ", %" #i ", 25\n"
#define P_ROT2(i) "v_lshrrev_b32 %9, 25, %" #i "\nv_lshl_or_b32 %" #i ", %" #i ", 7, %9\n"
#define P_AX_PERM(i) "v_add_u32 %" #i ", %" #i ", %8\nv_xor_b32 %" #i ", %" #i ", %8\nv_perm_b32 %" #i ", %" #i ", %" #i ", %8\n"
// runs of the same class: k adds, then k xors, then k rotates (k independent chains), to see whether the
// price of mixing the full-rate and the half-rate class depends on how often the stream switches
#define A_(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define X_(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define R_(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 25\n"
#define RUN2 A_(0) A_(1) X_(0) X_(1) R_(0) R_(1) A_(2) A_(3) X_(2) X_(3) R_(2) R_(3) A_(4) A_(5) X_(4) X_(5) R_(4) R_(5) A_(6) A_(7) X_(6) X_(7) R_(6) R_(7)
#define RUN4 A_(0) A_(1) A_(2) A_(3) X_(0) X_(1) X_(2) X_(3) R_(0) R_(1) R_(2) R_(3) A_(4) A_(5) A_(6) A_(7) X_(4) X_(5) X_(6) X_(7) R_(4) R_(5) R_(6) R_(7)
#define RUN8 A_(0) A_(1) A_(2) A_(3) A_(4) A_(5) A_(6) A_(7) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) R_(0) R_(1) R_(2) R_(3) R_(4) R_(5) R_(6) R_(7)
// 16 of each: two passes over the 8 chains per class (adds of pass 2 depend on adds of pass 1: chain length 2)
#define RUN16 A_(0) A_(1) A_(2) A_(3) A_(4) A_(5) A_(6) A_(7) A_(0) A_(1) A_(2) A_(3) A_(4) A_(5) A_(6) A_(7) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) X_(0) X_(1) X_(2) X_(3) X_(4) X_(5) X_(6) X_(7) R_(0) R_(1) R_(2) R_(3) R_(4) R_(5) R_(6) R_(7) R_(0) R_(1) R_(2) R_(3) R_(4) R_(5) R_(6) R_(7)
#define P_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %8 bitop3:0x96\n"

enum Pat { PT_ADD_E64, PT_ADD_LIT, PT_PERM, PT_ALIGNBYTE, PT_LSHL_OR, PT_XAD, PT_ADD3, PT_BFI, PT_LSHL, PT_XOR_SDWA,
           PT_MOV_DPP, PT_PK_ADD16, PT_MUL24, PT_MAD24, PT_MULLO, PT_AXR, PT_AX, PT_AR, PT_AAR, PT_ROT2, PT_AX_PERM,
           PT_BITOP3, PT_RUN2, PT_RUN4, PT_RUN8, PT_RUN16 };

#define PBODY(PAT, N)                                                                                        \
    asm volatile(N(PAT)                                                                                      \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(k), "v"(tmp))
#define PRUN(STR)                                                                                            \
    asm volatile(STR                                                                                         \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(k), "v"(tmp))
#define X8(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT)
#define X4(PAT) REP8(PAT) REP8(PAT) REP8(PAT) REP8(PAT)
#define X2(PAT) REP8(PAT) REP8(PAT)

template <int PT>
__global__ void __launch_bounds__(1024) k_pattern(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        r[i] = seed * (threadIdx.x + 1u) + (unsigned)i * 0x9E3779B9u;
    unsigned k = seed | 0x01020301u, tmp = 0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
        // every body is 64 instructions
        if (PT == PT_ADD_E64) PBODY(P_ADD_E64, X8);
        else if (PT == PT_ADD_LIT) PBODY(P_ADD_LIT, X8);
        else if (PT == PT_PERM) PBODY(P_PERM, X8);
        else if (PT == PT_ALIGNBYTE) PBODY(P_ALIGNBYTE, X8);
        else if (PT == PT_LSHL_OR) PBODY(P_LSHL_OR, X8);
        else if (PT == PT_XAD) PBODY(P_XAD, X8);
        else if (PT == PT_ADD3) PBODY(P_ADD3, X8);
        else if (PT == PT_BFI) PBODY(P_BFI, X8);
        else if (PT == PT_LSHL) PBODY(P_LSHL, X8);
        else if (PT == PT_XOR_SDWA) PBODY(P_XOR_SDWA, X8);
        else if (PT == PT_MOV_DPP) PBODY(P_MOV_DPP, X8);
        else if (PT == PT_PK_ADD16) PBODY(P_PK_ADD16, X8);
        else if (PT == PT_MUL24) PBODY(P_MUL24, X8);
        else if (PT == PT_MAD24) PBODY(P_MAD24, X8);
        else if (PT == PT_MULLO) PBODY(P_MULLO, X8);
        else if (PT == PT_AX) PBODY(P_AX, X4);
        else if (PT == PT_AR) PBODY(P_AR, X4);
        else if (PT == PT_ROT2) PBODY(P_ROT2, X4);
        else if (PT == PT_AAR) PBODY(P_AAR, X2);
        else if (PT == PT_BITOP3) PBODY(P_BITOP3, X8);
        else if (PT == PT_RUN2) { PRUN(RUN2); PRUN(RUN2); }                     // 48 instructions per trip
        else if (PT == PT_RUN4) { PRUN(RUN4); PRUN(RUN4); }
        else if (PT == PT_RUN8) { PRUN(RUN8); PRUN(RUN8); }
        else if (PT == PT_RUN16) { PRUN(RUN16); }
        else if (PT == PT_AXR) { PBODY(P_AXR, X2); PBODY(P_AX, REP8); }          // 48 + 16
        else if (PT == PT_AX_PERM) { PBODY(P_AX_PERM, X2); PBODY(P_AX, REP8); }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = tmp;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc ^= r[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

// The ChaCha double round with the instruction ORDER forced: B blocks per lane (4*B independent quarter
// rounds at a time), each of a quarter round's 12 steps done for all 4*B quarter rounds before the next
// step starts (__builtin_amdgcn_sched_barrier(0) between steps), so the stream is runs of 4*B adds,
// 4*B xors, 4*B rotates ...  -- does the cost of mixing the full-rate and the half-rate class fall
// when the stream switches class less often?
#define ST1(a, b, c, d) a += b;
#define ST2(a, b, c, d) d ^= a;
#define ST3(a, b, c, d) d = rotl32(d, 16);
#define ST4(a, b, c, d) c += d;
#define ST5(a, b, c, d) b ^= c;
#define ST6(a, b, c, d) b = rotl32(b, 12);
#define ST7(a, b, c, d) a += b;
#define ST8(a, b, c, d) d ^= a;
#define ST9(a, b, c, d) d = rotl32(d, 8);
#define ST10(a, b, c, d) c += d;
#define ST11(a, b, c, d) b ^= c;
#define ST12(a, b, c, d) b = rotl32(b, 7);
#define COLQ(ST, X) ST(X[0], X[4], X[8], X[12]) ST(X[1], X[5], X[9], X[13]) ST(X[2], X[6], X[10], X[14]) ST(X[3], X[7], X[11], X[15])
#define DIAQ(ST, X) ST(X[0], X[5], X[10], X[15]) ST(X[1], X[6], X[11], X[12]) ST(X[2], X[7], X[8], X[13]) ST(X[3], X[4], X[9], X[14])
#define SB __builtin_amdgcn_sched_barrier(0);
#define ALLB(Q, ST)                          \
    _Pragma("unroll") for (int b = 0; b < BLOCKS; ++b) { Q(ST, x[b]) } \
    SB
#define HALF(Q) ALLB(Q, ST1) ALLB(Q, ST2) ALLB(Q, ST3) ALLB(Q, ST4) ALLB(Q, ST5) ALLB(Q, ST6) ALLB(Q, ST7) ALLB(Q, ST8) ALLB(Q, ST9) ALLB(Q, ST10) ALLB(Q, ST11) ALLB(Q, ST12)

template <int BLOCKS>
__global__ void __launch_bounds__(1024) k_chacha_grouped(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];
    unsigned x[BLOCKS][16];
#pragma unroll
    for (int b = 0; b < BLOCKS; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            x[b][i] = seed * (threadIdx.x + 1u) + (unsigned)(b * 16 + i) * 0x9E3779B9u;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
        HALF(COLQ)
        HALF(DIAQ)
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
#pragma unroll
    for (int b = 0; b < BLOCKS; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc ^= x[b][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

// Do two VGPR source operands from the same register bank (index mod 4) cost an extra cycle?  64
// v_add_u32 / v_xor_b32 with EXPLICIT physical registers: dst = src0 = v[32+i], src1 = v[48+i+SKEW]
// (i = 0..7): SKEW 0 -> src0 and src1 in the same bank, SKEW 1 -> neighbouring banks.
template <int SKEW, int XOR>
__global__ void __launch_bounds__(1024) k_bank(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_mov_b32 v32, %0\nv_mov_b32 v33, %0\nv_mov_b32 v34, %0\nv_mov_b32 v35, %0\nv_mov_b32 v36, %0\nv_mov_b32 v37, %0\nv_mov_b32 v38, %0\nv_mov_b32 v39, %0\n"
                 "v_mov_b32 v48, %0\nv_mov_b32 v49, %0\nv_mov_b32 v50, %0\nv_mov_b32 v51, %0\nv_mov_b32 v52, %0\nv_mov_b32 v53, %0\nv_mov_b32 v54, %0\nv_mov_b32 v55, %0\nv_mov_b32 v56, %0\n"
                 :: "v"(seed * (threadIdx.x + 1u))
                 : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56");
#define BK_OP(OPC, d, s) OPC " v" #d ", v" #d ", v" #s "\n"
#define BK8(OPC, a0, a1, a2, a3, a4, a5, a6, a7) BK_OP(OPC, 32, a0) BK_OP(OPC, 33, a1) BK_OP(OPC, 34, a2) BK_OP(OPC, 35, a3) BK_OP(OPC, 36, a4) BK_OP(OPC, 37, a5) BK_OP(OPC, 38, a6) BK_OP(OPC, 39, a7)
#define BK64(OPC, ...) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__) BK8(OPC, __VA_ARGS__)
#define BK_CLOB "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56"
    for (int t = 0; t < trips; ++t) {
        if (SKEW == 0 && !XOR) asm volatile(BK64("v_add_u32", 48, 49, 50, 51, 52, 53, 54, 55) ::: BK_CLOB);
        if (SKEW == 1 && !XOR) asm volatile(BK64("v_add_u32", 49, 50, 51, 52, 53, 54, 55, 56) ::: BK_CLOB);
        if (SKEW == 0 && XOR) asm volatile(BK64("v_xor_b32", 48, 49, 50, 51, 52, 53, 54, 55) ::: BK_CLOB);
        if (SKEW == 1 && XOR) asm volatile(BK64("v_xor_b32", 49, 50, 51, 52, 53, 54, 55, 56) ::: BK_CLOB);
    }
    unsigned acc;
    asm volatile("v_xor_b32 %0, v32, v39" : "=v"(acc) :: BK_CLOB);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

#include "valu_bench_chacha_asm.inc"
// The ChaCha double round on two blocks as ONE asm statement with explicit registers: runs of 8 per
// step, and (FRIENDLY) the four words of every quarter round in four different VGPR banks (register
// index mod 4) -- hipcc's own allocation puts x[i] and x[i+4] in the same bank, so every add and xor of a
// column round reads two registers of one bank.
template <int FRIENDLY>
__global__ void __launch_bounds__(1024) k_chacha_asm(unsigned *out, Stamp *stamps, unsigned seed, int trips)
{
    extern __shared__ unsigned char lds_pad[];
    const unsigned s0 = seed * (threadIdx.x + 1u);
    asm volatile("v_mov_b32 v64, %0\nv_add_u32 v65, 1, %0\nv_add_u32 v66, 2, %0\nv_add_u32 v67, 3, %0\nv_add_u32 v68, 4, %0\nv_add_u32 v69, 5, %0\nv_add_u32 v70, 6, %0\nv_add_u32 v71, 7, %0\n"
                 "v_add_u32 v72, 8, %0\nv_add_u32 v73, 9, %0\nv_add_u32 v74, 10, %0\nv_add_u32 v75, 11, %0\nv_add_u32 v76, 12, %0\nv_add_u32 v77, 13, %0\nv_add_u32 v78, 14, %0\nv_add_u32 v79, 15, %0\n"
                 "v_add_u32 v80, 16, %0\nv_add_u32 v81, 17, %0\nv_add_u32 v82, 18, %0\nv_add_u32 v83, 19, %0\nv_add_u32 v84, 20, %0\nv_add_u32 v85, 21, %0\nv_add_u32 v86, 22, %0\nv_add_u32 v87, 23, %0\n"
                 "v_add_u32 v88, 24, %0\nv_add_u32 v89, 25, %0\nv_add_u32 v90, 26, %0\nv_add_u32 v91, 27, %0\nv_add_u32 v92, 28, %0\nv_add_u32 v93, 29, %0\nv_add_u32 v94, 30, %0\nv_add_u32 v95, 31, %0\n"
                 :: "v"(s0) : CHACHA_ASM_CLOBBERS);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
        if (FRIENDLY)
            asm volatile(CHACHA_ASM_FRIENDLY ::: CHACHA_ASM_CLOBBERS);
        else
            asm volatile(CHACHA_ASM_NATURAL ::: CHACHA_ASM_CLOBBERS);
    }
    unsigned acc;
    asm volatile("v_xor_b32 %0, v64, v95\nv_xor_b32 %0, %0, v70\nv_xor_b32 %0, %0, v81" : "=v"(acc) :: CHACHA_ASM_CLOBBERS);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63u) == 0) {
        Stamp s;
        s.cyc = c1 - c0;
        s.rt = t1 - t0;
        stamps[((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
    }
}

struct Result {
    double ginstr_s, cyc_per_instr_simd, clock_ghz, ms;
};

template <typename Launch>
static Result run(const void *func, Launch launch, int waves_per_simd, double instr_per_trip, int trips, int cus, unsigned *d_out,
                  Stamp *d_stamps)
{
    // waves per SIMD w: w <= 4 -> one workgroup of 256*w threads per CU; w = 8 -> two of 1024.
    const int threads = waves_per_simd <= 4 ? 256 * waves_per_simd : 1024;
    const int wg_per_cu = waves_per_simd <= 4 ? 1 : 2;
    const size_t lds = wg_per_cu == 1 ? 96 * 1024 : 64 * 1024;     // 160 KB per CU: pins wg_per_cu
    const int grid = cus * wg_per_cu;
    CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch(grid, threads, lds, trips / 8 + 1);                     // warm-up
    CHECK(hipDeviceSynchronize());
    std::vector<double> ms_all;
    std::vector<Stamp> st((size_t)grid * threads / 64);
    double clock = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        launch(grid, threads, lds, trips);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms_all.push_back(ms);
    }
    CHECK(hipMemcpy(st.data(), d_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (auto &s : st)
        if (s.rt)
            clk.push_back((double)s.cyc / (double)s.rt * 0.1);     // 100 MHz real-time counter -> GHz
    std::sort(clk.begin(), clk.end());
    clock = clk.empty() ? 0 : clk[clk.size() / 2];
    std::sort(ms_all.begin(), ms_all.end());
    const double ms = ms_all[ms_all.size() / 2];
    const double waves = (double)grid * threads / 64.0;
    const double instr = waves * instr_per_trip * trips;
    Result r;
    r.ms = ms;
    r.ginstr_s = instr / (ms * 1e-3) / 1e9;
    r.clock_ghz = clock;
    // per SIMD: instructions issued per second / (SIMDs) -> cycles per instruction = clock / rate
    const double per_simd = instr / (ms * 1e-3) / ((double)cus * 4.0);
    r.cyc_per_instr_simd = clock * 1e9 / per_simd;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return r;
}

int main(int argc, char **argv)
{
    int dev = 0, cus = 256;
    CHECK(hipSetDevice(dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    unsigned *d_out;
    Stamp *d_stamps;
    CHECK(hipMalloc(&d_out, (size_t)cus * 2 * 1024 * sizeof(unsigned)));
    CHECK(hipMalloc(&d_stamps, (size_t)cus * 2 * 16 * sizeof(Stamp)));
    const int trips = argc > 1 ? atoi(argv[1]) : 40000;            // 64 instr per trip -> 2.56 M instr per wave
    printf("# %-22s %5s %8s %12s %10s %12s\n", "stream", "w/SIMD", "ms", "Gwave-instr/s", "clock GHz",
           "cyc/instr/SIMD");
    const int wlist[4] = {1, 2, 4, 8};
#define RUN_CHAIN(OP, ILP, NAME)                                                                              \
    for (int w : wlist) {                                                                                     \
        Result r = run((const void *)k_chain<OP, ILP>, [&](int g, int t, size_t l, int tr) {                  \
            hipLaunchKernelGGL((k_chain<OP, ILP>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);      \
        }, w, 64.0, trips, cus, d_out, d_stamps);                                                             \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_CHAIN(OP_ADD, 1, "v_add_u32 chain1")
    RUN_CHAIN(OP_ADD, 4, "v_add_u32 chain4")
    RUN_CHAIN(OP_ADD, 8, "v_add_u32 indep8")
    RUN_CHAIN(OP_XOR, 1, "v_xor_b32 chain1")
    RUN_CHAIN(OP_XOR, 4, "v_xor_b32 chain4")
    RUN_CHAIN(OP_XOR, 8, "v_xor_b32 indep8")
    RUN_CHAIN(OP_ALIGNBIT, 1, "v_alignbit_b32 chain1")
    RUN_CHAIN(OP_ALIGNBIT, 4, "v_alignbit_b32 chain4")
    RUN_CHAIN(OP_ALIGNBIT, 8, "v_alignbit_b32 indep8")
#define RUN_CHACHA(B, NAME)                                                                                   \
    for (int w : wlist) {                                                                                     \
        Result r = run((const void *)k_chacha<B>, [&](int g, int t, size_t l, int tr) {                       \
            hipLaunchKernelGGL((k_chacha<B>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);           \
        }, w, 96.0 * B, trips / (2 * B) + 1, cus, d_out, d_stamps);                                           \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_CHACHA(1, "chacha dround 1 blk")
    RUN_CHACHA(2, "chacha dround 2 blk")
#define RUN_CHACHA_G(B, NAME)                                                                                 \
    for (int w : {1, 2, 4}) {                                                                                 \
        Result r = run((const void *)k_chacha_grouped<B>, [&](int g, int t, size_t l, int tr) {               \
            hipLaunchKernelGGL((k_chacha_grouped<B>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);   \
        }, w, 96.0 * B, trips / (2 * B) + 1, cus, d_out, d_stamps);                                           \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
#define RUN_BANK(SK, X, NAME)                                                                                 \
    for (int w : {2, 8}) {                                                                                    \
        Result r = run((const void *)k_bank<SK, X>, [&](int g, int t, size_t l, int tr) {                     \
            hipLaunchKernelGGL((k_bank<SK, X>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);         \
        }, w, 64.0, trips / 2, cus, d_out, d_stamps);                                                         \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_BANK(0, 0, "add v,v,v same bank")
    RUN_BANK(1, 0, "add v,v,v other bank")
    RUN_BANK(0, 1, "xor v,v,v same bank")
    RUN_BANK(1, 1, "xor v,v,v other bank")
#define RUN_CHACHA_ASM(F, NAME)                                                                               \
    for (int w : {1, 2, 4, 8}) {                                                                              \
        Result r = run((const void *)k_chacha_asm<F>, [&](int g, int t, size_t l, int tr) {                   \
            hipLaunchKernelGGL((k_chacha_asm<F>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);       \
        }, w, 192.0, trips / 4 + 1, cus, d_out, d_stamps);                                                    \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_CHACHA_ASM(0, "chacha asm x8 natural")
    RUN_CHACHA_ASM(1, "chacha asm x8 banks")
    RUN_CHACHA_G(1, "chacha grouped by 4")
    RUN_CHACHA_G(2, "chacha grouped by 8")
    RUN_CHACHA_G(4, "chacha grouped by 16")
#define RUN_PAT(PT, NAME)                                                                                    \
    for (int w : {1, 4, 8}) {                                                                                 \
        Result r = run((const void *)k_pattern<PT>, [&](int g, int t, size_t l, int tr) {                     \
            hipLaunchKernelGGL((k_pattern<PT>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);         \
        }, w, 64.0, trips / 2, cus, d_out, d_stamps);                                                         \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_PAT(PT_ADD_E64, "v_add_u32_e64 (VOP3)")
    RUN_PAT(PT_ADD_LIT, "v_add_u32 +literal")
    RUN_PAT(PT_LSHL, "v_lshlrev_b32")
    RUN_PAT(PT_PERM, "v_perm_b32")
    RUN_PAT(PT_ALIGNBYTE, "v_alignbyte_b32")
    RUN_PAT(PT_LSHL_OR, "v_lshl_or_b32")
    RUN_PAT(PT_XAD, "v_xad_u32")
    RUN_PAT(PT_ADD3, "v_add3_u32")
    RUN_PAT(PT_BFI, "v_bfi_b32")
    RUN_PAT(PT_BITOP3, "v_bitop3_b32")
    RUN_PAT(PT_XOR_SDWA, "v_xor_b32_sdwa")
    RUN_PAT(PT_MOV_DPP, "v_mov_b32_dpp row_ror")
    RUN_PAT(PT_PK_ADD16, "v_pk_add_u16")
    RUN_PAT(PT_MUL24, "v_mul_u32_u24")
    RUN_PAT(PT_MAD24, "v_mad_u32_u24")
    RUN_PAT(PT_MULLO, "v_mul_lo_u32")
    RUN_PAT(PT_AX, "mix add,xor")
    RUN_PAT(PT_AR, "mix add,alignbit")
    RUN_PAT(PT_AAR, "mix add,xor,add,alignbit")
    RUN_PAT(PT_AXR, "mix add,xor,alignbit")
    RUN_PAT(PT_AX_PERM, "mix add,xor,perm")
    RUN_PAT(PT_ROT2, "rot = lshr + lshl_or")
#define RUN_PAT48(PT, NAME)                                                                                  \
    for (int w : {1, 4, 8}) {                                                                                 \
        Result r = run((const void *)k_pattern<PT>, [&](int g, int t, size_t l, int tr) {                     \
            hipLaunchKernelGGL((k_pattern<PT>), dim3(g), dim3(t), l, 0, d_out, d_stamps, 12345u, tr);         \
        }, w, 48.0, trips / 2, cus, d_out, d_stamps);                                                         \
        printf("  %-22s %5d %8.3f %12.1f %10.3f %12.2f\n", NAME, w, r.ms, r.ginstr_s, r.clock_ghz,            \
               r.cyc_per_instr_simd);                                                                         \
    }
    RUN_PAT48(PT_RUN2, "runs of 2: a,a,x,x,r,r")
    RUN_PAT48(PT_RUN4, "runs of 4")
    RUN_PAT48(PT_RUN8, "runs of 8")
    RUN_PAT48(PT_RUN16, "runs of 16")
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_stamps));
    return 0;
}
