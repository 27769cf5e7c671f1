// Device templates of the MFMA GEMM for gfx950 (included by gemm.hip: fp32 operands on v_mfma_f32_32x32x2_f32, and by
// gemm_bf16.hip: operands rounded to bf16 at LDS staging, v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
// Design (MI355X_MICROARCH: the f32 MFMA runs at the f32 vector rate, 64 cycles per 32x32x2 per SIMD, so the fp32 loop
// is MFMA-paced and LDS/global staging only has to stay out of the way; the bf16 loop is paced by operand traffic):
//   * square block tile BT x BT (128 or 64), 256 threads = 4 waves in a 2x2 grid, each wave owns (BT/2)x(BT/2) outputs
//     as (BT/64)^2 accumulators of 32x32; BK = 16 / 32 (fp32, by tile) or 32 / 64 (bf16);
//   * operands are staged global -> registers -> LDS in a row-major [BT][BK + 16 bytes] image of each operand: a
//     lane's MFMA fragment (4 fp32 or 8 bf16 consecutive k of its row) is one conflict-free ds_read_b128;
//   * S register stages of global loads are in flight under the MFMAs; loads are unconditional on clamped addresses;
//   * two-level addressing on every operand axis (see cgn_hip.h) covers transposed operands, overlapping conv
//     windows and [B,T+1,H] history buffers without copies;
//   * tall-K / few-tile problems are split along K into fp32 slabs, summed in fixed order by a second kernel
//     (bitwise reproducible, no float atomics);
//   * blockIdx -> tile mapping is XCD-aware: blocks that share an XCD (blockIdx % 8) walk neighbouring tiles.
#pragma once
#include <type_traits>

#include "common.h"

namespace cgn_gemm {

// Two-level axis, off(i) = (i / rpg) * gs + (i % rpg) * stride, evaluated WITHOUT a branch and without a hardware
// division (both would sit between the global loads of the K loop, see TileLoader::load): i / rpg is a multiply by
// magic = ceil(2^40 / rpg) and a shift, exact for i < 2^24 and rpg < 2^16 (checked on the host). A plain axis is the
// same formula with rpg = 2^30, gs = 0 (i / rpg == 0 for every i < 2^30).
struct Axis {
    long stride, gs;
    unsigned long magic;
    int rpg;
};

__device__ __forceinline__ long axis_off(const Axis& a, int i) {
    const int g = (int)(((unsigned long)(unsigned)i * a.magic) >> 40);
    return (long)g * a.gs + (long)(i - g * a.rpg) * a.stride;
}

static inline Axis make_axis(const cgn_axis& c) {
    Axis a;
    a.stride = c.stride;
    if (c.rpg > 0) {
        a.rpg = (int)c.rpg;
        a.gs = c.gs;
        a.magic = ((1UL << 40) + (unsigned long)c.rpg - 1) / (unsigned long)c.rpg;
    } else {
        a.rpg = 1 << 30;
        a.gs = 0;
        a.magic = 1UL << 10;
    }
    return a;
}

struct GemmParams {
    const float* A;
    const float* B;
    float* C;
    const float* bias0;
    const float* bias1;
    const float* a_row_param;
    const float* ep_logits;
    const float* ep_lse;
    const float* ep_rowdot;
    const int64_t* ep_target;
    float* ws;
    const float* dyn;  // optional device scalars {inv_tau, ce_scale}: override a_scale/ep_inv_tau and ep_ce_scale
    long ldc, ep_ld;
    long ldw;  // row stride of the split-K slabs: N rounded up to 4, so the reduce always reads 16-byte quads
    Axis am, ak, bk, bn;
    int M, N, K;
    int k_per_split, splitk;
    int a_vec, b_vec;
    int a_transform, epilogue;
    int tiles_m, tiles_n;
    int order;  // 0 / 1: XCD remap over (split, tile); 2: remap over tiles only (the shorter tile axis runs fastest in all)
    float alpha, beta, a_scale, ep_inv_tau, ep_ce_scale;
    // fast K addressing (template flag FK): both K axes are plain and both operands span < 2^31 bytes, so a lane's byte
    // offset is fixed for the whole K loop and the K position is a scalar (buffer loads: voffset + soffset)
    int fastk;
    unsigned a_bytes, b_bytes;  // bounds of the two buffer resources (loads beyond return 0)
    // bf16 STORAGE (cgn_gemm_desc.storage): A is a k-contiguous bf16 matrix (the kernel then addresses it in 4-byte words:
    // am/ak strides and the K positions handed to A's loader are in WORDS = pairs of k); C / ep_logits hold bf16;
    // ct = optional transposed bf16 copy of C (ct[n * ldct + m])
    int a16, c16, ep16;
    void* ct;
    long ldct;
    // CGN_EPI_LINEAR_STATS (template flag LSE): per (row, 64-column chunk) softmax statistics of ep_inv_tau * C
    float* stats;
    long ldstats;
};

constexpr int g_unroll_mult = 2;
constexpr int KROUND = 32;  // split-K ranges start on multiples of 32 (aligned k quads; the fp32 BKs divide it)
constexpr int NT = 256;

__device__ __forceinline__ float ld_bf16(const void* base, long i) {
    return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(base)[i] << 16);
}
__device__ __forceinline__ void st_bf16(void* base, long i, float v) {
    reinterpret_cast<__bf16*>(base)[i] = (__bf16)v;  // round to nearest even
}

__device__ __forceinline__ float apply_epilogue(const GemmParams& p, int row, int col, float acc) {
    if (p.epilogue == CGN_EPI_SOFTMAX_BWD) {
        const float inv_tau = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        const float ce_scale = p.dyn ? p.dyn[1] : p.ep_ce_scale;
        const long li = (long)row * p.ep_ld + col;
        const float pr = __expf((p.ep16 ? ld_bf16(p.ep_logits, li) : p.ep_logits[li]) * inv_tau - p.ep_lse[row]);
        const float onehot = (p.ep_target[row] == (int64_t)col) ? 1.f : 0.f;
        return pr * (acc - p.ep_rowdot[row]) * inv_tau + ce_scale * (pr - onehot);
    }
    float v = p.alpha * acc;
    if (p.bias0) v += p.bias0[col];
    if (p.bias1) v += p.bias1[col];
    if (p.beta != 0.f) v += p.beta * p.C[(long)row * p.ldc + col];
    return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest-even fp32 -> bf16 pairs (hipcc emits v_cvt_pk_bf16_f32 for the casts)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x2 pack_bf16x4(const f32x4& v) {
    u32x2 w;
    w[0] = pack_bf16x2(v[0], v[1]);
    w[1] = pack_bf16x2(v[2], v[3]);
    return w;
}

// X6: the exact three-way split of fp32 operands, x = p0 + p1 + p2 with every term a bf16 (round to nearest: p0 = bf16(x),
// p1 = bf16(x - p0), p2 = x - p0 - p1; both differences are exact in fp32 and the last one has at most 8 significant bits,
// so the sum is x bit for bit). A product then runs on the bf16 matrix pipe as SIX partial products
//   a.b ~= a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),    dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|
// accumulated in fp32: the accuracy of an fp32 product (unit roundoff 2^-24) at 6 x 32 instead of 8 x 64 matrix-pipe
// cycles per 32x32x16 block (2.67x less MFMA time). Here: two elements at a time, packed pairs per term.
__device__ __forceinline__ void split3_pk(float a, float b, unsigned& w0, unsigned& w1, unsigned& w2) {
    w0 = pack_bf16x2(a, b);
    float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = pack_bf16x2(ra, rb);
    ra -= __builtin_bit_cast(float, w1 << 16);
    rb -= __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = pack_bf16x2(ra, rb);
}

// Operand tile loader. The tile is ROWS (m or n index) x BK (k index); its LDS image is row-major [ROWS][LDK] with
// LDK = BK + 4 floats (144-byte rows): the MFMA loop reads a lane's 4 consecutive k of one row with ONE ds_read_b128
// (rows 144 B apart put 16 consecutive rows on 16 different 16-byte slots of the 256-byte bank row: conflict-free),
// and every staging write is a ds_write_b128 of 4 consecutive k of one row.
//   KC = true : memory is contiguous along k (row-major [rows][k]): thread = (row, k-quad), one float4 load per chunk
//   KC = false: memory is contiguous along the row index ([k][rows]): thread = one row, lanes run along rows, scalar
//               loads (a wave covers 64 consecutive rows = 256 contiguous bytes per k), 4 k per chunk
// FK: 0 = generic two-level addressing in the K loop; 1 = fast (plain K axis); 2 = fast, two-level K axis with
// rpg >= BK/2 (row-vector form only: the history-buffer operands of the dW products, k = (sentence, timestep))
// SRC16: the operand is a k-contiguous bf16 matrix addressed in 4-byte words (BK, strides and k positions count WORDS =
// pairs of k; BF is false: a loaded quad of words IS 8 consecutive bf16 of the bf16 LDS image, whose row stride
// (BK_elements + 8) / 2 words equals this loader's BK + 4). Only the exp transform looks inside a word.
// X6 (with BF): the operand is split into three bf16 terms at staging (split3_pk); the LDS image is three bf16 planes,
// PLANE words apart. A k-contiguous operand uses the bf16 row-major image per plane; a row-contiguous one a
// "pair-major" image [BK/2][ROWS] of 4-byte words (the two consecutive k of one row): the lane's quad of rows of one k
// pair is ONE conflict-free 16-byte write per plane, and the MFMA loop reads a lane's 8 k with four 4-byte reads, lanes
// along the rows (the row-major bf16 image needs transposing 4-byte writes, 8-way bank-conflicted).
template <int ROWS, bool KC, int BK, int S, bool VEC, bool EXP, bool BF, int FK = 0, bool SRC16 = false, bool X6 = false>
struct TileLoader {
    static_assert(!X6 || (BF && VEC && !SRC16), "X6: fp32 operand, 16-byte loads, bf16 planes in LDS");
    static_assert(!SRC16 || (KC && VEC && !BF), "bf16 storage: k-contiguous, 16-byte loads, raw copy into the bf16 image");
    static_assert(!FK || VEC, "fast K addressing is built for the 16-byte load paths");
    static_assert(FK != 2 || !KC, "the two-level fast form exists for row-contiguous operands");
    // LDS row stride in 4-byte words: BK + 4 floats, or (BF) BK + 8 bf16: 16 bytes of padding either way
    static constexpr int LDK = BF ? (BK + 8) / 2 : BK + 4;
    static constexpr int PLANE = ROWS * LDK;     // X6: words between the planes of the three terms
    static constexpr int QUADS = ROWS * BK / 4;  // 4-k chunks per tile
    static constexpr int PER = QUADS / NT;       // chunks per thread
    static_assert(QUADS % NT == 0, "tile/threads mismatch");
    static constexpr int KGROUPS = NT / ROWS;    // !KC: thread groups along k
    static_assert(KC || (BK / KGROUPS == 4 * PER), "k coverage");

    // !KC && VEC ("row-vector" form): memory is contiguous along the row index and rows come in aligned quads, so a
    // thread loads 4 consecutive rows of ONE k with a 16-byte load (RQ row quads x KG k groups of KP k each) and
    // transposes on the way into the k-major LDS image (4 x KP-float writes). The scalar form issued 4x the load
    // instructions and measured 30-55 % slower on every NN / TN product of the step (tools/gemm_bench.py).
    static constexpr int RQ = ROWS / 4;
    static constexpr int KG = NT / RQ;
    static constexpr int KP = BK / KG;
    static_assert(KC || !VEC || ((KP == 2 || KP == 4) && KP == PER), "row-vector loader geometry");

    f32x4 regs[S][PER];  // S register stages: S tiles of global loads in flight per thread
    // FK: buffer resource of the operand, the lane's fixed byte offsets and the byte stride of one k
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[FK ? (KC ? PER : KP) : 1];
    int kbytes;
    int c0, wrap_bytes;  // FK == 2: the lane's first k within the K step; byte jump at a group boundary
    long roff[KC ? PER : 1];
    bool rok[KC ? PER : 1];
    bool rok4[4];
    float rparam[KC ? PER : 1];

    __device__ __forceinline__ void init(const Axis& ar, int r0, int rlimit, const float* row_param) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
            constexpr int QPR = BK / 4;   // quads per row
            constexpr int RS = NT / QPR;  // row step between a thread's chunks
            const int r = tid / QPR;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int row = r0 + r + i * RS;
                rok[i] = row < rlimit;
                roff[i] = rok[i] ? axis_off(ar, row) : 0;  // clamped: row 0 of the operand is always addressable
                rparam[i] = EXP ? row_param[rok[i] ? row : 0] : 0.f;
            }
        } else if constexpr (VEC) {
            const int row = r0 + 4 * (tid % RQ);
            rok[0] = row < rlimit;
#pragma unroll
            for (int i = 0; i < 4; ++i) rok4[i] = row + i < rlimit;
            roff[0] = rok[0] ? axis_off(ar, row) : 0;  // a quad never splits a group: rpg % 4 == 0 (host check)
            rparam[0] = 0.f;
        } else {
            const int row = r0 + tid % ROWS;
            rok[0] = row < rlimit;
            roff[0] = rok[0] ? axis_off(ar, row) : 0;
            rparam[0] = 0.f;
        }
    }

    // Issue the global loads of the tile whose first k is k0 (k valid in [k0, kend); kbeg = first k of this block).
    // The code is BRANCH-FREE: every load is unconditional on a clamped (always valid) address and nothing here
    // consumes the loaded registers, so S tiles of loads stay in flight under the MFMAs. Invalid elements are zeroed
    // (and the optional exp transform applied) in store(), right before the LDS write. (Any branch around or between
    // the loads — even a block-uniform one — makes hipcc's wait-count pass fall back to vmcnt(0) at the next use and
    // serialises the L2 round trips; measured: 1.25 us per K step for a lone block.)
    //   VEC: one 16-byte load per chunk. A quad may straddle kend only at the end of the contiguous axis, where the
    //        host has checked that the padded row stride covers it; quads wholly beyond kend re-read the quad at kbeg.
    // FK only, after init(): the operand's buffer resource and this lane's byte offsets at k = 0
    __device__ __forceinline__ void init_fast(const float* base, unsigned bytes, const Axis& ak) {
        const int tid = threadIdx.x;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
        kbytes = (int)ak.stride * 4;
        if constexpr (KC) {
            constexpr int QPR = BK / 4;
#pragma unroll
            for (int i = 0; i < PER; ++i) voff[i] = ((int)roff[i] + (tid % QPR) * 4) * 4;
        } else {
#pragma unroll
            for (int j = 0; j < KP; ++j) voff[j] = (int)roff[0] * 4 + ((tid / RQ) * KP + j) * kbytes;
            c0 = (tid / RQ) * KP;
            wrap_bytes = (int)(ak.gs - (long)ak.rpg * ak.stride) * 4;
        }
    }

    __device__ __forceinline__ void load(int st, const float* __restrict__ base, const Axis& ak, int k0, int kbeg,
                                         int kend) {
        const int tid = threadIdx.x;
        if constexpr (FK) {
            // No address arithmetic in the K loop (the generic path below spends ~45 VALU instructions per K step on
            // 64-bit offset math, clustered between the MFMAs: 10-17 % of the kernel, tools/experiments/gemm_loop_probe).
            // Nothing is clamped: a load beyond the operand returns 0 (buffer bounds), one beyond kend but inside it reads
            // values that store<true>() masks (tiles that are stored unmasked lie wholly inside the K range).
            if constexpr (FK == 2) {
                // k = k0 + c, c < BK <= 2 rpg: the group and rest of k0 are scalar work, a lane crosses at most two
                // group boundaries from there (two compares and selects per load instead of the 64-bit offset math)
                const int ku = __builtin_amdgcn_readfirstlane(k0);
                const int g0 = (int)(((unsigned long)(unsigned)ku * ak.magic) >> 40);
                const int r0 = ku - g0 * ak.rpg;
                const int soff = (int)((long)g0 * ak.gs + (long)r0 * ak.stride) * 4;
                const int th1 = ak.rpg - r0, th2 = th1 + ak.rpg;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const int c = c0 + j;
                    const int w = (c >= th1 ? wrap_bytes : 0) + (c >= th2 ? wrap_bytes : 0);
                    regs[st][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j] + w, soff, 0));
                }
                return;
            }
            const int soff = __builtin_amdgcn_readfirstlane(k0) * kbytes;
            constexpr int NL = KC ? PER : KP;
#pragma unroll
            for (int i = 0; i < NL; ++i)
                regs[st][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], soff, 0));
            return;
        }
        if constexpr (KC) {
            constexpr int QPR = BK / 4;
            const int q = tid % QPR;
            const int k = k0 + q * 4;
            if constexpr (VEC) {
                const long koff = axis_off(ak, k < kend ? k : kbeg);
#pragma unroll
                for (int i = 0; i < PER; ++i) regs[st][i] = *reinterpret_cast<const f32x4*>(base + roff[i] + koff);
            } else {
                long koff[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) koff[j] = axis_off(ak, min(k + j, kend - 1));
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = base[roff[i] + koff[j]];
                    regs[st][i] = v;
                }
            }
        } else if constexpr (VEC) {
            const int kb = k0 + (tid / RQ) * KP;
            const float* rb = base + roff[0];
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int k = kb + j;
                regs[st][j] = *reinterpret_cast<const f32x4*>(rb + axis_off(ak, k < kend ? k : kbeg));
            }
        } else {
            const int kb = k0 + (tid / ROWS) * (4 * PER);  // wave-uniform
            const float* rb = base + roff[0];
            // consecutive k: one magic division, then step (group, rest) by one with a select
            const int kf = min(kb, kend - 1);
            int g = (int)(((unsigned long)(unsigned)kf * ak.magic) >> 40);
            int r = kf - g * ak.rpg;
            int kk = kf;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = rb[(long)g * ak.gs + (long)r * ak.stride];
                    const bool adv = kk < kend - 1;  // clamp at the last valid k
                    kk += adv ? 1 : 0;
                    r += adv ? 1 : 0;
                    const bool wrap = r == ak.rpg;
                    g += wrap ? 1 : 0;
                    r = wrap ? 0 : r;
                }
                regs[st][i] = v;
            }
        }
    }

    // Transform the staged registers and write them into the LDS image dst[ROWS][LDK]. MASK = true zeroes what lies
    // beyond kend (the K tail MUST contribute zeros) and beyond the last row; MASK = false is for tiles that lie wholly
    // inside [kbeg, kend): rows beyond the operand were loaded from a clamped (valid) address and only ever reach output
    // rows / columns that the epilogue never stores, so they need no zeroing - and the ~40 compare/select instructions
    // per K step that the masks cost leave the loop that feeds the MFMAs.
    template <bool MASK = true>
    __device__ __forceinline__ void store(int st, float* __restrict__ dst, int k0, int kend, float scale) const {
        const int tid = threadIdx.x;
        if constexpr (KC) {
            constexpr int QPR = BK / 4;
            constexpr int RS = NT / QPR;
            const int q = tid % QPR;
            const int r = tid / QPR;
            const int k = k0 + q * 4;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v = regs[st][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j];
                    // hardware exponential (v_exp_f32 on the base-2 argument): ~1e-6 relative, two instructions
                    // instead of expf's ~15 in the loop that feeds the MFMAs (x <= 0 here: x*scale - lse)
                    if constexpr (EXP && SRC16) {
                        const unsigned w = __builtin_bit_cast(unsigned, x);
                        const float lo = __expf(__builtin_bit_cast(float, w << 16) * scale - rparam[i]);
                        const float hi = __expf(__builtin_bit_cast(float, w & 0xffff0000u) * scale - rparam[i]);
                        x = __builtin_bit_cast(float, pack_bf16x2(lo, hi));
                    } else if constexpr (EXP) {
                        x = __expf(x * scale - rparam[i]);
                    }
                    if constexpr (MASK) x = (rok[i] && k + j < kend) ? x : 0.f;
                    v[j] = x;
                }
                if constexpr (X6) {
                    unsigned t0[2], t1[2], t2[2];
                    split3_pk(v[0], v[1], t0[0], t1[0], t2[0]);
                    split3_pk(v[2], v[3], t0[1], t1[1], t2[1]);
                    const u32x2 w0 = {t0[0], t0[1]}, w1 = {t1[0], t1[1]}, w2 = {t2[0], t2[1]};
                    float* d = dst + (r + i * RS) * LDK + q * 2;
                    *reinterpret_cast<u32x2*>(d) = w0;
                    *reinterpret_cast<u32x2*>(d + PLANE) = w1;
                    *reinterpret_cast<u32x2*>(d + 2 * PLANE) = w2;
                } else if constexpr (BF)
                    *reinterpret_cast<u32x2*>(dst + (r + i * RS) * LDK + q * 2) = pack_bf16x4(v);
                else
                    *reinterpret_cast<f32x4*>(dst + (r + i * RS) * LDK + q * 4) = v;
            }
        } else if constexpr (VEC && X6) {
            // pair-major planes [BK/2][ROWS]: word = (k, k+1) of one row
            const int row = 4 * (tid % RQ);
            const int kl = (tid / RQ) * KP;
#pragma unroll
            for (int j = 0; j < KP; j += 2) {
                f32x4 lo = regs[st][j], hi = regs[st][j + 1];
                if constexpr (MASK) {
                    const bool k0ok = k0 + kl + j < kend, k1ok = k0 + kl + j + 1 < kend;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[i] = (rok4[i] && k0ok) ? lo[i] : 0.f;
                        hi[i] = (rok4[i] && k1ok) ? hi[i] : 0.f;
                    }
                }
                unsigned t0[4], t1[4], t2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) split3_pk(lo[i], hi[i], t0[i], t1[i], t2[i]);
                const u32x4_t w0 = {t0[0], t0[1], t0[2], t0[3]}, w1 = {t1[0], t1[1], t1[2], t1[3]},
                              w2 = {t2[0], t2[1], t2[2], t2[3]};
                float* d = dst + ((kl + j) / 2) * ROWS + row;
                *reinterpret_cast<u32x4_t*>(d) = w0;
                *reinterpret_cast<u32x4_t*>(d + PLANE) = w1;
                *reinterpret_cast<u32x4_t*>(d + 2 * PLANE) = w2;
            }
        } else if constexpr (VEC && !BF) {
            // k-major image [BK][ROWS]: the lane's quad (4 consecutive rows of one k) goes out as ONE 16-byte write, lanes
            // along the rows (conflict-free). The row-major image needed 4 transposing 8-byte writes per quad, 8-way
            // bank-conflicted (rows 4 apart are 16 banks apart whatever the padding): 78 % of the TN kernel's LDS cycles.
            const int row = 4 * (tid % RQ);
            const int kl = (tid / RQ) * KP;
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                f32x4 v = regs[st][j];
                if constexpr (MASK) {
                    const bool kok = k0 + kl + j < kend;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (rok4[i] && kok) ? v[i] : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + (kl + j) * ROWS + row) = v;
            }
        } else if constexpr (VEC) {
            const int row = 4 * (tid % RQ);
            const int kl = (tid / RQ) * KP;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[KP];
#pragma unroll
                for (int j = 0; j < KP; ++j)
                    v[j] = (!MASK || (rok4[i] && k0 + kl + j < kend)) ? regs[st][j][i] : 0.f;
                if constexpr (BF) {
#pragma unroll
                    for (int j = 0; j < KP; j += 2)
                        reinterpret_cast<unsigned*>(dst)[(row + i) * LDK + (kl + j) / 2] = pack_bf16x2(v[j], v[j + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < KP; j += 2) {
                        f32x2 w;
                        w[0] = v[j];
                        w[1] = v[j + 1];
                        *reinterpret_cast<f32x2*>(dst + (row + i) * LDK + kl + j) = w;
                    }
                }
            }
        } else if constexpr (!BF) {
            const int row = tid % ROWS;  // k-major image, lanes along the rows: 4-byte writes, conflict-free
            const int kl = (tid / ROWS) * (4 * PER);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = regs[st][i][j];
                    dst[(kl + 4 * i + j) * ROWS + row] = (!MASK || (rok[0] && k0 + kl + 4 * i + j < kend)) ? x : 0.f;
                }
            }
        } else {
            const int row = tid % ROWS;
            const int kl = (tid / ROWS) * (4 * PER);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v = regs[st][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (!MASK || (rok[0] && k0 + kl + 4 * i + j < kend)) ? v[j] : 0.f;
                if constexpr (BF)
                    *reinterpret_cast<u32x2*>(dst + row * LDK + (kl + 4 * i) / 2) = pack_bf16x4(v);
                else
                    *reinterpret_cast<f32x4*>(dst + row * LDK + kl + 4 * i) = v;
            }
        }
    }
};

// A wave's staged 64x64 bf16 image (row stride RS elements, 16-byte aligned rows) -> dst[(row_base + r) * ld + col_base + c]:
// a lane moves 8 consecutive columns of one row (16 bytes), 8 lanes per row, 8 rows per pass; ragged edges and
// unaligned destinations fall back to 2-byte stores.
template <int RS>
__device__ __forceinline__ void flush_stage(const unsigned short* stage, int lane, void* dst, long ld, int row_base,
                                            int row_lim, int col_base, int col_lim) {
    const bool vec_ok = (ld % 8 == 0) && ((uintptr_t)dst % 16 == 0) && (col_base % 8 == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rl = it * 8 + (lane >> 3), cl = (lane & 7) * 8;
        const int row = row_base + rl, col = col_base + cl;
        const u32x4_t q = *reinterpret_cast<const u32x4_t*>(stage + rl * RS + cl);
        if (row >= row_lim || col >= col_lim) continue;
        unsigned short* out = reinterpret_cast<unsigned short*>(dst) + (long)row * ld + col;
        if (vec_ok && col + 7 < col_lim) {
            *reinterpret_cast<u32x4_t*>(out) = q;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col + e < col_lim) out[e] = (unsigned short)((e & 1) ? (q[e >> 1] >> 16) : (q[e >> 1] & 0xffffu));
        }
    }
}

// XCD-aware bijective remap: blocks with equal (bid % 8) share an XCD's L2 (observed round-robin dispatch), so
// give each of the 8 residue classes one contiguous chunk of the tile list.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// BK: 16 for the 128-tile (3 blocks/CU by LDS and registers), 32 for the 64-tile (measured, tools/gemm_bench.py).
// S: register stages = K tiles of global loads in flight per thread. Mid-size products of the CGN step have only
// ~1 block per CU (260-700 tiles), so a wave must cover the L2/HBM round trip (~1-2 us) by itself: with one tile in
// flight it stalls every K step (0.43 us of MFMA work per step).
// VEC: bit 0 = operand A is read with 16-byte loads, bit 1 = operand B (each along its own contiguous axis)
// SMB: the batched softmax-backward epilogue is compiled in (only the dlogits product's variant: inside the shared
// variant its registers cost the vocabulary projection 10 %).
template <int BT, bool A_KC, bool B_KC, int VEC, bool EXP, int S, int BK, bool SMB = false, bool BF = false,
          int FK = 0, bool A16 = false, bool X6 = false, bool LSE = false>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int split, const int t) {
    static_assert(!LSE || (X6 && BT == 128 && !SMB), "the statistics epilogue is built for the 128-tile X6 kernel");
    static_assert(!A16 || (BF && A_KC && (VEC & 1)), "bf16-stored A: bf16 MFMA operands, k-contiguous, 16-byte loads");
    static_assert(!X6 || (BF && !A16 && VEC == 3), "X6: fp32 operands in memory, three bf16 planes in LDS");
    constexpr int NP = X6 ? 3 : 1;   // planes per operand image
    constexpr int TW = BT / 64;  // 32x32 accumulators per wave along each dim
    constexpr int LDK = BF ? (BK + 8) / 2 : BK + 4;  // LDS row stride (4-byte words) of both operand images
    // One LDS array (both operands, two buffers): keeps every carve offset explicit and 16-byte aligned.
    constexpr int TILE = BT * LDK;  // floats per operand image (per plane)
    constexpr int OPI = NP * TILE;  // one operand's image
    __shared__ __attribute__((aligned(16))) float lds[4 * OPI];

    int tile_m, tile_n;
    // the SHORTER tile axis runs fastest: tiles that are neighbours in the list (same XCD after the remap, resident
    // together) share a panel of the operand that spans the long axis - dW_fc's 107 MB dlogits are fetched once per row
    // panel instead of once per column tile (PMC: 445 -> see profiles/r02_pmc_traffic.txt; the kernel's time is unchanged)
    if (p.tiles_n < p.tiles_m) {
        tile_n = t % p.tiles_n;
        tile_m = t / p.tiles_n;
    } else {
        tile_m = t % p.tiles_m;
        tile_n = t / p.tiles_m;
    }
    const int m0 = tile_m * BT, n0 = tile_n * BT;

    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const float a_scale = (EXP && p.dyn) ? p.dyn[0] : p.a_scale;

    static_assert(!FK || VEC == 3, "fast K addressing needs 16-byte loads on both operands");
    // (FK == 2: only B's K axis is two-level.) A16: A's loader works in words, KA = 2 k per word
    constexpr int KA = A16 ? 2 : 1;
    TileLoader<BT, A_KC, BK / KA, S, (VEC & 1) != 0, EXP, BF && !A16, (FK ? 1 : 0), A16, X6> la;
    TileLoader<BT, B_KC, BK, S, (VEC & 2) != 0, false, BF, FK, false, X6> lb;
    la.init(p.am, m0, p.M, p.a_row_param);
    lb.init(p.bn, n0, p.N, nullptr);
    if constexpr (FK) {
        la.init_fast(p.A, p.a_bytes, p.ak);
        lb.init_fast(p.B, p.b_bytes, p.bk);
    }

    // the 64-tile has one 32x32 output per wave: two accumulator chains (even / odd k) halve the dependent-MFMA chain
    constexpr int NC = (TW == 1) ? 2 : 1;
    f32x16 acc[NC][TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c][i][j][r] = 0.f;

    // Pipeline (one barrier per K step, LDS double-buffered, S tiles of global loads in flight):
    //   tile j lives in register stage j % S from its load until it is written to LDS buffer j & 1.
    //   A K step is NBLK blocks of MFMAs (8 k each for fp32, 16 for bf16 operands); the fragments of a block are read
    //   from LDS one block ahead (register double buffer fa/fb), ACROSS the step boundary too:
    //   step kt:  block 0:         read fragments of block 1; write tile kt+1 (loaded during step kt-S) into
    //                              buf[(kt+1)&1]; issue the global loads of tile kt+1+S; MFMAs of block 0
    //             blocks 1..NBLK-2: read fragments of the next block; MFMAs
    //             block NBLK-1:    barrier (tile kt+1 is complete, every read of tile kt has returned); read the
    //                              fragments of block 0 of tile kt+1; MFMAs of block NBLK-1
    //   so a wave never sits in front of an LDS round trip with an empty MFMA pipe: the reads issued behind the barrier
    //   land under the last block's MFMAs. (Before this ordering a lone block per CU - every mid-size product of the
    //   CGN step - ran at half its MFMA rate: barrier, LDS store, LDS read, THEN the first MFMA of the step.)
    // Tiles that lie wholly inside [kbeg, kend) are written to LDS unmasked (hot loop); the last steps, whose next tile
    // is partial or past the end, run the same body with the masked store. Both loops are unrolled by U = lcm(S, 2)
    // multiples so that stage and buffer indices are compile-time constants.
    // Loads and LDS writes past the last tile are not skipped (that would need a branch, see TileLoader::load): they
    // re-read the block's first quads (L1/L2 hits) and write a buffer nobody reads again.
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;  // tiles wholly inside the K range
    if (nk > 0) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            la.load(j, p.A, p.ak, (kbeg + j * BK) / KA, kbeg / KA, kend / KA);
            lb.load(j, p.B, p.bk, kbeg + j * BK, kbeg, kend);
        }
        la.template store<true>(0, lds, kbeg / KA, kend / KA, a_scale);
        lb.template store<true>(0, lds + OPI, kbeg, kend, 0.f);
        la.load(0, p.A, p.ak, (kbeg + S * BK) / KA, kbeg / KA, kend / KA);
        lb.load(0, p.B, p.bk, kbeg + S * BK, kbeg, kend);
    }
    __syncthreads();

    const int fr = lane & 31;  // fragment row/col
    const int fk = lane >> 5;  // k-slice within the MFMA's k step
    constexpr int KB = BF ? 16 : 8;   // k per block of MFMAs
    constexpr int NBLK = BK / KB;     // blocks per K step
    static_assert(NBLK == 1 || NBLK % 2 == 0, "the fragment double buffer wraps at the step boundary");
    // fragment registers: [double buffer][tile index]; fp32: 4 consecutive k of one row = one ds_read_b128, bf16: 8 k
    typedef typename std::conditional<BF, bf16x8, f32x4>::type frag_t;
    frag_t fa[2][NP][TW], fb[2][NP][TW];
    // lane's fragment base inside an operand image (4-byte words)
    const int a_lane = (wm * (BT / 2) + fr) * LDK + 4 * fk;
    const int b_lane = (wn * (BT / 2) + fr) * LDK + 4 * fk;
    // A row-contiguous fp32 operand has a k-major LDS image [BK][BT] (TileLoader::store): the same 4 k per lane come
    // from 4 reads of 4 bytes, lanes along the rows (two runs of 32 consecutive floats: conflict-free).
    constexpr bool A_KM = !A_KC && !BF, B_KM = !B_KC && !BF;
    const int a_lane_km = (4 * fk) * BT + wm * (BT / 2) + fr;
    const int b_lane_km = (4 * fk) * BT + wn * (BT / 2) + fr;
    // X6, row-contiguous operand: pair-major planes [BK/2][BT] (TileLoader::store): word e of the lane's fragment is the
    // pair (2e, 2e+1) of its 8 k = four 4-byte reads at the same lane offsets as the k-major fp32 image
    constexpr bool A_PM = !A_KC && X6, B_PM = !B_KC && X6;
    auto read_frags = [&](const float* As_, const float* Bs_, int jb, int slot) {
        // block jb: k = jb*KB .. jb*KB+KB-1; lane half fk holds 4 (fp32) / 8 (bf16) consecutive k = 4 words either way
        constexpr int WPB = BF ? 8 : 8;  // words per block per row: fp32 8 floats; bf16 16 halves = 8 words
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                if constexpr (A_PM) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = As_[pl * TILE + a_lane_km + (jb * 8 + e) * BT + i * 32];
                    fa[slot][pl][i] = __builtin_bit_cast(frag_t, w);
                } else if constexpr (A_KM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[slot][pl][i][e] = As_[a_lane_km + (jb * 8 + e) * BT + i * 32];
                } else {
                    fa[slot][pl][i] = *reinterpret_cast<const frag_t*>(As_ + pl * TILE + a_lane + i * 32 * LDK + jb * WPB);
                }
            }
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                if constexpr (B_PM) {
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = Bs_[pl * TILE + b_lane_km + (jb * 8 + e) * BT + j * 32];
                    fb[slot][pl][j] = __builtin_bit_cast(frag_t, w);
                } else if constexpr (B_KM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[slot][pl][j][e] = Bs_[b_lane_km + (jb * 8 + e) * BT + j * 32];
                } else {
                    fb[slot][pl][j] = *reinterpret_cast<const frag_t*>(Bs_ + pl * TILE + b_lane + j * 32 * LDK + jb * WPB);
                }
            }
        }
    };
    // X6: six partial products per accumulator, term-major (the same accumulator comes round every TW*TW MFMAs; the
    // 64-tile alternates its two chains), small terms first; terms [q0, q1) of the block in fragment slot `slot`
    auto mfma_x6_terms = [&](int slot, int q0, int q1) {
        constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q < q0 || q >= q1) continue;
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[q % NC][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][X6 ? TA[q] : 0][i],
                                                                               fb[slot][X6 ? TB[q] : 0][j],
                                                                               acc[q % NC][i][j], 0, 0, 0);
        }
    };
    auto mfma_block = [&](int slot) {
        if constexpr (X6) {
            mfma_x6_terms(slot, 0, 6);
        } else if constexpr (BF) {
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][0][i], fb[slot][0][j], acc[0][i][j], 0, 0, 0);
        } else {
            // K order inside a block: lane half fk holds k = 4fk + e for the e-th MFMA (any permutation of k is legal as
            // long as A and B use the same one)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TW; ++i)
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        acc[e % NC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][0][i][e], fb[slot][0][j][e],
                                                                                 acc[e % NC][i][j], 0, 0, 0);
        }
    };
    if (nk > 0) read_frags(lds, lds + OPI, 0, 0);

    // one K step; u = step index within the unrolled trip (compile-time after unrolling), MASK = masked LDS store
    auto k_step = [&](auto mask_tag, int u, int kt) {
        constexpr bool MASK = decltype(mask_tag)::value;
        const float* As = lds + (u & 1) * 2 * OPI;
        const float* Bs = As + OPI;
        float* An = lds + ((u + 1) & 1) * 2 * OPI;
        const int st = (u + 1) % S;
        const int k1 = kbeg + (kt + 1) * BK;
        if constexpr (NBLK == 1) {
            // One block of MFMAs per K step (X6, 128-tile: 24 MFMAs = 768 matrix-pipe cycles per wave and step); the
            // fragment slot alternates with the step. A wave issues in order, so its own staging work of the next tile
            // (the three-way split: ~100 VALU instructions, 12 LDS writes, 4 global loads) is INTERLEAVED with the first
            // five terms' MFMAs (7 issue slots under each 32-cycle MFMA): two resident blocks that run in lock step - both
            // splitting, then both multiplying - halve the matrix pipe's duty (measured: 0.43). QA = terms issued in front
            // of the barrier, the rest behind it (covering the round trip of the next tile's fragment reads); swept with
            // the VALU count per slot in the step (tools/r03_x6_sched.sh, profiles/r03_x6_sched_variants.txt): all six
            // terms in front with five VALU per slot is best by 0.3-0.5 %, eight per slot or four terms in front lose 1 %
            // - the other resident block covers the reads.
#ifndef CGN_X6_QA
#define CGN_X6_QA 6
#endif
#ifndef CGN_X6_VALU_PER_SLOT
#define CGN_X6_VALU_PER_SLOT 5
#endif
            constexpr int QA = CGN_X6_QA;  // (tuning probes: tools/r03_x6_sched.sh builds variants with -D)
            la.template store<MASK>(st, An, k1 / KA, kend / KA, a_scale);
            lb.template store<MASK>(st, An + OPI, k1, kend, 0.f);
            la.load(st, p.A, p.ak, (k1 + S * BK) / KA, kbeg / KA, kend / KA);
            lb.load(st, p.B, p.bk, k1 + S * BK, kbeg, kend);
            mfma_x6_terms(u & 1, 0, QA);
#pragma unroll
            for (int g = 0; g < QA * TW * TW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, CGN_X6_VALU_PER_SLOT, 0);  // VALU of the split
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // an LDS write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // a global load
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            read_frags(An, An + OPI, 0, (u + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_x6_terms(u & 1, QA, 6);
            return;
        }
#pragma unroll
        for (int jb = 0; jb < NBLK; ++jb) {
            if (jb + 1 < NBLK) {
                read_frags(As, Bs, jb + 1, (jb + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);  // issued BEFORE this block's MFMAs, which then cover the round trip
            }
            if (jb == 0) {
                la.template store<MASK>(st, An, k1 / KA, kend / KA, a_scale);
                lb.template store<MASK>(st, An + OPI, k1, kend, 0.f);
                la.load(st, p.A, p.ak, (k1 + S * BK) / KA, kbeg / KA, kend / KA);
                lb.load(st, p.B, p.bk, k1 + S * BK, kbeg, kend);
            }
            if (jb == NBLK - 1) {
                // pin the order: this block's MFMAs stay BELOW the barrier and the reads behind it (left alone, the
                // scheduler hoists the MFMAs above the barrier and the wave meets the LDS round trip with an empty pipe)
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                read_frags(An, An + OPI, 0, 0);  // block 0 of the next tile: lands under this block's MFMAs
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_block(jb & 1);
        }
    };
    // U steps per trip, a multiple of lcm(S, 2): hipcc's wait-count pass is exact inside the unrolled body but
    // waits for ALL outstanding loads at the first LDS store after the loop header, so make trips long.
    constexpr int U = g_unroll_mult * ((S % 2 == 0) ? S : 2 * S);
    // step kt writes tile kt+1: unmasked while kt + 2 <= nfull
    const int n_hot = nfull >= 2 ? ((nfull - 1) / U) * U : 0;
    for (int kt0 = 0; kt0 < n_hot; kt0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) k_step(std::false_type{}, u, kt0 + u);
    }
    for (int kt0 = n_hot; kt0 < nk; kt0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kt0 + u >= nk) break;  // block-uniform
            k_step(std::true_type{}, u, kt0 + u);
        }
    }

    if constexpr (NC == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][0][r] += acc[1][0][0][r];
    }

    // Epilogue. C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    if constexpr (SMB) {
        // Fused softmax/CE backward (dlogits written over the logits): per 32-row slab the per-row operands (lse,
        // rowdot, target) are loaded once and ALL logits of the slab are requested (clamped addresses, no branch)
        // before the first exp, instead of four dependent loads per output element.
        // bf16 C (cgn_gemm_desc.storage): the wave's 64x64 sub-tile is staged through LDS (free after the K loop) and
        // leaves as 16-byte row stores; the optional transposed copy ct is staged the same way from the packed values.
        static_assert(BT == 128, "the staged bf16 epilogue is laid out for the 128-tile (4 waves x 64x64)");
        constexpr int RS = 72;  // staging row stride in bf16 elements (144 B: 16-byte aligned rows)
        static_assert(4 * 64 * RS * 2 <= 4 * OPI * 4, "staging images must fit the operand buffers");
        unsigned short* stage = reinterpret_cast<unsigned short*>(lds) + wave * (64 * RS);
        const float inv_tau = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        const float ce_scale = p.dyn ? p.dyn[1] : p.ep_ce_scale;
        u32x2 pk[TW][TW][4];  // bf16 C only: the lane's values, four consecutive rows of one column per entry
        if (p.c16) __syncthreads();  // (uniform) every wave is done with the operand images
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            float lse_r[16], rd_r[16], lg[TW][16];
            int tg_r[16];
            bool rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                rv[r] = row < p.M;
                const int rc = rv[r] ? row : 0;
                lse_r[r] = p.ep_lse[rc];
                rd_r[r] = p.ep_rowdot[rc];
                tg_r[r] = (int)p.ep_target[rc];
            }
            // (ONE uniform branch around the whole batch of loads: a select per element would put a branch between the
            // loads and serialise their round trips, see TileLoader::load)
            if (p.ep16) {
                const unsigned short* lgp = reinterpret_cast<const unsigned short*>(p.ep_logits);
                unsigned short raw[TW][16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const long ro = (long)(rv[r] ? row : 0) * p.ep_ld;
#pragma unroll
                    for (int j = 0; j < TW; ++j) {
                        const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                        raw[j][r] = lgp[ro + (col < p.N ? col : 0)];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int j = 0; j < TW; ++j) lg[j][r] = __builtin_bit_cast(float, (unsigned)raw[j][r] << 16);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const long ro = (long)(rv[r] ? row : 0) * p.ep_ld;
#pragma unroll
                    for (int j = 0; j < TW; ++j) {
                        const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                        lg[j][r] = p.ep_logits[ro + (col < p.N ? col : 0)];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = __expf(lg[j][r] * inv_tau - lse_r[r]);
                    const float onehot = (tg_r[r] == col) ? 1.f : 0.f;
                    v[r] = pr * (acc[0][i][j][r] - rd_r[r]) * inv_tau + ce_scale * (pr - onehot);
                }
                if (!p.c16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                        if (rv[r] && col < p.N) p.C[(long)row * p.ldc + col] = v[r];
                    }
                    continue;
                }
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    pk[i][j][rq][0] = pack_bf16x2(v[4 * rq], v[4 * rq + 1]);
                    pk[i][j][rq][1] = pack_bf16x2(v[4 * rq + 2], v[4 * rq + 3]);
                }
                // row-major image: lanes along the columns (2-byte writes, conflict-free)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const unsigned w = pk[i][j][r >> 2][(r & 3) >> 1];
                    stage[rl * RS + j * 32 + fr] = (unsigned short)((r & 1) ? (w >> 16) : (w & 0xffffu));
                }
            }
        }
        if (!p.c16) return;
        // (all of this wave's logits were read above: the in-place stores below cannot run ahead of a read)
        const int sub_m = m0 + wm * (BT / 2), sub_n = n0 + wn * (BT / 2);
        __syncthreads();
        flush_stage<RS>(stage, lane, p.C, p.ldc, sub_m, p.M, sub_n, p.N);
        if (p.ct) {
            __syncthreads();  // (uniform) the row-major image has been read
            // transposed image: stage[column][row], a lane's four consecutive rows of one column are one 8-byte write
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int j = 0; j < TW; ++j)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq)
                        *reinterpret_cast<u32x2*>(stage + (j * 32 + fr) * RS + i * 32 + 8 * rq + 4 * fk) = pk[i][j][rq];
            __syncthreads();
            flush_stage<RS>(stage, lane, p.ct, p.ldct, sub_n, p.N, sub_m, p.M);
        }
        return;
    }
    if (p.c16) {
        // bf16 C (cgn_gemm_desc.storage; host: no split-K, beta == 0). Its own loop nest: a c16 test inside the fp32
        // store loop below cost the fp32 products 13-25 % (the stores no longer issued back to back).
        if constexpr (BT == 128) {
            if (p.epilogue == CGN_EPI_LINEAR) {  // the vocabulary projection: staged through LDS like the fused epilogue
                constexpr int RS = 72;
                static_assert(4 * 64 * RS * 2 <= 4 * OPI * 4, "staging images must fit the operand buffers");
                unsigned short* stage = reinterpret_cast<unsigned short*>(lds) + wave * (64 * RS);
                __syncthreads();  // every wave is done with the operand images
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                    const int cc = col < p.N ? col : 0;
                    const float b = (p.bias0 ? p.bias0[cc] : 0.f) + (p.bias1 ? p.bias1[cc] : 0.f);
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                            const bf16x2 h = {(__bf16)(p.alpha * acc[0][i][j][r] + b), (__bf16)0.f};
                            stage[rl * RS + j * 32 + fr] = (unsigned short)(__builtin_bit_cast(unsigned, h) & 0xffffu);
                        }
                }
                __syncthreads();
                flush_stage<RS>(stage, lane, p.C, p.ldc, m0 + wm * (BT / 2), p.M, n0 + wn * (BT / 2), p.N);
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TW; ++i) {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                if (col >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    if (row >= p.M) continue;
                    const float v = apply_epilogue(p, row, col, acc[0][i][j][r]);
                    st_bf16(p.C, (long)row * p.ldc + col, v);
                    if (p.ct) st_bf16(p.ct, (long)col * p.ldct + row, v);
                }
            }
        }
        return;
    }
    if constexpr (LSE) {
        // CGN_EPI_LINEAR_STATS (host: unsplit, beta == 0): the wave's 64x64 sub-tile of s * C goes through LDS (free after
        // the K loop; rows 68 floats apart: a lane's 16-byte reads of its row are conflict-free), then lane l forms the
        // maximum and the exp-sum of row l over the chunk's 64 columns in registers - no cross-lane reduction.
        constexpr int RSF = 68;
        static_assert(4 * 64 * RSF * 4 <= 4 * OPI * 4, "staging images must fit the operand buffers");
        float* stage = lds + wave * (64 * RSF);
        const float sc = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        // No workgroup barrier: behind the K loop's last barrier the waves only read the operand images (fragments of a
        // tile past the end, which no MFMA consumes), and a wave's staging area is read by that wave alone.
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int col = n0 + wn * (BT / 2) + j * 32 + fr;
            const bool cok = col < p.N;
            const int cc = cok ? col : 0;
            const float b0 = p.bias0 ? p.bias0[cc] : 0.f, b1 = p.bias1 ? p.bias1[cc] : 0.f;
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const float v = (p.alpha * acc[0][i][j][r] + b0) + b1;  // (the linear nest's order: same C bits)
                    stage[rl * RSF + j * 32 + fr] = cok ? v * sc : -INFINITY;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes have landed (in-order LDS queue)
        {
            const float* rowp = stage + lane * RSF;
            f32x4 x[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) x[q] = *reinterpret_cast<const f32x4*>(rowp + q * 4);
            float mx = -INFINITY;
#pragma unroll
            for (int q = 0; q < 16; ++q) mx = fmaxf(fmaxf(fmaxf(x[q][0], x[q][1]), fmaxf(x[q][2], x[q][3])), mx);
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += __expf(x[q][e] - mx);
            const int row = m0 + wm * (BT / 2) + lane;
            const int chunk = (n0 + wn * (BT / 2)) / 64;
            if (row < p.M && chunk * 64 < p.N) {
                f32x2 st = {mx, sum};
                *reinterpret_cast<f32x2*>(p.stats + ((long)row * p.ldstats + chunk) * 2) = st;
            }
        }
        // (C itself leaves through the linear nest below)
    }
    // The common epilogues get LEAN loop nests of their own (one uniform branch picks the nest; per element: one bounds
    // test, one fma, one store off a per-column base pointer). The all-purpose apply_epilogue() inlined 64 times made
    // this part of the kernel ~60 KB of straight-line code: more than the instruction cache, run once per tile.
    const int row_w = m0 + wm * (BT / 2) + 4 * fk, col_w = n0 + wn * (BT / 2) + fr;
    if (p.splitk > 1) {  // K slab: the raw partial sums
        float* wsp = p.ws + (long)split * p.M * p.ldw;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int col = col_w + j * 32;
            if (col >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) wsp[(long)row * p.ldw + col] = acc[0][i][j][r];
                }
        }
        return;
    }
    if (p.epilogue == CGN_EPI_LINEAR && p.beta == 0.f) {  // C = alpha * acc + bias0 + bias1
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int col = col_w + j * 32;
            if (col >= p.N) continue;
            const float b0 = p.bias0 ? p.bias0[col] : 0.f, b1 = p.bias1 ? p.bias1[col] : 0.f;
            float* cp = p.C + col;
#pragma unroll
            for (int i = 0; i < TW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) cp[(long)row * p.ldc] = (p.alpha * acc[0][i][j][r] + b0) + b1;  // (apply_epilogue's order)
                }
        }
        return;
    }
    if (p.epilogue == CGN_EPI_LINEAR) {  // beta != 0: C = alpha * acc + bias0 + bias1 + beta * C (loads batched per slab)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int col = col_w + j * 32;
            if (col >= p.N) continue;
            const float b0 = p.bias0 ? p.bias0[col] : 0.f, b1 = p.bias1 ? p.bias1[col] : 0.f;
            float* cp = p.C + col;
#pragma unroll
            for (int i = 0; i < TW; ++i) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
                    old[r] = cp[(long)(row < p.M ? row : m0) * p.ldc];  // (clamped: m0 < M always)
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) cp[(long)row * p.ldc] = ((p.alpha * acc[0][i][j][r] + b0) + b1) + p.beta * old[r];
                }
            }
        }
        return;
    }
    // what is left (the softmax backward epilogue of the variants without the dedicated fused one: small shapes)
#pragma unroll
    for (int i = 0; i < TW; ++i) {
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int col = col_w + j * 32;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                p.C[(long)row * p.ldc + col] = apply_epilogue(p, row, col, acc[0][i][j][r]);
            }
        }
    }
}

// (X6, 128-tile: BK = 16, 72 KB of LDS: two blocks per CU, whose waves cover each other's staging work)
template <int BT, bool A_KC, bool B_KC, int VEC, bool EXP, int S, int BK = (BT == 128 ? 16 : 32), bool SMB = false,
          bool BF = false, int FK = 0, bool A16 = false, bool X6 = false, bool LSE = false>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmParams p) {
    // Work item = (K split, tile). Default (order 2): the XCD remap runs over the tile list, the shorter tile axis
    // fastest (gemm_tile: neighbouring tiles share a panel of the operand that spans the long axis). Orders 0/1 remap over the whole (split, tile) list so that an XCD
    // owns whole K ranges of a split-K product: that cuts the memory-side fetch of the three tall-K products (PMC:
    // their 107 MB logits-sized operand is fetched ~4x, profiles/r01_pmc_traffic.txt) but measured no faster to 15 %
    // slower (tools/gemm_bench.py, GEMM_ORDER=0,1,2): these kernels are not fetch-bound, the Infinity Cache absorbs
    // the re-reads. Kept as an experiment knob (cgn_debug_set_stages bits 4..7).
    const int ntiles = p.tiles_m * p.tiles_n;
    int split, t;
    if (p.order == 2) {
        split = blockIdx.y;
        t = xcd_remap(blockIdx.x, ntiles);
    } else {
        const int w = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, ntiles * p.splitk);
        split = w / ntiles;
        t = w - split * ntiles;
    }
    gemm_tile<BT, A_KC, B_KC, VEC, EXP, S, BK, SMB, BF, FK, A16, X6, LSE>(p, split, t);
}

// Grouped launch: up to GROUP_MAX independent products (same kernel variant) in ONE launch; block b works on the
// product whose block range contains b. Members may be split along K (slabs + one grouped reduce launch): the
// text-CNN's products have ~100 tiles of 30-50 K steps in total, a latency chain per CU without the split. For chains of small same-shaped products (the six char-TDNN widths, the
// text-CNN's filter widths) the launch boundaries and the drain/fill between them cost more than the arithmetic.
constexpr int GROUP_MAX = 8;
struct GemmGroup {
    GemmParams p[GROUP_MAX];
    int tile_end[GROUP_MAX];  // running sum of block counts (tiles x K splits)
    int n;
};

template <int BT, bool A_KC, bool B_KC, int VEC, int S, int BK = (BT == 128 ? 16 : 32), bool BF = false, int FK = 0,
          bool X6 = false>
__global__ __launch_bounds__(NT, 2) void gemm_group_kernel(const GemmGroup g) {
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.tile_end[j]) ++j;  // block-uniform
    const int local = blockIdx.x - (j ? g.tile_end[j - 1] : 0);  // (split, tile) of member j, tile fastest
    const int ntiles = g.p[j].tiles_m * g.p[j].tiles_n;
    const int split = local / ntiles;
    gemm_tile<BT, A_KC, B_KC, VEC, false, S, BK, false, BF, FK, false, X6>(g.p[j], split, local - split * ntiles);
}

// Sums the split-K slabs in fixed order z = 0..splitk-1 (bitwise reproducible). Slab rows are ldw = ceil4(N) floats
// apart, so every thread reads 16-byte quads whatever N is (N = 825 made the scalar form the common one: 13 us against
// 6 us per launch). The slab loads of 4 splits are issued together before they are added: with one load per trip the
// kernel was a chain of L2 round trips (12.7 us for 16 MB of slabs).
__device__ __forceinline__ void splitk_reduce_body(const GemmParams& p) {
    const long total = (long)p.M * p.ldw;
    const long nvec = total / 4;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long idx = v * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < p.splitk; z0 += 4) {
            f32x4 part[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = min(z0 + u, p.splitk - 1);
                part[u] = *reinterpret_cast<const f32x4*>(p.ws + (long)z * total + idx);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (z0 + u < p.splitk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += part[u][j];
                }
            }
        }
        const int row = (int)(idx / p.ldw), col = (int)(idx % p.ldw);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j < p.N) p.C[(long)row * p.ldc + col + j] = apply_epilogue(p, row, col + j, s[j]);
    }
}

static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) { splitk_reduce_body(p); }

// one launch for all split members of a group: blockIdx.y = member
static __global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const GemmGroup g) {
    if (g.p[blockIdx.y].splitk > 1) splitk_reduce_body(g.p[blockIdx.y]);
}

// BK per (tile, operand type): fp32 16 / 32 (measured, see above); bf16 operands take half the LDS per k and their MFMA
// covers 16 k, so the K step is doubled (fewer barriers per MFMA).
template <int BT, bool BF>
constexpr int bk_of() {
    return BF ? (BT == 128 ? 32 : 64) : (BT == 128 ? 16 : 32);
}

template <int BT, int VEC, bool EXP, bool BF, int FK>
void launch_vk(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    constexpr int S = 2;  // register stages (1, 3, 4 were swept in round 1: 2 is best or equal on every step shape)
    constexpr int BK = bk_of<BT, BF>();
    if constexpr (EXP) {  // the exp transform needs a k-contiguous A (checked by the caller)
        if (b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, true, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<BT, true, false, VEC, true, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
    } else {
        if constexpr (BT == 128 && VEC == 3) {
            if (a_kc && b_kc && p.epilogue == CGN_EPI_SOFTMAX_BWD && p.splitk == 1) {
                hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, false, S, BK, true, BF, FK>), grid, dim3(NT), 0, s, p);
                return;
            }
        }
        if (a_kc && b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, false, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
        else if (a_kc && !b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, false, VEC, false, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
        else if (!a_kc && b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, false, true, VEC, false, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<BT, false, false, VEC, false, S, BK, false, BF, FK>), grid, dim3(NT), 0, s, p);
    }
}

template <int BT, int VEC, bool EXP, bool BF>
void launch_v(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    if constexpr (VEC == 3) {
        if (p.fastk == 1) {
            launch_vk<BT, VEC, EXP, BF, 1>(p, a_kc, b_kc, grid, s);
            return;
        }
        if constexpr (!EXP) {
            if (p.fastk == 2) {  // (prepare_gemm: neither operand k-contiguous, B's K axis two-level)
                constexpr int BK = bk_of<BT, BF>();
                hipLaunchKernelGGL((gemm_kernel<BT, false, false, VEC, false, 2, BK, false, BF, 2>), grid, dim3(NT), 0, s, p);
                return;
            }
        }
    }
    launch_vk<BT, VEC, EXP, BF, 0>(p, a_kc, b_kc, grid, s);
}

template <int BT, bool EXP, bool BF>
void launch_e(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    // each operand is read with 16-byte loads along its contiguous axis when its addressing allows, scalar otherwise
    switch ((p.a_vec ? 1 : 0) | (p.b_vec ? 2 : 0)) {
        case 0: launch_v<BT, 0, EXP, BF>(p, a_kc, b_kc, grid, s); break;
        case 1: launch_v<BT, 1, EXP, BF>(p, a_kc, b_kc, grid, s); break;
        case 2: launch_v<BT, 2, EXP, BF>(p, a_kc, b_kc, grid, s); break;
        default: launch_v<BT, 3, EXP, BF>(p, a_kc, b_kc, grid, s); break;
    }
}

// One product: bt = 64 | 128.
template <bool BF>
void launch_gemm(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    const bool ex = p.a_transform == CGN_ATRANS_EXP;
    if (bt == 128) {
        if (ex) launch_e<128, true, BF>(p, a_kc, b_kc, grid, s);
        else launch_e<128, false, BF>(p, a_kc, b_kc, grid, s);
    } else {
        if (ex) launch_e<64, true, BF>(p, a_kc, b_kc, grid, s);
        else launch_e<64, false, BF>(p, a_kc, b_kc, grid, s);
    }
}

// bf16-stored A (k-contiguous), bf16 MFMA operands, 16-byte loads on both operands (checked by prepare_gemm)
template <int BT, bool EXP, int FK>
void launch_a16_k(const GemmParams& p, bool b_kc, dim3 grid, hipStream_t s) {
    constexpr int BK = bk_of<BT, true>();
    if (b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, true, true, 3, EXP, 2, BK, false, true, FK, true>), grid, dim3(NT), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_kernel<BT, true, false, 3, EXP, 2, BK, false, true, FK, true>), grid, dim3(NT), 0, s, p);
}

template <int BT>
void launch_a16_t(const GemmParams& p, bool b_kc, dim3 grid, hipStream_t s) {
    const bool ex = p.a_transform == CGN_ATRANS_EXP;
    if (p.fastk == 1) {
        if (ex) launch_a16_k<BT, true, 1>(p, b_kc, grid, s);
        else launch_a16_k<BT, false, 1>(p, b_kc, grid, s);
    } else {
        if (ex) launch_a16_k<BT, true, 0>(p, b_kc, grid, s);
        else launch_a16_k<BT, false, 0>(p, b_kc, grid, s);
    }
}

// X6 variants (gemm_x6.hip): fp32 operands with 16-byte loads and fast K addressing only; false = no such variant (the
// caller runs the fp32-MFMA kernel instead)
template <int BT>
bool launch_x6_t(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    constexpr int BK = BT == 128 ? 16 : 32;
    if (!(p.a_vec && p.b_vec) || p.fastk == 0 || p.a16 || p.c16 || p.ep16) return false;
    const bool ex = p.a_transform == CGN_ATRANS_EXP;
    if (p.fastk == 2) {
        if (ex) return false;
        hipLaunchKernelGGL((gemm_kernel<BT, false, false, 3, false, 2, BK, false, true, 2, false, true>), grid, dim3(NT), 0, s, p);
        return true;
    }
    if (ex) {
        if (!a_kc || b_kc) return false;  // (the soft embedding: exp(logits) . E)
        hipLaunchKernelGGL((gemm_kernel<BT, true, false, 3, true, 2, BK, false, true, 1, false, true>), grid, dim3(NT), 0, s, p);
        return true;
    }
    if constexpr (BT == 128) {
        if (a_kc && b_kc && p.epilogue == CGN_EPI_SOFTMAX_BWD && p.splitk == 1) {
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, 3, false, 2, BK, true, true, 1, false, true>), grid, dim3(NT), 0, s, p);
            return true;
        }
        if (p.stats) {  // CGN_EPI_LINEAR_STATS (run_gemm has checked: both operands k-contiguous, unsplit)
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, 3, false, 2, BK, false, true, 1, false, true, true>), grid, dim3(NT), 0,
                               s, p);
            return true;
        }
    }
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, true, true, 3, false, 2, BK, false, true, 1, false, true>), grid, dim3(NT), 0, s, p);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, true, false, 3, false, 2, BK, false, true, 1, false, true>), grid, dim3(NT), 0, s, p);
    else if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, false, false, 3, false, 2, BK, false, true, 1, false, true>), grid, dim3(NT), 0, s, p);
    else
        return false;
    return true;
}

// X6 grouped launch (128-tile; gemm_x6.hip): sibling products that each leave most of the chip idle - the encoder's two
// W_ih products, the per-direction / per-layer dW products - as ONE launch. fk: the members' common fast-K mode.
inline bool launch_group_x6_t(const GemmGroup& g, bool a_kc, bool b_kc, int fk, unsigned blocks, hipStream_t s) {
    if (fk == 2) {
        if (a_kc || b_kc) return false;
        hipLaunchKernelGGL((gemm_group_kernel<128, false, false, 3, 2, 16, true, 2, true>), dim3(blocks), dim3(NT), 0, s, g);
        return true;
    }
    if (fk != 1) return false;
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<128, true, true, 3, 2, 16, true, 1, true>), dim3(blocks), dim3(NT), 0, s, g);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<128, true, false, 3, 2, 16, true, 1, true>), dim3(blocks), dim3(NT), 0, s, g);
    else if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<128, false, false, 3, 2, 16, true, 1, true>), dim3(blocks), dim3(NT), 0, s, g);
    else
        return false;
    return true;
}

template <int VEC, bool BF, int FK>
void launch_group_vk(const GemmGroup& g, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    constexpr int BK = bk_of<64, BF>();
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, true, true, VEC, 2, BK, BF, FK>), dim3(blocks), dim3(NT), 0, s, g);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, true, false, VEC, 2, BK, BF, FK>), dim3(blocks), dim3(NT), 0, s, g);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, false, true, VEC, 2, BK, BF, FK>), dim3(blocks), dim3(NT), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_group_kernel<64, false, false, VEC, 2, BK, BF, FK>), dim3(blocks), dim3(NT), 0, s, g);
}

template <int VEC, bool BF>
void launch_group_v(const GemmGroup& g, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    if constexpr (VEC == 3) {
        bool fk = true;  // fast K addressing only if every member qualifies (one kernel variant per group)
        for (int i = 0; i < g.n; ++i) fk = fk && g.p[i].fastk == 1;
        if (fk) {
            launch_group_vk<VEC, BF, 1>(g, a_kc, b_kc, blocks, s);
            return;
        }
    }
    launch_group_vk<VEC, BF, 0>(g, a_kc, b_kc, blocks, s);
}

template <bool BF>
void launch_group(const GemmGroup& g, int vec, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    switch (vec) {
        case 0: launch_group_v<0, BF>(g, a_kc, b_kc, blocks, s); break;
        case 1: launch_group_v<1, BF>(g, a_kc, b_kc, blocks, s); break;
        case 2: launch_group_v<2, BF>(g, a_kc, b_kc, blocks, s); break;
        default: launch_group_v<3, BF>(g, a_kc, b_kc, blocks, s); break;
    }
}

}  // namespace cgn_gemm

// Defined in gemm.hip (fp32 operands) and gemm_bf16.hip (bf16 operands): one translation unit per operand type keeps the
// two sets of kernel variants compiling in parallel.
namespace cgn_gemm {
void launch_gemm_f32ops(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);
void launch_gemm_bf16ops(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);
void launch_gemm_a16(const GemmParams& p, int bt, bool b_kc, dim3 grid, hipStream_t s);
bool launch_gemm_x6(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);
bool launch_group_x6(const GemmGroup& g, bool a_kc, bool b_kc, int fk, unsigned blocks, hipStream_t s);
void launch_group_f32ops(const GemmGroup& g, int vec, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s);
void launch_group_bf16ops(const GemmGroup& g, int vec, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s);
}  // namespace cgn_gemm
