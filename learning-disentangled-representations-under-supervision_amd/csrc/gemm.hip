// fp32 MFMA GEMM for gfx950: C = epilogue(alpha * A·B + beta*C + bias), exact-fp32 v_mfma_f32_32x32x2_f32.
//
// Design (MI355X_MICROARCH: the f32 MFMA runs at the f32 vector rate, 64 cycles per 32x32x2 per SIMD, so the loop
// is MFMA-paced and LDS/global staging only has to stay out of the way):
//   * square block tile BT x BT (128 or 64), BK = 32, 256 threads = 4 waves in a 2x2 grid, each wave owns
//     (BT/2)x(BT/2) outputs as (BT/64)^2 accumulators of 32x32;
//   * operands are staged global -> registers -> LDS in a k-major [BK][BT+1] image: the MFMA fragment read
//     (lane l: row l&31 of k-slice l>>5) is then 32 consecutive floats per half-wave = bank-conflict free;
//   * the next K tile's global loads are issued before the current tile's MFMAs (register prefetch);
//   * two-level addressing on every operand axis (see cgn_hip.h) covers transposed operands, overlapping conv
//     windows and [B,T+1,H] history buffers without copies;
//   * tall-K / few-tile problems are split along K into fp32 slabs, summed in fixed order by a second kernel
//     (bitwise reproducible, no float atomics);
//   * blockIdx -> tile mapping is XCD-aware: blocks that share an XCD (blockIdx % 8) walk neighbouring tiles.
#include "common.h"

namespace {

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

Axis make_axis(const cgn_axis& c) {
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
    Axis am, ak, bk, bn;
    int M, N, K;
    int k_per_split, splitk;
    int a_vec, b_vec;
    int a_transform, epilogue;
    int tiles_m, tiles_n;
    int order;  // 0: remap over (split, tile), shorter tile axis fastest; 1: same remap, m fastest; 2: remap over tiles only
    float alpha, beta, a_scale, ep_inv_tau, ep_ce_scale;
};

constexpr int g_unroll_mult = 2;
constexpr int KROUND = 32;  // split-K ranges are multiples of every BK in use
constexpr int NT = 256;

__device__ __forceinline__ float apply_epilogue(const GemmParams& p, int row, int col, float acc) {
    if (p.epilogue == CGN_EPI_SOFTMAX_BWD) {
        const float inv_tau = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        const float ce_scale = p.dyn ? p.dyn[1] : p.ep_ce_scale;
        const float pr = __expf(p.ep_logits[(long)row * p.ep_ld + col] * inv_tau - p.ep_lse[row]);
        const float onehot = (p.ep_target[row] == (int64_t)col) ? 1.f : 0.f;
        return pr * (acc - p.ep_rowdot[row]) * inv_tau + ce_scale * (pr - onehot);
    }
    float v = p.alpha * acc;
    if (p.bias0) v += p.bias0[col];
    if (p.bias1) v += p.bias1[col];
    if (p.beta != 0.f) v += p.beta * p.C[(long)row * p.ldc + col];
    return v;
}

// Operand tile loader. The tile is ROWS (m or n index) x BK (k index); its LDS image is row-major [ROWS][LDK] with
// LDK = BK + 4 floats (144-byte rows): the MFMA loop reads a lane's 4 consecutive k of one row with ONE ds_read_b128
// (rows 144 B apart put 16 consecutive rows on 16 different 16-byte slots of the 256-byte bank row: conflict-free),
// and every staging write is a ds_write_b128 of 4 consecutive k of one row.
//   KC = true : memory is contiguous along k (row-major [rows][k]): thread = (row, k-quad), one float4 load per chunk
//   KC = false: memory is contiguous along the row index ([k][rows]): thread = one row, lanes run along rows, scalar
//               loads (a wave covers 64 consecutive rows = 256 contiguous bytes per k), 4 k per chunk
template <int ROWS, bool KC, int BK, int S, bool VEC, bool EXP>
struct TileLoader {
    static constexpr int LDK = BK + 4;
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
    static_assert(KC || !VEC || (KP == 2 && KP == PER), "row-vector loader geometry");

    f32x4 regs[S][PER];  // S register stages: S tiles of global loads in flight per thread
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
    __device__ __forceinline__ void load(int st, const float* __restrict__ base, const Axis& ak, int k0, int kbeg,
                                         int kend) {
        const int tid = threadIdx.x;
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

    // Mask/transform the staged registers and write them into the LDS image dst[ROWS][LDK].
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
                    if constexpr (EXP) x = __expf(x * scale - rparam[i]);
                    v[j] = (rok[i] && k + j < kend) ? x : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + (r + i * RS) * LDK + q * 4) = v;
            }
        } else if constexpr (VEC) {
            const int row = 4 * (tid % RQ);
            const int kl = (tid / RQ) * KP;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 v;
                v[0] = (rok4[i] && k0 + kl < kend) ? regs[st][0][i] : 0.f;
                v[1] = (rok4[i] && k0 + kl + 1 < kend) ? regs[st][1][i] : 0.f;
                *reinterpret_cast<f32x2*>(dst + (row + i) * LDK + kl) = v;
            }
        } else {
            const int row = tid % ROWS;
            const int kl = (tid / ROWS) * (4 * PER);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v = regs[st][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (rok[0] && k0 + kl + 4 * i + j < kend) ? v[j] : 0.f;
                *reinterpret_cast<f32x4*>(dst + row * LDK + kl + 4 * i) = v;
            }
        }
    }
};

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
template <int BT, bool A_KC, bool B_KC, int VEC, bool EXP, int S, int BK, bool SMB = false>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int split, const int t) {
    constexpr int TW = BT / 64;  // 32x32 accumulators per wave along each dim
    constexpr int LDK = BK + 4;  // LDS row stride (floats) of both operand images
    // One LDS array (both operands, two buffers): keeps every carve offset explicit and 16-byte aligned.
    constexpr int TILE = BT * LDK;  // floats per operand image
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];

    int tile_m, tile_n;
    if (p.order == 0 && p.tiles_n < p.tiles_m) {
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

    TileLoader<BT, A_KC, BK, S, (VEC & 1) != 0, EXP> la;
    TileLoader<BT, B_KC, BK, S, (VEC & 2) != 0, false> lb;
    la.init(p.am, m0, p.M, p.a_row_param);
    lb.init(p.bn, n0, p.N, nullptr);

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
    //   tile j lives in register stage j % S from its load until it is written to LDS buffer j & 1
    //   step kt:  write tile kt+1 (loaded during step kt-S) into buf[(kt+1)&1]
    //             issue the global loads of tile kt+1+S into the stage just freed
    //             MFMAs on buf[kt&1]
    //             barrier
    // The loop is unrolled by U = lcm(S, 2) so that stage and buffer indices are compile-time constants.
    // Loads and LDS writes past the last tile are not skipped (that would need a branch, see TileLoader::load): they
    // re-read the block's first quads (L1/L2 hits) and write a buffer nobody reads again.
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            la.load(j, p.A, p.ak, kbeg + j * BK, kbeg, kend);
            lb.load(j, p.B, p.bk, kbeg + j * BK, kbeg, kend);
        }
        la.store(0, lds, kbeg, kend, a_scale);
        lb.store(0, lds + TILE, kbeg, kend, 0.f);
        la.load(0, p.A, p.ak, kbeg + S * BK, kbeg, kend);
        lb.load(0, p.B, p.bk, kbeg + S * BK, kbeg, kend);
    }
    __syncthreads();

    const int fr = lane & 31;  // fragment row/col
    const int fk = lane >> 5;  // k-slice within the 32x32x2 step
    // U steps per trip, a multiple of lcm(S, 2): hipcc's wait-count pass is exact inside the unrolled body but
    // waits for ALL outstanding loads at the first LDS store after the loop header, so make trips long.
    constexpr int U = g_unroll_mult * ((S % 2 == 0) ? S : 2 * S);
    for (int kt0 = 0; kt0 < nk; kt0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;  // block-uniform
            const float* As = lds + (u & 1) * 2 * TILE;
            const float* Bs = As + TILE;
            {
                const int st = (u + 1) % S;
                float* An = lds + ((u + 1) & 1) * 2 * TILE;
                const int k1 = kbeg + (kt + 1) * BK;
                la.store(st, An, k1, kend, a_scale);
                lb.store(st, An + TILE, k1, kend, 0.f);
                la.load(st, p.A, p.ak, k1 + S * BK, kbeg, kend);
                lb.load(st, p.B, p.bk, k1 + S * BK, kbeg, kend);
            }
            // K order inside a step: lane half fk of block j holds k = 8j + 4fk + i for the i-th MFMA (any permutation
            // of k is legal as long as A and B use the same one), so a lane's 4 values are ONE 16-byte LDS read per tile.
            const float* ap = As + (wm * (BT / 2) + fr) * LDK + 4 * fk;
            const float* bp = Bs + (wn * (BT / 2) + fr) * LDK + 4 * fk;
            f32x4 a4[2][TW], b4[2][TW];
#pragma unroll
            for (int i = 0; i < TW; ++i) a4[0][i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < TW; ++j) b4[0][j] = *reinterpret_cast<const f32x4*>(bp + j * 32 * LDK);
#pragma unroll
            for (int jb = 0; jb < BK / 8; ++jb) {
                const int cur = jb & 1;
                if (jb + 1 < BK / 8) {  // fragments of the next 8-k block are in flight under this block's MFMAs
#pragma unroll
                    for (int i = 0; i < TW; ++i)
                        a4[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * LDK + (jb + 1) * 8);
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        b4[cur ^ 1][j] = *reinterpret_cast<const f32x4*>(bp + j * 32 * LDK + (jb + 1) * 8);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int j = 0; j < TW; ++j)
                            acc[e % NC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur][i][e], b4[cur][j][e], acc[e % NC][i][j], 0,
                                                                             0, 0);
            }
            __syncthreads();
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
        const float inv_tau = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        const float ce_scale = p.dyn ? p.dyn[1] : p.ep_ce_scale;
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
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int col = n0 + wn * (BT / 2) + j * 32 + fr;
                    lg[j][r] = p.ep_logits[(long)rc * p.ep_ld + (col < p.N ? col : 0)];
                }
            }
#pragma unroll
            for (int j = 0; j < TW; ++j) {
                const int col = n0 + wn * (BT / 2) + j * 32 + fr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const float pr = __expf(lg[j][r] * inv_tau - lse_r[r]);
                    const float onehot = (tg_r[r] == col) ? 1.f : 0.f;
                    const float v = pr * (acc[0][i][j][r] - rd_r[r]) * inv_tau + ce_scale * (pr - onehot);
                    if (rv[r] && col < p.N) p.C[(long)row * p.ldc + col] = v;
                }
            }
        }
        return;
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
                if (p.splitk > 1) {
                    p.ws[((long)split * p.M + row) * p.N + col] = acc[0][i][j][r];
                } else {
                    p.C[(long)row * p.ldc + col] = apply_epilogue(p, row, col, acc[0][i][j][r]);
                }
            }
        }
    }
}

template <int BT, bool A_KC, bool B_KC, int VEC, bool EXP, int S, int BK = (BT == 128 ? 16 : 32), bool SMB = false>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const GemmParams p) {
    // Work item = (K split, tile). Default (order 2): the XCD remap runs over the tile list, m fastest (neighbouring
    // tiles share the B, usually the weight, panel). Orders 0/1 remap over the whole (split, tile) list so that an XCD
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
    gemm_tile<BT, A_KC, B_KC, VEC, EXP, S, BK, SMB>(p, split, t);
}

// Grouped launch: up to GROUP_MAX independent products (same kernel variant, no split-K) in ONE launch; block b works
// on the product whose tile range contains b. For chains of small same-shaped products (the six char-TDNN widths, the
// text-CNN's filter widths) the launch boundaries and the drain/fill between them cost more than the arithmetic.
constexpr int GROUP_MAX = 8;
struct GemmGroup {
    GemmParams p[GROUP_MAX];
    int tile_end[GROUP_MAX];  // running sum of tile counts
    int n;
};

template <int BT, bool A_KC, bool B_KC, int VEC, int S, int BK = (BT == 128 ? 16 : 32)>
__global__ __launch_bounds__(NT, 2) void gemm_group_kernel(const GemmGroup g) {
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.tile_end[j]) ++j;  // block-uniform
    const int t = blockIdx.x - (j ? g.tile_end[j - 1] : 0);
    gemm_tile<BT, A_KC, B_KC, VEC, false, S, BK>(g.p[j], 0, t);
}

// Sums the split-K slabs in fixed order z = 0..splitk-1 (bitwise reproducible). VEC: four consecutive columns per
// thread (16-byte slab loads, N % 4 == 0). The slab loads of 4 splits are issued together before they are added: with
// one load per trip the kernel was a chain of L2 round trips (12.7 us for 16 MB of slabs).
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    constexpr int W = VEC ? 4 : 1;
    const long total = (long)p.M * p.N;
    const long nvec = total / W;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long idx = v * W;
        float s[W];
#pragma unroll
        for (int j = 0; j < W; ++j) s[j] = 0.f;
        for (int z0 = 0; z0 < p.splitk; z0 += 4) {
            float part[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = min(z0 + u, p.splitk - 1);
                const float* src = p.ws + (long)z * total + idx;
                if constexpr (VEC) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int j = 0; j < 4; ++j) part[u][j] = t[j];
                } else {
                    part[u][0] = *src;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (z0 + u < p.splitk) {
#pragma unroll
                    for (int j = 0; j < W; ++j) s[j] += part[u][j];
                }
            }
        }
        const int row = (int)(idx / p.N), col = (int)(idx % p.N);
#pragma unroll
        for (int j = 0; j < W; ++j) p.C[(long)row * p.ldc + col + j] = apply_epilogue(p, row, col + j, s[j]);
    }
}

// 16-byte loads along `contig` (extent n): aligned strides, and a quad that straddles the end of the axis must stay
// inside memory the caller owns: n % 4 == 0, or plain rows whose padded stride covers the rounded-up extent.
bool axis_vec_ok(const cgn_axis& contig, const cgn_axis& other, int n) {
    if (contig.stride != 1) return false;
    if (n % 4 != 0 && !(contig.rpg == 0 && other.rpg == 0 && other.stride >= (n + 3) / 4 * 4)) return false;
    if (contig.rpg != 0 && (contig.rpg % 4 != 0 || contig.gs % 4 != 0)) return false;
    if (other.stride % 4 != 0) return false;
    if (other.rpg != 0 && other.gs % 4 != 0) return false;
    return true;
}

int g_stages = 0;  // 0 = default; tools/gemm_bench.py overrides it through cgn_debug_set_stages
int g_order = 2;   // tile-order experiment knob (cgn_debug_set_stages bits 4..7); 2 = default

template <int BT, int VEC, bool EXP>
void launch_v(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    constexpr int S = 2;  // register stages (1, 3, 4 were swept in round 1: 2 is best or equal on every step shape)
    if constexpr (EXP) {  // the exp transform needs a k-contiguous A (checked by the caller)
        if (b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, true, S>), grid, dim3(NT), 0, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<BT, true, false, VEC, true, S>), grid, dim3(NT), 0, s, p);
    } else {
        if constexpr (BT == 128 && VEC == 3) {
            if (a_kc && b_kc && p.epilogue == CGN_EPI_SOFTMAX_BWD && p.splitk == 1) {
                hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, false, S, 16, true>), grid, dim3(NT), 0, s, p);
                return;
            }
        }
        if (a_kc && b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, true, VEC, false, S>), grid, dim3(NT), 0, s, p);
        else if (a_kc && !b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, true, false, VEC, false, S>), grid, dim3(NT), 0, s, p);
        else if (!a_kc && b_kc)
            hipLaunchKernelGGL((gemm_kernel<BT, false, true, VEC, false, S>), grid, dim3(NT), 0, s, p);
        else
            hipLaunchKernelGGL((gemm_kernel<BT, false, false, VEC, false, S>), grid, dim3(NT), 0, s, p);
    }
}

template <int BT, bool EXP>
void launch_e(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    // each operand is read with 16-byte loads along its contiguous axis when its addressing allows, scalar otherwise
    switch ((p.a_vec ? 1 : 0) | (p.b_vec ? 2 : 0)) {
        case 0: launch_v<BT, 0, EXP>(p, a_kc, b_kc, grid, s); break;
        case 1: launch_v<BT, 1, EXP>(p, a_kc, b_kc, grid, s); break;
        case 2: launch_v<BT, 2, EXP>(p, a_kc, b_kc, grid, s); break;
        default: launch_v<BT, 3, EXP>(p, a_kc, b_kc, grid, s); break;
    }
}

template <int BT>
void launch(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    if (p.a_transform == CGN_ATRANS_EXP)
        launch_e<BT, true>(p, a_kc, b_kc, grid, s);
    else
        launch_e<BT, false>(p, a_kc, b_kc, grid, s);
}

// Tile / split-K heuristic: MI355X has 256 CUs and this kernel fits ~3-4 blocks per CU; aim for >= ~512 blocks.
int plan_gemm(const cgn_gemm_desc* d, int* bt_out, int* splitk_out, int* kps_out) {
    // Measured on MI355X (tools/gemm_bench.py): the kernel needs >= ~4 blocks per CU to keep the MFMA pipes fed, the
    // 128-tile is the better choice once there are enough of them, and tall-K products want K split across blocks.
    const long t128 = (long)cgn_cdiv(d->M, 128) * cgn_cdiv(d->N, 128);
    const long t64 = (long)cgn_cdiv(d->M, 64) * cgn_cdiv(d->N, 64);
    int bt = d->force_tile;
    int splitk = d->force_splitk;
    if (bt == 0) {
        if (t128 >= 256 || (d->K >= 4096 && t128 >= 24))
            bt = 128;
        else
            bt = 64;
    }
    const long tiles = bt == 128 ? t128 : t64;
    if (splitk <= 0 && bt == 128 && d->K >= 4096) {  // tall-K only: a split of the K = 1344 dW_fc product (sk 2) won 11 %
        // alone but lost it again inside the step (123 MB of slab traffic beside the LSTM backward on the main stream)
        // 128-tile: a CU works its blocks off at a constant rate whether 1 or 2 are resident, so the time is
        // ceil(blocks / 256 CUs) block-times of K/splitk steps each. Pick the split count that minimises that product
        // (plus ~2 K steps of fill/drain per block and the slab traffic of each extra split). Measured on the three
        // logits-sized tall-K products of the CGN step (tools/gemm_bench.py sweeps): 44 tiles: sk 11 -> 335 us, 12 ->
        // 406, 16 -> 341, 23 -> 325; 33 tiles: sk 15 -> 250, 16 -> 290; 628 tiles, K = 1344: sk 1 -> 394, 2 -> 352.
        const int max_sk = (int)(d->K / 512 > 64 ? 64 : d->K / 512);
        long best_cost = -1;
        int best = 1;
        for (int sk = 1; sk <= (max_sk < 1 ? 1 : max_sk); ++sk) {
            const long kps = (long)cgn_cdiv(cgn_cdiv(d->K, sk), KROUND) * KROUND;
            const long per_cu = (tiles * sk + 255) / 256;
            const long cost = per_cu * (kps + 64) + (sk > 1 ? 8L * sk : 0);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = sk;
            }
        }
        splitk = best;
    }
    if (splitk <= 0) {
        // Measured (tools/gemm_bench.py sweeps): with >= 2 blocks per CU a split only adds slab traffic; below that,
        // splitting K until ~1024 blocks exist wins as long as every split keeps a few K steps. Products with a
        // handful of tiles (e2mu/e2logvar, d[z,c], the text-CNN windows) are pure latency chains of K/BK dependent
        // steps on one CU each: split them hard, whatever their FLOP count.
        const bool few = tiles < 64;
        const int min_k = few ? 128 : (bt == 128 ? 512 : 256);
        long want = tiles >= 512 ? 1 : (1024 + tiles - 1) / tiles;
        if (few && want > 16) want = 16;
        const long max_by_k = d->K / min_k;
        if (want > max_by_k) want = max_by_k;
        // a split costs a slab round trip plus a reduce launch (~5 us): skip it for small many-tile products
        if (!few && (double)d->M * d->N * d->K < 1.25e8) want = 1;
        if (want < 1) want = 1;
        splitk = (int)want;
    }
    if (splitk > 1) {
        const long need = (long)splitk * d->M * d->N * (long)sizeof(float);
        if (!d->workspace || d->workspace_bytes < need) {
            if (d->force_splitk > 0) return CGN_ERR_WORKSPACE;
            const long fit = d->workspace ? d->workspace_bytes / ((long)d->M * d->N * (long)sizeof(float)) : 0;
            splitk = fit >= 2 ? (int)fit : 1;
        }
    }
    int kps = cgn_cdiv(cgn_cdiv(d->K, splitk), KROUND) * KROUND;
    if (kps < KROUND) kps = KROUND;
    splitk = d->K > 0 ? cgn_cdiv(d->K, kps) : 1;
    *bt_out = bt; *splitk_out = splitk; *kps_out = kps;
    return CGN_OK;
}

}  // namespace

extern "C" int cgn_abi_version(void) { return 2; }

extern "C" void cgn_debug_set_stages(int v) {
    g_stages = v & 15;  // kept for ABI stability: the stage count is fixed at 2 since round 1
    g_order = (v >> 4) & 15;
    if (v == 0) g_order = 2;
}

extern "C" int cgn_gemm_plan(const cgn_gemm_desc* d, int32_t* tile, int32_t* splitk, int32_t* a_kcontig,
                             int32_t* b_kcontig) {
    CGN_CHECK_ARG(d && tile && splitk && a_kcontig && b_kcontig);
    CGN_CHECK_ARG(d->force_tile == 0 || d->force_tile == 64 || d->force_tile == 128);
    int bt, sk, kps;
    const int rc = plan_gemm(d, &bt, &sk, &kps);
    if (rc) return rc;
    *tile = bt; *splitk = sk;
    *a_kcontig = (d->a_k.stride == 1 || d->a_m.stride != 1) ? 1 : 0;
    *b_kcontig = (d->b_k.stride == 1 || d->b_n.stride != 1) ? 1 : 0;
    return CGN_OK;
}

// Validates a descriptor and fills the kernel parameters (everything except the tile / split-K plan).
static int prepare_gemm(const cgn_gemm_desc* d, GemmParams& p, bool& a_kc, bool& b_kc) {
    CGN_CHECK_ARG(d != nullptr);
    CGN_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0);
    if (d->M == 0 || d->N == 0) return CGN_OK;
    CGN_CHECK_ARG(d->A && d->B && d->C);
    CGN_CHECK_ARG(d->ldc >= d->N);
    // index ranges of the magic-number division in axis_off()
    CGN_CHECK_ARG(d->M < (1 << 30) && d->N < (1 << 30) && d->K < (1 << 30));
    CGN_CHECK_ARG(d->a_m.rpg >= 0 && d->a_k.rpg >= 0 && d->b_k.rpg >= 0 && d->b_n.rpg >= 0);
    CGN_CHECK_ARG(d->a_m.rpg < (1 << 16) && d->a_k.rpg < (1 << 16) && d->b_k.rpg < (1 << 16) && d->b_n.rpg < (1 << 16));
    CGN_CHECK_ARG((d->a_m.rpg == 0 || d->M < (1 << 24)) && (d->b_n.rpg == 0 || d->N < (1 << 24)) &&
                  ((d->a_k.rpg == 0 && d->b_k.rpg == 0) || d->K < (1 << 24)));
    CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE || (d->a_transform == CGN_ATRANS_EXP && d->a_row_param));
    CGN_CHECK_ARG(d->epilogue == CGN_EPI_LINEAR || d->epilogue == CGN_EPI_SOFTMAX_BWD);
    if (d->epilogue == CGN_EPI_SOFTMAX_BWD)
        CGN_CHECK_ARG(d->ep_logits && d->ep_lse && d->ep_rowdot && d->ep_target && d->ep_ld >= d->N);
    CGN_CHECK_ARG(d->force_tile == 0 || d->force_tile == 64 || d->force_tile == 128);
    // an operand is "k-contiguous" when its k stride is 1; otherwise lanes run along its row index
    a_kc = d->a_k.stride == 1 || d->a_m.stride != 1;
    b_kc = d->b_k.stride == 1 || d->b_n.stride != 1;
    CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE || a_kc);

    p.A = d->A; p.B = d->B; p.C = d->C;
    p.bias0 = d->bias0; p.bias1 = d->bias1; p.a_row_param = d->a_row_param;
    p.ep_logits = d->ep_logits; p.ep_lse = d->ep_lse; p.ep_rowdot = d->ep_rowdot; p.ep_target = d->ep_target;
    p.ldc = d->ldc; p.ep_ld = d->ep_ld;
    p.am = make_axis(d->a_m); p.ak = make_axis(d->a_k); p.bk = make_axis(d->b_k); p.bn = make_axis(d->b_n);
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.a_transform = d->a_transform; p.epilogue = d->epilogue;
    p.alpha = d->alpha; p.beta = d->beta; p.a_scale = d->a_scale;
    p.ep_inv_tau = d->ep_inv_tau; p.ep_ce_scale = d->ep_ce_scale;
    // 16-byte loads run along the operand's contiguous axis: k for a k-contiguous operand, the row index otherwise
    p.a_vec = ((uintptr_t)d->A % 16 == 0) &&
              (a_kc ? axis_vec_ok(d->a_k, d->a_m, d->K) : axis_vec_ok(d->a_m, d->a_k, d->M));
    p.b_vec = ((uintptr_t)d->B % 16 == 0) &&
              (b_kc ? axis_vec_ok(d->b_k, d->b_n, d->K) : axis_vec_ok(d->b_n, d->b_k, d->N));
    p.ws = d->workspace;
    p.dyn = d->dyn_scalars;
    p.order = g_order;
    return CGN_OK;
}

extern "C" int cgn_gemm_f32(void* stream, const cgn_gemm_desc* d) {
    GemmParams p;
    bool a_kc = false, b_kc = false;
    {
        const int rc = prepare_gemm(d, p, a_kc, b_kc);
        if (rc) return rc;
    }
    if (d->M == 0 || d->N == 0) return CGN_OK;
    int bt, splitk, kps;
    {
        const int rc = plan_gemm(d, &bt, &splitk, &kps);
        if (rc) return rc;
    }
    p.tiles_m = cgn_cdiv(d->M, bt);
    p.tiles_n = cgn_cdiv(d->N, bt);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    p.k_per_split = kps;
    p.splitk = splitk;

    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)splitk);
    if (bt == 128)
        launch<128>(p, a_kc, b_kc, grid, s);
    else
        launch<64>(p, a_kc, b_kc, grid, s);
    CGN_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total = (long)d->M * d->N;
        const bool vec = d->N % 4 == 0;  // slabs are dense [M, N] at a 256-byte aligned base: rows of 4k columns align
        int blocks = cgn_cdiv(total / (vec ? 4 : 1), 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        if (vec)
            hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, p);
        CGN_LAUNCH_CHECK();
    }
    return CGN_OK;
}

template <int VEC>
static void launch_group(const GemmGroup& g, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, true, true, VEC, 2>), dim3(blocks), dim3(NT), 0, s, g);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, true, false, VEC, 2>), dim3(blocks), dim3(NT), 0, s, g);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_group_kernel<64, false, true, VEC, 2>), dim3(blocks), dim3(NT), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_group_kernel<64, false, false, VEC, 2>), dim3(blocks), dim3(NT), 0, s, g);
}

extern "C" int cgn_gemm_group_f32(void* stream, const cgn_gemm_desc* descs, int32_t n) {
    CGN_CHECK_ARG(descs && n >= 1 && n <= GROUP_MAX);
    GemmGroup g;
    g.n = 0;
    bool a_kc0 = false, b_kc0 = false, va = true, vb = true;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        const cgn_gemm_desc* d = descs + i;
        GemmParams p;
        bool a_kc = false, b_kc = false;
        const int rc = prepare_gemm(d, p, a_kc, b_kc);
        if (rc) return rc;
        CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE && d->epilogue == CGN_EPI_LINEAR);
        if (d->M == 0 || d->N == 0) continue;
        // One kernel variant per group: the orientation flags of the first member are used for all. The scalar
        // loaders address any strides (orientation only decides which index runs along the lanes), so a member whose
        // own orientation differs is still computed correctly once 16-byte loads are off for that operand.
        if (g.n == 0) {
            a_kc0 = a_kc; b_kc0 = b_kc;
        }
        va = va && p.a_vec && a_kc == a_kc0;  // 16-byte loads only if every member allows them
        vb = vb && p.b_vec && b_kc == b_kc0;
        p.tiles_m = cgn_cdiv(d->M, 64);
        p.tiles_n = cgn_cdiv(d->N, 64);
        p.splitk = 1;
        p.k_per_split = cgn_cdiv(d->K > 0 ? d->K : 1, KROUND) * KROUND;
        p.order = 2;
        blocks += (long)p.tiles_m * p.tiles_n;
        CGN_CHECK_ARG(blocks < (1L << 30));
        g.p[g.n] = p;
        g.tile_end[g.n] = (int)blocks;
        ++g.n;
    }
    if (g.n == 0) return CGN_OK;
    hipStream_t s = (hipStream_t)stream;
    switch ((va ? 1 : 0) | (vb ? 2 : 0)) {
        case 0: launch_group<0>(g, a_kc0, b_kc0, (unsigned)blocks, s); break;
        case 1: launch_group<1>(g, a_kc0, b_kc0, (unsigned)blocks, s); break;
        case 2: launch_group<2>(g, a_kc0, b_kc0, (unsigned)blocks, s); break;
        default: launch_group<3>(g, a_kc0, b_kc0, (unsigned)blocks, s); break;
    }
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
