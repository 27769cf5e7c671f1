// bf16 x bf16 -> fp32 | bf16 products for gfx950 (cgn_gemm_bb, cgn_cast_bf16_f32; include/cgn_hip.h): the optional bf16
// mode's nn.Linear / LSTM input products with both operands HELD in bf16.
//
// Why its own kernel: with bf16 operands the 128x128 tile of gemm_core.h is paced by LDS traffic, not by the matrix
// pipe (per 16 k a block reads 16 KB of fragments and writes 8 KB for 128 MFMA cycles per SIMD), and converting fp32
// operands at staging costs 4 bytes per element from L2 plus a cvt per pair, in every block that touches the tile.
// Here:
//   * block tile 256x256, 8 waves in a 2x4 grid, each wave 128x64 = 4x2 accumulators of 32x32 (128 accumulator
//     registers): 6 fragment reads per 8 MFMAs, 48 KB of LDS reads + 16 KB of writes per 512 MFMA cycles per SIMD;
//     from the same template a 256x128 tile (4 waves of 128x64, two blocks per CU) and a 128x128 tile (4 waves of
//     64x64, three blocks per CU) for products with few tiles or narrow outputs (the planner scores tile x split by
//     fill, padding and the tile's measured rate);
//   * operands are bf16 in memory, k-contiguous: a thread's 16-byte global load IS 8 consecutive k of one row and goes
//     to LDS as one ds_write_b128, nothing is converted in the loop;
//   * LDS image per operand and stage: [rows][32 k] bf16 = 64-byte rows WITHOUT padding, the 16-byte chunk c of row r
//     stored at chunk position c ^ ((r >> 2) & 3): the staging writes (4 consecutive rows x 4 chunks per 16 lanes) and
//     the fragment reads (16 consecutive rows, one chunk) both touch every bank once;
//   * THREE LDS stages, K step 32, one barrier per step: while a step multiplies tile u, tile u+2 is written and the
//     first fragments of tile u+1 (complete since the previous barrier) are read BEFORE the barrier, so no wave waits
//     for an LDS round trip with an empty matrix pipe; global loads run two more steps ahead in registers;
//   * buffer loads with a fixed per-lane offset and a scalar k offset; k chunks past the end of the K range and tiles
//     past the last one are loads from an out-of-range offset (return 0): no masks, no branches in the loop;
//   * blocks of one XCD (blockIdx % 8) walk 8-tile-tall column groups: an XCD's 32 resident blocks share 8 + 4 operand
//     panels through its L2;
//   * few-tile / tall-K products are split along K into fp32 slabs and summed in fixed order (bitwise reproducible);
//   * two variants for the logits-sized products over bf16 logits: an exp transform between A's global load and its LDS
//     write (the soft embedding), and the fused softmax / CE backward epilogue, done on row chunks (see there).
// Measured (tools/gemm_bb_probe.py, profiles/r03_bb_probe.txt): 1.09 PFLOP/s at 4096^3, 1.16 at 8192^3 - the chip's power
// limit for this pipe (0.62 matrix-pipe busy x 1.82 GHz; tools/r03_bb_pmc.sh) -, 0.82-0.92 on the configs[3] products.
#include "common.h"

namespace cgn_bb {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Params {
    const void* A;
    const void* B;
    void* C;
    const float* bias0;
    const float* bias1;
    float* ws;
    long lda, ldb, ldc, ldw;
    int M, N, K;
    int kps, splitk;  // k per split (multiple of 32)
    int tiles_m, tiles_n;
    float alpha, beta;
    int c16;
    int c_vec;  // bf16 C: rows allow 16-byte stores
    unsigned a_bytes, b_bytes;
    // EXPA: A := exp(A * a_scale - a_row_param[m]); SMB: the fused softmax / CE backward epilogue (cgn_gemm_bb_desc)
    const float* a_row_param;
    const float* dyn;  // optional device scalars {inv_tau, ce_scale}: override a_scale / ep_inv_tau and ep_ce_scale
    const void* ep_logits;
    const float* ep_lse;
    const float* ep_rowdot;
    const int64_t* ep_target;
    void* ct;
    long ep_ld, ldct;
    int map_whole;  // block -> (split, tile) run over the whole list (few tiles per split), see bb_kernel
    int ct_vec;
    float a_scale, ep_inv_tau, ep_ce_scale;
};

#ifndef CGN_BB_SCHED
#define CGN_BB_SCHED 1
#endif
constexpr int BK = 32;
constexpr int ROWB = BK * 2;  // bytes per LDS row
constexpr int NS = 3;         // LDS stages
constexpr int OOB = (int)0x80000000;

__device__ __forceinline__ unsigned pack2(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// EXPA: the A operand holds bf16 logits and the product takes A := exp(A * a_scale - a_row_param[m]) (the softmax
// probabilities of the soft embedding, model/cgn.py:1103-1106), formed between the global load and the LDS write.
// SMB: the fused softmax / CE backward epilogue over bf16 logits, in place, with the optional transposed copy.
// Waves per SIMD the variants are built for (register budget 512 / waves): 256x256 = one block of 8 waves per CU;
// 256x128 = two blocks of 4 waves (one block's epilogue runs beside the other's K loop); 128x128 = three blocks.
template <int WTM, int WTN, int WVM, int WVN, bool EXPA = false, bool SMB = false>
__global__ __launch_bounds__(64 * WVM * WVN, (WTM == 2 && WVM * WVN == 4 ? 3 : 2)) void bb_kernel(const Params p) {
    constexpr int BM = WVM * WTM * 32, BN = WVN * WTN * 32, NTH = 64 * WVM * WVN;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LA = BM * 4 / NTH, LB = BN * 4 / NTH;  // 16-byte chunks per thread and tile
    constexpr int RSTEP = NTH / 4;                       // rows between a thread's chunks
    static_assert(BM * 4 % NTH == 0 && BN * 4 % NTH == 0 && RSTEP % 16 == 0, "loader geometry");
    // the staged bf16 epilogues flush 64 rows x 64 columns per wave; the 128x384 tile (wave tile 64x96) writes fp32 only
    constexpr bool C16OK = WTN == 2 && WTM % 2 == 0;
    static_assert(C16OK || !SMB, "the fused epilogue writes bf16");
    // (the fused epilogue's wave-local images, 13312 bytes per wave, outgrow the three stages of the 256x256 tile)
    constexpr int EPI_BYTES = SMB ? (NTH / 64) * (32 * 68 * 4 + 32 * 72 * 2) : 0;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WVN, wn = wave % WVN;
    const int fr = lane & 31, fk = lane >> 5;

    // block -> tile: XCD x (= blockIdx % 8) owns a contiguous run of logical ids; logical ids walk groups of GM tile rows
    // Split launches (splits on blockIdx.y): blockIdx.x % 8 is the XCD of a block of split 0 only - the hardware deals
    // blocks in linear order, x fastest. Products with FEW tiles per split (<= 100: map_whole) take their run over the
    // whole (split, tile) list in that linear order, so that an XCD works on ONE K range and its blocks share that
    // range's operand panels: dH1 at configs[1] 66.5 -> 58 us, the highway dW at configs[3] shapes 77 -> 69, the LSTM dW
    // 102 -> 99. Products with many tiles per split keep the per-split run although the PMC passes show their big
    // operand fetched 1.6-3.7 times (tools/r03_cfg4_pmc.sh): with the whole-list run the soft embedding at configs[3]
    // shapes took 756 -> 803 us and dH1 1319 -> 1382 - the re-reads of XCDs that walk the same K range at the same time
    // are served by the memory-side cache, and that lockstep is worth more than the fetches it costs.
    int tm, tn, split;
    {
        int l;
        if (p.map_whole) {
            const int nb = gridDim.x * gridDim.y, per = nb >> 3, rem = nb & 7;
            const int bid = blockIdx.x + gridDim.x * blockIdx.y;
            const int x = bid & 7, q = bid >> 3;
            const int lw = x * per + (x < rem ? x : rem) + q;
            split = lw / (int)gridDim.x;
            l = lw - split * (int)gridDim.x;
        } else {
            const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
            const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
            l = x * per + (x < rem ? x : rem) + q;
            split = blockIdx.y;
        }
        constexpr int GM = 8;
        const int span = GM * p.tiles_n;
        const int g = l / span, in = l - g * span;
        const int first = g * GM;
        const int gsz = p.tiles_m - first < GM ? p.tiles_m - first : GM;
        tm = first + in % gsz;
        tn = in / gsz;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * p.kps;
    const int kend = p.K < kbeg + p.kps ? p.K : kbeg + p.kps;
    const int nk = (kend - kbeg + BK - 1) / BK;

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    const int lc = tid & 3, lr = tid >> 2;
    int voa[LA], vob[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        int row = m0 + lr + i * RSTEP;
        row = row < p.M ? row : p.M - 1;  // clamped: rows past the end are never stored
        voa[i] = (int)((long)row * p.lda * 2) + lc * 16;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        int row = n0 + lr + i * RSTEP;
        row = row < p.N ? row : p.N - 1;
        vob[i] = (int)((long)row * p.ldb * 2) + lc * 16;
    }
    const int lds_w = lr * ROWB + ((lc ^ ((lr >> 2) & 3)) << 4);  // (RSTEP % 16 == 0: the same swizzle for every chunk)

    float rp[EXPA ? LA : 1];  // EXPA: the row parameter (log-sum-exp) of this thread's rows
    float ascale = 0.f;
    if constexpr (EXPA) {
        ascale = p.dyn ? p.dyn[0] : p.a_scale;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int row = m0 + lr + i * RSTEP;
            rp[i] = p.a_row_param[row < p.M ? row : p.M - 1];
        }
    }
    u32x4 ga[2][LA], gb[2][LB];
    bool oka[2] = {false, false};
    auto gload = [&](int R, int kt) {  // global loads of tile kt into register stage R
        const int k0 = kbeg + kt * BK;
        const bool ok = k0 + lc * 8 < kend;
        oka[R] = ok;
        const int so = __builtin_amdgcn_readfirstlane(k0) * 2;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            ga[R][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? voa[i] : OOB, so, 0));
#pragma unroll
        for (int i = 0; i < LB; ++i)
            gb[R][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? vob[i] : OOB, so, 0));
    };
    auto lstore = [&](int R, int st) {  // register stage R -> LDS stage at byte offset st
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            u32x4 v = ga[R][i];
            if constexpr (EXPA) {  // (chunks past the K range must stay zero: exp(0 - lse) is not)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float lo = __builtin_bit_cast(float, v[w] << 16), hi = __builtin_bit_cast(float, v[w] & 0xffff0000u);
                    const unsigned e = pack2(__expf(lo * ascale - rp[i]), __expf(hi * ascale - rp[i]));
                    v[w] = oka[R] ? e : 0u;
                }
            }
            *reinterpret_cast<u32x4*>(lds + st + lds_w + i * RSTEP * ROWB) = v;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<u32x4*>(lds + st + A_BYTES + lds_w + i * RSTEP * ROWB) = gb[R][i];
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // lane's fragment offsets inside an operand image: row fr of its 32-row tile, chunk (2s + fk) of sub-step s
    const int sw = (fr >> 2) & 3;
    int fo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) fo[s] = fr * ROWB + (((2 * s + fk) ^ sw) << 4);
    const int a_base = wm * (WTM * 32) * ROWB, b_base = A_BYTES + wn * (WTN * 32) * ROWB;
    bf16x8 fa[2][WTM], fb[2][WTN];
    auto fread = [&](int s, int st) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
            fa[s][i] = *reinterpret_cast<const bf16x8*>(lds + st + a_base + fo[s] + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < WTN; ++j)
            fb[s][j] = *reinterpret_cast<const bf16x8*>(lds + st + b_base + fo[s] + j * 32 * ROWB);
    };
    auto mfma = [&](int s) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    };

    // prologue: tiles 0 and 1 into LDS stages 0 and 1, tiles 2 and 3 on their way in the two register stages
    gload(0, 0);
    gload(1, 1);
    lstore(0, 0);
    gload(0, 2);
    lstore(1, STAGE);
    gload(1, 3);
    __syncthreads();
    fread(0, 0);

    // step u: multiplies tile u (stage cur); writes tile u+2 (register stage u & 1) into stage st; loads tile u+4
    int cur = 0, nxt = STAGE, st = 2 * STAGE;
    auto step = [&](int R, int kt) {
        // Two phases, fenced for the scheduler (left alone hipcc hoists all twelve fragment reads and the four writes
        // to the top of the step and issues the two MFMAs of each accumulator back to back: every wave of the block then
        // queues at the LDS behind the barrier while the matrix pipe idles - 0.59 matrix-pipe busy at 8192^3):
        //   phase 1: MFMAs of sub-step 0, one per slot, with the reads of sub-step 1 and the writes of tile u+2
        //   phase 2: MFMAs of sub-step 1, with the reads of tile u+1's sub-step 0 (complete since the barrier that ended
        //            step u-1) and the global loads of tile u+4
        constexpr int NM = WTM * WTN, NR = WTM + WTN, NL = LA + LB;
        static_assert(NR <= NM && NL <= NM, "one LDS read, one LDS write / global load per MFMA slot");
        fread(1, cur);
        lstore(R, st);
        mfma(0);
#if CGN_BB_SCHED
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g < NL) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        gload(R, kt + 4);
        fread(0, nxt);
        mfma(1);
#if CGN_BB_SCHED
#pragma unroll
        for (int g = 0; g < NM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (g < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (g < NL) {
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        __syncthreads();
        const int t = cur;
        cur = nxt;
        nxt = st;
        st = t;
    };
    // (an odd tile count runs one more step on a tile of zeros: loads past kend return 0)
    for (int kt = 0; kt < nk; kt += 2) {
        step(0, kt);
        step(1, kt + 1);
    }

    const int row0 = m0 + wm * (WTM * 32), col0 = n0 + wn * (WTN * 32);
    if (p.splitk > 1) {  // slab of this split: plain partial sums, the reduce kernel applies the epilogue
        float* slab = p.ws + (long)split * p.M * p.ldw;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < WTN; ++j) {
                        const int col = col0 + j * 32 + fr;
                        if (col < p.N) slab[(long)row * p.ldw + col] = acc[i][j][r];
                    }
                }
            }
        return;
    }
    if constexpr (SMB) {
        // Fused softmax / CE backward (model/cgn.py:1086-1108 backwards; host: bf16 C, no split): with
        // pr = exp(logits * inv_tau - lse[m]), C = pr * (acc - rowdot[m]) * inv_tau + ce_scale * (pr - [n == target[m]]),
        // written over the logits. 525 M elements at configs[3] shapes: the epilogue is instruction-bound, so the
        // arithmetic runs on ROW CHUNKS, not in the accumulators' layout (one column x 4 rows per lane: 2-byte LDS
        // traffic per element, ~40 instructions each, 1.2 ms). Per 32-row slab of the wave:
        //   1. the accumulators go to a wave-local fp32 image [32][68] (4-byte writes, lanes along the columns);
        //   2. a lane owns 8 consecutive columns of one row, four times: its logits are ONE 16-byte global load (issued a
        //      slab ahead), its sums two ds_read_b128; the 8 results leave as one 16-byte store to C and one
        //      ds_write_b128 into a bf16 image [32][72];
        //   3. the transposed copy reads that image down the columns (8 two-byte reads per 16-byte store to ct).
        // (Where the transposed copy's 0.39 ms at configs[3] shapes go, measured with timing-only builds: the LDS reads
        // cost nothing (stores left out: +9 us over the product without the copy); the same bytes stored row-major
        // instead of as 64-byte column pieces: 1188 against 1272 us. It is the third GB of traffic itself, in a kernel
        // whose 256x256 blocks do their memory phase after their matrix phase.)
        constexpr int RSF = 68, RSR = 72, WAVE_BYTES = 32 * RSF * 4 + 32 * RSR * 2;
        __syncthreads();  // every wave is done with the operand stages
        float* facc = reinterpret_cast<float*>(lds + wave * WAVE_BYTES);
        unsigned short* simg = reinterpret_cast<unsigned short*>(lds + wave * WAVE_BYTES + 32 * RSF * 4);
        const float inv_tau = p.dyn ? p.dyn[0] : p.ep_inv_tau;
        const float ce_scale = p.dyn ? p.dyn[1] : p.ep_ce_scale;
        const unsigned short* lgp = reinterpret_cast<const unsigned short*>(p.ep_logits);
        unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
        unsigned short* Ct = reinterpret_cast<unsigned short*>(p.ct);
        const bool lvec = p.ep_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(p.ep_logits) & 15) == 0;
        const int crl = lane >> 3, ccc = (lane & 7) * 8;  // the lane's row (+ 8 ps) and first column of its chunks
        u32x4 lq[4];
        auto load_slab = [&](int i) {
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = row0 + i * 32 + ps * 8 + crl, col = col0 + ccc;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (row < p.M && col < p.N) {
                    const unsigned short* src = lgp + (long)row * p.ep_ld + col;
                    if (lvec && col + 8 <= p.N) {
                        v = *reinterpret_cast<const u32x4*>(src);
                    } else {
                        unsigned short e[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) e[q] = col + q < p.N ? src[q] : (unsigned short)0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = (unsigned)e[2 * q] | ((unsigned)e[2 * q + 1] << 16);
                    }
                }
                lq[ps] = v;
            }
        };
        load_slab(0);
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) facc[((r & 3) + 8 * (r >> 2) + 4 * fk) * RSF + j * 32 + fr] = acc[i][j][r];
            float lse_r[4], rd_r[4];
            int tg_r[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = row0 + i * 32 + ps * 8 + crl;
                const int rc = row < p.M ? row : 0;
                lse_r[ps] = p.ep_lse[rc];
                rd_r[ps] = p.ep_rowdot[rc];
                tg_r[ps] = (int)p.ep_target[rc];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local images
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int rl = ps * 8 + crl;
                const int row = row0 + i * 32 + rl, col = col0 + ccc;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(facc + rl * RSF + ccc);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(facc + rl * RSF + ccc + 4);
                const int tl = tg_r[ps] - col;  // the target's position inside the chunk, if any
                u32x4 o;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float l0 = __builtin_bit_cast(float, lq[ps][w] << 16);
                    const float l1 = __builtin_bit_cast(float, lq[ps][w] & 0xffff0000u);
                    const float p0 = __expf(l0 * inv_tau - lse_r[ps]), p1 = __expf(l1 * inv_tau - lse_r[ps]);
                    const float a0 = w < 2 ? s0[2 * w] : s1[2 * w - 4], a1 = w < 2 ? s0[2 * w + 1] : s1[2 * w - 3];
                    const float v0 = p0 * (a0 - rd_r[ps]) * inv_tau + ce_scale * (p0 - (tl == 2 * w ? 1.f : 0.f));
                    const float v1 = p1 * (a1 - rd_r[ps]) * inv_tau + ce_scale * (p1 - (tl == 2 * w + 1 ? 1.f : 0.f));
                    o[w] = pack2(v0, v1);
                }
                *reinterpret_cast<u32x4*>(simg + rl * RSR + ccc) = o;
                // (this chunk's logits sit in this lane's registers: the in-place store cannot run ahead of a read)
                if (row < p.M && col < p.N) {
                    if (p.c_vec && col + 8 <= p.N) {
                        *reinterpret_cast<u32x4*>(Ch + (long)row * p.ldc + col) = o;
                    } else {
                        const unsigned short* e = reinterpret_cast<const unsigned short*>(&o);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (col + q < p.N) Ch[(long)row * p.ldc + col + q] = e[q];
                    }
                }
            }
            if (i + 1 < WTM) load_slab(i + 1);
            if (Ct) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int cl = ps * 16 + (lane >> 2), rr = (lane & 3) * 8;
                    const int col = col0 + cl, row = row0 + i * 32 + rr;
                    u32x4 o;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        o[w] = (unsigned)simg[(rr + 2 * w) * RSR + cl] | ((unsigned)simg[(rr + 2 * w + 1) * RSR + cl] << 16);
                    if (col < p.N && row < p.M) {
                        if (p.ct_vec && row + 8 <= p.M) {
                            *reinterpret_cast<u32x4*>(Ct + (long)col * p.ldct + row) = o;
                        } else {
                            const unsigned short* e = reinterpret_cast<const unsigned short*>(&o);
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (row + q < p.M) Ct[(long)col * p.ldct + row + q] = e[q];
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    float bias[WTN];
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int col = col0 + j * 32 + fr;
        const int cc = col < p.N ? col : 0;
        bias[j] = (p.bias0 ? p.bias0[cc] : 0.f) + (p.bias1 ? p.bias1[cc] : 0.f);
    }
    if (!p.c16) {
        float* Cf = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < p.M) {
#pragma unroll
                    for (int j = 0; j < WTN; ++j) {
                        const int col = col0 + j * 32 + fr;
                        if (col < p.N) {
                            float v = p.alpha * acc[i][j][r] + bias[j];
                            if (p.beta != 0.f) v += p.beta * Cf[(long)row * p.ldc + col];
                            Cf[(long)row * p.ldc + col] = v;
                        }
                    }
                }
            }
        return;
    }
    if constexpr (C16OK) {
    // bf16 C: 64 rows x 64 columns of the wave at a time through LDS (free after the loop), out as 16-byte row stores
    constexpr int RS = 72;  // staging row stride in elements (144 bytes)
    static_assert((NTH / 64) * 64 * RS * 2 <= NS * STAGE, "staging images must fit the operand stages");
    __syncthreads();  // every wave is done with the operand stages
    unsigned short* stg = reinterpret_cast<unsigned short*>(lds) + wave * (64 * RS);
    unsigned short* Ch = reinterpret_cast<unsigned short*>(p.C);
#pragma unroll
    for (int h = 0; h < WTM / 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // rows r and r + 1 of one column: two 2-byte writes (lanes along the columns: conflict-free)
                    const int rl = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    const unsigned w = pack2(p.alpha * acc[2 * h + ii][j][r] + bias[j],
                                             p.alpha * acc[2 * h + ii][j][r + 1] + bias[j]);
                    stg[rl * RS + j * 32 + fr] = (unsigned short)(w & 0xffffu);
                    stg[(rl + 1) * RS + j * 32 + fr] = (unsigned short)(w >> 16);
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local image: written by this wave's lanes only
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int rl = ps * 8 + (lane >> 3), cc = (lane & 7) * 8;
            const int row = row0 + h * 64 + rl, col = col0 + cc;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + rl * RS + cc);
            if (row < p.M && col < p.N) {
                if (p.c_vec && col + 8 <= p.N) {
                    *reinterpret_cast<u32x4*>(Ch + (long)row * p.ldc + col) = v;
                } else {
                    const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (col + q < p.N) Ch[(long)row * p.ldc + col + q] = e[q];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    }  // C16OK
}

// Sums the slabs in fixed order and applies the linear epilogue (fp32 C only).
__global__ __launch_bounds__(256) void bb_reduce_kernel(const Params p) {
    const long total = (long)p.M * p.ldw, nvec = total / 4;
    float* Cf = reinterpret_cast<float*>(p.C);
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
        const long idx = v * 4;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int z0 = 0; z0 < p.splitk; z0 += 4) {
            f32x4 part[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int z = z0 + u < p.splitk ? z0 + u : p.splitk - 1;
                part[u] = *reinterpret_cast<const f32x4*>(p.ws + (long)z * total + idx);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (z0 + u < p.splitk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[j] += part[u][j];
                }
        }
        const int row = (int)(idx / p.ldw), col = (int)(idx % p.ldw);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (col + j < p.N) {
                float o = p.alpha * s[j];
                if (p.bias0) o += p.bias0[col + j];
                if (p.bias1) o += p.bias1[col + j];
                const long ci = (long)row * p.ldc + col + j;
                if (p.beta != 0.f) o += p.beta * Cf[ci];
                Cf[ci] = o;
            }
    }
}

// ---- casts ---------------------------------------------------------------------------------------------------------
struct RowAxis {
    long stride, gs;
    int rpg;  // 0 = plain
};
__device__ __forceinline__ long row_off(const RowAxis& a, int r) {
    if (a.rpg == 0) return (long)r * a.stride;
    const int g = r / a.rpg;
    return (long)g * a.gs + (long)(r - g * a.rpg) * a.stride;
}

// dst[r][c] = bf16(src[r][c]), 8 columns per thread; columns n_cols..ld_dst-1 zero
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, RowAxis ax, int n_rows, int n_cols,
                                                        unsigned short* __restrict__ dst, long ld_dst, int vec) {
    const int cpr = (int)(ld_dst / 8);  // 8-column chunks per row
    const long total = (long)n_rows * cpr;
    for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long)gridDim.x * blockDim.x) {
        const int r = (int)(v / cpr), c = (int)(v - (long)r * cpr) * 8;
        const float* s = src + row_off(ax, r) + c;
        float x[8];
        if (vec && c + 8 <= n_cols) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(s), hi = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = lo[q], x[4 + q] = hi[q];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = c + q < n_cols ? s[q] : 0.f;
        }
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2(x[2 * q], x[2 * q + 1]);
        *reinterpret_cast<u32x4*>(dst + (long)r * ld_dst + c) = o;
    }
}

// dst[c][r] = bf16(src[r][c]): 64 x 64 tiles through LDS; columns n_rows..ld_dst-1 of every dst row zero. A thread
// reads two consecutive rows at four columns (16-byte loads where the source allows), packs the row PAIR of each column
// into one word and writes it to the [c][r / 2] image (33-word rows: the 64 lanes of a write touch 64 banks); a dst row
// then leaves as 16-byte stores of 8 rows each.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, RowAxis ax, int n_rows,
                                                             int n_cols, unsigned short* __restrict__ dst, long ld_dst,
                                                             int vec) {
    __shared__ unsigned t[64][33];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int rp = ps * 16 + (tid >> 4), cl = (tid & 15) * 4;  // row pair, first of 4 columns
        float x[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r0 + 2 * rp + h;
#pragma unroll
            for (int q = 0; q < 4; ++q) x[h][q] = 0.f;
            if (r < n_rows) {
                const float* sp = src + row_off(ax, r) + c0 + cl;
                if (vec && c0 + cl + 4 <= n_cols) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(sp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[h][q] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + cl + q < n_cols) x[h][q] = sp[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) t[cl + q][rp] = pack2(x[0][q], x[1][q]);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int cl = ps * 32 + (tid >> 3), j = tid & 7;  // dst row, chunk of 8 rows
        const int c = c0 + cl;
        if (c < n_cols && r0 + 8 * j < ld_dst) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = t[cl][4 * j + e];
            *reinterpret_cast<u32x4*>(dst + (long)c * ld_dst + r0 + 8 * j) = o;  // (rows past n_rows were read as zeros)
        }
    }
}

struct Tile {
    int code, bm, bn, slots;  // slots: blocks resident on 256 CUs (registers and LDS of the variant)
    double rate;              // steady-state rate relative to the 256x256 tile (measured at 4096^3 ... 8192^3)
};
// code = cgn_gemm_bb_desc.force_tile: 256 -> 256x256 (8 waves), 192 -> 256x128 (4 waves of 128x64), 128 -> 128x128,
// 384 -> 128x384 (8 waves of 64x96, fp32 C only): ONE column tile for N <= 384, so the A operand is read - and, with the
// exp loader, exponentiated - once instead of once per 128 columns. The 256x128 tile spends 32 exp per thread and 32-k
// step on 16 MFMAs and is VALU-bound with that loader (430 TFLOP/s on the soft embedding); here it is 8 exp on 12 MFMAs.
constexpr Tile TILES[4] = {{256, 256, 256, 256, 1.0}, {192, 256, 128, 512, 0.93}, {128, 128, 128, 768, 0.80},
                           {384, 128, 384, 256, 0.90}};

static inline bool plan(const cgn_gemm_bb_desc* d, bool smb, Tile* t_out, int* sk_out, int* kps_out) {
    const int K = d->K;
    const bool expa = d->a_transform == CGN_ATRANS_EXP;
    // share of the block slots of the last round that hold a block
    auto fill = [&](long blocks, int slots) { return (double)blocks / ((double)cgn_cdiv(blocks, slots) * slots); };
    // splits along K that the caller's workspace can hold and that keep at least 512 k each
    const long slab = (long)d->M * ((d->N + 3) / 4 * 4) * 4;
    auto can = [&](int s) {
        if (d->force_splitk) return s == d->force_splitk;
        return s == 1 || (!d->c_bf16 && d->workspace != nullptr && K / s >= 512 && slab * s <= d->workspace_bytes);
    };
    double best = -1.0;
    for (const Tile& t : TILES) {
        if (d->force_tile && d->force_tile != t.code) continue;
        if (smb && t.code == 128) continue;  // (the fused epilogue is built for the 128x64 wave tile)
        // the wide tile: fp32 results of at most one column tile; by request, or for the exp loader (see TILES)
#ifndef CGN_BB_WIDE_EXPA
#define CGN_BB_WIDE_EXPA 1  // (0: A/B builds, tools/build_variant.sh)
#endif
        const bool wide_expa = CGN_BB_WIDE_EXPA && expa;
        if (t.code == 384 && (d->c_bf16 || smb || d->N > 384 || !(d->force_tile == 384 || wide_expa))) continue;
        if (wide_expa && !d->force_tile && d->N <= 384 && t.code != 384 && !d->c_bf16) continue;
        const long tiles = (long)cgn_cdiv(d->M, t.bm) * cgn_cdiv(d->N, t.bn);
        const double pad = (double)d->M * d->N / ((double)tiles * t.bm * t.bn);
        for (int s = 1; s <= 16; ++s) {
            if (!can(s)) continue;
            // a split costs a slab round trip per block and a reduce pass: worth it only for a clearly better fill
            const double score = t.rate * pad * fill(tiles * s, t.slots) * (1.0 - 0.02 * (s - 1));
            if (score > best + 1e-9) {
                best = score;
                *t_out = t;
                *sk_out = s;
            }
        }
    }
    if (best < 0.0) return false;
    int sk = *sk_out;
    if (sk > 1 && (d->c_bf16 || d->workspace == nullptr)) return false;
    const int kps = cgn_cdiv(cgn_cdiv(K, sk), BK) * BK;
    sk = cgn_cdiv(K, kps);
    if (sk > 1 && slab * sk > d->workspace_bytes) return false;
    *sk_out = sk;
    *kps_out = kps;
    return true;
}

template <bool EXPA, bool SMB>
static void launch(const Tile& t, dim3 grid, hipStream_t s, const Params& p) {
    if (t.code == 256)
        hipLaunchKernelGGL((bb_kernel<4, 2, 2, 4, EXPA, SMB>), grid, dim3(512), 0, s, p);
    else if (t.code == 192)
        hipLaunchKernelGGL((bb_kernel<4, 2, 2, 2, EXPA, SMB>), grid, dim3(256), 0, s, p);
    else if (t.code == 384) {
        if constexpr (!SMB) hipLaunchKernelGGL((bb_kernel<2, 3, 2, 4, EXPA, false>), grid, dim3(512), 0, s, p);
    } else if constexpr (!SMB)
        hipLaunchKernelGGL((bb_kernel<2, 2, 2, 2, EXPA, false>), grid, dim3(256), 0, s, p);
}

}  // namespace cgn_bb

using namespace cgn_bb;

extern "C" int cgn_gemm_bb_wants(int32_t M, int32_t N, int32_t K) { return (M >= 128 && N >= 128 && K >= 64) ? 1 : 0; }

extern "C" int cgn_gemm_bb(void* stream, const cgn_gemm_bb_desc* d, void* ev_start, void* ev_stop, int32_t* tile,
                           int32_t* splitk) {
    CGN_CHECK_ARG(d != nullptr);
    CGN_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0 && d->M < (1 << 30) && d->N < (1 << 30) && d->K < (1 << 30));
    if (tile) *tile = 0;
    if (splitk) *splitk = 0;
    if (d->M == 0 || d->N == 0) return CGN_OK;
    CGN_CHECK_ARG(d->A && d->B && d->C && d->K > 0 && d->K % 8 == 0);
    CGN_CHECK_ARG(d->lda % 8 == 0 && d->ldb % 8 == 0 && d->lda >= d->K && d->ldb >= d->K && d->ldc >= d->N);
    CGN_CHECK_ARG((uintptr_t)d->A % 16 == 0 && (uintptr_t)d->B % 16 == 0 && (uintptr_t)d->workspace % 16 == 0);
    CGN_CHECK_ARG(d->force_tile == 0 || d->force_tile == 128 || d->force_tile == 192 || d->force_tile == 256 ||
                  d->force_tile == 384);
    CGN_CHECK_ARG(d->force_splitk >= 0 && d->force_splitk <= 16);
    CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE || (d->a_transform == CGN_ATRANS_EXP && d->a_row_param));
    CGN_CHECK_ARG(d->epilogue == CGN_EPI_LINEAR || d->epilogue == CGN_EPI_SOFTMAX_BWD);
    const bool smb = d->epilogue == CGN_EPI_SOFTMAX_BWD, expa = d->a_transform == CGN_ATRANS_EXP;
    if (smb) {
        CGN_CHECK_ARG(d->c_bf16 && !expa && d->ep_logits && d->ep_lse && d->ep_rowdot && d->ep_target && d->ep_ld >= d->N);
        CGN_CHECK_ARG(d->force_tile != 128 && (uintptr_t)d->ep_logits % 2 == 0 && (uintptr_t)d->c_t % 2 == 0);
        CGN_CHECK_ARG(d->c_t == nullptr || d->ld_ct >= d->M);
    } else {
        CGN_CHECK_ARG(d->c_t == nullptr);
    }
    if (d->c_bf16) CGN_CHECK_ARG(d->beta == 0.f && (uintptr_t)d->C % 2 == 0 && d->force_splitk <= 1);
    else CGN_CHECK_ARG((uintptr_t)d->C % 4 == 0);
    const long ea = ((long)(d->M - 1) * d->lda + d->K) * 2, eb = ((long)(d->N - 1) * d->ldb + d->K) * 2;
    CGN_CHECK_ARG(ea < (1L << 31) - 64 && eb < (1L << 31) - 64);
    Tile t = TILES[0];
    int sk, kps;
    CGN_CHECK_ARG(plan(d, smb, &t, &sk, &kps));
    Params p;
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.bias0 = d->bias0; p.bias1 = d->bias1; p.ws = d->workspace;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldw = (d->N + 3) / 4 * 4;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.kps = kps; p.splitk = sk;
    p.tiles_m = cgn_cdiv(d->M, t.bm); p.tiles_n = cgn_cdiv(d->N, t.bn);
    p.map_whole = (sk > 1 && p.tiles_m * p.tiles_n <= 100) ? 1 : 0;
    p.alpha = d->alpha; p.beta = d->beta;
    p.c16 = d->c_bf16 ? 1 : 0;
    p.c_vec = (d->c_bf16 && d->ldc % 8 == 0 && (uintptr_t)d->C % 16 == 0) ? 1 : 0;
    p.a_bytes = (unsigned)ea; p.b_bytes = (unsigned)eb;
    p.a_row_param = d->a_row_param; p.dyn = d->dyn_scalars;
    p.ep_logits = d->ep_logits; p.ep_lse = d->ep_lse; p.ep_rowdot = d->ep_rowdot; p.ep_target = d->ep_target;
    p.ct = d->c_t; p.ep_ld = d->ep_ld; p.ldct = d->ld_ct;
    p.ct_vec = (d->c_t && d->ld_ct % 8 == 0 && (uintptr_t)d->c_t % 16 == 0) ? 1 : 0;
    p.a_scale = d->a_scale; p.ep_inv_tau = d->ep_inv_tau; p.ep_ce_scale = d->ep_ce_scale;
    if (tile) *tile = t.code;
    if (splitk) *splitk = sk;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((long)p.tiles_m * p.tiles_n), (unsigned)sk);
    if (ev_start && hipEventRecord((hipEvent_t)ev_start, s) != hipSuccess) return CGN_ERR_HIP_BASE + (int)hipGetLastError();
    if (smb)
        launch<false, true>(t, grid, s, p);
    else if (expa)
        launch<true, false>(t, grid, s, p);
    else
        launch<false, false>(t, grid, s, p);
    CGN_LAUNCH_CHECK();
    if (ev_stop && hipEventRecord((hipEvent_t)ev_stop, s) != hipSuccess) return CGN_ERR_HIP_BASE + (int)hipGetLastError();
    if (sk > 1) {
        int blocks = cgn_cdiv((long)d->M * p.ldw / 4, 256);
        blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
        hipLaunchKernelGGL(bb_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
        CGN_LAUNCH_CHECK();
    }
    return CGN_OK;
}

extern "C" int cgn_cast_bf16_f32(void* stream, const float* src, cgn_axis rows, int32_t n_rows, int32_t n_cols, void* dst,
                                 int64_t ld_dst, int32_t transpose) {
    CGN_CHECK_ARG(n_rows >= 0 && n_cols >= 0);
    if (n_rows == 0 || n_cols == 0) return CGN_OK;
    CGN_CHECK_ARG(src && dst && ld_dst % 8 == 0 && (uintptr_t)dst % 16 == 0);
    CGN_CHECK_ARG(rows.stride >= 0 && rows.rpg >= 0 && rows.gs >= 0);
    CGN_CHECK_ARG(ld_dst >= (transpose ? n_rows : n_cols));
    RowAxis ax;
    ax.stride = rows.stride; ax.gs = rows.gs; ax.rpg = (int)rows.rpg;
    hipStream_t s = (hipStream_t)stream;
    if (!transpose) {
        const int vec = ((uintptr_t)src % 16 == 0 && rows.stride % 4 == 0 && (rows.rpg == 0 || rows.gs % 4 == 0)) ? 1 : 0;
        const long total = (long)n_rows * (ld_dst / 8);
        int blocks = cgn_cdiv(total, 256);
        blocks = blocks > 8192 ? 8192 : blocks;
        hipLaunchKernelGGL(cast_rows_kernel, dim3(blocks), dim3(256), 0, s, src, ax, n_rows, n_cols,
                           reinterpret_cast<unsigned short*>(dst), (long)ld_dst, vec);
    } else {
        // the row tiles cover the padded row count so that the padding columns of dst are written (as zeros)
        const dim3 grid((unsigned)cgn_cdiv(ld_dst, 64), (unsigned)cgn_cdiv(n_cols, 64));
        CGN_CHECK_ARG(grid.y < 65536);
        const int vec = ((uintptr_t)src % 16 == 0 && rows.stride % 4 == 0 && (rows.rpg == 0 || rows.gs % 4 == 0)) ? 1 : 0;
        hipLaunchKernelGGL(cast_transpose_kernel, grid, dim3(256), 0, s, src, ax, n_rows, n_cols,
                           reinterpret_cast<unsigned short*>(dst), (long)ld_dst, vec);
    }
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
