// LSTM recurrence for gfx950: forward and BPTT step kernels (replaces the cuDNN RNN behind nn.LSTM at
// model/encoder.py:17-21,40 and model/generator.py:17-20,58 of the reference).
//
// The time-batched input projections (x · W_ihᵀ + b_ih + b_hh) are MFMA GEMMs done by cgn_gemm_f32 beforehand; what
// remains per timestep is gates = xg[:, t] + h_{t-1} · W_hhᵀ and the cell update. Per step that is a
// [B,H]x[H,4H] product (134 MFLOP at B=64, H=512): latency-bound, so it is decomposed for parallelism, not tile size:
//   * one workgroup = 16 batch rows x 16 hidden units (all four gates of those units), grid = (H/16, B/16, nseq)
//     -> 256 workgroups (one per CU) for the encoder's two directions at cfg2;
//   * its 8 waves split K eight ways, each running v_mfma_f32_16x16x4_f32 chains on 4 independent accumulators
//     (i, f, g, o tiles), then sum through LDS; the i/f/g/o of one (row, unit) land in the same lane/register of the
//     four tiles, so the gate nonlinearities and the cell update need no cross-lane traffic;
//   * operands are read straight from L2 into registers as float4 along k: lane (r, q) feeds k = 16s + 4q + i to the
//     i-th MFMA of chunk s (any permutation of k is legal as long as A and B agree), so every load is a 16-byte
//     access of a 64-byte contiguous row segment;
//   * independent sequences (the encoder's two directions) advance in the same launch (blockIdx.z).
// One launch per timestep: on MI355X a dependent kernel boundary (~1.5 us) is cheaper than a grid-wide barrier
// (~4-5 us, MI355X_MICROARCH price list), so the seam between timesteps is a launch, not an in-kernel barrier.
//
// Buffers (all batch-major, fp32):
//   xg    [B, T, 4H]  input projections incl. both biases, gate order i,f,g,o (torch layout)
//   hbuf  [B, T+1, H] hidden history; forward sequences keep the initial state in slot 0 and h_t in slot t+1,
//                      reverse sequences keep the initial state in slot T and h_t in slot t
//   cbuf  [B, T+1, H] same for the cell state
//   gates [B, T, 4H]  activated gates, saved for the backward pass
#include "common.h"

namespace {

constexpr int LSTM_MAX_SEQ = 4;

struct FwdSeq {
    const float* xg;
    const float* xg_extra;
    const float* w_hh;
    float* hbuf;
    float* cbuf;
    float* gates;
    int reverse;
    int init_zero;
    unsigned short* hx;  // bf16 hand-off scratch (operand mode 2), see lstm_persist_fwd_rt_kernel
};
struct FwdParams {
    FwdSeq s[LSTM_MAX_SEQ];
    int B, T, H, step, nseq;
};

struct BwdSeq {
    const float* w_hh_t;
    const float* w_hh;
    const float* cbuf;
    const float* gates;
    const float* dh_ext;
    const float* dh_seed;
    float* dgates;
    float* dc_state;
    long dh_bs, dh_ts;
    int reverse;
    unsigned short* hx;  // bf16 hand-off scratch (operand mode 2)
};
struct BwdParams {
    BwdSeq s[LSTM_MAX_SEQ];
    int B, T, H, step, nseq;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// XCD-aware work mapping. Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one), and every
// (batch tile, sequence) of a hidden-unit tile reads the same W_hh rows. Give each XCD residue class a contiguous
// range of hidden-unit tiles, so an XCD's L2 sees only 1/8 of W_hh instead of all of it (otherwise every step
// re-streams 8 copies of the recurrent weights from the Infinity Cache). Placement only affects speed.
struct WorkId {
    int jt, bt, seq;
};
// blockIdx.x -> position in a work list of nwg items such that the blocks of one XCD residue class (b % 8) take a
// CONTIGUOUS range of the list; bijective for every nwg
__device__ __forceinline__ int xcd_contiguous(int nwg, int identity = 0) {
    const int bid = blockIdx.x;
    if (identity) return bid;  // CGN_LSTM_XCD_MAP=0: the round-2 placement (A/B measurements only)
    const int qq = nwg / 8, rr = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    return ((xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
}
__device__ __forceinline__ WorkId lstm_work(int nj, int nb, int nseq) {
    const int w = xcd_contiguous(nj * nb * nseq);
    WorkId o;
    o.jt = w / (nb * nseq);
    const int rest = w % (nb * nseq);
    o.bt = rest / nseq;
    o.seq = rest % nseq;
    return o;
}

constexpr int LSTM_NW = 8;            // waves per workgroup: K is split 8 ways
constexpr int LSTM_NT = LSTM_NW * 64;

// All global loads of a batch of chunks are issued unconditionally on clamped addresses BEFORE any of them is
// consumed, so a wave has 20-32 16-byte loads in flight at once; the per-step latency is then ~one L2 round trip plus
// the MFMA chain instead of one round trip per chunk.
__global__ __launch_bounds__(LSTM_NT) void lstm_step_fwd_kernel(const FwdParams p) {
    __shared__ float red[LSTM_NW][16][64];
    const int B = p.B, T = p.T, H = p.H;
    const WorkId wk = lstm_work((H + 15) / 16, (B + 15) / 16, p.nseq);
    const FwdSeq& sq = p.s[wk.seq];
    const int t = sq.reverse ? (T - 1 - p.step) : p.step;
    const int slot_prev = sq.reverse ? t + 1 : t;
    const int slot_out = sq.reverse ? t : t + 1;
    const int j0 = wk.jt * 16, b0 = wk.bt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;

    const int nchunks = (H + 15) / 16;
    const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;
    const int c_beg = wave * cpw, c_end = min(nchunks, c_beg + cpw);

    const int bA = min(b0 + fr, B - 1);  // clamped operand row / column: out-of-range results are never stored
    const int jB = min(j0 + fr, H - 1);
    const float* hrow = sq.hbuf + ((long)bA * (T + 1) + slot_prev) * H;
    const float* wrow = sq.w_hh + (long)jB * H;
    const long gate_stride = (long)H * H;

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};

    // Operands of the cell update (independent of the product): requested NOW, on clamped addresses and by every wave
    // (no branch), so their round trip overlaps the W_hh / h loads instead of following the LDS reduction.
    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1), ejc = min(ej, H - 1);
    float pre_x[4], pre_e[4];
    {
        const float* xg = sq.xg + ((long)ebc * T + t) * 4 * H + ejc;
        const float* xe = sq.xg_extra ? sq.xg_extra + (long)ebc * 4 * H + ejc : xg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pre_x[g] = xg[g * H];
            pre_e[g] = xe[g * H];
        }
    }
    const float cprev_pf = sq.cbuf[((long)ebc * (T + 1) + slot_prev) * H + ejc];
    const bool has_extra = sq.xg_extra != nullptr;

    constexpr int CH = 4;
    for (int c0 = c_beg; c0 < c_end; c0 += CH) {
        f32x4 a[CH], w[4][CH];
        bool kok[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int kb = (c0 + u) * 16 + q * 4;
            kok[u] = (c0 + u < c_end) && (kb < H);  // H % 4 == 0 (host check): kb < H implies kb + 3 < H
            const int kc = kok[u] ? kb : 0;
            a[u] = ld4(hrow + kc);
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g][u] = ld4(wrow + g * gate_stride + kc);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 au = kok[u] ? a[u] : z;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(au[i], w[g][u][i], acc[g], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][lane] = acc[g][r];
    __syncthreads();

    // thread (lane', r) of the first 4 waves finishes element (row = (lane'>>4)*4 + r, unit = lane'&15)
    if (wave >= 4) return;
    const int r = wave;
    const int b = b0 + (lane >> 4) * 4 + r;
    const int j = j0 + (lane & 15);
    if (b < B && j < H) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < LSTM_NW; ++wv) v += red[wv][g * 4 + r][lane];
            v += pre_x[g];
            if (has_extra) v += pre_e[g];
            pre[g] = v;
        }
        const float ig = cgn_sigmoid(pre[0]);
        const float fg = cgn_sigmoid(pre[1]);
        const float gg = tanhf(pre[2]);
        const float og = cgn_sigmoid(pre[3]);
        const float cprev = cprev_pf;
        const float cn = fg * cprev + ig * gg;
        const float hn = og * tanhf(cn);
        float* go = sq.gates + ((long)b * T + t) * 4 * H;
        go[j] = ig;
        go[H + j] = fg;
        go[2 * H + j] = gg;
        go[3 * H + j] = og;
        sq.cbuf[((long)b * (T + 1) + slot_out) * H + j] = cn;
        sq.hbuf[((long)b * (T + 1) + slot_out) * H + j] = hn;
    }
}

__global__ __launch_bounds__(LSTM_NT) void lstm_step_bwd_kernel(const BwdParams p) {
    __shared__ float red[LSTM_NW][4][64];
    const int B = p.B, T = p.T, H = p.H;
    const WorkId wk = lstm_work((H + 15) / 16, (B + 15) / 16, p.nseq);
    const BwdSeq& sq = p.s[wk.seq];
    // processing order is the reverse of the forward pass
    const int t = sq.reverse ? p.step : (T - 1 - p.step);
    const int t_next = sq.reverse ? t - 1 : t + 1;  // the timestep processed just before this one
    const bool first = p.step == 0;
    const int slot_prev = sq.reverse ? t + 1 : t;
    const int slot_out = sq.reverse ? t : t + 1;
    const int j0 = wk.jt * 16, b0 = wk.bt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int K = 4 * H;

    // operands of the gate-gradient update, requested before the product (see the forward kernel)
    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1), ejc = min(ej, H - 1);
    const float* gt_pf = sq.gates + ((long)ebc * T + t) * K + ejc;
    const float ig_pf = gt_pf[0], fg_pf = gt_pf[H], gg_pf = gt_pf[2 * H], og_pf = gt_pf[3 * H];
    const float cprev_pf = sq.cbuf[((long)ebc * (T + 1) + slot_prev) * H + ejc];
    const float cout_pf = sq.cbuf[((long)ebc * (T + 1) + slot_out) * H + ejc];
    const float dcs_pf = sq.dc_state[(long)ebc * H + ejc];
    const float dhx_pf = sq.dh_ext ? sq.dh_ext[(long)ebc * sq.dh_bs + (long)t * sq.dh_ts + ejc] : 0.f;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (!first) {
        const int nchunks = K / 16;  // H % 4 == 0 => 4H % 16 == 0: every chunk is whole
        const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;
        const int c_beg = wave * cpw, c_end = min(nchunks, c_beg + cpw);
        const int bA = min(b0 + fr, B - 1), jB = min(j0 + fr, H - 1);
        const float* arow = sq.dgates + ((long)bA * T + t_next) * K;
        const float* wrow = sq.w_hh_t + (long)jB * K;
        constexpr int CH = 8;
        for (int c0 = c_beg; c0 < c_end; c0 += CH) {
            f32x4 a[CH], w[CH];
            bool kok[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                kok[u] = c0 + u < c_end;
                const int kc = kok[u] ? (c0 + u) * 16 + q * 4 : 0;
                a[u] = ld4(arow + kc);
                w[u] = ld4(wrow + kc);
            }
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 a0 = kok[u] ? a[u] : z, a1 = kok[u + 1] ? a[u + 1] : z;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], w[u][i], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], w[u + 1][i], acc1, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
    __syncthreads();

    if (wave >= 4) return;
    const int r = wave;
    const int b = b0 + (lane >> 4) * 4 + r;
    const int j = j0 + (lane & 15);
    if (b < B && j < H) {
        float dh = 0.f;
#pragma unroll
        for (int wv = 0; wv < LSTM_NW; ++wv) dh += red[wv][r][lane];
        if (first && sq.dh_seed) dh += sq.dh_seed[(long)b * H + j];
        dh += dhx_pf;
        const float ig = ig_pf, fg = fg_pf, gg = gg_pf, og = og_pf;
        const float cprev = cprev_pf;
        const float tc = tanhf(cout_pf);
        const float dc = dcs_pf + dh * og * (1.f - tc * tc);
        float* dg = sq.dgates + ((long)b * T + t) * K;
        dg[j] = dc * gg * ig * (1.f - ig);
        dg[H + j] = dc * cprev * fg * (1.f - fg);
        dg[2 * H + j] = dc * ig * (1.f - gg * gg);
        dg[3 * H + j] = dh * tc * og * (1.f - og);
        sq.dc_state[(long)b * H + j] = dc * fg;
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent recurrence: ONE launch runs all T timesteps.
//
// Same decomposition as the step kernels (workgroup = 16 batch rows x 16 hidden units, 8 waves split K), but the
// workgroup's slice of W_hh (forward: 64 gate rows x H; backward: 16 rows of W_hhᵀ x 4H = 128 KB at H = 512) is loaded
// ONCE into registers (64 VGPRs per lane) and stays there across timesteps; c_t (forward) and dc_t (backward) of the
// workgroup's own 16x16 elements stay in registers too. What crosses workgroups per step is h_t (dgates_t): the
// workgroups that share a batch tile form a GROUP, and only a group synchronises (never the grid):
//   producer: h_t stored write-through (sc1), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE
//             lane adds 1 to the group's counter (agent-scope atomic);
//   consumer: ONE lane polls the counter (relaxed sc1 loads, bounded) until NJ*step arrivals, workgroup barrier,
//             then EVERY load of the handed-off rows is an sc1 buffer load (L2-served, never this CU's L1).
// This is the all-sc1 hand-off of MI355X_MICROARCH.md (visibility table, first row): placement-independent, no
// L1/L2 invalidate per step.
// Placement (speed only): the group-major work list is dealt so that every XCD residue class of blockIdx takes a
// contiguous range of it (xcd_contiguous) - at cfg2 each group of 32 workgroups IS one XCD's 32 blocks. sc1 stores drop
// their line from the L2, so the first sc1 load of an h_t row on an XCD fetches it over the fabric and the other
// workgroups of that XCD hit the L2: with a group spread over all 8 XCDs every row crossed the fabric 8 times.
// Counters: the caller zeroes the scratch ONCE; every launch leaves its counters at zero again (persist_checkout: the
// last workgroup of a hand-off domain to leave re-arms them), so no memset node sits in front of a launch.
// Residency: grid <= number of CUs (host check), 512-thread workgroups; every spin is bounded: a wait that runs out
// raises the STICKY word, which ends every other wait of this and of every later launch after at most 64 polls, so the
// grid always drains (results are invalid from then on, the host-visible word says so, and the caller clears it).
// ------------------------------------------------------------------------------------------------------------------
constexpr unsigned PERSIST_SPIN_LIMIT = 1u << 19;
constexpr int SYNC_STRIDE = 32;  // u32 words per counter: one 128-byte line each
constexpr int SYNC_EXIT = 16;    // word of a domain's first counter line that counts the workgroups that have left

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool group_wait(unsigned* cnt, unsigned target, unsigned* sticky) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (spins > PERSIST_SPIN_LIMIT ||
            ((spins & 63u) == 0u && __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // never cleared by the library
            return false;
        }
    }
    return true;
}

// Last act of every workgroup of a persistent launch (reached on the timeout path too): check out of the hand-off
// domain; the last of its `members` to leave sets the domain's counters and the exit word back to zero. Nobody polls
// them any more (every member has left its time loop), and this workgroup's own arrival adds have been performed
// (vmcnt(0)) before it checks out, so no add can land behind the reset.
__device__ __forceinline__ void persist_checkout(unsigned* c0, unsigned* c1, unsigned members) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* ex = c0 + SYNC_EXIT;
        const unsigned old = __hip_atomic_fetch_add(ex, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == members - 1u) {
            __hip_atomic_store(c0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c1) __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ex, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);  // aux 16 = sc1
    return __builtin_bit_cast(f32x4, v);
}

// the same with a wave-uniform (scalar) byte offset on top of the per-lane one
__device__ __forceinline__ f32x4 ld4_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off, int scalar_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, scalar_off, 16);  // aux 16 = sc1
    return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bf16 hand-off (operand mode 2, "X1"). At H = 1024 a step of the row-tiled recurrence is paced by L2 REQUESTS, not by
// latency or the matrix pipe: every workgroup re-reads the 64 rows of its batch tile (h_t: 256 KB, dgates_t: 1 MB) as
// 64-byte row segments of the fp32 history buffers (halving the loads of the backward kernel took it from 33 to 20 us
// per step; deeper prefetch changed nothing). So the producers ALSO write what they hand over as bf16 into a small
// scratch, blocked [parity][k / 8][B][8]: a consumer lane's 8 k of one row are ONE 16-byte load, the 16 rows of a
// wave's load are 256 contiguous bytes (whole 128-byte lines), a quarter of the bytes and an eighth of the requests.
// (A tile-major variant, [parity][batch tile][k group][16 rows][..] = 1 KB of contiguous memory per wave load, measured
// the same or slower on one box: two-layer backward 161 -> 168 us, the step 2.480 -> 2.486 ms.)
// Two parities (step & 1): a workgroup writes step s+1's values only after every member of its group has finished
// step s, i.e. has read step s-1's. The fp32 history buffers are still written (the GEMMs behind the recurrence read
// them); nothing in the kernel reads them back.
__device__ __forceinline__ long hx_off(int par, int KG, int kg, int B, int b) {  // element offset of (parity, k group, row)
    return ((long)(par * KG + kg) * B + b) * 8;
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.f * fast_sigmoid(2.f * x) - 1.f; }
// element offset of row b, k group kg (4 fp32) in the TILE-major fp32 scratch of the row-tiled kernels:
// [parity][batch tile of 16 rows][KG4][16 rows][4] - with B = 256 rows the k groups of the layout above lie 4 KB apart
// and a lane's successive loads 16 KB; here a wave's load is 1 KB of contiguous memory whatever B is
__device__ __forceinline__ long hx32_off(int par, int NBT, int KG4, int b, int kg) {
    return ((((long)par * NBT + (b >> 4)) * KG4 + kg) * 16 + (b & 15)) * 4;
}
__device__ __forceinline__ unsigned pack_bf16x2_lstm(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
// the value of lane + 1 (same row, next unit) joins this lane's: even lanes store the pair as one 4-byte sc1 word
__device__ __forceinline__ void hx_store_pair(unsigned short* hx, long elem_off, float v, int lane) {
    const float nb = __shfl_down(v, 1, 64);
    if (!(lane & 1)) st_sc1(reinterpret_cast<unsigned*>(hx + elem_off), pack_bf16x2_lstm(v, nb));
}

// ------------------------------------------------------------------------------------------------------------------
// X6: the recurrent products on the bf16 matrix cores at fp32 accuracy.
// A step's product is MFMA-paced at B = 64 (16 x 64 x H MACs per workgroup on the fp32 MFMA = 1.7 us at H = 512, a
// third of the step), and gfx950's bf16 MFMA has 16x the fp32 rate. Every fp32 operand is split EXACTLY into three bf16
// parts, x = p0 + p1 + p2 (8 + 8 + 8 significant bits: p0 = bf16(x), p1 = bf16(x - p0), p2 = x - p0 - p1; the two
// subtractions are exact in fp32), and a.b is accumulated in fp32 from the six partial products of weight >= 2^-16:
//   a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0      (dropped: a1b2, a2b1, a2b2 <= 2^-24 |a||b|, below fp32's own rounding)
// = 6 v_mfma_f32_16x16x32_bf16 (16 cycles each per SIMD) for 32 k against 8 v_mfma_f32_16x16x4_f32 (32 cycles each):
// 2.7x less MFMA time, same result to fp32 rounding (tests/test_kernels_gpu.py holds both forms to nn.LSTM in fp64 at
// 1e-5). The weight slices are split once per launch (they live in registers as three bf16x8 per 8 k instead of two
// f32x4: 1.5x the registers), h_t / dgates_t after each hand-off load. A lane's 8 k of a 32-k step are the quads of
// TWO neighbouring 16-k chunks (any assignment of k to the MFMA's k slots is legal as long as A and B agree).
// Which kernels use it: lstm_x6() / lstm_x6_bwd() below.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct Frag3 {
    bf16x8_t p0, p1, p2;
};
__device__ __forceinline__ Frag3 split3(const f32x4& u, const f32x4& v) {
    Frag3 f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? u[i] : v[i - 4];
        const __bf16 a = (__bf16)x;
        const float r1 = x - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        f.p0[i] = a;
        f.p1[i] = b;
        f.p2[i] = (__bf16)r2;
    }
    return f;
}
#define CGN_MFMA_BF16(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A_), (B_), (C_), 0, 0, 0)
// NG accumulators that share the A operand (the four gates): term-major, so consecutive MFMAs are independent
// NP = 3: the six products above (fp32 accuracy). NP = 2 ("X3", the optional bf16 mode's row-tiled recurrences at
// H > 512, cgn_lstm_set_operand_mode): two-term operands, a0b0 + a0b1 + a1b0, ~2^-16 relative per product - 250x finer
// than bf16 operands, half the registers of the three-term weight slice (the same 128 VGPRs as the fp32 slice at H = 1024).
// NP = 1 ("X1", operand mode 2): plain bf16 operands, one product (only the p0 parts are read; the others are dead code).
template <int NG, int NP = 3>
__device__ __forceinline__ void mfma_x6(const Frag3& a, const Frag3 (&b)[NG], f32x4 (&acc)[NG]) {
    if constexpr (NP == 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p0, b[g].p0, acc[g]);
        return;
    }
    if constexpr (NP == 3) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p0, b[g].p2, acc[g]);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p2, b[g].p0, acc[g]);
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p1, b[g].p1, acc[g]);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p0, b[g].p1, acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p1, b[g].p0, acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = CGN_MFMA_BF16(a.p0, b[g].p0, acc[g]);
}
// one accumulator chain (backward kernels: two chains, even / odd chunk pairs)
template <int NP = 3>
__device__ __forceinline__ f32x4 mfma_x6(const Frag3& a, const Frag3& b, f32x4 acc) {
    if constexpr (NP == 1) return CGN_MFMA_BF16(a.p0, b.p0, acc);
    if constexpr (NP == 3) {
        acc = CGN_MFMA_BF16(a.p0, b.p2, acc);
        acc = CGN_MFMA_BF16(a.p2, b.p0, acc);
        acc = CGN_MFMA_BF16(a.p1, b.p1, acc);
    }
    acc = CGN_MFMA_BF16(a.p0, b.p1, acc);
    acc = CGN_MFMA_BF16(a.p1, b.p0, acc);
    acc = CGN_MFMA_BF16(a.p0, b.p0, acc);
    return acc;
}

// CPW: chunks (of 16 k) per wave held in registers; H <= 128 * CPW. X6: see above.
template <int CPW, bool X6>
__global__ __launch_bounds__(LSTM_NT) void lstm_persist_fwd_kernel(const FwdParams p, unsigned* sync, unsigned* sticky) {
    static_assert(CPW % 2 == 0, "X6 pairs neighbouring chunks");
    __shared__ float red[LSTM_NW][16][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int NJ = H / 16, NB = (B + 15) / 16;
    const int wid = xcd_contiguous(gridDim.x, p.step);  // p.step: 1 = identity placement (A/B runs)
    const int seq = wid / (NB * NJ);
    const int bt = (wid % (NB * NJ)) / NJ, jt = wid % NJ;
    const FwdSeq& sq = p.s[seq];
    unsigned* cnt = sync + (seq * NB + bt) * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = H / 16;
    const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;  // <= CPW (host check)
    const int c_beg = wave * cpw;

    // this workgroup's W_hh rows (4 gates x 16 units), this wave's k range: registers for the whole sequence
    const float* wrow = sq.w_hh + (long)(j0 + fr) * H;
    const long gate_stride = (long)H * H;
    f32x4 w[X6 ? 1 : 4][X6 ? 1 : CPW];
    Frag3 w3[X6 ? CPW / 2 : 1][X6 ? 4 : 1];  // [chunk pair][gate]
    int koff[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        const bool ok = (u < cpw) && (c_beg + u < nchunks);
        const int kc = ok ? (c_beg + u) * 16 + q * 4 : 0;
        koff[u] = kc * 4;
    }
#pragma unroll
    for (int u = 0; u < CPW; u += 2) {
        f32x4 v[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = (u + h < cpw) && (c_beg + u + h < nchunks);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 x = ld4(wrow + g * gate_stride + koff[u + h] / 4);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v[h][g] = ok ? x : z;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (X6) {
                w3[u / 2][g] = split3(v[0][g], v[1][g]);
            } else {
                w[g][u] = v[0][g];
                w[g][u + 1] = v[1][g];
            }
        }
    }
    const int bA = min(b0 + fr, B - 1);
    const __amdgpu_buffer_rsrc_t hres =
        __builtin_amdgcn_make_buffer_rsrc((void*)sq.hbuf, 0, B * (T + 1) * H * 4, 0x00020000);
    // fp32 hand-off scratch (nullable), blocked [parity][k / 4][B][4]: the 16 rows of a wave's load are 256 contiguous
    // bytes (whole 128-byte lines) instead of sixteen 64-byte segments of the history buffer - the same bytes in half
    // the L2 requests (see hx_off). Same values, same k order: results do not change by a bit.
    float* hx32 = reinterpret_cast<float*>(sq.hx);
    const int KG4 = H / 4;
    const __amdgpu_buffer_rsrc_t xres =
        __builtin_amdgcn_make_buffer_rsrc(hx32 ? (void*)hx32 : (void*)sq.hbuf, 0, 2 * KG4 * B * 16, 0x00020000);

    // the cell-update element of this thread (waves 0..3): row (lane>>4)*4 + wave, unit lane&15 -- fixed for all t
    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1);
    const bool evalid = (wave < 4) && (eb < B);
    const int init_slot = sq.reverse ? T : 0;
    float c_reg = 0.f;
    if (sq.init_zero) {
        // zero initial state (nn.LSTM without (h0, c0)): nothing is read; the slots are written for the later readers
        // of the history buffers (dW_hh's h_{t-1} operand, the backward pass's c_{t-1})
        if (evalid) {
            sq.hbuf[((long)eb * (T + 1) + init_slot) * H + ej] = 0.f;
            sq.cbuf[((long)eb * (T + 1) + init_slot) * H + ej] = 0.f;
        }
    } else {
        c_reg = sq.cbuf[((long)ebc * (T + 1) + init_slot) * H + ej];
    }
    float pre_e[4] = {0.f, 0.f, 0.f, 0.f};
    if (sq.xg_extra) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_e[g] = sq.xg_extra[(long)ebc * 4 * H + g * H + ej];
    }

    for (int step = 0; step < T; ++step) {
        const int t = sq.reverse ? (T - 1 - step) : step;
        const int slot_prev = sq.reverse ? t + 1 : t;
        const int slot_out = sq.reverse ? t : t + 1;
        float pre_x[4];
        {
            const float* xg = sq.xg + ((long)ebc * T + t) * 4 * H + ej;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre_x[g] = xg[g * H];
        }
        if (step > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt, (unsigned)(NJ * step), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
        }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
        if (step > 0 || !sq.init_zero) {  // uniform: h_{-1} = 0 contributes nothing
            const int hoff = (bA * (T + 1) + slot_prev) * H * 4;
            f32x4 a[CPW];
            if (hx32 && step > 0) {  // (uniform; step 0 reads the given initial state from the history buffer)
                // k group kc / 4 of row bA: byte offset ((kc / 4) * B + bA) * 16 = koff * B + bA * 16
                const int xoff = (((step + 1) & 1) * KG4 * B + bA) * 16;
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[u] = ld4_sc1(xres, xoff + koff[u] * B);
            } else {
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[u] = ld4_sc1(hres, hoff + koff[u]);
            }
            if constexpr (X6) {
#pragma unroll
                for (int u = 0; u < CPW; u += 2) mfma_x6<4>(split3(a[u], a[u + 1]), w3[u / 2], acc);
            } else {
#pragma unroll
                for (int u = 0; u < CPW; ++u) {  // chunks beyond this wave's range hold zero weights: no branch
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w[g][u][i], acc[g], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][lane] = acc[g][r];
        __syncthreads();
        if (evalid) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < LSTM_NW; ++wv) v += red[wv][g * 4 + er][lane];
                pre[g] = v + pre_x[g] + pre_e[g];
            }
            const float ig = cgn_sigmoid(pre[0]);
            const float fg = cgn_sigmoid(pre[1]);
            const float gg = tanhf(pre[2]);
            const float og = cgn_sigmoid(pre[3]);
            const float cn = fg * c_reg + ig * gg;
            const float hn = og * tanhf(cn);
            c_reg = cn;
            float* go = sq.gates + ((long)eb * T + t) * 4 * H;
            go[ej] = ig;
            go[H + ej] = fg;
            go[2 * H + ej] = gg;
            go[3 * H + ej] = og;
            sq.cbuf[((long)eb * (T + 1) + slot_out) * H + ej] = cn;
            st_sc1(sq.hbuf + ((long)eb * (T + 1) + slot_out) * H + ej, hn);
            if (hx32) st_sc1(hx32 + ((long)((step & 1) * KG4 + (ej >> 2)) * B + eb) * 4 + (ej & 3), hn);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival is signalled
        __syncthreads();
        if (threadIdx.x == 0 && step + 1 < T)
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    persist_checkout(cnt, nullptr, (unsigned)NJ);
}

// CPW: chunks (of 16 k over K = 4H) per wave held in registers; H <= 32 * CPW.
template <int CPW, bool X6>
__global__ __launch_bounds__(LSTM_NT) void lstm_persist_bwd_kernel(const BwdParams p, unsigned* sync, unsigned* sticky) {
    static_assert(CPW % 4 == 0, "two accumulator chains over chunk pairs");
    __shared__ float red[LSTM_NW][4][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int K = 4 * H;
    const int NJ = H / 16, NB = (B + 15) / 16;
    const int wid = xcd_contiguous(gridDim.x, p.step);  // p.step: 1 = identity placement (A/B runs)
    const int seq = wid / (NB * NJ);
    const int bt = (wid % (NB * NJ)) / NJ, jt = wid % NJ;
    const BwdSeq& sq = p.s[seq];
    unsigned* cnt = sync + (seq * NB + bt) * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = K / 16;
    const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;  // <= CPW (host check)
    const int c_beg = wave * cpw;

    // B operand = W_hhᵀ rows (j0 + fr), this wave's k range. Loaded once, so the untransposed [4H,H] weight serves as
    // well (16 lanes = 64 contiguous bytes per k) and the caller needs no transposed copy.
    f32x4 w[X6 ? 1 : CPW];
    Frag3 w3[X6 ? CPW / 2 : 1];
    int koff[CPW];
    {
        f32x4 prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            const bool ok = (u < cpw) && (c_beg + u < nchunks);
            const int kc = ok ? (c_beg + u) * 16 + q * 4 : 0;
            koff[u] = kc * 4;
            f32x4 v;
            if (sq.w_hh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sq.w_hh[(long)(kc + i) * H + j0 + fr];
            } else {
                v = ld4(sq.w_hh_t + (long)(j0 + fr) * K + kc);
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v = ok ? v : z;
            if constexpr (X6) {
                if (u & 1) w3[u / 2] = split3(prev, v);
                prev = v;
            } else {
                w[u] = v;
            }
        }
    }
    const int bA = min(b0 + fr, B - 1);
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)sq.dgates, 0, B * T * K * 4, 0x00020000);
    float* hx32 = reinterpret_cast<float*>(sq.hx);  // fp32 hand-off scratch [parity][k / 4][B][4], see the forward kernel
    const int KG4 = K / 4;
    const __amdgpu_buffer_rsrc_t xres =
        __builtin_amdgcn_make_buffer_rsrc(hx32 ? (void*)hx32 : (void*)sq.dgates, 0, 2 * KG4 * B * 16, 0x00020000);

    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1);
    const bool evalid = (wave < 4) && (eb < B);
    float dc_reg = sq.dc_state[(long)ebc * H + ej];
    const float seed = sq.dh_seed ? sq.dh_seed[(long)ebc * H + ej] : 0.f;

    for (int step = 0; step < T; ++step) {
        const int t = sq.reverse ? step : (T - 1 - step);
        const int t_next = sq.reverse ? t - 1 : t + 1;
        const int slot_prev = sq.reverse ? t + 1 : t;
        const int slot_out = sq.reverse ? t : t + 1;
        // operands of the gate-gradient update (written by earlier kernels): requested before the wait
        const float* gt = sq.gates + ((long)ebc * T + t) * K + ej;
        const float ig = gt[0], fg = gt[H], gg = gt[2 * H], og = gt[3 * H];
        const float cprev = sq.cbuf[((long)ebc * (T + 1) + slot_prev) * H + ej];
        const float cout = sq.cbuf[((long)ebc * (T + 1) + slot_out) * H + ej];
        const float dhx = sq.dh_ext ? sq.dh_ext[(long)ebc * sq.dh_bs + (long)t * sq.dh_ts + ej] : 0.f;

        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt, (unsigned)(NJ * step), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
            const int aoff = (bA * T + t_next) * K * 4;
            f32x4 a[CPW];
            if (hx32) {  // (uniform)
                const int xoff = (((step + 1) & 1) * KG4 * B + bA) * 16;
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[u] = ld4_sc1(xres, xoff + koff[u] * B);
            } else {
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[u] = ld4_sc1(dres, aoff + koff[u]);
            }
            if constexpr (X6) {
#pragma unroll
                for (int u = 0; u < CPW; u += 4) {
                    acc0 = mfma_x6(split3(a[u], a[u + 1]), w3[u / 2], acc0);
                    acc1 = mfma_x6(split3(a[u + 2], a[u + 3]), w3[u / 2 + 1], acc1);
                }
            } else {
#pragma unroll
                for (int u = 0; u < CPW; u += 2) {  // chunks beyond this wave's range hold zero weights: no branch
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w[u][i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1][i], w[u + 1][i], acc1, 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
        __syncthreads();
        if (evalid) {
            float dh = 0.f;
#pragma unroll
            for (int wv = 0; wv < LSTM_NW; ++wv) dh += red[wv][er][lane];
            if (step == 0) dh += seed;
            dh += dhx;
            const float tc = tanhf(cout);
            const float dc = dc_reg + dh * og * (1.f - tc * tc);
            float* dg = sq.dgates + ((long)eb * T + t) * K;
            const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * cprev * fg * (1.f - fg);
            const float d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
            st_sc1(dg + ej, d0);
            st_sc1(dg + H + ej, d1);
            st_sc1(dg + 2 * H + ej, d2);
            st_sc1(dg + 3 * H + ej, d3);
            if (hx32) {  // k = gate * H + unit: the gates' k groups are H / 4 groups apart
                float* xo = hx32 + ((long)((step & 1) * KG4 + (ej >> 2)) * B + eb) * 4 + (ej & 3);
                const long gs = (long)(H / 4) * B * 4;
                st_sc1(xo, d0);
                st_sc1(xo + gs, d1);
                st_sc1(xo + 2 * gs, d2);
                st_sc1(xo + 3 * gs, d3);
            }
            dc_reg = dc * fg;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && step + 1 < T)
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (evalid) sq.dc_state[(long)eb * H + ej] = dc_reg;
    persist_checkout(cnt, nullptr, (unsigned)NJ);
}

// ------------------------------------------------------------------------------------------------------------------
// The persistent recurrence for shapes with more (batch tile x unit tile) pairs than CUs, or H > 512 (BASELINE
// configs[3]: H = 1024, B = 256): a workgroup owns RT batch tiles of 16 rows for its 16 hidden units, so
// nseq * ceil(B / 16RT) * H/16 workgroups fit the CUs, and its weight slice (H <= 1024: 128 VGPRs per lane) still lives
// in registers for the whole sequence. The row tiles of a step are worked off one after the other through the same
// 32 KB reduction buffer; hand-off, counters and timeouts are those of the kernels above (one group = the workgroups
// that share a 16RT-row batch tile). RT = 1, H <= 512 keeps the kernels above.
// ------------------------------------------------------------------------------------------------------------------
// (The fp32 scratch layout of the kernels above, [parity][k / 4][B][4], made the fp32 forms at H = 1024, B = 256 SLOWER,
// forward 21.4 -> 24.2 and backward 34.6 -> 44.0 us per step: with 256 rows per k group a lane's successive loads are
// 16 KB apart. The row-tiled kernels use the tile-major layout of hx32_off instead: backward 33.9 -> 28.5 us per step at
// H = 1024, 10.1 -> 9.5 at H = 512, B = 256; the forward launches do not move.)
// XM: 0 = fp32 MFMA, 3 = X6, 2 = X3 (two-term operands), 1 = X1 (plain bf16 operands; the bf16 mode at H > 512). X1's
// weight slice takes a quarter of the registers of the fp32 one (a bf16x8 per 8 k and gate instead of two f32x4), which
// buys what the recurrence at H = 1024 lacks: loads in flight. Forward: the h rows of ALL row tiles of a step are
// requested up to two row tiles ahead (three register buffers: one exposed L2 round trip per step instead of one per
// row tile); backward: the gradient rows travel 8 chunks at a time through three register buffers, the next two batches
// (of this or the next row tile) on their way while the current one is multiplied (one exposed round trip per step
// instead of sixteen).
template <int CPW, int RT, int XM>
__global__ __launch_bounds__(LSTM_NT) void lstm_persist_fwd_rt_kernel(const FwdParams p, unsigned* sync, unsigned* sticky) {
    constexpr bool X6 = XM != 0;
    constexpr int NPT = XM == 2 ? 2 : XM == 1 ? 1 : 3;  // bf16 terms per operand
    constexpr bool PRE = XM == 1;                       // all row tiles' loads in flight at once
    static_assert(XM >= 0 && XM <= 3, "operand mode");
    static_assert(CPW % 2 == 0 && (XM != 3 || CPW <= 4), "X6: chunk pairs; the three-term weight slice of H = 1024 does not fit the registers");
    __shared__ float red[LSTM_NW][16][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int NJ = H / 16, NB = (B + 16 * RT - 1) / (16 * RT);
    const int wid = xcd_contiguous(gridDim.x, p.step);  // p.step: 1 = identity placement (A/B runs)
    const int seq = wid / (NB * NJ);
    const int bt = (wid % (NB * NJ)) / NJ, jt = wid % NJ;
    const FwdSeq& sq = p.s[seq];
    unsigned* cnt = sync + (seq * NB + bt) * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16 * RT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = H / 16;
    const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;  // <= CPW (host check)
    const int c_beg = wave * cpw;

    const float* wrow = sq.w_hh + (long)(j0 + fr) * H;
    const long gate_stride = (long)H * H;
    f32x4 w[X6 ? 1 : 4][X6 ? 1 : CPW];
    Frag3 w3[X6 ? CPW / 2 : 1][X6 ? 4 : 1];  // [chunk pair][gate]
    // X1: a lane's 8 k of a chunk pair are CONSECUTIVE (k = 32 pair + 8 q ...: one 16-byte load of the bf16 hand-off
    // scratch); the other forms take quad q of both chunks. (host: X1 only with whole chunk pairs per wave, H % 256 == 0)
    int koff[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        const bool ok = (u < cpw) && (c_beg + u < nchunks);
        const int kc = !ok ? 0 : (XM == 1 ? (c_beg + (u & ~1)) * 16 + q * 8 + 4 * (u & 1) : (c_beg + u) * 16 + q * 4);
        koff[u] = kc * 4;
    }
#pragma unroll
    for (int u = 0; u < CPW; u += 2) {
        f32x4 v[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = (u + h < cpw) && (c_beg + u + h < nchunks);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 x = ld4(wrow + g * gate_stride + koff[u + h] / 4);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                v[h][g] = ok ? x : z;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (X6) {
                w3[u / 2][g] = split3(v[0][g], v[1][g]);
            } else {
                w[g][u] = v[0][g];
                w[g][u + 1] = v[1][g];
            }
        }
    }
    const __amdgpu_buffer_rsrc_t hres =
        __builtin_amdgcn_make_buffer_rsrc((void*)sq.hbuf, 0, B * (T + 1) * H * 4, 0x00020000);
    const int KG = H / 8;  // X1: k groups of the hand-off scratch [2][KG][B][8] bf16
    // the other forms: an fp32 scratch (nullable; the same bits with and without), TILE-major here,
    // [parity][batch tile of 16 rows][H / 4][16][4]: a wave's load is 1 KB of contiguous memory (hx32_off)
    float* hx32 = XM == 1 ? nullptr : reinterpret_cast<float*>(sq.hx);
    const int KG4 = H / 4, NBT = (B + 15) / 16;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(sq.hx ? (void*)sq.hx : (void*)sq.hbuf), 0, XM == 1 ? 2 * KG * B * 16 : 2 * NBT * KG4 * 256, 0x00020000);

    // cell-update elements of this thread (waves 0..3): row (lane>>4)*4 + wave of EVERY row tile, unit lane&15
    const int er = wave & 3;
    const int ej = j0 + (lane & 15);
    const int init_slot = sq.reverse ? T : 0;
    float c_reg[RT], pre_e[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int eb = b0 + rt * 16 + (lane >> 4) * 4 + er;
        const int ebc = min(eb, B - 1);
        const bool evalid = (wave < 4) && (eb < B);
        c_reg[rt] = 0.f;
        if (sq.init_zero) {
            if (evalid) {
                sq.hbuf[((long)eb * (T + 1) + init_slot) * H + ej] = 0.f;
                sq.cbuf[((long)eb * (T + 1) + init_slot) * H + ej] = 0.f;
            }
        } else {
            c_reg[rt] = sq.cbuf[((long)ebc * (T + 1) + init_slot) * H + ej];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_e[rt][g] = sq.xg_extra ? sq.xg_extra[(long)ebc * 4 * H + g * H + ej] : 0.f;
    }

    for (int step = 0; step < T; ++step) {
        const int t = sq.reverse ? (T - 1 - step) : step;
        const int slot_prev = sq.reverse ? t + 1 : t;
        const int slot_out = sq.reverse ? t : t + 1;
        float pre_x[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int ebc = min(b0 + rt * 16 + (lane >> 4) * 4 + er, B - 1);
            const float* xg = sq.xg + ((long)ebc * T + t) * 4 * H + ej;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre_x[rt][g] = xg[g * H];
        }
        if (step > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt, (unsigned)(NJ * step), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
        }
        constexpr int NBUF = PRE ? 3 : 1;  // PRE: the h rows of the next two row tiles are in flight under the current one
        f32x4 a[NBUF][PRE ? CPW / 2 : CPW];  // X1: a 16-byte load = the lane's 8 bf16 of one chunk pair
        auto issue = [&](int buf, int rt) {
            if constexpr (XM == 1) {
                const int xoff = (int)hx_off((step + 1) & 1, KG, c_beg * 2 + q, B, min(b0 + rt * 16 + fr, B - 1)) * 2;
#pragma unroll
                for (int u = 0; u < CPW / 2; ++u) {
                    // (pairs beyond this wave's range hold zero weights; their loads re-read pair 0: finite values)
                    const int pu = (2 * u < cpw && c_beg + 2 * u < nchunks) ? u : 0;
                    a[buf][u] = ld4_sc1(xres, xoff + pu * (4 * B * 16));
                }
            } else if (hx32 && step > 0) {  // (uniform; step 0 reads the given initial state from the history buffer)
                // k group kc / 4 sits kc / 4 * 256 = koff * 16 bytes into the row's tile
                const int xoff = (int)hx32_off((step + 1) & 1, NBT, KG4, min(b0 + rt * 16 + fr, B - 1), 0) * 4;
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[buf][u] = ld4_sc1(xres, xoff + koff[u] * 16);
            } else {
                const int hoff = (min(b0 + rt * 16 + fr, B - 1) * (T + 1) + slot_prev) * H * 4;
#pragma unroll
                for (int u = 0; u < CPW; ++u) a[buf][u] = ld4_sc1(hres, hoff + koff[u]);
            }
        };
        if constexpr (PRE) {
            if (step > 0 || !sq.init_zero) {
#pragma unroll
                for (int rt = 0; rt < NBUF - 1 && rt < RT; ++rt) issue(rt, rt);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
            if (step > 0 || !sq.init_zero) {
                constexpr int ar = PRE ? 1 : 0;
                if constexpr (PRE) {
                    if (rt + NBUF - 1 < RT) issue((rt + NBUF - 1) % NBUF, rt + NBUF - 1);
                } else {
                    issue(0, rt);
                }
                if constexpr (XM == 1) {
#pragma unroll
                    for (int u = 0; u < CPW; u += 2) {
                        Frag3 f;
                        f.p0 = __builtin_bit_cast(bf16x8_t, a[rt % NBUF][u / 2]);
                        mfma_x6<4, 1>(f, w3[u / 2], acc);
                    }
                } else if constexpr (X6) {
#pragma unroll
                    for (int u = 0; u < CPW; u += 2)
                        mfma_x6<4, NPT>(split3(a[ar * (rt % NBUF)][u], a[ar * (rt % NBUF)][u + 1]), w3[u / 2], acc);
                } else {
#pragma unroll
                    for (int u = 0; u < CPW; ++u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][u][i], w[g][u][i], acc[g], 0, 0, 0);
                        }
                    }
                }
            }
            if (rt > 0) __syncthreads();  // the previous row tile's sums have been read
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][lane] = acc[g][r];
            __syncthreads();
            const int eb = b0 + rt * 16 + (lane >> 4) * 4 + er;
            if (wave < 4 && eb < B) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v = 0.f;
#pragma unroll
                    for (int wv = 0; wv < LSTM_NW; ++wv) v += red[wv][g * 4 + er][lane];
                    pre[g] = v + pre_x[rt][g] + pre_e[rt][g];
                }
                // X1 (the bf16 mode): v_exp_f32 / v_rcp_f32 forms, ~1e-7 absolute - far below the operands' rounding;
                // expf, the division and tanhf are ~150 instructions per element and row tile on the step's chain
                float ig, fg, gg, og, cn, hn;
                if constexpr (XM == 1) {
                    ig = fast_sigmoid(pre[0]);
                    fg = fast_sigmoid(pre[1]);
                    gg = fast_tanh(pre[2]);
                    og = fast_sigmoid(pre[3]);
                    cn = fg * c_reg[rt] + ig * gg;
                    hn = og * fast_tanh(cn);
                } else {
                    ig = cgn_sigmoid(pre[0]);
                    fg = cgn_sigmoid(pre[1]);
                    gg = tanhf(pre[2]);
                    og = cgn_sigmoid(pre[3]);
                    cn = fg * c_reg[rt] + ig * gg;
                    hn = og * tanhf(cn);
                }
                c_reg[rt] = cn;
                float* go = sq.gates + ((long)eb * T + t) * 4 * H;
                go[ej] = ig;
                go[H + ej] = fg;
                go[2 * H + ej] = gg;
                go[3 * H + ej] = og;
                sq.cbuf[((long)eb * (T + 1) + slot_out) * H + ej] = cn;
                st_sc1(sq.hbuf + ((long)eb * (T + 1) + slot_out) * H + ej, hn);
                if constexpr (XM == 1) hx_store_pair(sq.hx, hx_off(step & 1, KG, ej >> 3, B, eb) + (ej & 7), hn, lane);
                else if (hx32) st_sc1(hx32 + hx32_off(step & 1, NBT, KG4, eb, ej >> 2) + (ej & 3), hn);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival is signalled
        __syncthreads();
        if (threadIdx.x == 0 && step + 1 < T)
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    persist_checkout(cnt, nullptr, (unsigned)NJ);
}

// CPW: chunks (of 16 k over K = 4H) per wave held in registers (H <= 32 * CPW); the gradient rows are loaded CB chunks at
// a time (32 VGPRs) so that the 128-VGPR weight slice of H = 1024 leaves room for them.
template <int CPW, int RT, int XM>  // XM: see lstm_persist_fwd_rt_kernel
__global__ __launch_bounds__(LSTM_NT) void lstm_persist_bwd_rt_kernel(const BwdParams p, unsigned* sync, unsigned* sticky) {
    constexpr bool X6 = XM != 0;
    constexpr int NPT = XM == 2 ? 2 : XM == 1 ? 1 : 3;
    constexpr bool PIPE = XM == 1;  // NBUF register buffers of CB chunks: the next NBUF-1 batches in flight under the current one
    constexpr int CB = CPW < 8 ? CPW : 8;
    constexpr int NBUF = PIPE ? 3 : 1;
    static_assert(CPW % CB == 0 && CB % 4 == 0, "chunk batches; two accumulator chains over chunk pairs");
    static_assert(XM == 0 || XM == 1 || XM == 2 || (XM == 3 && CPW <= 16), "the three-term weight slice of H = 1024 does not fit the registers");
    __shared__ float red[LSTM_NW][4][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int K = 4 * H;
    const int NJ = H / 16, NB = (B + 16 * RT - 1) / (16 * RT);
    const int wid = xcd_contiguous(gridDim.x, p.step);  // p.step: 1 = identity placement (A/B runs)
    const int seq = wid / (NB * NJ);
    const int bt = (wid % (NB * NJ)) / NJ, jt = wid % NJ;
    const BwdSeq& sq = p.s[seq];
    unsigned* cnt = sync + (seq * NB + bt) * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16 * RT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = K / 16;
    const int cpw = (nchunks + LSTM_NW - 1) / LSTM_NW;  // <= CPW (host check)
    const int c_beg = wave * cpw;

    f32x4 w[X6 ? 1 : CPW];
    Frag3 w3[X6 ? CPW / 2 : 1];
    int kbase;  // byte offset of this wave's first chunk + the lane's quad
    {
        kbase = (c_beg * 16 + q * 4) * 4;
        f32x4 prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            const bool ok = (u < cpw) && (c_beg + u < nchunks);
            // (X1: the lane's 8 k of a chunk pair are consecutive, as in the forward kernel)
            const int kc = !ok ? 0 : (XM == 1 ? (c_beg + (u & ~1)) * 16 + q * 8 + 4 * (u & 1) : (c_beg + u) * 16 + q * 4);
            f32x4 v;
            if (sq.w_hh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = sq.w_hh[(long)(kc + i) * H + j0 + fr];
            } else {
                v = ld4(sq.w_hh_t + (long)(j0 + fr) * K + kc);
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v = ok ? v : z;
            if constexpr (X6) {
                if (u & 1) w3[u / 2] = split3(prev, v);
                prev = v;
            } else {
                w[u] = v;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t dres = __builtin_amdgcn_make_buffer_rsrc((void*)sq.dgates, 0, B * T * K * 4, 0x00020000);
    const int KG = K / 8;  // X1: k groups of the hand-off scratch [2][KG][B][8] bf16
    float* hx32 = XM == 1 ? nullptr : reinterpret_cast<float*>(sq.hx);  // the other forms: tile-major fp32 scratch (hx32_off)
    const int KG4 = K / 4, NBT = (B + 15) / 16;
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(sq.hx ? (void*)sq.hx : (void*)sq.dgates), 0, XM == 1 ? 2 * KG * B * 16 : 2 * NBT * KG4 * 256, 0x00020000);

    const int er = wave & 3;
    const int ej = j0 + (lane & 15);
    float dc_reg[RT], seed[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int ebc = min(b0 + rt * 16 + (lane >> 4) * 4 + er, B - 1);
        dc_reg[rt] = sq.dc_state[(long)ebc * H + ej];
        seed[rt] = sq.dh_seed ? sq.dh_seed[(long)ebc * H + ej] : 0.f;
    }

    for (int step = 0; step < T; ++step) {
        const int t = sq.reverse ? step : (T - 1 - step);
        const int t_next = sq.reverse ? t - 1 : t + 1;
        const int slot_prev = sq.reverse ? t + 1 : t;
        const int slot_out = sq.reverse ? t : t + 1;
        if (step > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt, (unsigned)(NJ * step), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
        }
        constexpr int NBR = CPW / CB;  // batches per row tile
        f32x4 ab[NBUF][PIPE ? CB / 2 : CB];  // X1: a 16-byte load = the lane's 8 bf16 of one chunk pair
        auto issue = [&](int buf, int rt, int bi) {  // the loads of batch bi of row tile rt into register buffer buf
            if constexpr (XM == 1) {
                const int xoff = (int)hx_off((step + 1) & 1, KG, c_beg * 2 + q, B, min(b0 + rt * 16 + fr, B - 1)) * 2;
#pragma unroll
                for (int u = 0; u < CB / 2; ++u) {
                    const int c = bi * CB + 2 * u;  // first chunk of the pair; pairs beyond the wave's range re-read pair 0
                    const int pu = (c < cpw && c_beg + c < nchunks) ? c / 2 : 0;
                    ab[buf][u] = ld4_sc1(xres, xoff + pu * (4 * B * 16));
                }
                return;
            }
            const int aoff = (min(b0 + rt * 16 + fr, B - 1) * T + t_next) * K * 4 + kbase;
            // tile-major scratch: k group (c_beg + cu) * 4 + q of the row, 256 bytes per k group
            const int xoff = (int)hx32_off((step + 1) & 1, NBT, KG4, min(b0 + rt * 16 + fr, B - 1), c_beg * 4 + q) * 4;
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                // chunks beyond this wave's range hold zero weights; their (clamped) loads re-read chunk 0
                const int cu = (bi * CB + u < cpw && c_beg + bi * CB + u < nchunks) ? (bi * CB + u) : 0;
                ab[buf][u] = hx32 ? ld4_sc1(xres, xoff + cu * 1024) : ld4_sc1(dres, aoff + cu * 64);
            }
        };
        if constexpr (PIPE) {
            if (step > 0) {
#pragma unroll
                for (int n = 0; n < NBUF - 1 && n < RT * NBR; ++n) issue(n, n / NBR, n % NBR);
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int eb = b0 + rt * 16 + (lane >> 4) * 4 + er;
            const int ebc = min(eb, B - 1);
            // operands of the gate-gradient update (written by earlier kernels)
            const float* gt = sq.gates + ((long)ebc * T + t) * K + ej;
            const float ig = gt[0], fg = gt[H], gg = gt[2 * H], og = gt[3 * H];
            const float cprev = sq.cbuf[((long)ebc * (T + 1) + slot_prev) * H + ej];
            const float cout = sq.cbuf[((long)ebc * (T + 1) + slot_out) * H + ej];
            const float dhx = sq.dh_ext ? sq.dh_ext[(long)ebc * sq.dh_bs + (long)t * sq.dh_ts + ej] : 0.f;

            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (step > 0) {
#pragma unroll
                for (int bi = 0; bi < NBR; ++bi) {
                    const int c0 = bi * CB;
                    const int n = rt * NBR + bi;  // batch number within the step
                    const int cur = PIPE ? (n % NBUF) : 0;
                    if constexpr (PIPE) {
                        constexpr int D = NBUF - 1;
                        if (n + D < RT * NBR) issue((n + D) % NBUF, (n + D) / NBR, (n + D) % NBR);
                    } else {
                        issue(0, rt, bi);
                    }
                    auto& a = ab[cur];
                    if constexpr (XM == 1) {
#pragma unroll
                        for (int u = 0; u < CB; u += 4) {
                            Frag3 f0, f1;
                            f0.p0 = __builtin_bit_cast(bf16x8_t, a[u / 2]);
                            f1.p0 = __builtin_bit_cast(bf16x8_t, a[u / 2 + 1]);
                            acc0 = mfma_x6<1>(f0, w3[(c0 + u) / 2], acc0);
                            acc1 = mfma_x6<1>(f1, w3[(c0 + u) / 2 + 1], acc1);
                        }
                    } else if constexpr (X6) {
#pragma unroll
                        for (int u = 0; u < CB; u += 4) {
                            acc0 = mfma_x6<NPT>(split3(a[u], a[u + 1]), w3[(c0 + u) / 2], acc0);
                            acc1 = mfma_x6<NPT>(split3(a[u + 2], a[u + 3]), w3[(c0 + u) / 2 + 1], acc1);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < CB; u += 2) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w[c0 + u][i], acc0, 0, 0, 0);
                                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1][i], w[c0 + u + 1][i], acc1, 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (rt > 0) __syncthreads();  // the previous row tile's sums have been read
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
            __syncthreads();
            if (wave < 4 && eb < B) {
                float dh = 0.f;
#pragma unroll
                for (int wv = 0; wv < LSTM_NW; ++wv) dh += red[wv][er][lane];
                if (step == 0) dh += seed[rt];
                dh += dhx;
                const float tc = XM == 1 ? fast_tanh(cout) : tanhf(cout);  // (X1: as in the forward kernel)
                const float dc = dc_reg[rt] + dh * og * (1.f - tc * tc);
                float* dg = sq.dgates + ((long)eb * T + t) * K;
                const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * cprev * fg * (1.f - fg);
                const float d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
                st_sc1(dg + ej, d0);
                st_sc1(dg + H + ej, d1);
                st_sc1(dg + 2 * H + ej, d2);
                st_sc1(dg + 3 * H + ej, d3);
                if constexpr (XM == 1) {
                    const long xo = hx_off(step & 1, KG, ej >> 3, B, eb) + (ej & 7);
                    const long gs = (long)(H / 8) * B * 8;  // elements between the gates' k groups (k = g H + unit)
                    hx_store_pair(sq.hx, xo, d0, lane);
                    hx_store_pair(sq.hx, xo + gs, d1, lane);
                    hx_store_pair(sq.hx, xo + 2 * gs, d2, lane);
                    hx_store_pair(sq.hx, xo + 3 * gs, d3, lane);
                } else if (hx32) {
                    float* xo = hx32 + hx32_off(step & 1, NBT, KG4, eb, ej >> 2) + (ej & 3);
                    const long gs = (long)H * 16;  // H / 4 k groups of 16 rows x 4 values between the gates
                    st_sc1(xo, d0);
                    st_sc1(xo + gs, d1);
                    st_sc1(xo + 2 * gs, d2);
                    st_sc1(xo + 3 * gs, d3);
                }
                dc_reg[rt] = dc * fg;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && step + 1 < T)
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int eb = b0 + rt * 16 + (lane >> 4) * 4 + er;
        if (wave < 4 && eb < B) sq.dc_state[(long)eb * H + ej] = dc_reg[rt];
    }
    persist_checkout(cnt, nullptr, (unsigned)NJ);
}

// ------------------------------------------------------------------------------------------------------------------
// Two stacked layers in ONE persistent launch (the generator's nn.LSTM(num_layers=2), model/generator.py:17-20,58).
//
// Layer 1 of timestep t needs h0_t, layer 0 of timestep t+1 needs h0_t too: the two layers form a wavefront, layer 1
// one timestep behind layer 0. Run back to back (round 1) each layer occupied half the chip for a whole sequence pass
// and a time-batched x1·W_ih1ᵀ GEMM sat between them; here both layers' workgroups are resident together (2·NB·NJ <=
// CUs) and layer 1's input projection moves INTO the recurrence: its workgroups multiply [h0_t ; h1_{t-1}] (K = 2H) with
// their slice of [W_ih1 | W_hh1], resident in registers for the whole sequence like W_hh0 is for layer 0.
// Hand-off: the same all-sc1 form as above, with one arrival counter per (layer, batch tile); a layer-1 workgroup waits
// for its own group's step t-1 AND for layer 0's step t. Layer 0 never waits for layer 1 (the history buffers hold
// every timestep), so it simply runs ahead.
// Backward is the mirror image: the top layer's gate gradients of timestep t feed the bottom layer's dh0_t =
// W_ih1ᵀ·dg1_t + W_hh0ᵀ·dg0_{t+1} (K = 8H, the dH0 GEMM of round 1 folded in), bottom one step behind top.
// Zero initial state, forward direction, equal hidden sizes; final-state gradients are zero (the step discards the
// generator's final state, model/cgn.py:1082).
// ------------------------------------------------------------------------------------------------------------------
struct Fwd2Params {
    const float* xg0;
    const float* xg0_extra;
    const float* w_hh0;
    const float* w_ih1;
    const float* w_hh1;
    const float* b_ih1;
    const float* b_hh1;
    float* hbuf[2];
    float* cbuf[2];
    float* gates[2];
    int B, T, H, flat;
};

// CPH: chunks (of 16 k) per wave and per operand half held in registers: H <= 128 * CPH. Every wave owns CPH chunks
// of the h0_t half (layer 1 only; multiplied with W_ih1) and CPH chunks of the h_{t-1} half (W_hh), so that the two
// halves can run as two PHASES of a timestep: layer 0 runs ahead of layer 1, its h0_t has long arrived, and the
// W_ih1·h0_t half is done while this group's own step t-1 is still being handed over; only the W_hh1·h1_{t-1} half
// (the same work as a single-layer step) sits behind the own-group wait.
template <int CPH, bool X6>
__global__ __launch_bounds__(LSTM_NT) void lstm2_persist_fwd_kernel(const Fwd2Params p, unsigned* sync, unsigned* sticky) {
    static_assert(CPH % 2 == 0, "X6 pairs neighbouring chunks");
    __shared__ float red[LSTM_NW][16][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int NJ = H / 16, NB = (B + 15) / 16;
    // layer-0 and layer-1 groups of one batch tile on NEIGHBOURING list ranges: (bt, layer, jt)
    const int wid = xcd_contiguous(gridDim.x, p.flat);
    const int layer = (wid / NJ) % 2;
    const int bt = wid / (2 * NJ), jt = wid % NJ;
    unsigned* cnt_own = sync + (layer * NB + bt) * SYNC_STRIDE;
    unsigned* cnt_below = sync + bt * SYNC_STRIDE;  // layer 0's group of the same batch tile
    const int j0 = jt * 16, b0 = bt * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: chunk ranges are wave-uniform
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = H / 16;                            // per half
    const int cph = (nchunks + LSTM_NW - 1) / LSTM_NW;     // <= CPH (host check)
    const int c_beg = wave * cph;
    // chunk u of this wave covers columns colb(u) .. colb(u)+15 of the half's weight / operand rows; wave-uniform, so
    // the chunk's byte offset travels as the SCALAR offset of the buffer loads (no per-chunk VGPRs)
    auto chunk_ok = [&](int u) { return (u < cph) && (c_beg + u < nchunks); };
    auto colb = [&](int u) { return chunk_ok(u) ? (c_beg + u) * 16 : 0; };

    // [gate][chunk]: W_ih1 (layer 1 only) and W_hh slices, resident for all T steps; X6: [chunk pair][gate] in bf16 parts
    f32x4 w_lo[X6 ? 1 : 4][X6 ? 1 : CPH], w_hi[X6 ? 1 : 4][X6 ? 1 : CPH];
    Frag3 w3_lo[X6 ? CPH / 2 : 1][X6 ? 4 : 1], w3_hi[X6 ? CPH / 2 : 1][X6 ? 4 : 1];
    {
        const float* w_own = layer ? p.w_hh1 : p.w_hh0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < CPH; u += 2) {
            f32x4 vh[2][4], vl[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = chunk_ok(u + h);
                const long off = (long)(j0 + fr) * H + colb(u + h) + q * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 xh = ld4(w_own + (long)g * H * H + off);
                    vh[h][g] = ok ? xh : z;
                    vl[h][g] = z;
                    if (layer) {
                        const f32x4 xl = ld4(p.w_ih1 + (long)g * H * H + off);
                        vl[h][g] = ok ? xl : z;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (X6) {
                    w3_hi[u / 2][g] = split3(vh[0][g], vh[1][g]);
                    w3_lo[u / 2][g] = split3(vl[0][g], vl[1][g]);
                } else {
                    w_hi[g][u] = vh[0][g];
                    w_hi[g][u + 1] = vh[1][g];
                    w_lo[g][u] = vl[0][g];
                    w_lo[g][u + 1] = vl[1][g];
                }
            }
        }
    }
    const int bA = min(b0 + fr, B - 1);
    const int hbytes = B * (T + 1) * H * 4;
    const __amdgpu_buffer_rsrc_t hres_own = __builtin_amdgcn_make_buffer_rsrc((void*)p.hbuf[layer], 0, hbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t hres_below = __builtin_amdgcn_make_buffer_rsrc((void*)p.hbuf[0], 0, hbytes, 0x00020000);

    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1);
    const bool evalid = (wave < 4) && (eb < B);
    float* hb = p.hbuf[layer];
    float* cb = p.cbuf[layer];
    float* gb = p.gates[layer];
    if (evalid) {  // zero initial state: written for the later readers of the history buffers (dW_hh, backward)
        hb[((long)eb * (T + 1)) * H + ej] = 0.f;
        cb[((long)eb * (T + 1)) * H + ej] = 0.f;
    }
    float c_reg = 0.f;
    float pre_c[4] = {0.f, 0.f, 0.f, 0.f};  // time-independent part of the pre-activations
    if (layer == 0) {
        if (p.xg0_extra) {
#pragma unroll
            for (int g = 0; g < 4; ++g) pre_c[g] = p.xg0_extra[(long)ebc * 4 * H + g * H + ej];
        }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) pre_c[g] = p.b_ih1[g * H + ej] + p.b_hh1[g * H + ej];
    }

    bool alive = true;
    for (int t = 0; t < T && alive; ++t) {
        float pre_x[4] = {0.f, 0.f, 0.f, 0.f};
        if (layer == 0) {
            const float* xg = p.xg0 + ((long)ebc * T + t) * 4 * H + ej;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre_x[g] = xg[g * H];
        }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
        // ---- phase A (layer 1): W_ih1 · h0_t; h0_t sits in slot t+1 of layer 0's history
        if (layer == 1) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt_below, (unsigned)(NJ * (t + 1)), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
            const int off = (bA * (T + 1) + t + 1) * H * 4 + q * 16;
            f32x4 a[CPH];
#pragma unroll
            for (int u = 0; u < CPH; ++u) a[u] = ld4_sc1(hres_below, off, colb(u) * 4);
            if constexpr (X6) {
#pragma unroll
                for (int u = 0; u < CPH; u += 2) mfma_x6<4>(split3(a[u], a[u + 1]), w3_lo[u / 2], acc);
            } else {
#pragma unroll
                for (int u = 0; u < CPH; ++u) {  // chunks beyond this wave's range hold zero weights: no branch
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w_lo[g][u][i], acc[g], 0, 0, 0);
                    }
                }
            }
            __syncthreads();  // s_ok is rewritten below
        }
        // ---- phase B: W_hh · h_{t-1} of this layer (slot t of its own history); h_{-1} = 0 contributes nothing
        if (t > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt_own, (unsigned)(NJ * t), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
            const int off = (bA * (T + 1) + t) * H * 4 + q * 16;
            f32x4 a[CPH];
#pragma unroll
            for (int u = 0; u < CPH; ++u) a[u] = ld4_sc1(hres_own, off, colb(u) * 4);
            if constexpr (X6) {
#pragma unroll
                for (int u = 0; u < CPH; u += 2) mfma_x6<4>(split3(a[u], a[u + 1]), w3_hi[u / 2], acc);
            } else {
#pragma unroll
                for (int u = 0; u < CPH; ++u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w_hi[g][u][i], acc[g], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][g * 4 + r][lane] = acc[g][r];
        __syncthreads();
        if (evalid) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < LSTM_NW; ++wv) v += red[wv][g * 4 + er][lane];
                pre[g] = v + pre_x[g] + pre_c[g];
            }
            const float ig = cgn_sigmoid(pre[0]);
            const float fg = cgn_sigmoid(pre[1]);
            const float gg = tanhf(pre[2]);
            const float og = cgn_sigmoid(pre[3]);
            const float cn = fg * c_reg + ig * gg;
            const float hn = og * tanhf(cn);
            c_reg = cn;
            float* go = gb + ((long)eb * T + t) * 4 * H;
            go[ej] = ig;
            go[H + ej] = fg;
            go[2 * H + ej] = gg;
            go[3 * H + ej] = og;
            cb[((long)eb * (T + 1) + t + 1) * H + ej] = cn;
            st_sc1(hb + ((long)eb * (T + 1) + t + 1) * H + ej, hn);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the arrival is signalled
        __syncthreads();
        // layer 0 signals its last step too: layer 1's step T-1 waits for it
        if (threadIdx.x == 0 && (layer == 0 || t + 1 < T))
            __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // hand-off domain = both layers' groups of this batch tile (layer 1 polls layer 0's counter too)
    persist_checkout(sync + bt * SYNC_STRIDE, sync + (NB + bt) * SYNC_STRIDE, 2u * (unsigned)NJ);
}

struct Bwd2Params {
    const float* w_hh0;
    const float* w_ih1;
    const float* w_hh1;
    const float* cbuf[2];
    const float* gates[2];
    const float* dh_ext1;
    float* dgates[2];
    long dh_bs, dh_ts;
    int B, T, H, flat;
    // nullable: blocked fp32 copies [T][4H / 4][B][4] of both layers' gate gradients, what the workgroups read from each
    // other (whole cache lines per load, see hx_off). Full histories, not two parities: the top layer runs ahead of the
    // bottom one by any number of steps.
    float* hx[2];
};

// role 0 = the TOP layer (layer 1: dh1_t = dH_ext[t] + W_hh1ᵀ·dg1_{t+1}), role 1 = the BOTTOM layer (layer 0:
// dh0_t = W_ih1ᵀ·dg1_t + W_hh0ᵀ·dg0_{t+1}). CPH: chunks (of 16 k) per wave and per operand half held in registers:
// 4H <= 128 * CPH. Two phases per timestep like the forward kernel: the W_ih1ᵀ·dg1_t half (the top layer runs ahead)
// is done before this group's own previous step is waited for.
template <int CPH, bool X6>
__global__ __launch_bounds__(LSTM_NT) void lstm2_persist_bwd_kernel(const Bwd2Params p, unsigned* sync, unsigned* sticky) {
    constexpr int CB = CPH < 8 ? CPH : 8;  // X6: the gradient rows are loaded CB chunks at a time (registers)
    static_assert(CPH % CB == 0 && CB % 4 == 0, "chunk batches; two accumulator chains over chunk pairs");
    __shared__ float red[LSTM_NW][4][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int K4 = 4 * H;
    const int NJ = H / 16, NB = (B + 15) / 16;
    const int wid = xcd_contiguous(gridDim.x, p.flat);  // list order (bt, role, jt), as in the forward kernel
    const int role = (wid / NJ) % 2;  // 0: layer 1 (top), 1: layer 0 (bottom)
    const int layer = 1 - role;
    const int bt = wid / (2 * NJ), jt = wid % NJ;
    unsigned* cnt_own = sync + (role * NB + bt) * SYNC_STRIDE;
    unsigned* cnt_top = sync + bt * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: chunk ranges are wave-uniform
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = K4 / 16;                           // per half
    const int cph = (nchunks + LSTM_NW - 1) / LSTM_NW;     // <= CPH (host check)
    const int c_beg = wave * cph;
    auto chunk_ok = [&](int u) { return (u < cph) && (c_beg + u < nchunks); };
    auto colb = [&](int u) { return chunk_ok(u) ? (c_beg + u) * 16 : 0; };  // scalar

    // B operands = rows (j0 + fr) of W_ih1ᵀ (bottom layer only) and of this layer's W_hhᵀ, read once from the
    // untransposed [4H,H] weights (16 lanes = 64 contiguous bytes per k)
    f32x4 w_lo[X6 ? 1 : CPH], w_hi[X6 ? 1 : CPH];
    Frag3 w3_lo[X6 ? CPH / 2 : 1], w3_hi[X6 ? CPH / 2 : 1];
    {
        const float* w_own = role ? p.w_hh0 : p.w_hh1;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 ph = z, pl = z;
#pragma unroll
        for (int u = 0; u < CPH; ++u) {
            const bool ok = chunk_ok(u);
            const int col = colb(u) + q * 4;
            f32x4 vh, vl = z;
#pragma unroll
            for (int i = 0; i < 4; ++i) vh[i] = w_own[(long)(col + i) * H + j0 + fr];
            if (role) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vl[i] = p.w_ih1[(long)(col + i) * H + j0 + fr];
            }
            vh = ok ? vh : z;
            vl = ok ? vl : z;
            if constexpr (X6) {
                if (u & 1) {
                    w3_hi[u / 2] = split3(ph, vh);
                    w3_lo[u / 2] = split3(pl, vl);
                }
                ph = vh;
                pl = vl;
            } else {
                w_hi[u] = vh;
                w_lo[u] = vl;
            }
        }
    }
    const int bA = min(b0 + fr, B - 1);
    const int dbytes = B * T * K4 * 4;
    const bool blk = p.hx[0] != nullptr;  // blocked hand-off copies: the same values, same k order (bit-identical results)
    const __amdgpu_buffer_rsrc_t dres_own =
        __builtin_amdgcn_make_buffer_rsrc(blk ? (void*)p.hx[layer] : (void*)p.dgates[layer], 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dres_top =
        __builtin_amdgcn_make_buffer_rsrc(blk ? (void*)p.hx[1] : (void*)p.dgates[1], 0, dbytes, 0x00020000);
    const int KG4 = K4 / 4;
    // byte offset of (timestep, row bA, quad q) and the scalar step per column: history [B,T,4H] | blocked [T][4H/4][B][4]
    auto lane_off = [&](int tt) { return blk ? ((tt * KG4 + q) * B + bA) * 16 : (bA * T + tt) * K4 * 4 + q * 16; };
    const int col_mul = blk ? 4 * B : 4;

    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1);
    const bool evalid = (wave < 4) && (eb < B);
    const float* gates = p.gates[layer];
    const float* cbuf = p.cbuf[layer];
    float* dgo = p.dgates[layer];
    float dc_reg = 0.f;

    for (int step = 0; step < T; ++step) {
        const int t = T - 1 - step;
        // operands of the gate-gradient update (written by earlier kernels): requested before the waits
        const float* gt = gates + ((long)ebc * T + t) * K4 + ej;
        const float ig = gt[0], fg = gt[H], gg = gt[2 * H], og = gt[3 * H];
        const float cprev = cbuf[((long)ebc * (T + 1) + t) * H + ej];
        const float cout = cbuf[((long)ebc * (T + 1) + t + 1) * H + ej];
        const float dhx = role == 0 && p.dh_ext1 ? p.dh_ext1[(long)ebc * p.dh_bs + (long)t * p.dh_ts + ej] : 0.f;

        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // ---- phase A (bottom layer): W_ih1ᵀ · dg1_t, the top layer's gate gradients of THIS timestep
        if (role == 1) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt_top, (unsigned)(NJ * (step + 1)), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
            const int off = lane_off(t);
            if constexpr (X6) {
#pragma unroll
                for (int c0 = 0; c0 < CPH; c0 += CB) {
                    f32x4 a[CB];
#pragma unroll
                    for (int u = 0; u < CB; ++u) a[u] = ld4_sc1(dres_top, off, colb(c0 + u) * col_mul);
#pragma unroll
                    for (int u = 0; u < CB; u += 4) {
                        acc0 = mfma_x6(split3(a[u], a[u + 1]), w3_lo[(c0 + u) / 2], acc0);
                        acc1 = mfma_x6(split3(a[u + 2], a[u + 3]), w3_lo[(c0 + u) / 2 + 1], acc1);
                    }
                }
            } else {
                f32x4 a[CPH];
#pragma unroll
                for (int u = 0; u < CPH; ++u) a[u] = ld4_sc1(dres_top, off, colb(u) * col_mul);
#pragma unroll
                for (int u = 0; u < CPH; u += 2) {  // chunks beyond this wave's range hold zero weights: no branch
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w_lo[u][i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1][i], w_lo[u + 1][i], acc1, 0, 0, 0);
                    }
                }
            }
            __syncthreads();  // s_ok is rewritten below
        }
        // ---- phase B: W_hhᵀ · dg_{t+1} of this layer; the first step has none
        if (step > 0) {
            if (threadIdx.x == 0) s_ok = group_wait(cnt_own, (unsigned)(NJ * step), sticky) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;  // workgroup-uniform
            const int off = lane_off(t + 1);
            if constexpr (X6) {
#pragma unroll
                for (int c0 = 0; c0 < CPH; c0 += CB) {
                    f32x4 a[CB];
#pragma unroll
                    for (int u = 0; u < CB; ++u) a[u] = ld4_sc1(dres_own, off, colb(c0 + u) * col_mul);
#pragma unroll
                    for (int u = 0; u < CB; u += 4) {
                        acc0 = mfma_x6(split3(a[u], a[u + 1]), w3_hi[(c0 + u) / 2], acc0);
                        acc1 = mfma_x6(split3(a[u + 2], a[u + 3]), w3_hi[(c0 + u) / 2 + 1], acc1);
                    }
                }
            } else {
                f32x4 a[CPH];
#pragma unroll
                for (int u = 0; u < CPH; ++u) a[u] = ld4_sc1(dres_own, off, colb(u) * col_mul);
#pragma unroll
                for (int u = 0; u < CPH; u += 2) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w_hi[u][i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1][i], w_hi[u + 1][i], acc1, 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc0[r] + acc1[r];
        __syncthreads();
        if (evalid) {
            float dh = 0.f;
#pragma unroll
            for (int wv = 0; wv < LSTM_NW; ++wv) dh += red[wv][er][lane];
            dh += dhx;
            const float tc = tanhf(cout);
            const float dc = dc_reg + dh * og * (1.f - tc * tc);
            float* dg = dgo + ((long)eb * T + t) * K4;
            const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * cprev * fg * (1.f - fg);
            const float d2 = dc * ig * (1.f - gg * gg), d3 = dh * tc * og * (1.f - og);
            st_sc1(dg + ej, d0);
            st_sc1(dg + H + ej, d1);
            st_sc1(dg + 2 * H + ej, d2);
            st_sc1(dg + 3 * H + ej, d3);
            if (blk) {  // k = gate * H + unit: the gates' k groups are H / 4 groups apart
                float* xo = p.hx[layer] + ((long)(t * KG4 + (ej >> 2)) * B + eb) * 4 + (ej & 3);
                const long gs = (long)(H / 4) * B * 4;
                st_sc1(xo, d0);
                st_sc1(xo + gs, d1);
                st_sc1(xo + 2 * gs, d2);
                st_sc1(xo + 3 * gs, d3);
            }
            dc_reg = dc * fg;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the top layer signals its last step too: the bottom layer's last step waits for it
        if (threadIdx.x == 0 && (role == 0 || step + 1 < T))
            __hip_atomic_fetch_add(cnt_own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    persist_checkout(sync + bt * SYNC_STRIDE, sync + (NB + bt) * SYNC_STRIDE, 2u * (unsigned)NJ);
}

int g_num_cus[16] = {0};

// CGN_LSTM_XCD_MAP=0 keeps blockIdx order (hand-off groups spread over all XCDs) for A/B measurements
// X6 (six bf16 partial products instead of the fp32 MFMA) is ON in the forward kernels and OFF in the backward ones:
// measured at cfg2 (tools/lstm_bench.py, profiles/r03_lstm_bench.txt) forward 5.1 -> 4.5 us per step, two-layer forward
// 7.5 -> 6.7; backward 6.0 -> 6.4 and 9.7 -> 11.5: a backward step hands over 4x the values (dgates, K = 4H), every
// consumer wave splits its share again (64 values per lane and step: ~1 us of VALU work on a SIMD that two waves
// share), which eats the MFMA time saved. Splitting once at the producer and handing the three bf16 parts over through
// a blocked scratch (6 bytes per value, three 16-byte loads per chunk pair that ARE the MFMA operands) was built and
// measured too: encoder backward 104.5 -> 122.3 us, one generator layer 96.6 -> 116.5 - a backward step is paced by
// the hand-off loads, not by the matrix pipe, and that form loads 1.5x the bytes. The backward kernels stay on the fp32
// MFMA with the fp32 scratch. CGN_LSTM_X6=0 / CGN_LSTM_X6_BWD=1 override (A/B measurements).
// 0 = every recurrence at fp32 accuracy (default); the optional bf16 mode: the row-tiled recurrences at H > 512 take
// two-term (1) or plain (2) bf16 operands (cgn_lstm_set_operand_mode, include/cgn_hip.h)
int g_lstm_operand_mode = 0;

bool lstm_x6() {
    static const bool v = [] { const char* e = getenv("CGN_LSTM_X6"); return !(e && e[0] == '0'); }();
    return v;
}
bool lstm_x6_bwd() {
    static const bool v = [] { const char* e = getenv("CGN_LSTM_X6_BWD"); return e && e[0] == '1'; }();
    return v;
}
int xcd_identity() {
    static const int v = [] { const char* e = getenv("CGN_LSTM_XCD_MAP"); return (e && e[0] == '0') ? 1 : 0; }();
    return v;
}

int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (g_num_cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        g_num_cus[dev] = n;
    }
    return g_num_cus[dev];
}

// shapes the persistent kernels take: whole 16-unit tiles, the register-resident W slice, one workgroup per CU at most
bool persist_ok(int nseq, int B, int T, int H) {
    if (nseq < 1 || nseq > LSTM_MAX_SEQ || B < 1 || T < 1 || H < 16 || H % 16 != 0 || H > 512) return false;
    const long wgs = (long)nseq * cgn_cdiv(B, 16) * (H / 16);
    if (wgs > device_cus()) return false;
    if ((long)B * (T + 1) * H * 4 >= (1L << 31) || (long)B * T * 4 * H * 4 >= (1L << 31)) return false;
    return true;
}

// Row tiles per workgroup for the persistent form: 1 (the kernels without row tiling, H <= 512) when the (batch tile,
// unit tile) pairs fit the CUs, else 2 or 4 (lstm_persist_*_rt_kernel: also H <= 1024); 0 = outside the persistent form.
int persist_rt(int nseq, int B, int T, int H) {
    if (nseq < 1 || nseq > LSTM_MAX_SEQ || B < 1 || T < 1 || H < 16 || H % 16 != 0 || H > 1024) return 0;
    if ((long)B * (T + 1) * H * 4 >= (1L << 31) || (long)B * T * 4 * H * 4 >= (1L << 31)) return 0;
    for (int rt = 1; rt <= 4; rt *= 2)
        if ((long)nseq * cgn_cdiv(B, 16 * rt) * (H / 16) <= device_cus()) return rt;
    return 0;
}

// scratch layout: one 128-byte line per group (word 0: arrival counter, word 16: exit counter of its hand-off domain),
// zeroed ONCE by the caller and left at zero by every launch; one reserved line; and, as the LAST line of whatever the
// caller passed, the sticky timeout flag that the library sets and never clears
long persist_sync_bytes(int nseq, int B) { return ((long)nseq * cgn_cdiv(B, 16) + 2) * SYNC_STRIDE * 4; }

}  // namespace

// two-layer wavefront form: both layers' workgroups resident together, the wider layer's weight slice in registers
static bool persist2_ok(int B, int T, int H) {
    if (B < 1 || T < 1 || H < 16 || H % 16 != 0 || H > 512) return false;
    const long wgs = 2L * cgn_cdiv(B, 16) * (H / 16);
    if (wgs > device_cus()) return false;
    if ((long)B * (T + 1) * H * 4 >= (1L << 31) || (long)B * T * 4 * H * 4 >= (1L << 31)) return false;
    return true;
}

namespace {
struct InitSlots {
    float* h[LSTM_MAX_SEQ];
    float* c[LSTM_MAX_SEQ];
    int reverse[LSTM_MAX_SEQ];
    int init_zero[LSTM_MAX_SEQ];
};
InitSlots p_init(const cgn_lstm_fwd_seq* seqs, int nseq) {
    InitSlots q = {};
    for (int i = 0; i < nseq; ++i) {
        q.h[i] = seqs[i].hbuf; q.c[i] = seqs[i].cbuf; q.reverse[i] = seqs[i].reverse; q.init_zero[i] = seqs[i].init_zero;
    }
    return q;
}
// zero the initial-state slot (t = 0, or t = T of a reversed sequence) of hbuf / cbuf [B, T+1, H]; H % 4 == 0
__global__ __launch_bounds__(256) void lstm_zero_initial_kernel(const InitSlots q, int B, int T, int H) {
    const int s = blockIdx.y;
    if (!q.init_zero[s]) return;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int hq = H / 4;
    if (v >= (long)B * hq) return;
    const int b = (int)(v / hq), j = (int)(v % hq) * 4;
    const long off = ((long)b * (T + 1) + (q.reverse[s] ? T : 0)) * H + j;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(q.h[s] + off) = z;
    *reinterpret_cast<f32x4*>(q.c[s] + off) = z;
}
}  // namespace

extern "C" int cgn_lstm_seq_fwd_f32(void* stream, const cgn_lstm_fwd_seq* seqs, int32_t nseq, int32_t B, int32_t T,
                                    int32_t H) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && H % 4 == 0);
    if (B == 0 || T == 0) return CGN_OK;
    FwdParams p;
    bool any_zero = false;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_fwd_seq& s = seqs[i];
        CGN_CHECK_ARG(s.xg && s.w_hh && s.hbuf && s.cbuf && s.gates);
        CGN_CHECK_ARG(((uintptr_t)s.hbuf % 16 == 0) && ((uintptr_t)s.w_hh % 16 == 0) && ((uintptr_t)s.cbuf % 16 == 0));
        p.s[i] = {s.xg, s.xg_extra, s.w_hh, s.hbuf, s.cbuf, s.gates, s.reverse, s.init_zero};
        any_zero = any_zero || s.init_zero;
    }
    // The step kernels read the initial-state slots: zero them with a kernel of our own. (hipMemset2DAsync did this until
    // round 2: as a node of a replayed hipGraph it was not reliably ordered against the step kernels around it on ROCm
    // 7.2 - replays that followed a device synchronise drifted from the eager step and ended in NaN after ~30 steps.)
    if (any_zero) {
        hipLaunchKernelGGL(lstm_zero_initial_kernel, dim3(cgn_cdiv((long)B * H / 4, 256), nseq), dim3(256), 0,
                           (hipStream_t)stream, p_init(seqs, nseq), B, T, H);
        CGN_LAUNCH_CHECK();
    }
    p.B = B; p.T = T; p.H = H; p.nseq = nseq;
    dim3 grid(cgn_cdiv(H, 16) * cgn_cdiv(B, 16) * nseq);
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(LSTM_NT), 0, (hipStream_t)stream, p);
    }
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lstm_seq_bwd_f32(void* stream, const cgn_lstm_bwd_seq* seqs, int32_t nseq, int32_t B, int32_t T,
                                    int32_t H) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && H % 4 == 0);
    if (B == 0 || T == 0) return CGN_OK;
    BwdParams p;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_bwd_seq& s = seqs[i];
        CGN_CHECK_ARG(s.w_hh_t && s.cbuf && s.gates && s.dgates && s.dc_state);
        CGN_CHECK_ARG(((uintptr_t)s.dgates % 16 == 0) && ((uintptr_t)s.w_hh_t % 16 == 0));
        p.s[i] = {s.w_hh_t, s.w_hh, s.cbuf, s.gates, s.dh_ext, s.dh_seed, s.dgates, s.dc_state, s.dh_bs, s.dh_ts,
                  s.reverse};
    }
    p.B = B; p.T = T; p.H = H; p.nseq = nseq;
    dim3 grid(cgn_cdiv(H, 16) * cgn_cdiv(B, 16) * nseq);
    for (int step = 0; step < T; ++step) {
        p.step = step;
        hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, dim3(LSTM_NT), 0, (hipStream_t)stream, p);
    }
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

namespace {
// operand mode 2, sequences with a given initial state: bf16(h_init) into parity 1 of the hand-off scratch
// [2][H / 8][B][8] (the parity step 0 reads); two units per thread
__global__ __launch_bounds__(256) void lstm_handoff_init_kernel(const FwdParams p) {
    const FwdSeq& sq = p.s[blockIdx.y];
    if (sq.init_zero || !sq.hx) return;
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int hp = p.H / 2;
    if (v >= (long)p.B * hp) return;
    const int b = (int)(v / hp), j = (int)(v % hp) * 2;
    const float* h = sq.hbuf + ((long)b * (p.T + 1) + (sq.reverse ? p.T : 0)) * p.H + j;
    *reinterpret_cast<unsigned*>(sq.hx + hx_off(1, p.H / 8, j >> 3, p.B, b) + (j & 7)) = pack_bf16x2_lstm(h[0], h[1]);
}
}  // namespace

extern "C" int cgn_lstm_persistent_query(int32_t nseq, int32_t B, int32_t T, int32_t H, int64_t* sync_bytes) {
    CGN_CHECK_ARG(sync_bytes != nullptr);
    *sync_bytes = persist_rt(nseq, B, T, H) ? persist_sync_bytes(nseq, B) : 0;
    return CGN_OK;
}

extern "C" int cgn_lstm_seq_fwd_persistent_f32(void* stream, const cgn_lstm_fwd_seq* seqs, int32_t nseq, int32_t B,
                                               int32_t T, int32_t H, void* sync_ws, int64_t sync_ws_bytes) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0);
    if (B == 0 || T == 0) return CGN_OK;
    const int rt = persist_rt(nseq, B, T, H);
    CGN_CHECK_ARG(rt > 0);
    const long need = persist_sync_bytes(nseq, B);
    CGN_CHECK_ARG(sync_ws && sync_ws_bytes >= need && ((uintptr_t)sync_ws % 16 == 0));
    FwdParams p;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_fwd_seq& s = seqs[i];
        CGN_CHECK_ARG(s.xg && s.w_hh && s.hbuf && s.cbuf && s.gates);
        CGN_CHECK_ARG(((uintptr_t)s.hbuf % 16 == 0) && ((uintptr_t)s.w_hh % 16 == 0));
        CGN_CHECK_ARG((uintptr_t)s.handoff % 16 == 0);
        p.s[i] = {s.xg, s.xg_extra, s.w_hh, s.hbuf, s.cbuf, s.gates, s.reverse, s.init_zero, (unsigned short*)s.handoff};
    }
    p.B = B; p.T = T; p.H = H; p.nseq = nseq; p.step = xcd_identity();
    hipStream_t st = (hipStream_t)stream;
    unsigned* sticky = (unsigned*)((char*)sync_ws + (sync_ws_bytes / 128 - 1) * 128);
    dim3 grid(nseq * cgn_cdiv(B, 16 * rt) * (H / 16));
    unsigned* sw = (unsigned*)sync_ws;
    const bool x6 = lstm_x6();
    // operand mode 2 runs the plain-bf16 kernels where every sequence brings its hand-off scratch and the waves own
    // whole chunk pairs (H % 256 == 0); otherwise the two-term kernels
    bool x1 = g_lstm_operand_mode == 2 && H > 512 && H % 256 == 0;
    for (int i = 0; i < nseq; ++i) x1 = x1 && seqs[i].handoff != nullptr;
    if (x1) {
        bool any_init = false;
        for (int i = 0; i < nseq; ++i) any_init = any_init || !seqs[i].init_zero;
        if (any_init) {  // the given initial h as bf16 into parity 1 of the scratch: what step 0 reads
            hipLaunchKernelGGL(lstm_handoff_init_kernel, dim3(cgn_cdiv((long)B * H / 2, 256), nseq), dim3(256), 0, st, p);
            CGN_LAUNCH_CHECK();
        }
    }
#define CGN_FWD_K(CPW) do { if (x6) hipLaunchKernelGGL((lstm_persist_fwd_kernel<CPW, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky); \
                            else hipLaunchKernelGGL((lstm_persist_fwd_kernel<CPW, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky); } while (0)
#define CGN_FWD_RT(CPW, RT, X) hipLaunchKernelGGL((lstm_persist_fwd_rt_kernel<CPW, RT, X>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky)
#define CGN_FWD_RTX(CPW, RT) do { if (x6) CGN_FWD_RT(CPW, RT, 3); else CGN_FWD_RT(CPW, RT, 0); } while (0)
    // H > 512: the three-term weight slice does not fit the registers; the two-term one does (the bf16 mode only)
#define CGN_FWD_RTB(CPW, RT) do { if (x1) CGN_FWD_RT(CPW, RT, 1); else if (g_lstm_operand_mode != 0) CGN_FWD_RT(CPW, RT, 2); else CGN_FWD_RT(CPW, RT, 0); } while (0)
    if (rt == 1 && H <= 256) CGN_FWD_K(2);
    else if (rt == 1 && H <= 512) CGN_FWD_K(4);
    else if (rt == 1) CGN_FWD_RTB(8, 1);
    else if (rt == 2 && H <= 256) CGN_FWD_RTX(2, 2);
    else if (rt == 2 && H <= 512) CGN_FWD_RTX(4, 2);
    else if (rt == 2) CGN_FWD_RTB(8, 2);
    else if (H <= 256) CGN_FWD_RTX(2, 4);
    else if (H <= 512) CGN_FWD_RTX(4, 4);
    else CGN_FWD_RTB(8, 4);
#undef CGN_FWD_RTB
#undef CGN_FWD_RT
#undef CGN_FWD_RTX
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The same persistent forward launch for callers that cannot allocate (code under hipGraph capture, tight serving
// loops, harnesses that forbid hipMalloc): the arrival counters, the timeout word and the sticky flag live in a
// file-scope device array of this code object instead of a caller-owned sync_ws. Every GPU holds its own copy of the
// array at its own address (taken once per device with hipGetSymbolAddress, with that device current); every call on
// a GPU shares that GPU's copy, so calls that overlap on one GPU race (the counters are re-zeroed per call on the
// call's stream). No hipMalloc / hipFree / allocating library call on this path.
// ------------------------------------------------------------------------------------------------------------------
constexpr int NOALLOC_MAX_GROUPS = 256;  // nseq * ceil(B/16) <= 256: persist_ok() already caps workgroups at the CU count
constexpr int NOALLOC_LINES = NOALLOC_MAX_GROUPS + 2;  // + per-launch timeout word + sticky flag, one 128-byte line each
static __device__ __align__(256) unsigned g_lstm_noalloc_scratch[NOALLOC_LINES * SYNC_STRIDE];  // 33,024 bytes per GPU

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
unsigned* g_noalloc_ptr[NOALLOC_MAX_DEVICES] = {nullptr};

// device address of this GPU's copy of the scratch array (never the array's host-side name)
int noalloc_scratch(unsigned** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) return CGN_ERR_INVALID_ARG;
    unsigned* p = __atomic_load_n(&g_noalloc_ptr[dev], __ATOMIC_ACQUIRE);
    if (p == nullptr) {
        void* addr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&addr, HIP_SYMBOL(g_lstm_noalloc_scratch));
        if (e != hipSuccess || addr == nullptr) return CGN_ERR_HIP_BASE + (int)e;
        p = (unsigned*)addr;
        __atomic_store_n(&g_noalloc_ptr[dev], p, __ATOMIC_RELEASE);  // racing first calls compute the same address
    }
    *out = p;
    return CGN_OK;
}
}  // namespace

extern "C" int cgn_lstm_seq_fwd_persistent_noalloc_f32(void* stream, const cgn_lstm_fwd_seq* seqs, int32_t nseq,
                                                       int32_t B, int32_t T, int32_t H, uint32_t* status) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0);
    if (B == 0 || T == 0) return CGN_OK;
    CGN_CHECK_ARG(persist_ok(nseq, B, T, H));
    // the stated maximum of the built-in scratch: one counter line per (sequence, 16-row batch tile)
    CGN_CHECK_ARG((long)nseq * cgn_cdiv(B, 16) <= NOALLOC_MAX_GROUPS);
    unsigned* scratch = nullptr;
    {
        const int rc = noalloc_scratch(&scratch);
        if (rc) return rc;
    }
    FwdParams p;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_fwd_seq& s = seqs[i];
        CGN_CHECK_ARG(s.xg && s.w_hh && s.hbuf && s.cbuf && s.gates);
        CGN_CHECK_ARG(((uintptr_t)s.hbuf % 16 == 0) && ((uintptr_t)s.w_hh % 16 == 0));
        p.s[i] = {s.xg, s.xg_extra, s.w_hh, s.hbuf, s.cbuf, s.gates, s.reverse, s.init_zero};
    }
    p.B = B; p.T = T; p.H = H; p.nseq = nseq; p.step = xcd_identity();
    hipStream_t st = (hipStream_t)stream;
    // the counters are at zero: the array starts zeroed and every launch re-arms them on its way out (persist_checkout)
    unsigned* sticky = status ? status : scratch + (NOALLOC_LINES - 1) * SYNC_STRIDE;
    dim3 grid(nseq * cgn_cdiv(B, 16) * (H / 16));
    unsigned* sw = scratch;
    const bool x6 = lstm_x6();
    if (H <= 256) CGN_FWD_K(2);
    else CGN_FWD_K(4);
#undef CGN_FWD_K
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lstm_seq_bwd_persistent_f32(void* stream, const cgn_lstm_bwd_seq* seqs, int32_t nseq, int32_t B,
                                               int32_t T, int32_t H, void* sync_ws, int64_t sync_ws_bytes) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0);
    if (B == 0 || T == 0) return CGN_OK;
    const int rt = persist_rt(nseq, B, T, H);
    CGN_CHECK_ARG(rt > 0);
    const long need = persist_sync_bytes(nseq, B);
    CGN_CHECK_ARG(sync_ws && sync_ws_bytes >= need && ((uintptr_t)sync_ws % 16 == 0));
    BwdParams p;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_bwd_seq& s = seqs[i];
        CGN_CHECK_ARG((s.w_hh_t || s.w_hh) && s.cbuf && s.gates && s.dgates && s.dc_state);
        CGN_CHECK_ARG(((uintptr_t)s.dgates % 16 == 0) && ((uintptr_t)s.w_hh_t % 16 == 0));
        CGN_CHECK_ARG((uintptr_t)s.handoff % 16 == 0);
        p.s[i] = {s.w_hh_t, s.w_hh, s.cbuf, s.gates, s.dh_ext, s.dh_seed, s.dgates, s.dc_state, s.dh_bs, s.dh_ts,
                  s.reverse, (unsigned short*)s.handoff};
    }
    p.B = B; p.T = T; p.H = H; p.nseq = nseq; p.step = xcd_identity();
    hipStream_t st = (hipStream_t)stream;
    unsigned* sticky = (unsigned*)((char*)sync_ws + (sync_ws_bytes / 128 - 1) * 128);
    dim3 grid(nseq * cgn_cdiv(B, 16 * rt) * (H / 16));
    unsigned* sw = (unsigned*)sync_ws;
    const bool x6 = lstm_x6_bwd();
    bool x1 = g_lstm_operand_mode == 2 && H > 512 && H % 256 == 0;  // (as in the forward entry point)
    for (int i = 0; i < nseq; ++i) x1 = x1 && seqs[i].handoff != nullptr;
#define CGN_BWD_K(CPW) do { if (x6) hipLaunchKernelGGL((lstm_persist_bwd_kernel<CPW, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky); \
                            else hipLaunchKernelGGL((lstm_persist_bwd_kernel<CPW, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky); } while (0)
#define CGN_BWD_RT(CPW, RT, X) hipLaunchKernelGGL((lstm_persist_bwd_rt_kernel<CPW, RT, X>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky)
#define CGN_BWD_RTX(CPW, RT) do { if (x6) CGN_BWD_RT(CPW, RT, 3); else CGN_BWD_RT(CPW, RT, 0); } while (0)
#define CGN_BWD_RTB(CPW, RT) do { if (x1) CGN_BWD_RT(CPW, RT, 1); else if (g_lstm_operand_mode != 0) CGN_BWD_RT(CPW, RT, 2); else CGN_BWD_RT(CPW, RT, 0); } while (0)
    if (rt == 1 && H <= 256) CGN_BWD_K(8);
    else if (rt == 1 && H <= 512) CGN_BWD_K(16);
    else if (rt == 1) CGN_BWD_RTB(32, 1);
    else if (rt == 2 && H <= 256) CGN_BWD_RTX(8, 2);
    else if (rt == 2 && H <= 512) CGN_BWD_RTX(16, 2);
    else if (rt == 2) CGN_BWD_RTB(32, 2);
    else if (H <= 256) CGN_BWD_RTX(8, 4);
    else if (H <= 512) CGN_BWD_RTX(16, 4);
    else CGN_BWD_RTB(32, 4);
#undef CGN_BWD_RTB
#undef CGN_BWD_K
#undef CGN_BWD_RT
#undef CGN_BWD_RTX
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lstm2_persistent_query(int32_t B, int32_t T, int32_t H, int64_t* sync_bytes) {
    CGN_CHECK_ARG(sync_bytes != nullptr);
    *sync_bytes = persist2_ok(B, T, H) ? persist_sync_bytes(2, B) : 0;
    return CGN_OK;
}

extern "C" int cgn_lstm2_fwd_persistent_f32(void* stream, const cgn_lstm2_fwd* d, int32_t B, int32_t T, int32_t H,
                                            void* sync_ws, int64_t sync_ws_bytes) {
    CGN_CHECK_ARG(d != nullptr && B >= 0 && T >= 0 && H > 0);
    if (B == 0 || T == 0) return CGN_OK;
    CGN_CHECK_ARG(persist2_ok(B, T, H));
    const long need = persist_sync_bytes(2, B);
    CGN_CHECK_ARG(sync_ws && sync_ws_bytes >= need && ((uintptr_t)sync_ws % 16 == 0));
    CGN_CHECK_ARG(d->xg0 && d->w_hh0 && d->w_ih1 && d->w_hh1 && d->b_ih1 && d->b_hh1);
    CGN_CHECK_ARG(d->hbuf0 && d->cbuf0 && d->gates0 && d->hbuf1 && d->cbuf1 && d->gates1);
    CGN_CHECK_ARG(((uintptr_t)d->hbuf0 % 16 == 0) && ((uintptr_t)d->hbuf1 % 16 == 0) && ((uintptr_t)d->w_hh0 % 16 == 0) &&
                  ((uintptr_t)d->w_ih1 % 16 == 0) && ((uintptr_t)d->w_hh1 % 16 == 0));
    Fwd2Params p;
    p.xg0 = d->xg0; p.xg0_extra = d->xg0_extra; p.w_hh0 = d->w_hh0; p.w_ih1 = d->w_ih1; p.w_hh1 = d->w_hh1;
    p.b_ih1 = d->b_ih1; p.b_hh1 = d->b_hh1;
    p.hbuf[0] = d->hbuf0; p.cbuf[0] = d->cbuf0; p.gates[0] = d->gates0;
    p.hbuf[1] = d->hbuf1; p.cbuf[1] = d->cbuf1; p.gates[1] = d->gates1;
    p.B = B; p.T = T; p.H = H; p.flat = xcd_identity();
    hipStream_t st = (hipStream_t)stream;
    unsigned* sticky = (unsigned*)((char*)sync_ws + (sync_ws_bytes / 128 - 1) * 128);
    dim3 grid(2 * cgn_cdiv(B, 16) * (H / 16));
    unsigned* sw = (unsigned*)sync_ws;
    if (H <= 256 && lstm_x6()) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<2, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else if (H <= 256) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<2, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else if (lstm_x6()) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<4, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else hipLaunchKernelGGL((lstm2_persist_fwd_kernel<4, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lstm2_bwd_persistent_f32(void* stream, const cgn_lstm2_bwd* d, int32_t B, int32_t T, int32_t H,
                                            void* sync_ws, int64_t sync_ws_bytes) {
    CGN_CHECK_ARG(d != nullptr && B >= 0 && T >= 0 && H > 0);
    if (B == 0 || T == 0) return CGN_OK;
    CGN_CHECK_ARG(persist2_ok(B, T, H));
    const long need = persist_sync_bytes(2, B);
    CGN_CHECK_ARG(sync_ws && sync_ws_bytes >= need && ((uintptr_t)sync_ws % 16 == 0));
    CGN_CHECK_ARG(d->w_hh0 && d->w_ih1 && d->w_hh1 && d->cbuf0 && d->gates0 && d->cbuf1 && d->gates1 && d->dgates0 &&
                  d->dgates1);
    CGN_CHECK_ARG(((uintptr_t)d->dgates0 % 16 == 0) && ((uintptr_t)d->dgates1 % 16 == 0));
    Bwd2Params p;
    p.w_hh0 = d->w_hh0; p.w_ih1 = d->w_ih1; p.w_hh1 = d->w_hh1;
    p.cbuf[0] = d->cbuf0; p.gates[0] = d->gates0; p.cbuf[1] = d->cbuf1; p.gates[1] = d->gates1;
    p.dh_ext1 = d->dh_ext1; p.dh_bs = d->dh_bs; p.dh_ts = d->dh_ts;
    p.dgates[0] = d->dgates0; p.dgates[1] = d->dgates1;
    CGN_CHECK_ARG((uintptr_t)d->handoff % 16 == 0);
    p.hx[0] = (float*)d->handoff;
    p.hx[1] = d->handoff ? (float*)d->handoff + (long)B * T * 4 * H : nullptr;
    p.B = B; p.T = T; p.H = H; p.flat = xcd_identity();
    hipStream_t st = (hipStream_t)stream;
    unsigned* sticky = (unsigned*)((char*)sync_ws + (sync_ws_bytes / 128 - 1) * 128);
    dim3 grid(2 * cgn_cdiv(B, 16) * (H / 16));
    unsigned* sw = (unsigned*)sync_ws;
    if (H <= 256 && lstm_x6_bwd()) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<8, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else if (H <= 256) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<8, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else if (lstm_x6_bwd()) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<16, true>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    else hipLaunchKernelGGL((lstm2_persist_bwd_kernel<16, false>), grid, dim3(LSTM_NT), 0, st, p, sw, sticky);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lstm_set_operand_mode(int32_t mode) {
    CGN_CHECK_ARG(mode == 0 || mode == 1 || mode == 2);
    g_lstm_operand_mode = mode;
    return CGN_OK;
}
