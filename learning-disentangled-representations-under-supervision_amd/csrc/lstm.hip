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
};
struct FwdParams {
    FwdSeq s[LSTM_MAX_SEQ];
    int B, T, H, step, nseq;
};

struct BwdSeq {
    const float* w_hh_t;
    const float* cbuf;
    const float* gates;
    const float* dh_ext;
    const float* dh_seed;
    float* dgates;
    float* dc_state;
    long dh_bs, dh_ts;
    int reverse;
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
__device__ __forceinline__ WorkId lstm_work(int nj, int nb, int nseq) {
    const int nwg = nj * nb * nseq;
    const int bid = blockIdx.x;
    const int qq = nwg / 8, rr = nwg % 8;
    const int xcd = bid % 8, idx = bid / 8;
    const int w = ((xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;  // bijective
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

}  // namespace

extern "C" int cgn_lstm_seq_fwd_f32(void* stream, const cgn_lstm_fwd_seq* seqs, int32_t nseq, int32_t B, int32_t T,
                                    int32_t H) {
    CGN_CHECK_ARG(seqs && nseq >= 1 && nseq <= LSTM_MAX_SEQ);
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H > 0 && H % 4 == 0);
    if (B == 0 || T == 0) return CGN_OK;
    FwdParams p;
    for (int i = 0; i < nseq; ++i) {
        const cgn_lstm_fwd_seq& s = seqs[i];
        CGN_CHECK_ARG(s.xg && s.w_hh && s.hbuf && s.cbuf && s.gates);
        CGN_CHECK_ARG(((uintptr_t)s.hbuf % 16 == 0) && ((uintptr_t)s.w_hh % 16 == 0));
        p.s[i] = {s.xg, s.xg_extra, s.w_hh, s.hbuf, s.cbuf, s.gates, s.reverse};
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
        p.s[i] = {s.w_hh_t, s.cbuf, s.gates, s.dh_ext, s.dh_seed, s.dgates, s.dc_state, s.dh_bs, s.dh_ts, s.reverse};
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
