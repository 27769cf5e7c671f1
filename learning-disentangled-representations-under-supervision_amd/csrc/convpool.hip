// Multi-width 1-D convolution + max-over-time pooling around the implicit-GEMM convolution, for both users:
//   * the text-CNN discriminator (model/discriminator_sentiment.py:56-78): Conv2d(1,Co,(K,E)) + ReLU + max-pool,
//     weights [Co][K][E] (layout 0), one "sentence" = [L, E] word embeddings;
//   * the char-TDNN (selfModules/tdnn.py:40-43): conv1d(no bias) + tanh + max over positions, weights [od][Cc][kW]
//     (torch conv1d layout = layout 1), one "sentence" = one word = [W, Cc] char embeddings.
// The convolution itself is cgn_gemm_f32 over overlapping windows (two-level A addressing; for layout 1 also a
// two-level K axis on the weight operand). Here: activation + pooling with argmax, and the two backward routes through
// the pooled maximum. Max-over-time keeps ONE position per (sentence, map), so both backward routes are sparse
// (one K*E window per map) and run on the VALU with thread-exclusive accumulators (fixed summation order).
#include "common.h"

namespace {

constexpr int ACT_RELU = 0, ACT_TANH = 1;

// conv is [B*P, Co] (row (b,p)), bias included. feat[b, col0+co] = act(max_p conv); arg = argmax position, or -1 when
// the ReLU gate is closed (max <= 0: no gradient).
__global__ __launch_bounds__(256) void act_maxpool_kernel(const float* __restrict__ conv, int B, int P, int Co, int act,
                                                          float* __restrict__ feat, long ldf, int col0,
                                                          int32_t* __restrict__ arg, long lda) {
    const long total = (long)B * Co;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Co), co = (int)(i % Co);
        const float* p = conv + (long)b * P * Co + co;
        float best = p[0];
        int bi = 0;
        for (int t0 = 1; t0 < P; t0 += 8) {  // 8 positions of loads in flight; first maximum wins, as in torch
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)min(t0 + u, P - 1) * Co];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < P && v[u] > best) {
                    best = v[u];
                    bi = t0 + u;
                }
            }
        }
        if (act == ACT_TANH) {
            feat[(long)b * ldf + col0 + co] = tanhf(best);  // tanh is monotonic: max tanh(x) = tanh(max x)
            arg[(long)b * lda + col0 + co] = bi;
        } else {
            feat[(long)b * ldf + col0 + co] = fmaxf(best, 0.f);
            arg[(long)b * lda + col0 + co] = (best > 0.f) ? bi : -1;
        }
    }
}

// The same for up to 8 widths in ONE launch (each width has its own conv buffer, position count and column range).
struct PoolGroup {
    const float* conv[8];
    int P[8], Co[8], col0[8], end[8];  // end: running sum of Co
    int n;
};
__global__ __launch_bounds__(256) void act_maxpool_group_kernel(const PoolGroup g, int B, int act,
                                                                float* __restrict__ feat, long ldf,
                                                                int32_t* __restrict__ arg, long lda) {
    const int F = g.end[g.n - 1];
    const long total = (long)B * F;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / F), f = (int)(i % F);
        int j = 0;
        while (j + 1 < g.n && f >= g.end[j]) ++j;
        const int co = f - (j ? g.end[j - 1] : 0);
        const int P = g.P[j], Co = g.Co[j];
        const float* p = g.conv[j] + (long)b * P * Co + co;
        float best = p[0];
        int bi = 0;
        for (int t0 = 1; t0 < P; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)min(t0 + u, P - 1) * Co];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < P && v[u] > best) {
                    best = v[u];
                    bi = t0 + u;
                }
            }
        }
        const int col = g.col0[j] + co;
        if (act == ACT_TANH) {
            feat[(long)b * ldf + col] = tanhf(best);
            arg[(long)b * lda + col] = bi;
        } else {
            feat[(long)b * ldf + col] = fmaxf(best, 0.f);
            arg[(long)b * lda + col] = (best > 0.f) ? bi : -1;
        }
    }
}

constexpr int MAXW = 8;
struct ConvWidths {
    const float* W[MAXW];
    float* dW[MAXW];
    float* db[MAXW];
    int K[MAXW], Co[MAXW], col0[MAXW];
    int n, F, act, layout;
};

__device__ __forceinline__ float pooled_grad(const ConvWidths& w, const float* dfeat, const float* feat, long idx) {
    float g = dfeat[idx];
    if (w.act == ACT_TANH) {
        const float y = feat[idx];
        g *= (1.f - y * y);
    }
    return g;
}

// ---- dense route for dX -------------------------------------------------------------------------------------------
// The pooled gradient is first spread back over the convolution outputs (one non-zero per (sentence, map): the kept
// position), dcols_w = dconv_w · W_w then runs as ONE grouped MFMA launch over the widths (cgn_gemm_group_f32, the
// transposed convolution as a product), and the overlapping windows are summed per output element (col2im) in fixed
// order. Same arithmetic as conv_bwd_dx_kernel below; 2-3x faster where every map's window would otherwise be walked
// on the VALU (text-CNN 73 -> 26 us, char-TDNN 94 -> 50 us at cfg2).
struct ScatterGroup {
    float* dconv[MAXW];
    int P[MAXW], Co[MAXW], col0[MAXW];
    long end[MAXW];  // running sum of B*P*Co
    int n, F, act;
};
// blockIdx.y = width; a thread produces 4 consecutive elements of that width's [B*P, Co] matrix (one 16-byte store;
// one 32-bit division per thread instead of three 64-bit ones per element: 33 us -> see DESIGN.md for cfg2's char-TDNN).
__global__ __launch_bounds__(256) void pool_bwd_scatter_kernel(const ScatterGroup g, const float* __restrict__ dfeat,
                                                               const float* __restrict__ feat,
                                                               const int32_t* __restrict__ arg) {
    const int j = blockIdx.y;
    const unsigned size = (unsigned)(g.end[j] - (j ? g.end[j - 1] : 0));  // < 2^31 (host check)
    const unsigned Co = (unsigned)g.Co[j], P = (unsigned)g.P[j];
    float* out = g.dconv[j];
    for (unsigned li = (blockIdx.x * 256u + threadIdx.x) * 4u; li < size; li += gridDim.x * 1024u) {
        unsigned bp = li / Co;
        unsigned co = li - bp * Co;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned b = bp / P, p = bp - b * P;
            const bool in = li + u < size;
            const long fi = in ? (long)b * g.F + g.col0[j] + co : 0;
            float x = 0.f;
            if (in && arg[fi] == (int)p) {
                x = dfeat[fi];
                if (g.act == ACT_TANH) {
                    const float y = feat[fi];
                    x *= (1.f - y * y);
                }
            }
            v[u] = x;
            if (++co == Co) {
                co = 0;
                ++bp;
            }
        }
        if (li + 3 < size) {
            f32x4 w = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(out + li) = w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (li + u < size) out[li + u] = v[u];
        }
    }
}

struct Col2imGroup {
    const float* dcols[MAXW];
    int K[MAXW];
    int n;
};
// dx[b, t, e] = sum_w sum_k dcols_w[(b, t-k), k*E + e] over the windows (t-k in [0, L-K_w]) that cover t.
__global__ __launch_bounds__(256) void conv_col2im_kernel(const Col2imGroup g, int B, int L, int E,
                                                          float* __restrict__ dx, long sent_stride) {
    const long total = (long)B * L * E;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int e = (int)(i % E);
        const long bt = i / E;
        const int t = (int)(bt % L), b = (int)(bt / L);
        float acc = 0.f;
        for (int j = 0; j < g.n; ++j) {
            const int K = g.K[j], P = L - K + 1;
            const long ld = (long)K * E;
            const float* base = g.dcols[j] + (long)b * P * ld + e;
            for (int k0 = 0; k0 < K; k0 += 8) {  // 8 loads in flight, clamped addresses
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = k0 + u, p = t - k;
                    const bool ok = k < K && p >= 0 && p < P;
                    v[u] = ok ? base[(long)p * ld + (long)k * E] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
        dx[(long)b * sent_stride + (long)t * E + e] = acc;
    }
}

// dX[b, t, e] = sum over (width, map) whose kept window covers t of g[b,map] * W[map][t - t0][e].
// One wave owns 64 consecutive outputs (t,e) of one sentence (e fastest => coalesced weight reads) and walks all
// maps; the (argmax, gradient) pairs of 64 maps are loaded one per lane and broadcast with v_readlane. The walk is
// BRANCH-FREE in batches of DXU maps: every lane issues DXU weight loads on clamped addresses before the first one is
// consumed and a miss contributes an exact 0 (a branch per map made every load wait for the one before it: the kernel
// was pure L2 latency, 124 us for the char-TDNN). Every output element is owned by exactly one lane: no atomics, fixed
// summation order.
constexpr int DXU = 16;
__global__ __launch_bounds__(256) void conv_bwd_dx_kernel(const ConvWidths w, const float* __restrict__ dfeat,
                                                          const float* __restrict__ feat,
                                                          const int32_t* __restrict__ arg, int L, int E,
                                                          float* __restrict__ dx, long sent_stride) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int o = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;  // output index within the sentence
    const bool live = o < L * E;
    const int t = live ? o / E : 0, e = live ? o % E : 0;
    const int o_first = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (o_first >= L * E) return;  // wave-uniform
    float acc = 0.f;
    for (int wi = 0; wi < w.n; ++wi) {
        const int Kw = w.K[wi], Co = w.Co[wi];
        const float* Wb = w.W[wi];
        const int KE = Kw * E;
        // this lane's offset inside a map's window for k = 0, and the step per k
        const int e_off = (w.layout == 0) ? e : e * Kw;
        const int k_step = (w.layout == 0) ? E : 1;
        for (int c0 = 0; c0 < Co; c0 += 64) {
            int my_t0 = -(1 << 20);  // a window that covers nothing
            float my_g = 0.f;
            if (c0 + lane < Co) {
                const long fi = (long)b * w.F + w.col0[wi] + c0 + lane;
                const int t0 = arg[fi];
                if (t0 >= 0 && t0 + Kw <= L) {
                    my_t0 = t0;
                    my_g = pooled_grad(w, dfeat, feat, fi);
                }
            }
            const int nc = min(64, Co - c0);
            for (int c = 0; c < nc; c += DXU) {
                float wv[DXU], gv[DXU];
#pragma unroll
                for (int u = 0; u < DXU; ++u) {
                    const int cc = min(c + u, nc - 1);
                    const int t0 = __builtin_amdgcn_readlane(my_t0, cc);
                    const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_g), cc));
                    const int k = t - t0;
                    const bool hit = live && (c + u < nc) && k >= 0 && k < Kw;
                    const int off = hit ? e_off + k * k_step : 0;
                    wv[u] = Wb[(long)(c0 + cc) * KE + off];
                    gv[u] = hit ? g : 0.f;
                }
#pragma unroll
                for (int u = 0; u < DXU; ++u) acc += gv[u] * wv[u];
            }
        }
    }
    if (live) dx[(long)b * sent_stride + o] = acc;
}

// dW[map][window element] (+)= sum_b g[b,map] * x[b, t0(b,map) + k, e];  db[map] (+)= sum_b g[b,map].
// Workgroup = 64 window elements of one map; its NW waves split the batch NW ways and are summed through LDS in
// fixed order (NW = 16 for long batches: the char-TDNN's "batch" is every word of the step, and with 4 waves each
// walked 320 words in 20 dependent rounds of loads: 69 us). (t0, g) of 64 sentences are loaded one per lane and broadcast; the x loads of DWU sentences are issued
// together (independent addresses) before any is consumed, so the walk over the batch is not one L2 round trip per
// sentence.
constexpr int DWU = 16;
template <int NW>
__global__ __launch_bounds__(64 * NW) void conv_bwd_dw_kernel(const ConvWidths w, const float* __restrict__ dfeat,
                                                          const float* __restrict__ feat, const int32_t* __restrict__ arg,
                                                          const float* __restrict__ x, long sent_stride, int B, int L,
                                                          int E, float beta) {
    __shared__ float red[NW][64];
    __shared__ float redb[NW];
    const int wi = blockIdx.z, co = blockIdx.y;
    const int Kw = w.K[wi];
    const int KE = Kw * E;
    if (co >= w.Co[wi] || blockIdx.x * 64 >= KE) return;  // uniform per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;  // window element, order (k, e)
    const bool live = i < KE;
    const int il = live ? i : 0;
    const int per = (B + NW - 1) / NW;
    const int b_beg = wave * per, b_end = min(B, b_beg + per);
    float acc = 0.f, bsum = 0.f;
    for (int bb = b_beg; bb < b_end; bb += 64) {
        int my_t0 = 0;
        float my_g = 0.f;
        if (bb + lane < b_end) {
            const long fi = (long)(bb + lane) * w.F + w.col0[wi] + co;
            const int t0 = arg[fi];
            if (t0 >= 0 && t0 + Kw <= L) {
                my_t0 = t0;
                my_g = pooled_grad(w, dfeat, feat, fi);
            }
        }
        const int nb = min(64, b_end - bb);
        for (int c = 0; c < nb; c += DWU) {
            float xv[DWU], gv[DWU];
#pragma unroll
            for (int u = 0; u < DWU; ++u) {
                const int cc = min(c + u, nb - 1);
                const int t0 = __builtin_amdgcn_readlane(my_t0, cc);
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_g), cc));
                gv[u] = (c + u < nb) ? g : 0.f;  // g == 0 for closed gates
                xv[u] = x[(long)(bb + cc) * sent_stride + (long)t0 * E + il];
            }
#pragma unroll
            for (int u = 0; u < DWU; ++u) {
                bsum += gv[u];
                acc += gv[u] * xv[u];
            }
        }
    }
    red[wave][lane] = acc;
    if (lane == 0) redb[wave] = bsum;
    __syncthreads();
    if (wave == 0) {
        float tot = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) tot += red[v][lane];
        if (live) {
            const int di = (w.layout == 0) ? i : (i % E) * Kw + (i / E);
            float* d = w.dW[wi] + (long)co * KE + di;
            *d = (beta != 0.f) ? tot + beta * (*d) : tot;
        }
        if (blockIdx.x == 0 && lane == 0 && w.db[wi]) {
            float bs = 0.f;
#pragma unroll
            for (int v = 0; v < NW; ++v) bs += redb[v];
            float* d = w.db[wi] + co;
            *d = (beta != 0.f) ? bs + beta * (*d) : bs;
        }
    }
}

// Embedding-table gradient without atomics: dtable[v, :] (+)= sum over positions with idx == v of dx[pos, :].
// Stage 1: workgroup (v, chunk) scans its chunk of positions (16 position groups x 16 column lanes, fixed-order LDS
// reduction) into part[chunk][v][:]; stage 2 sums the chunks in order. Bitwise reproducible.
__global__ __launch_bounds__(256) void embedding_bwd_partial_kernel(const int64_t* __restrict__ idx,
                                                                    const float* __restrict__ dx, int n, int V, int E,
                                                                    int chunk_len, float* __restrict__ part) {
    __shared__ float red[16][17];
    const int v = blockIdx.x;
    const int p0 = blockIdx.y * chunk_len, p1 = min(n, p0 + chunk_len);
    const int pg = threadIdx.x >> 4, el = threadIdx.x & 15;
    for (int e0 = 0; e0 < E; e0 += 16) {
        const int e = e0 + el;
        float s = 0.f;
        if (e < E) {
            for (int pos = p0 + pg; pos < p1; pos += 16)
                if (idx[pos] == (int64_t)v) s += dx[(long)pos * E + e];
        }
        __syncthreads();
        red[pg][el] = s;
        __syncthreads();
        if (pg == 0 && e < E) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][el];
            part[((long)blockIdx.y * V + v) * E + e] = t;
        }
    }
}
__global__ __launch_bounds__(256) void embedding_bwd_final_kernel(const float* __restrict__ part, int nchunk, int VE,
                                                                  float beta, float* __restrict__ dtable) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= VE) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += part[(long)c * VE + i];
    dtable[i] = (beta != 0.f) ? s + beta * dtable[i] : s;
}

int fill_widths(ConvWidths& cw, const cgn_conv_widths* d, bool need_w, bool need_dw) {
    if (!d || d->n < 1 || d->n > MAXW) return CGN_ERR_INVALID_ARG;
    if (d->act != ACT_RELU && d->act != ACT_TANH) return CGN_ERR_INVALID_ARG;
    if (d->layout != 0 && d->layout != 1) return CGN_ERR_INVALID_ARG;
    cw.n = d->n; cw.F = d->F; cw.act = d->act; cw.layout = d->layout;
    for (int i = 0; i < MAXW; ++i) {
        cw.W[i] = nullptr; cw.dW[i] = nullptr; cw.db[i] = nullptr;
        cw.K[i] = 0; cw.Co[i] = 0; cw.col0[i] = 0;
    }
    for (int i = 0; i < d->n; ++i) {
        if (d->K[i] < 1 || d->Co[i] < 1 || d->col0[i] < 0 || d->col0[i] + d->Co[i] > d->F) return CGN_ERR_INVALID_ARG;
        if (need_w && !d->W[i]) return CGN_ERR_INVALID_ARG;
        if (need_dw && !d->dW[i]) return CGN_ERR_INVALID_ARG;
        cw.K[i] = d->K[i]; cw.Co[i] = d->Co[i]; cw.col0[i] = d->col0[i];
        cw.W[i] = d->W[i]; cw.dW[i] = d->dW[i]; cw.db[i] = d->db[i];
    }
    return CGN_OK;
}

}  // namespace

extern "C" int cgn_act_maxpool_f32(void* stream, const float* conv, int32_t B, int32_t P, int32_t Co, int32_t act,
                                   float* feat, int64_t ldf, int32_t col0, int32_t* arg, int64_t lda) {
    CGN_CHECK_ARG(B >= 0 && P >= 1 && Co >= 1 && col0 >= 0 && (act == ACT_RELU || act == ACT_TANH));
    if (B == 0) return CGN_OK;
    CGN_CHECK_ARG(conv && feat && arg && ldf >= col0 + Co && lda >= col0 + Co);
    int blocks = cgn_cdiv((long)B * Co, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, conv, B, P, Co, act, feat,
                       (long)ldf, col0, arg, (long)lda);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_act_maxpool_group_f32(void* stream, int32_t n, const float* const* conv, const int32_t* P,
                                         const int32_t* Co, const int32_t* col0, int32_t B, int32_t act, float* feat,
                                         int64_t ldf, int32_t* arg, int64_t lda) {
    CGN_CHECK_ARG(n >= 1 && n <= 8 && conv && P && Co && col0 && B >= 0 && (act == ACT_RELU || act == ACT_TANH));
    if (B == 0) return CGN_OK;
    CGN_CHECK_ARG(feat && arg);
    PoolGroup g;
    int end = 0;
    for (int i = 0; i < n; ++i) {
        CGN_CHECK_ARG(conv[i] && P[i] >= 1 && Co[i] >= 1 && col0[i] >= 0 && ldf >= col0[i] + Co[i] && lda >= col0[i] + Co[i]);
        g.conv[i] = conv[i]; g.P[i] = P[i]; g.Co[i] = Co[i]; g.col0[i] = col0[i];
        end += Co[i];
        g.end[i] = end;
    }
    g.n = n;
    int blocks = cgn_cdiv((long)B * end, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(act_maxpool_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, B, act, feat,
                       (long)ldf, arg, (long)lda);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_pool_bwd_scatter_f32(void* stream, const cgn_conv_widths* widths, const float* dfeat,
                                        const float* feat, const int32_t* arg, int32_t B, int32_t L,
                                        float* const* dconv) {
    CGN_CHECK_ARG(B >= 0 && L >= 1 && dconv);
    if (B == 0) return CGN_OK;
    ConvWidths cw;
    int rc = fill_widths(cw, widths, false, false);
    if (rc) return rc;
    CGN_CHECK_ARG(dfeat && arg && (cw.act == ACT_RELU || feat));
    ScatterGroup g;
    long end = 0, max_size = 0;
    for (int i = 0; i < cw.n; ++i) {
        CGN_CHECK_ARG(dconv[i] && cw.K[i] <= L && (uintptr_t)dconv[i] % 16 == 0);
        g.dconv[i] = dconv[i];
        g.P[i] = L - cw.K[i] + 1; g.Co[i] = cw.Co[i]; g.col0[i] = cw.col0[i];
        const long size = (long)B * g.P[i] * g.Co[i];
        CGN_CHECK_ARG(size < (1L << 31) - 4096L * 1024L);  // 32-bit element index inside a width (incl. the grid stride)
        if (size > max_size) max_size = size;
        end += size;
        g.end[i] = end;
    }
    g.n = cw.n; g.F = cw.F; g.act = cw.act;
    long blocks = (max_size / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pool_bwd_scatter_kernel, dim3((unsigned)blocks, (unsigned)g.n), dim3(256), 0, (hipStream_t)stream, g,
                       dfeat, feat, arg);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_conv_col2im_f32(void* stream, int32_t n, const float* const* dcols, const int32_t* K, int32_t B,
                                   int32_t L, int32_t E, float* dx, int64_t sent_stride) {
    CGN_CHECK_ARG(n >= 1 && n <= MAXW && dcols && K && B >= 0 && L >= 1 && E >= 1);
    if (B == 0) return CGN_OK;
    CGN_CHECK_ARG(dx && sent_stride >= (int64_t)L * E);
    Col2imGroup g;
    for (int i = 0; i < n; ++i) {
        CGN_CHECK_ARG(dcols[i] && K[i] >= 1 && K[i] <= L);
        g.dcols[i] = dcols[i];
        g.K[i] = K[i];
    }
    g.n = n;
    long blocks = ((long)B * L * E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(conv_col2im_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, B, L, E, dx,
                       (long)sent_stride);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_conv_bwd_dx_f32(void* stream, const cgn_conv_widths* widths, const float* dfeat, const float* feat,
                                   const int32_t* arg, int32_t B, int32_t L, int32_t E, float* dx, int64_t sent_stride) {
    CGN_CHECK_ARG(B >= 0 && L >= 1 && E >= 1);
    if (B == 0) return CGN_OK;
    ConvWidths cw;
    int rc = fill_widths(cw, widths, true, false);
    if (rc) return rc;
    CGN_CHECK_ARG(dfeat && arg && dx && sent_stride >= (int64_t)L * E && (cw.act == ACT_RELU || feat));
    CGN_CHECK_ARG(B <= 65535);
    dim3 grid(cgn_cdiv((long)L * E, 256), B);
    hipLaunchKernelGGL(conv_bwd_dx_kernel, grid, dim3(256), 0, (hipStream_t)stream, cw, dfeat, feat, arg, L, E, dx,
                       (long)sent_stride);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_conv_bwd_dw_f32(void* stream, const cgn_conv_widths* widths, const float* dfeat, const float* feat,
                                   const int32_t* arg, const float* x, int64_t sent_stride, int32_t B, int32_t L,
                                   int32_t E, float beta) {
    CGN_CHECK_ARG(B >= 0 && L >= 1 && E >= 1);
    ConvWidths cw;
    int rc = fill_widths(cw, widths, false, true);
    if (rc) return rc;
    CGN_CHECK_ARG(dfeat && arg && x && sent_stride >= (int64_t)L * E && (cw.act == ACT_RELU || feat));
    int maxco = 0, maxke = 0;
    for (int i = 0; i < cw.n; ++i) {
        if (cw.Co[i] > maxco) maxco = cw.Co[i];
        if (cw.K[i] * E > maxke) maxke = cw.K[i] * E;
    }
    CGN_CHECK_ARG(maxco <= 65535);
    dim3 grid(cgn_cdiv(maxke, 64), maxco, cw.n);
    if (B >= 512)
        hipLaunchKernelGGL(conv_bwd_dw_kernel<16>, grid, dim3(1024), 0, (hipStream_t)stream, cw, dfeat, feat, arg, x,
                           (long)sent_stride, B, L, E, beta);
    else
        hipLaunchKernelGGL(conv_bwd_dw_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, cw, dfeat, feat, arg, x,
                           (long)sent_stride, B, L, E, beta);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_embedding_bwd_f32(void* stream, const int64_t* idx, const float* dx, int32_t n, int32_t V, int32_t E,
                                     float beta, float* dtable, float* workspace, int64_t workspace_bytes) {
    CGN_CHECK_ARG(n >= 0 && V >= 1 && E >= 1 && V <= 65535);
    CGN_CHECK_ARG(idx && dx && dtable && workspace);
    int nchunk = cgn_cdiv(n, 512);
    if (nchunk > 64) nchunk = 64;
    if (nchunk < 1) nchunk = 1;
    if ((long)nchunk * V * E * 4 > workspace_bytes) return CGN_ERR_WORKSPACE;
    const int chunk_len = cgn_cdiv(n > 0 ? n : 1, nchunk);
    hipLaunchKernelGGL(embedding_bwd_partial_kernel, dim3(V, nchunk), dim3(256), 0, (hipStream_t)stream, idx, dx, n, V,
                       E, chunk_len, workspace);
    CGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(embedding_bwd_final_kernel, dim3(cgn_cdiv((long)V * E, 256)), dim3(256), 0, (hipStream_t)stream,
                       workspace, nchunk, V * E, beta, dtable);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
