// The launch-bound chains of the step's main stream as single kernels (round 3).
//
// Between its large products the encoder/generator step (model/cgn.py:1049-1154) runs chains of tiny dependent
// kernels, each worth >= 4.5 us of stream time whatever it computes (un-profiled phase stamps, profiles/r03_step_stamps_*):
//   * the code head  cgn.py:83-103, :1076-1080 + generator.py:54-56: [c_n fwd | c_n bwd] -> mu|logvar -> z, std, KLD ->
//     [z,c] -> its per-sentence contribution to the generator's layer-0 gates (2 copies, GEMM, split-K reduce,
//     reparametrisation, GEMM: 6 launches) and its backward (time sum, GEMM, reduce, reparam backward, GEMM, 2 copies);
//   * the c-reconstruction head behind the frozen discriminator's convolution products, cgn.py:1114-1118 (max-over-time,
//     label cast, fc + CE, its backward, scatter of the pooled gradient: 5 launches);
//   * col2im of the convolution's input gradient + the zero last timestep of cgn.py:1108 + the row dot product the fused
//     softmax backward needs (3 launches).
// One workgroup per sentence; everything a sentence needs between two products stays in LDS. Sums over the batch (KLD,
// mean CE) are formed in fixed order by the last workgroup to finish (ticket), so results are bitwise reproducible.
#include "common.h"

namespace {

constexpr int FZ_NT = 512;

// Batch reduction without a second launch: every workgroup publishes its row value (agent-scope store), checks in on
// the ticket, and the LAST one sums rows[0..n) in a fixed order and re-arms the ticket (the scratch is zeroed once by the
// caller). Returns true in every thread of that last workgroup.
__device__ __forceinline__ bool publish_and_check_last(float* rows, unsigned* ticket, int row, float value, unsigned n,
                                                       int* s_flag) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(rows + row, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (old == n - 1u) ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}
// fixed-order sum of rows[0..n) by wave 0 of the last workgroup (lane l takes l, l+64, ...; then the butterfly)
__device__ __forceinline__ float ordered_sum_wave0(const float* rows, unsigned n) {
    float s = 0.f;
    for (unsigned i = threadIdx.x; i < n; i += 64)
        s += __hip_atomic_load(rows + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return wave_sum(s);
}

// ---- code head, forward -------------------------------------------------------------------------------------------
constexpr int CH_MAX_2HE = 2048, CH_MAX_2Z = 256, CH_MAX_ZC = 144;
struct CodeHeadFwd {
    const float* cf;   // final cell state of the forward sequence, element (b, j) at cf[b * ldc + j]   (encoder.py:40-46)
    const float* cr;   // ... of the reverse sequence
    long ldc;
    const float* W2;   // [2Z, 2He] = e2mu.weight | e2logvar.weight stacked, row stride ldw2            (cgn.py:37-38)
    long ldw2;
    const float* b2;   // [2Z]
    const float* eps;  // [B, Z]                                                                         (cgn.py:96)
    const float* c;    // [B, C]
    const float* Wzc;  // nullable: generator W_ih_l0[:, E:], 4Hg rows of Z+C, row stride ldzc          (generator.py:54-56)
    long ldzc;
    float* ctx;        // [B, 2He] out (the encoder's context; the weight gradients of e2mu/e2logvar read it)
    float* mulv;       // [B, 2Z]  out
    float* zc;         // [B, Z+C] out
    float* stdv;       // [B, Z]   out
    float* kld;        // [1]      out: mean_b(-0.5 sum_z(logvar - mu^2 - e^logvar + 1))                 (cgn.py:1076)
    float* zg;         // [B, G4]  out (nullable with Wzc): [z,c] . W_ih_l0[:, E:]^T, added to the gates at every timestep
    float* rows;       // scratch: B floats
    unsigned* ticket;  // scratch: one counter, zero between launches
    int B, He, Z, C, G4;
};

// mu | logvar: a wave takes FOUR outputs at a time, lanes along k (16-byte loads of the weight rows); all loads of the
// four rows are issued before the first is consumed (one output at a time was a chain of L2 round trips: 16 outputs x
// ~1 us per wave). KI = quads per lane and row.
template <int KI>
__device__ __forceinline__ void mulv_phase(const CodeHeadFwd& p, const float* ctx_s, float* mulv_s, int b, int K2) {
    constexpr int OB = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nout = 2 * p.Z;
    for (int n0 = wave * OB; n0 < nout; n0 += (FZ_NT / 64) * OB) {
        f32x4 w[OB][KI];
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const float* wr = p.W2 + (long)min(n0 + o, nout - 1) * p.ldw2;
#pragma unroll
            for (int i = 0; i < KI; ++i) {
                const int k = lane * 4 + i * 256;
                w[o][i] = *reinterpret_cast<const f32x4*>(wr + (k < K2 ? k : 0));
            }
        }
        float acc[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) acc[o] = 0.f;
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < K2) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(ctx_s + k);
#pragma unroll
                for (int o = 0; o < OB; ++o)
                    acc[o] += w[o][i][0] * x[0] + w[o][i][1] * x[1] + w[o][i][2] * x[2] + w[o][i][3] * x[3];
            }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const float v = wave_sum(acc[o]);
            const int n = n0 + o;
            if (lane == 0 && n < nout) {
                const float r = v + p.b2[n];
                mulv_s[n] = r;
                p.mulv[(long)b * nout + n] = r;
            }
        }
    }
}

// the tiled [z,c] input's share of the generator's layer-0 gates, once per sentence: 4 lanes per gate row (16-byte
// loads), FOUR rows at a time per lane group; the pad columns of a row meet the zeros behind zc_s[Z+C). KQ = quads per lane
template <int KQ>
__device__ __forceinline__ void zg_phase(const CodeHeadFwd& p, const float* zc_s, int b, int ZC) {
    constexpr int OB = 4;
    const int tid = threadIdx.x, sub = tid & 3;
    for (int n0 = (tid >> 2) * OB; n0 < p.G4; n0 += (FZ_NT / 4) * OB) {
        f32x4 w[OB][KQ];
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            const float* wr = p.Wzc + (long)min(n0 + o, p.G4 - 1) * p.ldzc;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const int k0 = sub * 4 + i * 16;
                w[o][i] = *reinterpret_cast<const f32x4*>(wr + (k0 < ZC ? k0 : 0));
            }
        }
#pragma unroll
        for (int o = 0; o < OB; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < KQ; ++i) {
                const int k0 = sub * 4 + i * 16;
                if (k0 < ZC) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(zc_s + k0);
                    acc += w[o][i][0] * x[0] + w[o][i][1] * x[1] + w[o][i][2] * x[2] + w[o][i][3] * x[3];
                }
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (sub == 0 && n0 + o < p.G4) p.zg[(long)b * p.G4 + n0 + o] = acc;
        }
    }
}

__global__ __launch_bounds__(FZ_NT) void code_head_fwd_kernel(const CodeHeadFwd p) {
    __shared__ __attribute__((aligned(16))) float ctx_s[CH_MAX_2HE];
    __shared__ float mulv_s[CH_MAX_2Z];
    __shared__ __attribute__((aligned(16))) float zc_s[CH_MAX_ZC];
    __shared__ float red[FZ_NT / 64 + 1];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int He = p.He, Z = p.Z, C = p.C, K2 = 2 * He, ZC = Z + C;
    // context = cat(c_n forward, c_n reverse)                                                     (encoder.py:43-46)
    for (int j = tid; j < K2; j += FZ_NT) {
        const float v = j < He ? p.cf[(long)b * p.ldc + j] : p.cr[(long)b * p.ldc + j - He];
        ctx_s[j] = v;
        p.ctx[(long)b * K2 + j] = v;
    }
    for (int k = tid; k < CH_MAX_ZC; k += FZ_NT) zc_s[k] = 0.f;
    __syncthreads();
    // mu | logvar                                                                                  (cgn.py:91-92)
    if (K2 <= 1024) mulv_phase<4>(p, ctx_s, mulv_s, b, K2);
    else mulv_phase<8>(p, ctx_s, mulv_s, b, K2);
    __syncthreads();
    // z = eps * exp(0.5 logvar) + mu, [z, c], this sentence's KLD term                              (cgn.py:94-101, :1076-1080)
    float term = 0.f;
    if (tid < Z) {
        const float mu = mulv_s[tid], lv = mulv_s[Z + tid];
        const float sd = expf(0.5f * lv);
        const float z = p.eps[(long)b * Z + tid] * sd + mu;
        zc_s[tid] = z;
        p.zc[(long)b * ZC + tid] = z;
        p.stdv[(long)b * Z + tid] = sd;
        term = lv - mu * mu - expf(lv) + 1.f;
    } else if (tid < ZC) {
        const float cv = p.c[(long)b * C + tid - Z];
        zc_s[tid] = cv;
        p.zc[(long)b * ZC + tid] = cv;
    }
    term = block_sum<FZ_NT>(term, red);  // (its barriers also publish zc_s)
    // the tiled [z,c] input's share of the generator's layer-0 gates, once per sentence: 4 lanes per gate row, 16-byte
    // loads; the pad columns of a row are multiplied by the zeros behind zc_s[Z+C)
    if (p.Wzc) {
        if (ZC <= 80) zg_phase<5>(p, zc_s, b, ZC);
        else zg_phase<9>(p, zc_s, b, ZC);
    }
    if (publish_and_check_last(p.rows, p.ticket, b, -0.5f * term, (unsigned)p.B, &s_flag)) {
        if (wave == 0) {
            const float s = ordered_sum_wave0(p.rows, (unsigned)p.B);
            if (lane == 0) {
                p.kld[0] = s / (float)p.B;
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- code head, backward ------------------------------------------------------------------------------------------
constexpr int CH_MAX_G4 = 4096;
struct CodeHeadBwd {
    const float* dg0;   // [B, T, G4] gradients of the generator's layer-0 gates
    const float* Wzc;   // as above
    long ldzc;
    const float* mulv;  // [B, 2Z]
    const float* eps;   // [B, Z]
    const float* dyn_w; // nullable device scalar overriding kld_w (graph replay)
    const float* W2;
    long ldw2;
    float* dsum;        // [B, G4]  out: sum_t dg0 (dW_ih_l0[:, E:] = dsum^T [z,c] reads it)
    float* dzc;         // [B, Z+C] out: d(79 CE + recon)/d[z,c]; its first Z columns are g of cgn.py:1099, :1142
    float* dmulv;       // [B, 2Z]  out
    float* dcf;         // [B, He]  out: gradient wrt the forward sequence's final cell state
    float* dcr;         // [B, He]  out
    float kld_w;        // kld_coef / B
    int B, T, He, Z, C, G4;
};

__global__ __launch_bounds__(FZ_NT) void code_head_bwd_kernel(const CodeHeadBwd p) {
    __shared__ __attribute__((aligned(16))) float dsum_s[CH_MAX_G4];
    __shared__ __attribute__((aligned(16))) float part_s[FZ_NT / 64][CH_MAX_ZC];
    __shared__ float dzc_s[CH_MAX_ZC];
    __shared__ float dmulv_s[CH_MAX_2Z];
    __shared__ __attribute__((aligned(16))) float half_s[CH_MAX_2HE];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int He = p.He, Z = p.Z, ZC = p.Z + p.C, G4 = p.G4, T = p.T;
    // dsum = sum over time of the gate gradients (the [z,c] input is the same at every timestep): a thread owns four
    // consecutive gate columns, 8 timesteps of 16-byte loads in flight (G4 % 4 == 0)
    for (int n = tid * 4; n < G4; n += FZ_NT * 4) {
        const float* src = p.dg0 + (long)b * T * G4 + n;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < T; t0 += 8) {  // fixed summation order
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)min(t0 + u, T - 1) * G4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u < T) s += v[u];
        }
        *reinterpret_cast<f32x4*>(dsum_s + n) = s;
        *reinterpret_cast<f32x4*>(p.dsum + (long)b * G4 + n) = s;
    }
    __syncthreads();
    // d[z,c][k] = sum_n dsum[n] W[n, k]. A weight row is Z + C contiguous floats: LPR lanes take one quad each (a coalesced
    // row segment), 64 / LPR rows per wave instruction, 16 instructions in flight; the lanes that hold the same quad of
    // different rows are added at the end (shuffles), the 8 waves' row slices through LDS. Fixed summation order.
    {
        const int nq = (ZC + 3) / 4;                       // quads per row (the last may reach into the row's zero padding)
        const int lpr = nq <= 16 ? 16 : (nq <= 32 ? 32 : 64);
        const int rpi = 64 / lpr;                          // rows per wave instruction
        const int qi = lane % lpr, ri = lane / lpr;
        const bool qok = qi < nq;
        const int ns = (G4 + FZ_NT / 64 - 1) / (FZ_NT / 64);
        const int n0 = wave * ns, n1 = min(G4, n0 + ns);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int nb = n0; nb < n1; nb += 16 * rpi) {
            f32x4 w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n = min(nb + u * rpi + ri, n1 - 1);
                w[u] = *reinterpret_cast<const f32x4*>(p.Wzc + (long)n * p.ldzc + (qok ? qi * 4 : 0));
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n = nb + u * rpi + ri;
                const float dn = (n < n1 && qok) ? dsum_s[n] : 0.f;
                acc += w[u] * dn;
            }
        }
        for (int o = lpr; o < 64; o <<= 1) {  // add the rows that share a quad
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        }
        if (ri == 0 && qok) *reinterpret_cast<f32x4*>(&part_s[wave][qi * 4]) = acc;
    }
    __syncthreads();
    if (tid < ZC) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FZ_NT / 64; ++w) s += part_s[w][tid];
        dzc_s[tid] = s;
        p.dzc[(long)b * ZC + tid] = s;
    }
    __syncthreads();
    // through z = eps * exp(0.5 logvar) + mu, plus the KLD term's own gradient                     (cgn.py:1146-1148)
    if (tid < Z) {
        const float w = p.dyn_w ? p.dyn_w[0] : p.kld_w;
        const float mu = p.mulv[(long)b * 2 * Z + tid], lv = p.mulv[(long)b * 2 * Z + Z + tid];
        const float gz = dzc_s[tid];
        const float dmu = gz + w * mu;
        const float dlv = gz * p.eps[(long)b * Z + tid] * 0.5f * expf(0.5f * lv) + w * 0.5f * (expf(lv) - 1.f);
        dmulv_s[tid] = dmu;
        dmulv_s[Z + tid] = dlv;
        p.dmulv[(long)b * 2 * Z + tid] = dmu;
        p.dmulv[(long)b * 2 * Z + Z + tid] = dlv;
    }
    __syncthreads();
    // d(context)[j] = sum_n d[mu|logvar][n] W2[n, j], split into the two final cell states' gradients. A thread owns four
    // consecutive j (16-byte loads along the weight rows), the two halves of the workgroup take the two halves of n
    // (16 rows in flight each) and meet in LDS; (2He) % 4 == 0, 2He <= 4 * 256.
    {
        const int K2 = 2 * He, N2 = 2 * Z;
        const int half = tid >> 8, jt = (tid & 255) * 4;
        const int nh = (N2 + 1) / 2;
        const int nb0 = half * nh, nb1 = min(N2, nb0 + nh);
        for (int j = jt; j < K2; j += 1024) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int nb = nb0; nb < nb1; nb += 16) {
                f32x4 w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    w[u] = *reinterpret_cast<const f32x4*>(p.W2 + (long)min(nb + u, nb1 - 1) * p.ldw2 + j);
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (nb + u < nb1) acc += w[u] * dmulv_s[nb + u];
            }
            if (half == 1) *reinterpret_cast<f32x4*>(half_s + j) = acc;
            __syncthreads();  // (uniform: every thread runs the same number of j trips)
            if (half == 0) {
                acc += *reinterpret_cast<const f32x4*>(half_s + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int jj = j + e;
                    if (jj < He) p.dcf[(long)b * He + jj] = acc[e];
                    else p.dcr[(long)b * He + jj - He] = acc[e];
                }
            }
            __syncthreads();
        }
    }
}

// ---- the c-reconstruction head behind the frozen text-CNN ---------------------------------------------------------
constexpr int TH_MAX_F = 1024, TH_NT = 256;
struct TextCnnHead {
    const float* conv[8];  // per filter width: [B * P, Co] convolution outputs, bias included
    float* dconv[8];       // out: same shape, the pooled gradient spread back (one non-zero per (sentence, map))
    int P[8], Co[8], end[8];  // end: running sum of Co
    int n, F;
    const float* W;        // fc1.weight [2, F]                                      (discriminator_sentiment.py:49, :77)
    const float* bias;     // [2]
    const float* c;        // labels as floats, element b at c[b * ldc]: y = (int64) c   (cgn.py:1117 c.long())
    long ldc;
    float w_ce;            // lambda_c: dlogits = w_ce / B * (softmax - onehot)
    float* feat;           // [B, F] out: relu(max over time)
    int32_t* arg;          // [B, F] out: the kept position, -1 where the ReLU gate is closed
    float* stats;          // out: stats[0] = mean_b CE
    float* rows;
    unsigned* ticket;
    int B;
};

__global__ __launch_bounds__(TH_NT) void textcnn_head_kernel(const TextCnnHead p) {
    __shared__ float feat_s[TH_MAX_F];
    __shared__ int arg_s[TH_MAX_F];
    __shared__ float red[TH_NT / 64 + 1];
    __shared__ int s_flag;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int F = p.F;
    // ReLU + max over time with argmax (first maximum wins, as torch's max)          (discriminator_sentiment.py:66-70)
    for (int f = tid; f < F; f += TH_NT) {
        int j = 0;
        while (j + 1 < p.n && f >= p.end[j]) ++j;
        const int co = f - (j ? p.end[j - 1] : 0);
        const int P = p.P[j], Co = p.Co[j];
        const float* src = p.conv[j] + (long)b * P * Co + co;
        float best = src[0];
        int bi = 0;
        for (int t0 = 1; t0 < P; t0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)min(t0 + u, P - 1) * Co];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (t0 + u < P && v[u] > best) {
                    best = v[u];
                    bi = t0 + u;
                }
            }
        }
        const float y = fmaxf(best, 0.f);
        const int a = best > 0.f ? bi : -1;
        feat_s[f] = y;
        arg_s[f] = a;
        p.feat[(long)b * F + f] = y;
        p.arg[(long)b * F + f] = a;
    }
    __syncthreads();
    // fc1 (eval mode: no dropout) + 2-class CE against c.long()                                      (cgn.py:1114-1118)
    float a0 = 0.f, a1 = 0.f;
    for (int f = tid; f < F; f += TH_NT) {
        a0 += feat_s[f] * p.W[f];
        a1 += feat_s[f] * p.W[F + f];
    }
    const float s0 = block_sum<TH_NT>(a0, red) + p.bias[0];
    const float s1 = block_sum<TH_NT>(a1, red) + p.bias[1];
    const float mx = fmaxf(s0, s1);
    const float lse = mx + logf(expf(s0 - mx) + expf(s1 - mx));
    const float l0 = s0 - lse, l1 = s1 - lse;
    const float q0 = expf(l0), q1 = expf(l1);
    const int64_t yb = (int64_t)p.c[(long)b * p.ldc];
    const float ce = (yb == 1) ? -l1 : -l0;
    const float d0 = p.w_ce * (q0 - (yb == 0 ? 1.f : 0.f)) / (float)p.B;
    const float d1 = p.w_ce * (q1 - (yb == 1 ? 1.f : 0.f)) / (float)p.B;
    // back through fc1 and the pooling: the gradient of map f lands on its kept position, zeros elsewhere
    for (int f = tid; f < F; f += TH_NT) {
        int j = 0;
        while (j + 1 < p.n && f >= p.end[j]) ++j;
        const int co = f - (j ? p.end[j - 1] : 0);
        const int P = p.P[j], Co = p.Co[j];
        const float g = d0 * p.W[f] + d1 * p.W[F + f];
        const int a = arg_s[f];
        float* dst = p.dconv[j] + (long)b * P * Co + co;
        for (int t = 0; t < P; ++t) dst[(long)t * Co] = (t == a) ? g : 0.f;
    }
    if (publish_and_check_last(p.rows, p.ticket, b, ce, (unsigned)p.B, &s_flag)) {
        if (tid < 64) {
            const float s = ordered_sum_wave0(p.rows, (unsigned)p.B);
            if (tid == 0) {
                p.stats[0] = s / (float)p.B;
                __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- col2im of the convolution's input gradient + zero rows + row dot product ------------------------------------
struct Col2imRow {
    const float* dcols[8];
    int K[8];
    int n;
};
// dx[(b*T + t), e] = beta * dx + sum_w sum_k dcols_w[(b, t-k), k*E + e] for t < L, (beta * dx +) 0 for L <= t < T (the
// soft embedding's last timestep is dropped, cgn.py:1108); rowdot[b*T + t] = sum_e dx[.., e] * other[.., e] (the
// softmax Jacobian's  sum_v p_v dp_v  of the fused backward epilogue, cgn.py:1103-1106 reversed). One workgroup per row,
// one thread per element (all of an element's window loads in flight at once; a wave-per-row walk over e was a chain
// of L2 round trips: 16 us against col2im's 6), the row's dot product through LDS.
constexpr int C2R_MAX_WAVES = 16;
__global__ __launch_bounds__(64 * C2R_MAX_WAVES) void col2im_rowdot_kernel(const Col2imRow g, int B, int L, int T, int E,
                                                                            float beta, float* __restrict__ dx,
                                                                            const float* __restrict__ other, long ldo,
                                                                            float* __restrict__ rowdot) {
    __shared__ float red[C2R_MAX_WAVES];
    const int r = blockIdx.x;
    const int b = r / T, t = r % T;
    float dot = 0.f;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float acc = 0.f;
        if (t < L) {
            for (int j = 0; j < g.n; ++j) {
                const int K = g.K[j], P = L - K + 1;
                const long ld = (long)K * E;
                const float* base = g.dcols[j] + (long)b * P * ld + e;
                for (int k0 = 0; k0 < K; k0 += 8) {  // 8 loads in flight, clamped addresses
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int k = k0 + u, pp = t - k;
                        const bool ok = k < K && pp >= 0 && pp < P;
                        v[u] = ok ? base[(long)pp * ld + (long)k * E] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u];
                }
            }
        }
        const long di = (long)r * E + e;
        if (beta != 0.f) acc += beta * dx[di];
        dx[di] = acc;
        dot += acc * other[(long)r * ldo + e];
    }
    dot = wave_sum(dot);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w];  // fixed order
        rowdot[r] = s;
    }
}

// ---- log-softmax / CE from the vocabulary projection's per-chunk statistics ------------------------------------------
// (model/cgn.py:1086-1092 behind generator.py:61; the product's CGN_EPI_LINEAR_STATS epilogue left, per row and 64-column
// chunk, the maximum of logits/tau and the exp-sum around it.) One wave per row: lse = M + log(sum_c s_c e^{m_c - M}),
// nll = lse - logits[row, target]/tau - the only logits read here; mean NLL in fixed order by the last workgroup.
constexpr int LS_NT = 256;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(LS_NT) void softmax_nll_from_stats_kernel(const float* __restrict__ stats, long ld_stats,
                                                                       int chunks, const float* __restrict__ logits,
                                                                       long ld, const long* __restrict__ target, int M,
                                                                       float inv_tau, const float* __restrict__ dyn,
                                                                       float* __restrict__ lse, float* __restrict__ nll,
                                                                       float* __restrict__ loss, float* rows,
                                                                       unsigned* ticket) {
    __shared__ float part[LS_NT / 64];
    __shared__ int s_flag;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * (LS_NT / 64) + wave;
    const float sc = dyn ? dyn[0] : inv_tau;
    float v = 0.f;
    if (row < M) {
        const f32x2* st = reinterpret_cast<const f32x2*>(stats + (long)row * ld_stats * 2);
        float m = -INFINITY, s = 0.f;
        if (chunks <= 64 * 8) {  // one pass: the row's statistics in registers (all loads in flight at once)
            f32x2 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = lane + 64 * u;
                q[u] = st[c < chunks ? c : 0];
                if (c >= chunks) q[u][0] = -INFINITY, q[u][1] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmaxf(m, q[u][0]);
            m = wave_max(m);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += q[u][1] * __expf(q[u][0] - m);  // (exp(-inf) = 0 for the empty slots)
        } else {
            for (int c = lane; c < chunks; c += 64) m = fmaxf(m, st[c][0]);
            m = wave_max(m);
            for (int c = lane; c < chunks; c += 64) {
                const f32x2 q = st[c];
                s += q[1] * __expf(q[0] - m);
            }
        }
        s = wave_sum(s);
        const float l = m + logf(s);
        if (lane == 0) {
            const float n = l - logits[(long)row * ld + target[row]] * sc;
            lse[row] = l;
            nll[row] = n;
            v = n;
        }
    }
    if (lane == 0) part[wave] = v;
    __syncthreads();
    float bsum = 0.f;
#pragma unroll
    for (int w = 0; w < LS_NT / 64; ++w) bsum += part[w];
    if (publish_and_check_last(rows, ticket, blockIdx.x, bsum, gridDim.x, &s_flag)) {
        if (wave == 0) {
            const float s = ordered_sum_wave0(rows, gridDim.x);
            if (lane == 0) {
                loss[0] = s / (float)M;
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" int cgn_code_head_fwd_f32(void* stream, const cgn_code_head_fwd* d) {
    CGN_CHECK_ARG(d != nullptr);
    CGN_CHECK_ARG(d->B > 0 && d->He > 0 && d->Z > 0 && d->C >= 0);
    CGN_CHECK_ARG(2 * d->He <= CH_MAX_2HE && 2 * d->Z <= CH_MAX_2Z && d->Z + d->C <= CH_MAX_ZC - 4 && d->Z <= FZ_NT &&
                  d->Z + d->C <= FZ_NT);
    CGN_CHECK_ARG(d->He % 2 == 0 && d->ldw2 % 4 == 0 && d->ldw2 >= 2 * d->He && d->ldc >= d->He);
    CGN_CHECK_ARG(d->cf && d->cr && d->W2 && d->b2 && d->eps && d->ctx && d->mulv && d->zc && d->stdv && d->kld &&
                  d->scratch);
    CGN_CHECK_ARG(d->C == 0 || d->c != nullptr);
    CGN_CHECK_ARG((uintptr_t)d->W2 % 16 == 0 && (uintptr_t)d->scratch % 4 == 0);
    CGN_CHECK_ARG(d->scratch_bytes >= (int64_t)(d->B + 1) * 4);
    if (d->Wzc) {
        CGN_CHECK_ARG(d->zg != nullptr && d->G4 > 0 && d->ldzc % 4 == 0 && (uintptr_t)d->Wzc % 16 == 0 &&
                      d->ldzc >= (d->Z + d->C + 3) / 4 * 4);
    }
    CodeHeadFwd p;
    p.cf = d->cf; p.cr = d->cr; p.ldc = d->ldc; p.W2 = d->W2; p.ldw2 = d->ldw2; p.b2 = d->b2; p.eps = d->eps; p.c = d->c;
    p.Wzc = d->Wzc; p.ldzc = d->ldzc; p.ctx = d->ctx; p.mulv = d->mulv; p.zc = d->zc; p.stdv = d->stdv; p.kld = d->kld;
    p.zg = d->zg; p.rows = (float*)d->scratch + 1; p.ticket = (unsigned*)d->scratch;
    p.B = d->B; p.He = d->He; p.Z = d->Z; p.C = d->C; p.G4 = d->G4;
    hipLaunchKernelGGL(code_head_fwd_kernel, dim3(d->B), dim3(FZ_NT), 0, S_(stream), p);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_code_head_bwd_f32(void* stream, const cgn_code_head_bwd* d) {
    CGN_CHECK_ARG(d != nullptr);
    CGN_CHECK_ARG(d->B > 0 && d->T > 0 && d->He > 0 && d->Z > 0 && d->C >= 0 && d->G4 > 0);
    CGN_CHECK_ARG(d->G4 <= CH_MAX_G4 && 2 * d->Z <= CH_MAX_2Z && d->Z + d->C <= CH_MAX_ZC && d->Z + d->C <= FZ_NT);
    CGN_CHECK_ARG(d->dg0 && d->Wzc && d->mulv && d->eps && d->W2 && d->dsum && d->dzc && d->dmulv && d->dcf && d->dcr);
    CGN_CHECK_ARG(d->ldzc >= (d->Z + d->C + 3) / 4 * 4 && d->ldzc % 4 == 0 && d->ldw2 >= 2 * d->He && d->ldw2 % 4 == 0);
    CGN_CHECK_ARG(d->G4 % 4 == 0 && d->He % 2 == 0 && 2 * d->He <= CH_MAX_2HE);
    CGN_CHECK_ARG((uintptr_t)d->Wzc % 16 == 0 && (uintptr_t)d->W2 % 16 == 0 && (uintptr_t)d->dg0 % 16 == 0 &&
                  (uintptr_t)d->dsum % 16 == 0);
    CodeHeadBwd p;
    p.dg0 = d->dg0; p.Wzc = d->Wzc; p.ldzc = d->ldzc; p.mulv = d->mulv; p.eps = d->eps; p.dyn_w = d->dyn_kld_w;
    p.W2 = d->W2; p.ldw2 = d->ldw2; p.dsum = d->dsum; p.dzc = d->dzc; p.dmulv = d->dmulv; p.dcf = d->dcf; p.dcr = d->dcr;
    p.kld_w = d->kld_w; p.B = d->B; p.T = d->T; p.He = d->He; p.Z = d->Z; p.C = d->C; p.G4 = d->G4;
    hipLaunchKernelGGL(code_head_bwd_kernel, dim3(d->B), dim3(FZ_NT), 0, S_(stream), p);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_textcnn_head_f32(void* stream, int32_t n, const float* const* conv, float* const* dconv,
                                    const int32_t* P, const int32_t* Co, int32_t B, const float* W, const float* bias,
                                    const float* c, int64_t ldc, float w_ce, float* feat, int32_t* arg, float* stats,
                                    void* scratch, int64_t scratch_bytes) {
    CGN_CHECK_ARG(n >= 1 && n <= 8 && conv && dconv && P && Co && B > 0);
    CGN_CHECK_ARG(W && bias && c && feat && arg && stats && scratch && ldc >= 1);
    CGN_CHECK_ARG((uintptr_t)scratch % 4 == 0 && scratch_bytes >= (int64_t)(B + 1) * 4);
    TextCnnHead p;
    int F = 0;
    for (int i = 0; i < n; ++i) {
        CGN_CHECK_ARG(conv[i] && dconv[i] && P[i] >= 1 && Co[i] >= 1);
        p.conv[i] = conv[i]; p.dconv[i] = dconv[i]; p.P[i] = P[i]; p.Co[i] = Co[i];
        F += Co[i];
        p.end[i] = F;
    }
    CGN_CHECK_ARG(F <= TH_MAX_F);
    p.n = n; p.F = F; p.W = W; p.bias = bias; p.c = c; p.ldc = ldc; p.w_ce = w_ce; p.feat = feat; p.arg = arg;
    p.stats = stats; p.rows = (float*)scratch + 1; p.ticket = (unsigned*)scratch; p.B = B;
    hipLaunchKernelGGL(textcnn_head_kernel, dim3(B), dim3(TH_NT), 0, S_(stream), p);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_conv_col2im_rowdot_f32(void* stream, int32_t n, const float* const* dcols, const int32_t* K,
                                          int32_t B, int32_t L, int32_t T, int32_t E, float beta, float* dx,
                                          const float* other, int64_t ldo, float* rowdot) {
    CGN_CHECK_ARG(n >= 1 && n <= 8 && dcols && K && B > 0 && L > 0 && T >= L && E > 0);
    CGN_CHECK_ARG(dx && other && rowdot && ldo >= E);
    CGN_CHECK_ARG((long)B * T < (1L << 31));
    Col2imRow g;
    for (int i = 0; i < n; ++i) {
        CGN_CHECK_ARG(dcols[i] && K[i] >= 1 && K[i] <= L);
        g.dcols[i] = dcols[i]; g.K[i] = K[i];
    }
    g.n = n;
    const int nt = 64 * (cgn_cdiv(E, 64) < C2R_MAX_WAVES ? cgn_cdiv(E, 64) : C2R_MAX_WAVES);
    hipLaunchKernelGGL(col2im_rowdot_kernel, dim3(B * T), dim3(nt), 0, S_(stream), g, B, L, T, E, beta, dx, other, (long)ldo,
                       rowdot);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_softmax_nll_from_stats_f32(void* stream, const float* stats, int64_t ld_stats, int32_t chunks,
                                              const float* logits, int64_t ld, const int64_t* target, int32_t M, int32_t V,
                                              float inv_tau, const float* dyn_inv_tau, float* lse, float* nll, float* loss,
                                              void* scratch, int64_t scratch_bytes) {
    CGN_CHECK_ARG(M > 0 && V > 0 && chunks == cgn_cdiv(V, 64) && ld_stats >= chunks && ld >= V);
    CGN_CHECK_ARG(stats && logits && target && lse && nll && loss && scratch && (uintptr_t)stats % 8 == 0);
    const int blocks = cgn_cdiv(M, LS_NT / 64);
    CGN_CHECK_ARG(scratch_bytes >= (int64_t)(blocks + 1) * 4);
    unsigned* ticket = (unsigned*)scratch;
    hipLaunchKernelGGL(softmax_nll_from_stats_kernel, dim3(blocks), dim3(LS_NT), 0, S_(stream), stats, (long)ld_stats, chunks,
                       logits, (long)ld, (const long*)target, M, inv_tau, dyn_inv_tau, lse, nll, loss, (float*)(ticket + 1),
                       ticket);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
