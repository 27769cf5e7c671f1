// Memory-bound glue kernels of the CGN step (gfx950): gathers, highway gate, reparameterisation + KLD, reductions,
// the 2-class discriminator head with CE / entropy, transposes and fused Adam. All are HBM/L2-streaming kernels:
// coalesced lanes along the contiguous dimension, float4 where alignment allows, fixed-order reductions.
#include "common.h"

namespace {

constexpr int PW_NT = 256;

inline int pw_blocks(long n, int per_block = PW_NT, int cap = 4096) {
    long b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- strided 2-D copy / scale -------------------------------------------------------------------------------
__global__ __launch_bounds__(PW_NT) void copy2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst,
                                                       long ldd, int rows, int cols, float scale, float beta) {
    const long total = (long)rows * cols;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < total; i += (long)gridDim.x * PW_NT) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const float v = (scale == 0.f) ? 0.f : src[(long)r * lds_ + c] * scale;  // scale 0 = strided zero-fill
        float* d = dst + (long)r * ldd + c;
        *d = (beta != 0.f) ? v + beta * (*d) : v;
    }
}

// ---- strided [B,T,H] copy (LSTM history slots <-> batch-first outputs) ------------------------------------------
__global__ __launch_bounds__(PW_NT) void copy3d_kernel(const float* __restrict__ src, long s_bs, long s_ts,
                                                       float* __restrict__ dst, long d_bs, long d_ts, int B, int T,
                                                       int H, float beta) {
    const long total = (long)B * T * H;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < total; i += (long)gridDim.x * PW_NT) {
        const int h = (int)(i % H);
        const long bt = i / H;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float* d = dst + (long)b * d_bs + (long)t * d_ts + h;
        const float v = src[(long)b * s_bs + (long)t * s_ts + h];
        *d = (beta != 0.f) ? v + beta * (*d) : v;
    }
}

// ---- 2-D transpose through LDS --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols,
                                                        float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) dst[(long)c * rows + r] = tile[tx][i];
    }
}

// ---- embedding row gather (frozen word table), optional elementwise mask (word dropout, generator.py:52) ------
__global__ __launch_bounds__(PW_NT) void gather_rows_kernel(const float* __restrict__ table, int V, int E,
                                                            const int64_t* __restrict__ idx, int n,
                                                            const float* __restrict__ mask, float* __restrict__ out,
                                                            long ldo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = blockIdx.x * (PW_NT / 64) + wave; row < n; row += gridDim.x * (PW_NT / 64)) {
        const int64_t id = idx[row];
        const bool ok = id >= 0 && id < V;  // out-of-range index: zeros, never an out-of-bounds read
        const float* src = table + (long)id * E;
        float* dst = out + (long)row * ldo;
        for (int e = lane; e < E; e += 64) {
            float v = ok ? src[e] : 0.f;
            if (mask) v *= mask[(long)row * E + e];
            dst[e] = v;
        }
    }
}

// ---- highway gate (selfModules/highway.py:38-44) ---------------------------------------------------------------
// pre is [N, 3*Sp] (Sp >= S, the 16-byte aligned block stride): columns [0,S) nonlinear pre-activation,
// [Sp,Sp+S) linear, [2Sp,2Sp+S) gate pre-activation.
__global__ __launch_bounds__(PW_NT) void highway_fwd_kernel(const float* __restrict__ pre, int N, int S, int Sp,
                                                            float* __restrict__ out, long ldo) {
    const long total = (long)N * S;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < total; i += (long)gridDim.x * PW_NT) {
        const int r = (int)(i / S), c = (int)(i % S);
        const float* p = pre + (long)r * 3 * Sp;
        const float nl = fmaxf(p[c], 0.f), li = p[Sp + c], g = cgn_sigmoid(p[2 * Sp + c]);
        out[(long)r * ldo + c] = g * nl + (1.f - g) * li;
    }
}

__global__ __launch_bounds__(PW_NT) void highway_bwd_kernel(const float* __restrict__ pre, int N, int S, int Sp,
                                                            const float* __restrict__ dy, long ldy,
                                                            float* __restrict__ dpre) {
    // runs over the padded width: the pad columns [S, Sp) of each block are written as zeros, so dpre can feed GEMMs
    // whose K or M axis spans the whole padded [N, 3*Sp] buffer (fused N|L|G products)
    const long total = (long)N * Sp;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < total; i += (long)gridDim.x * PW_NT) {
        const int r = (int)(i / Sp), c = (int)(i % Sp);
        float* d = dpre + (long)r * 3 * Sp;
        if (c >= S) {
            d[c] = 0.f;
            d[Sp + c] = 0.f;
            d[2 * Sp + c] = 0.f;
            continue;
        }
        const float* p = pre + (long)r * 3 * Sp;
        const float n = p[c], li = p[Sp + c], g = cgn_sigmoid(p[2 * Sp + c]);
        const float g_y = dy[(long)r * ldy + c];
        d[c] = (n > 0.f) ? g_y * g : 0.f;
        d[Sp + c] = g_y * (1.f - g);
        d[2 * Sp + c] = g_y * (fmaxf(n, 0.f) - li) * g * (1.f - g);
    }
}

// ---- reparameterisation + KLD (model/cgn.py:94-101, :1076-1080) ---------------------------------------------------
// mulv is [B, 2Z] = [mu | logvar]; zc is [B, Z+C] = [z | c]; kld[0] = mean_b(-0.5 * sum_z(logvar - mu^2 - e^logvar + 1)).
__global__ __launch_bounds__(256) void reparam_kld_fwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                              const float* __restrict__ c, int B, int Z, int Cc,
                                                              float* __restrict__ zc, float* __restrict__ stdv,
                                                              float* __restrict__ kld) {
    __shared__ float red[8];
    float acc = 0.f;
    const int total = B * Z;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int b = i / Z, k = i % Z;
        const float mu = mulv[(long)b * 2 * Z + k], lv = mulv[(long)b * 2 * Z + Z + k];
        const float sd = expf(0.5f * lv);
        zc[(long)b * (Z + Cc) + k] = eps[i] * sd + mu;
        stdv[i] = sd;
        acc += lv - mu * mu - expf(lv) + 1.f;
    }
    for (int i = threadIdx.x; i < B * Cc; i += 256) {
        const int b = i / Cc, k = i % Cc;
        zc[(long)b * (Z + Cc) + Z + k] = c[i];
    }
    acc = block_sum<256>(acc, red);
    if (threadIdx.x == 0) kld[0] = -0.5f * acc / (float)B;
}

// d(mulv) given g = dL/dz (row stride ldg, nullable) and the KLD term weight w = kld_coef / B:
//   dmu = g + w*mu ; dlogvar = g*eps*0.5*std + w*0.5*(e^logvar - 1)
__global__ __launch_bounds__(PW_NT) void reparam_kld_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                                const float* __restrict__ g, long ldg, float w_imm,
                                                                const float* __restrict__ dyn_w, int B, int Z,
                                                                float* __restrict__ dmulv) {
    const float w = dyn_w ? dyn_w[0] : w_imm;
    const int total = B * Z;
    for (int i = blockIdx.x * PW_NT + threadIdx.x; i < total; i += gridDim.x * PW_NT) {
        const int b = i / Z, k = i % Z;
        const float mu = mulv[(long)b * 2 * Z + k], lv = mulv[(long)b * 2 * Z + Z + k];
        const float gz = g ? g[(long)b * ldg + k] : 0.f;
        dmulv[(long)b * 2 * Z + k] = gz + w * mu;
        dmulv[(long)b * 2 * Z + Z + k] = gz * eps[i] * 0.5f * expf(0.5f * lv) + w * 0.5f * (expf(lv) - 1.f);
    }
}

// ---- column sums (bias gradients): two fixed-order stages ---------------------------------------------------------
template <bool IN16>
__device__ __forceinline__ float colsum_at(const void* x, long i) {
    if constexpr (IN16)
        return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(x)[i] << 16);
    else
        return reinterpret_cast<const float*>(x)[i];
}
// IN16: x holds bf16 (the bf16 mode's dlogits, cgn_gemm_desc.storage)
template <bool IN16>
__global__ __launch_bounds__(PW_NT) void colsum_partial_kernel(const void* __restrict__ x, long ld, int R, int Cn,
                                                               int rows_per_chunk, float* __restrict__ part) {
    const int c = blockIdx.x * PW_NT + threadIdx.x;
    if (c >= Cn) return;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    // four independent chains keep 4+ loads in flight per thread; the summation order is fixed (reproducible)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        s0 += colsum_at<IN16>(x, (long)r * ld + c);
        s1 += colsum_at<IN16>(x, (long)(r + 1) * ld + c);
        s2 += colsum_at<IN16>(x, (long)(r + 2) * ld + c);
        s3 += colsum_at<IN16>(x, (long)(r + 3) * ld + c);
    }
    for (; r < r1; ++r) s0 += colsum_at<IN16>(x, (long)r * ld + c);
    part[(long)blockIdx.y * Cn + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(PW_NT) void colsum_final_kernel(const float* __restrict__ part, int nchunk, int Cn,
                                                             float beta, float* __restrict__ out) {
    const int c = blockIdx.x * PW_NT + threadIdx.x;
    if (c >= Cn) return;
    float s = 0.f;
    for (int k0 = 0; k0 < nchunk; k0 += 8) {  // 8 loads in flight; fixed summation order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long)min(k0 + u, nchunk - 1) * Cn + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k0 + u < nchunk) ? v[u] : 0.f;
    }
    out[c] = (beta != 0.f) ? s + beta * out[c] : s;
}

// ---- sum over the time axis: x[B, T, C] -> out[B, C] --------------------------------------------------------------
__global__ __launch_bounds__(PW_NT) void sum_time_kernel(const float* __restrict__ x, int B, int T, int Cn,
                                                         float* __restrict__ out) {
    const long total = (long)B * Cn;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < total; i += (long)gridDim.x * PW_NT) {
        const int b = (int)(i / Cn), c = (int)(i % Cn);
        const float* p = x + (long)b * T * Cn + c;
        float s = 0.f;
        for (int t0 = 0; t0 < T; t0 += 8) {  // 8 loads in flight; fixed summation order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)min(t0 + u, T - 1) * Cn];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (t0 + u < T) ? v[u] : 0.f;
        }
        out[i] = s;
    }
}

// ---- row-wise dot product: out[r] = sum_c a[r,c]*b[r,c] (one wave per row) ----------------------------------------
__global__ __launch_bounds__(PW_NT) void rowdot_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b,
                                                       long ldb, int R, int Cn, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = blockIdx.x * (PW_NT / 64) + wave; r < R; r += gridDim.x * (PW_NT / 64)) {
        float s = 0.f;
        for (int c = lane; c < Cn; c += 64) s += a[(long)r * lda + c] * b[(long)r * ldb + c];
        s = wave_sum(s);
        if (lane == 0) out[r] = s;
    }
}

// ---- Adam (torch.optim.Adam defaults: bias-corrected, no weight decay, no amsgrad; model/cgn.py:267,1046-1047) -----
__global__ __launch_bounds__(PW_NT) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                     float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                     const int* __restrict__ dyn_step, float grad_scale) {
    if (dyn_step) {  // device-resident step count: same double-precision bias corrections as the host path
        const double t = (double)dyn_step[0];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < n; i += (long)gridDim.x * PW_NT) {
        const float gi = g[i] * grad_scale;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);  // torch: exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

// The same update, four elements per thread and trip (16-byte loads / stores: 4 loads of 16 bytes in flight per thread
// instead of 4 of 4 bytes); element by element the arithmetic is adam_kernel's, so the results are bit-identical. The
// generator's arena (14.3 M parameters, 400 MB of traffic) is the step's largest HBM-bound kernel.
__global__ __launch_bounds__(PW_NT) void adam_vec4_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                          float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                          const int* __restrict__ dyn_step, float grad_scale) {
    if (dyn_step) {
        const double t = (double)dyn_step[0];
        bc1 = (float)(1.0 - pow((double)b1, t));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    }
    const float step_size = lr / bc1;
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < n4; i += (long)gridDim.x * PW_NT) {
        const f32x4 g4 = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 m4 = reinterpret_cast<const f32x4*>(m)[i];
        f32x4 v4 = reinterpret_cast<const f32x4*>(v)[i];
        f32x4 p4 = reinterpret_cast<const f32x4*>(p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = g4[e] * grad_scale;
            const float mi = m4[e] + (gi - m4[e]) * (1.f - b1);
            const float vi = b2 * v4[e] + (1.f - b2) * gi * gi;
            m4[e] = mi;
            v4[e] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p4[e] = p4[e] - step_size * (mi / denom);
        }
        reinterpret_cast<f32x4*>(m)[i] = m4;
        reinterpret_cast<f32x4*>(v)[i] = v4;
        reinterpret_cast<f32x4*>(p)[i] = p4;
    }
    if (blockIdx.x == 0) {  // n % 4 leftover elements
        const long i = n4 * 4 + threadIdx.x;
        if (i < n) {
            const float gi = g[i] * grad_scale;
            const float mi = m[i] + (gi - m[i]) * (1.f - b1);
            const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            p[i] = p[i] - step_size * (mi / denom);
        }
    }
}

__global__ void incr_kernel(int* counter) { counter[0] += 1; }

// ---- 2-class discriminator head: fc1 + CE / entropy losses and their gradient wrt the logits ---------------------
// (model/discriminator_sentiment.py:76-77, model/cgn.py:167-170, :306, :318-334, :1118)
// One workgroup; B and F are small (B <= a few hundred, F = |Ks|*Co).
//   featm = feat * mask (mask nullable)        logits[b,c] = featm[b,:] . W[c,:] + bias[c]
//   ce  = mean_b( lse_b - logits[b, y_b] )     ent = -mean_b sum_c p log p
//   dlogits[b,c] = w_ce/B * (p_c - [c == y_b]) + w_ent/B * ( -p_c (log p_c + H_b) )
constexpr int DH_NT = 1024;
__global__ __launch_bounds__(DH_NT) void disc_head_kernel(const float* __restrict__ feat, long ldf,
                                                        const float* __restrict__ mask, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const int64_t* __restrict__ y,
                                                        int B, int F, float w_ce, float w_ent,
                                                        float* __restrict__ logits, float* __restrict__ p1,
                                                        float* __restrict__ dlogits, float* __restrict__ stats) {
    __shared__ float red[DH_NT / 64 + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float ce_acc = 0.f, ent_acc = 0.f;
    // A wave takes its rows four at a time (B = 64: all of them): the loads of the four rows are issued together, so the
    // kernel pays one L2 round trip per group instead of one per row (it is a latency chain on the discriminator's
    // stream: 21 -> 13 us). Per row the sums are formed in the same order as before.
    constexpr int RB = 4;
    for (int bg = wave; bg < B; bg += RB * (DH_NT / 64)) {
        float s0r[RB], s1r[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) s0r[r] = s1r[r] = 0.f;
        // batches of 8 chunks: all loads of a batch are issued (clamped addresses) before the first is consumed
        for (int f0 = 0; f0 < F; f0 += 512) {
            float xv[RB][8], mv[RB][8], w0v[8], w1v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * 64 + lane;
                const int fc = f < F ? f : 0;
                w0v[u] = W[fc];
                w1v[u] = W[F + fc];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int b = min(bg + r * (DH_NT / 64), B - 1);
                    xv[r][u] = feat[(long)b * ldf + fc];
                    mv[r][u] = mask ? mask[(long)b * F + fc] : 1.f;
                }
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float x = (f0 + u * 64 + lane < F) ? xv[r][u] * mv[r][u] : 0.f;
                    s0r[r] += x * w0v[u];
                    s1r[r] += x * w1v[u];
                }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
        const int b = bg + r * (DH_NT / 64);
        if (b >= B) break;  // (wave-uniform)
        const float s0 = wave_sum(s0r[r]) + bias[0];
        const float s1 = wave_sum(s1r[r]) + bias[1];
        if (lane == 0) {
            const float mx = fmaxf(s0, s1);
            const float lse = mx + logf(expf(s0 - mx) + expf(s1 - mx));
            const float l0 = s0 - lse, l1 = s1 - lse;
            const float q0 = expf(l0), q1 = expf(l1);
            const float Hb = -(q0 * l0 + q1 * l1);
            logits[b * 2] = s0;
            logits[b * 2 + 1] = s1;
            if (p1) p1[b] = q1;
            float d0 = 0.f, d1 = 0.f;
            if (y) {
                const int64_t yb = y[b];
                ce_acc += (yb == 1) ? -l1 : -l0;
                d0 += w_ce * (q0 - (yb == 0 ? 1.f : 0.f));
                d1 += w_ce * (q1 - (yb == 1 ? 1.f : 0.f));
            }
            ent_acc += Hb;
            d0 += w_ent * (-q0 * (l0 + Hb));
            d1 += w_ent * (-q1 * (l1 + Hb));
            if (dlogits) {
                dlogits[b * 2] = d0 / (float)B;
                dlogits[b * 2 + 1] = d1 / (float)B;
            }
        }
        }
    }
    ce_acc = block_sum<DH_NT>(ce_acc, red);
    ent_acc = block_sum<DH_NT>(ent_acc, red);
    if (threadIdx.x == 0 && stats) {
        stats[0] = ce_acc / (float)B;
        stats[1] = ent_acc / (float)B;
    }
}

// Backward of the head: dW[c,f] (+)= sum_b dlogits[b,c]*featm[b,f]; db[c] (+)= sum_b dlogits[b,c];
// dfeat[b,f] = mask * sum_c dlogits[b,c]*W[c,f]. Grid over f.
__global__ __launch_bounds__(PW_NT) void disc_head_bwd_kernel(const float* __restrict__ feat, long ldf,
                                                              const float* __restrict__ mask, const float* __restrict__ W,
                                                              const float* __restrict__ dlogits, int B, int F, float beta,
                                                              float* __restrict__ dW, float* __restrict__ db,
                                                              float* __restrict__ dfeat) {
    const int f = blockIdx.x * PW_NT + threadIdx.x;
    if (f < F) {
        float a0 = 0.f, a1 = 0.f;
        const float w0 = W[f], w1 = W[F + f];
        constexpr int RBW = 16;  // rows of loads in flight (the walk was one L2 round trip per row; 8 rows: 20 us at B = 64)
        for (int b0 = 0; b0 < B; b0 += RBW) {
            float mk[RBW], xv[RBW], d0[RBW], d1[RBW];
#pragma unroll
            for (int u = 0; u < RBW; ++u) {
                const int b = min(b0 + u, B - 1);
                mk[u] = mask ? mask[(long)b * F + f] : 1.f;
                xv[u] = feat[(long)b * ldf + f];
                d0[u] = dlogits[b * 2];
                d1[u] = dlogits[b * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < RBW; ++u) {
                if (b0 + u < B) {
                    const float x = xv[u] * mk[u];
                    a0 += d0[u] * x;
                    a1 += d1[u] * x;
                    if (dfeat) dfeat[(long)(b0 + u) * F + f] = (d0[u] * w0 + d1[u] * w1) * mk[u];
                }
            }
        }
        if (dW) {
            dW[f] = (beta != 0.f) ? a0 + beta * dW[f] : a0;
            dW[F + f] = (beta != 0.f) ? a1 + beta * dW[F + f] : a1;
        }
    }
    if (db && blockIdx.x == 0 && threadIdx.x < 2) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dlogits[b * 2 + threadIdx.x];
        db[threadIdx.x] = (beta != 0.f) ? s + beta * db[threadIdx.x] : s;
    }
}

// c ~ Bernoulli(p1) with the uniform draw given (model/cgn.py:170)
__global__ __launch_bounds__(PW_NT) void bernoulli_kernel(const float* __restrict__ p1, const float* __restrict__ u, int n,
                                                          float* __restrict__ c) {
    const int i = blockIdx.x * PW_NT + threadIdx.x;
    if (i < n) c[i] = (u[i] < p1[i]) ? 1.f : 0.f;
}

// out = a * b elementwise over [R, C] (explicit dropout masks on the modular path)
__global__ __launch_bounds__(PW_NT) void mul2d_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < n; i += (long)gridDim.x * PW_NT) out[i] = a[i] * b[i];
}

// labels = (int64) c  — `c.long()` of model/cgn.py:1117
__global__ __launch_bounds__(PW_NT) void float_to_index_kernel(const float* __restrict__ x, int n,
                                                               int64_t* __restrict__ out) {
    const int i = blockIdx.x * PW_NT + threadIdx.x;
    if (i < n) out[i] = (int64_t)x[i];
}

// z-reconstruction loss (model/cgn.py:47-57, :1134-1137): with (mu1, std1) constants and (mu2, logvar2) from the
// second encoder pass, dist_b = sqrt(sum_z (mu1-mu2)^2 / ((std1+std2)/2)), z_loss = sum_b dist_b / B;
// dmulv2 = weight * d z_loss / d[mu2 | logvar2]. One wave per sentence, single workgroup (B is small).
__global__ __launch_bounds__(256) void mahalanobis_kernel(const float* __restrict__ mulv1, const float* __restrict__ mulv2,
                                                          int B, int Z, float weight, float* __restrict__ z_loss,
                                                          float* __restrict__ dmulv2) {
    __shared__ float red[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float total = 0.f;
    for (int b = wave; b < B; b += 4) {
        const float* m1 = mulv1 + (long)b * 2 * Z;
        const float* m2 = mulv2 + (long)b * 2 * Z;
        float acc = 0.f;
        for (int k = lane; k < Z; k += 64) {
            const float d = m1[k] - m2[k];
            const float s = 0.5f * (expf(0.5f * m1[Z + k]) + expf(0.5f * m2[Z + k]));
            acc += d * d / s;
        }
        acc = wave_sum(acc);
        const float dist = sqrtf(acc);
        if (lane == 0) total += dist;
        const float inv = (dist > 0.f) ? weight / ((float)B * dist) : 0.f;  // d sqrt at 0 is undefined; emit 0
        for (int k = lane; k < Z; k += 64) {
            const float d = m1[k] - m2[k];
            const float sd2 = expf(0.5f * m2[Z + k]);
            const float s = 0.5f * (expf(0.5f * m1[Z + k]) + sd2);
            dmulv2[(long)b * 2 * Z + k] = -d / s * inv;
            dmulv2[(long)b * 2 * Z + Z + k] = -(d * d) / (4.f * s * s) * inv * 0.5f * sd2;
        }
    }
    total = block_sum<256>(total, red);
    if (threadIdx.x == 0) z_loss[0] = total / (float)B;
}

struct LinComb {
    const float* src[4];
    float coef[4];
    int n;
};
// out[0] = sum_i coef[i] * src[i][0] — the scalar loss combinations returned by the step closures
// (model/cgn.py:329,334,1151-1152) without a host round trip.
__global__ void lincomb_kernel(const LinComb lc, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < lc.n; ++i) s += lc.coef[i] * lc.src[i][0];
        out[0] = s;
    }
}

__global__ __launch_bounds__(PW_NT) void fill_kernel(float* __restrict__ p, long n, float v) {
    for (long i = (long)blockIdx.x * PW_NT + threadIdx.x; i < n; i += (long)gridDim.x * PW_NT) p[i] = v;
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" int cgn_copy2d_f32(void* stream, const float* src, int64_t lds_, float* dst, int64_t ldd, int32_t rows,
                              int32_t cols, float scale, float beta) {
    CGN_CHECK_ARG(rows >= 0 && cols >= 0);
    if (rows == 0 || cols == 0) return CGN_OK;
    CGN_CHECK_ARG(src && dst && lds_ >= cols && ldd >= cols);
    hipLaunchKernelGGL(copy2d_kernel, dim3(pw_blocks((long)rows * cols)), dim3(PW_NT), 0, S_(stream), src, (long)lds_,
                       dst, (long)ldd, rows, cols, scale, beta);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_copy3d_f32(void* stream, const float* src, int64_t s_bs, int64_t s_ts, float* dst, int64_t d_bs,
                              int64_t d_ts, int32_t B, int32_t T, int32_t H, float beta) {
    CGN_CHECK_ARG(B >= 0 && T >= 0 && H >= 0);
    if (B == 0 || T == 0 || H == 0) return CGN_OK;
    CGN_CHECK_ARG(src && dst && (s_ts >= H || s_ts == 0) && d_ts >= H && s_bs >= 0 && d_bs >= 0);  // s_ts 0 = broadcast over t
    hipLaunchKernelGGL(copy3d_kernel, dim3(pw_blocks((long)B * T * H)), dim3(PW_NT), 0, S_(stream), src, (long)s_bs,
                       (long)s_ts, dst, (long)d_bs, (long)d_ts, B, T, H, beta);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_transpose_f32(void* stream, const float* src, int32_t rows, int32_t cols, float* dst) {
    CGN_CHECK_ARG(rows >= 0 && cols >= 0);
    if (rows == 0 || cols == 0) return CGN_OK;
    CGN_CHECK_ARG(src && dst && src != dst);
    hipLaunchKernelGGL(transpose_kernel, dim3(cgn_cdiv(cols, 32), cgn_cdiv(rows, 32)), dim3(256), 0, S_(stream), src,
                       rows, cols, dst);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_gather_rows_f32(void* stream, const float* table, int32_t V, int32_t E, const int64_t* idx, int32_t n,
                                   const float* mask, float* out, int64_t ldo) {
    CGN_CHECK_ARG(n >= 0 && V > 0 && E > 0);
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(table && idx && out && ldo >= E);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(pw_blocks(n, PW_NT / 64)), dim3(PW_NT), 0, S_(stream), table, V, E, idx,
                       n, mask, out, (long)ldo);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_highway_gate_fwd_f32(void* stream, const float* pre, int32_t N, int32_t S, int32_t Sp, float* out,
                                        int64_t ldo) {
    CGN_CHECK_ARG(N >= 0 && S > 0 && Sp >= S);
    if (N == 0) return CGN_OK;
    CGN_CHECK_ARG(pre && out && ldo >= S);
    hipLaunchKernelGGL(highway_fwd_kernel, dim3(pw_blocks((long)N * S)), dim3(PW_NT), 0, S_(stream), pre, N, S, Sp, out,
                       (long)ldo);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_highway_gate_bwd_f32(void* stream, const float* pre, int32_t N, int32_t S, int32_t Sp,
                                        const float* dy, int64_t ldy, float* dpre) {
    CGN_CHECK_ARG(N >= 0 && S > 0 && Sp >= S);
    if (N == 0) return CGN_OK;
    CGN_CHECK_ARG(pre && dy && dpre && ldy >= S);
    hipLaunchKernelGGL(highway_bwd_kernel, dim3(pw_blocks((long)N * Sp)), dim3(PW_NT), 0, S_(stream), pre, N, S, Sp, dy,
                       (long)ldy, dpre);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_reparam_kld_fwd_f32(void* stream, const float* mulv, const float* eps, const float* c, int32_t B,
                                       int32_t Z, int32_t Cc, float* zc, float* stdv, float* kld) {
    CGN_CHECK_ARG(B > 0 && Z > 0 && Cc >= 0);
    CGN_CHECK_ARG(mulv && eps && zc && stdv && kld && (Cc == 0 || c));
    hipLaunchKernelGGL(reparam_kld_fwd_kernel, dim3(1), dim3(256), 0, S_(stream), mulv, eps, c, B, Z, Cc, zc, stdv, kld);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_reparam_kld_bwd_f32(void* stream, const float* mulv, const float* eps, const float* g, int64_t ldg,
                                       float kld_weight, const float* dyn_kld_weight, int32_t B, int32_t Z,
                                       float* dmulv) {
    CGN_CHECK_ARG(B > 0 && Z > 0);
    CGN_CHECK_ARG(mulv && eps && dmulv && (!g || ldg >= Z));
    hipLaunchKernelGGL(reparam_kld_bwd_kernel, dim3(pw_blocks((long)B * Z)), dim3(PW_NT), 0, S_(stream), mulv, eps, g,
                       (long)ldg, kld_weight, dyn_kld_weight, B, Z, dmulv);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

static int colsum_any(void* stream, const void* x, bool in16, int64_t ld, int32_t R, int32_t Cn, float beta, float* out,
                      float* workspace, int64_t workspace_bytes) {
    CGN_CHECK_ARG(R >= 0 && Cn > 0);
    CGN_CHECK_ARG(x && out && ld >= Cn && workspace);
    int nchunk = cgn_cdiv(R, 16);  // short row chunks: enough blocks to fill 256 CUs even for narrow matrices
    if (nchunk > 64) nchunk = 64;
    if (nchunk < 1) nchunk = 1;
    if ((long)nchunk * Cn * 4 > workspace_bytes) return CGN_ERR_WORKSPACE;
    const int rpc = cgn_cdiv(R, nchunk);
    const dim3 grid(cgn_cdiv(Cn, PW_NT), nchunk);
    if (in16)
        hipLaunchKernelGGL(colsum_partial_kernel<true>, grid, dim3(PW_NT), 0, S_(stream), x, (long)ld, R, Cn, rpc, workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel<false>, grid, dim3(PW_NT), 0, S_(stream), x, (long)ld, R, Cn, rpc, workspace);
    CGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cgn_cdiv(Cn, PW_NT)), dim3(PW_NT), 0, S_(stream), workspace, nchunk, Cn,
                       beta, out);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_colsum_f32(void* stream, const float* x, int64_t ld, int32_t R, int32_t Cn, float beta, float* out,
                              float* workspace, int64_t workspace_bytes) {
    return colsum_any(stream, x, false, ld, R, Cn, beta, out, workspace, workspace_bytes);
}

extern "C" int cgn_colsum_bf16(void* stream, const void* x, int64_t ld, int32_t R, int32_t Cn, float beta, float* out,
                               float* workspace, int64_t workspace_bytes) {
    CGN_CHECK_ARG((uintptr_t)x % 2 == 0);
    return colsum_any(stream, x, true, ld, R, Cn, beta, out, workspace, workspace_bytes);
}

extern "C" int cgn_sum_time_f32(void* stream, const float* x, int32_t B, int32_t T, int32_t Cn, float* out) {
    CGN_CHECK_ARG(B >= 0 && T >= 0 && Cn > 0);
    if (B == 0) return CGN_OK;
    CGN_CHECK_ARG(x && out);
    hipLaunchKernelGGL(sum_time_kernel, dim3(pw_blocks((long)B * Cn)), dim3(PW_NT), 0, S_(stream), x, B, T, Cn, out);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_rowdot_f32(void* stream, const float* a, int64_t lda, const float* b, int64_t ldb, int32_t R,
                              int32_t Cn, float* out) {
    CGN_CHECK_ARG(R >= 0 && Cn > 0);
    if (R == 0) return CGN_OK;
    CGN_CHECK_ARG(a && b && out && lda >= Cn && ldb >= Cn);
    hipLaunchKernelGGL(rowdot_kernel, dim3(pw_blocks(R, PW_NT / 64)), dim3(PW_NT), 0, S_(stream), a, (long)lda, b,
                       (long)ldb, R, Cn, out);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_adam_f32(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, int32_t step, const int32_t* dyn_step, float grad_scale) {
    CGN_CHECK_ARG(n >= 0 && (dyn_step || step >= 1));
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(p && g && m && v);
    // bias corrections in double on the host, as torch does for python-scalar steps; with a device-resident step
    // count (graph replay) the kernel forms them itself, also in double
    const double bc1 = dyn_step ? 0.0 : 1.0 - pow((double)beta1, (double)step);
    const double bc2 = dyn_step ? 0.0 : 1.0 - pow((double)beta2, (double)step);
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0) && n >= 4;
    if (vec)
        hipLaunchKernelGGL(adam_vec4_kernel, dim3(pw_blocks(n / 4, PW_NT, 8192)), dim3(PW_NT), 0, S_(stream), p, g, m, v,
                           (long)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), dyn_step, grad_scale);
    else
        hipLaunchKernelGGL(adam_kernel, dim3(pw_blocks(n, PW_NT, 8192)), dim3(PW_NT), 0, S_(stream), p, g, m, v, (long)n,
                           lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), dyn_step, grad_scale);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_incr_i32(void* stream, int32_t* counter) {
    CGN_CHECK_ARG(counter != nullptr);
    hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(1), 0, S_(stream), counter);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

// Measurement aid: one lane writes the device's constant-rate wall clock into *slot when the stream reaches this point.
// Works as a node of a captured hipGraph, which HIP events and rocprofv3's per-kernel trace do not (un-profiled replays).
static __global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int cgn_stamp_u64(void* stream, uint64_t* slot) {
    CGN_CHECK_ARG(slot != nullptr);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, S_(stream), (unsigned long long*)slot);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_disc_head_f32(void* stream, const float* feat, int64_t ldf, const float* mask, const float* W,
                                 const float* bias, const int64_t* y, int32_t B, int32_t F, float w_ce, float w_ent,
                                 float* logits, float* p1, float* dlogits, float* stats) {
    CGN_CHECK_ARG(B > 0 && F > 0 && ldf >= F);
    CGN_CHECK_ARG(feat && W && bias && logits);
    hipLaunchKernelGGL(disc_head_kernel, dim3(1), dim3(DH_NT), 0, S_(stream), feat, (long)ldf, mask, W, bias, y, B, F, w_ce,
                       w_ent, logits, p1, dlogits, stats);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_disc_head_bwd_f32(void* stream, const float* feat, int64_t ldf, const float* mask, const float* W,
                                     const float* dlogits, int32_t B, int32_t F, float beta, float* dW, float* db,
                                     float* dfeat) {
    CGN_CHECK_ARG(B > 0 && F > 0 && ldf >= F);
    CGN_CHECK_ARG(feat && W && dlogits);
    hipLaunchKernelGGL(disc_head_bwd_kernel, dim3(cgn_cdiv(F, PW_NT)), dim3(PW_NT), 0, S_(stream), feat, (long)ldf, mask,
                       W, dlogits, B, F, beta, dW, db, dfeat);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_bernoulli_f32(void* stream, const float* p1, const float* u, int32_t n, float* c) {
    CGN_CHECK_ARG(n >= 0);
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(p1 && u && c);
    hipLaunchKernelGGL(bernoulli_kernel, dim3(cgn_cdiv(n, PW_NT)), dim3(PW_NT), 0, S_(stream), p1, u, n, c);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_fill_f32(void* stream, float* p, int64_t n, float v) {
    CGN_CHECK_ARG(n >= 0);
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(p);
    hipLaunchKernelGGL(fill_kernel, dim3(pw_blocks(n)), dim3(PW_NT), 0, S_(stream), p, (long)n, v);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_lincomb_f32(void* stream, int32_t n, const float* const* src, const float* coef, float* out) {
    CGN_CHECK_ARG(n >= 1 && n <= 4 && src && coef && out);
    LinComb lc;
    lc.n = n;
    for (int i = 0; i < 4; ++i) {
        lc.src[i] = i < n ? src[i] : nullptr;
        lc.coef[i] = i < n ? coef[i] : 0.f;
        if (i < n) CGN_CHECK_ARG(src[i] != nullptr);
    }
    hipLaunchKernelGGL(lincomb_kernel, dim3(1), dim3(64), 0, S_(stream), lc, out);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_mul_f32(void* stream, const float* a, const float* b, float* out, int64_t n) {
    CGN_CHECK_ARG(n >= 0);
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(a && b && out);
    hipLaunchKernelGGL(mul2d_kernel, dim3(pw_blocks(n)), dim3(PW_NT), 0, S_(stream), a, b, out, (long)n);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_float_to_index_f32(void* stream, const float* x, int32_t n, int64_t* out) {
    CGN_CHECK_ARG(n >= 0);
    if (n == 0) return CGN_OK;
    CGN_CHECK_ARG(x && out);
    hipLaunchKernelGGL(float_to_index_kernel, dim3(cgn_cdiv(n, PW_NT)), dim3(PW_NT), 0, S_(stream), x, n, out);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_mahalanobis_f32(void* stream, const float* mulv1, const float* mulv2, int32_t B, int32_t Z,
                                   float weight, float* z_loss, float* dmulv2) {
    CGN_CHECK_ARG(B > 0 && Z > 0);
    CGN_CHECK_ARG(mulv1 && mulv2 && z_loss && dmulv2);
    hipLaunchKernelGGL(mahalanobis_kernel, dim3(1), dim3(256), 0, S_(stream), mulv1, mulv2, B, Z, weight, z_loss, dmulv2);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
