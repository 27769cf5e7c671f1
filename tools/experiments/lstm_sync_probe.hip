// Ablation probe for the persistent LSTM forward recurrence (NOT part of libcgn_hip.so): a stand-alone copy of the
// kernel's per-timestep structure (csrc/lstm.hip lstm_persist_fwd_kernel<4>) with compile-time switches that take one
// piece at a time OUT of the step, to see what the 4.6 us per timestep (1.7 us of it MFMA) are made of. Variants with a
// piece removed compute garbage by design; only variant 0 is the real algorithm. Timing only, no results are kept.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/lstm_sync_probe.hip -o tools/experiments/probe_lstm_sync
//   tools/experiments/probe_lstm_sync [nseq B T H]      (default 2 64 20 512 = the encoder biLSTM at cfg2)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 8, NT = NW * 64, SYNC_STRIDE = 32;
constexpr unsigned SPIN_LIMIT = 1u << 19;

// switches (bit set = piece removed / changed)
constexpr int NO_WAIT = 1;      // do not wait for the other workgroups' h_{t-1}
constexpr int NO_MFMA = 2;      // skip the MFMAs
constexpr int NO_HLOAD = 4;     // no h_{t-1} loads
constexpr int NO_GSTORE = 8;    // no gates / c stores
constexpr int NO_SIGNAL = 16;   // no drain + barrier + arrival add
constexpr int H_FIRST = 32;     // h store issued BEFORE the gates / c stores, drained alone, signalled, then the rest
constexpr int WAVE_SIGNAL = 64; // every storing wave signals for itself (no barrier before the add); target x4
constexpr int NO_SLEEP = 128;   // poll without s_sleep
constexpr int FAST_TANH = 512;  // tanh through one hardware exponential instead of libm's tanhf
constexpr int SENTINEL = 256;   // no counters at all: the history buffer is pre-filled with a NaN pattern that no
                                // computation produces, a wave re-loads the quads of ITS k range that still hold it

struct Seq {
    const float* xg;
    const float* w_hh;
    float* hbuf;
    float* cbuf;
    float* gates;
};
struct Params {
    Seq s[4];
    int B, T, H, nseq;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
template <int PROBE>
__device__ __forceinline__ float tanh_p(float x) {
    if constexpr (PROBE & 512) return 1.f - 2.f / (1.f + __expf(2.f * x));
    else return tanhf(x);
}

template <int PROBE>
__device__ __forceinline__ bool group_wait(unsigned* cnt, unsigned target, unsigned* tmo) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (!(PROBE & NO_SLEEP)) __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (spins > SPIN_LIMIT ||
            ((spins & 63u) == 0u && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <int PROBE>
__global__ __launch_bounds__(NT) void probe_kernel(const Params p, unsigned* sync) {
    constexpr int CPW = 4;
    __shared__ float red[NW][16][64];
    __shared__ int s_ok;
    const int B = p.B, T = p.T, H = p.H;
    const int NJ = H / 16, NB = (B + 15) / 16;
    const int wid = blockIdx.x;
    const int seq = wid / (NB * NJ);
    const int bt = (wid % (NB * NJ)) / NJ, jt = wid % NJ;
    const Seq& sq = p.s[seq];
    unsigned* cnt = sync + (seq * NB + bt) * SYNC_STRIDE;
    unsigned* tmo = sync + p.nseq * NB * SYNC_STRIDE;
    const int j0 = jt * 16, b0 = bt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, q = lane >> 4;
    const int nchunks = H / 16;
    const int cpw = (nchunks + NW - 1) / NW;
    const int c_beg = wave * cpw;
    const float* wrow = sq.w_hh + (long)(j0 + fr) * H;
    const long gate_stride = (long)H * H;
    f32x4 w[4][CPW];
    int koff[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        const bool ok = (u < cpw) && (c_beg + u < nchunks);
        const int kc = ok ? (c_beg + u) * 16 + q * 4 : 0;
        koff[u] = kc * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + g * gate_stride + kc);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            w[g][u] = ok ? v : z;
        }
    }
    const int bA = min(b0 + fr, B - 1);
    const __amdgpu_buffer_rsrc_t hres = __builtin_amdgcn_make_buffer_rsrc((void*)sq.hbuf, 0, B * (T + 1) * H * 4, 0x00020000);
    const int er = wave & 3;
    const int eb = b0 + (lane >> 4) * 4 + er, ej = j0 + (lane & 15);
    const int ebc = min(eb, B - 1);
    const bool evalid = (wave < 4) && (eb < B);
    float c_reg = 0.f;
    if (evalid) {
        sq.hbuf[((long)eb * (T + 1)) * H + ej] = 0.f;
        sq.cbuf[((long)eb * (T + 1)) * H + ej] = 0.f;
    }
    const unsigned per_step = (PROBE & WAVE_SIGNAL) ? 4u * NJ : (unsigned)NJ;
    for (int step = 0; step < T; ++step) {
        const int t = step;
        float pre_x[4];
        {
            const float* xg = sq.xg + ((long)ebc * T + t) * 4 * H + ej;
#pragma unroll
            for (int g = 0; g < 4; ++g) pre_x[g] = xg[g * H];
        }
        if (step > 0 && !(PROBE & (NO_WAIT | SENTINEL))) {
            if (threadIdx.x == 0) s_ok = group_wait<PROBE>(cnt, per_step * step, tmo) ? 1 : 0;
            __syncthreads();
            if (!s_ok) break;
        }
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            const int hoff = (bA * (T + 1) + t) * H * 4;
            f32x4 a[CPW];
#pragma unroll
            for (int u = 0; u < CPW; ++u) {
                if (PROBE & NO_HLOAD) {
                    a[u] = {pre_x[0], pre_x[1], pre_x[2], pre_x[3]};
                } else {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(hres, hoff + koff[u], 0, 16);
                    a[u] = __builtin_bit_cast(f32x4, v);
                }
            }
            if (PROBE & SENTINEL) {
                unsigned spins = 0;
                while (true) {
                    bool bad = false;
#pragma unroll
                    for (int u = 0; u < CPW; ++u) {
                        const u32x4 b = __builtin_bit_cast(u32x4, a[u]);
                        const bool bu = (b[0] == 0xFFFFFFFFu) | (b[1] == 0xFFFFFFFFu) | (b[2] == 0xFFFFFFFFu) | (b[3] == 0xFFFFFFFFu);
                        if (bu) {  // only the lanes whose quad has not arrived load again
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(hres, hoff + koff[u], 0, 16);
                            a[u] = __builtin_bit_cast(f32x4, v);
                        }
                        bad |= bu;
                    }
                    if (!__any(bad) || ++spins > SPIN_LIMIT) break;
                }
            }
            if (!(PROBE & NO_MFMA)) {
#pragma unroll
                for (int u = 0; u < CPW; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], w[g][u][i], acc[g], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < CPW; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] += a[u] * w[g][u];
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
                for (int wv = 0; wv < NW; ++wv) v += red[wv][g * 4 + er][lane];
                pre[g] = v + pre_x[g];
            }
            const float ig = sigm(pre[0]), fg = sigm(pre[1]), gg = tanh_p<PROBE>(pre[2]), og = sigm(pre[3]);
            const float cn = fg * c_reg + ig * gg;
            const float hn = og * tanh_p<PROBE>(cn);
            c_reg = cn;
            float* go = sq.gates + ((long)eb * T + t) * 4 * H;
            float* hp = sq.hbuf + ((long)eb * (T + 1) + t + 1) * H + ej;
            if (PROBE & H_FIRST) {
                __hip_atomic_store(hp, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (!(PROBE & NO_GSTORE)) {
                    go[ej] = ig, go[H + ej] = fg, go[2 * H + ej] = gg, go[3 * H + ej] = og;
                    sq.cbuf[((long)eb * (T + 1) + t + 1) * H + ej] = cn;
                }
                __hip_atomic_store(hp, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (PROBE & H_FIRST) {
                // drain the h store alone, signal, and only then issue the stores nobody waits for
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (PROBE & WAVE_SIGNAL) {
                    if (lane == 0 && step + 1 < T) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!(PROBE & NO_GSTORE)) {
                    go[ej] = ig, go[H + ej] = fg, go[2 * H + ej] = gg, go[3 * H + ej] = og;
                    sq.cbuf[((long)eb * (T + 1) + t + 1) * H + ej] = cn;
                }
            }
        }
        if (!(PROBE & (NO_SIGNAL | SENTINEL)) && !((PROBE & H_FIRST) && (PROBE & WAVE_SIGNAL))) {
            if (PROBE & WAVE_SIGNAL) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wave < 4 && lane == 0 && step + 1 < T)
                    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (!(PROBE & H_FIRST)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0 && step + 1 < T) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int PROBE>
void run(const char* what, const Params& p, unsigned* sync, long sync_bytes, int grid) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipMemsetAsync(sync, 0, sync_bytes, 0));
        if (PROBE & SENTINEL)
            for (int i = 0; i < p.nseq; ++i) CK(hipMemsetAsync(p.s[i].hbuf, 0xFF, (size_t)p.B * (p.T + 1) * p.H * 4, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(probe_kernel<PROBE>, dim3(grid), dim3(NT), 0, 0, p, sync);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2 && ms < best) best = ms;
    }
    unsigned tmo = 0;
    CK(hipMemcpy(&tmo, (char*)sync + sync_bytes - 128, 4, hipMemcpyDeviceToHost));
    printf("%3d  %-58s %7.1f us  %5.2f us/step%s\n", PROBE, what, best * 1e3f, best * 1e3f / p.T, tmo ? "  TIMEOUT" : "");
}

int main(int argc, char** argv) {
    const int nseq = argc > 4 ? atoi(argv[1]) : 2, B = argc > 4 ? atoi(argv[2]) : 64;
    const int T = argc > 4 ? atoi(argv[3]) : 20, H = argc > 4 ? atoi(argv[4]) : 512;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int NB = (B + 15) / 16, grid = nseq * NB * (H / 16);
    if (nseq < 1 || nseq > 4 || H % 16 || H > 512 || grid > cus) {
        fprintf(stderr, "shape outside the probe (grid %d, CUs %d)\n", grid, cus);
        return 1;
    }
    Params p = {};
    p.B = B, p.T = T, p.H = H, p.nseq = nseq;
    std::vector<float> hw((size_t)4 * H * H), hx((size_t)B * T * 4 * H);
    srand(1);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.08f;
    for (auto& v : hx) v = (rand() / (float)RAND_MAX - 0.5f);
    for (int i = 0; i < nseq; ++i) {
        float *xg, *w, *h, *c, *g;
        CK(hipMalloc(&xg, hx.size() * 4));
        CK(hipMalloc(&w, hw.size() * 4));
        CK(hipMalloc(&h, (size_t)B * (T + 1) * H * 4));
        CK(hipMalloc(&c, (size_t)B * (T + 1) * H * 4));
        CK(hipMalloc(&g, hx.size() * 4));
        CK(hipMemcpy(xg, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(h, 0, (size_t)B * (T + 1) * H * 4));
        p.s[i] = {xg, w, h, c, g};
    }
    const long sync_bytes = ((long)nseq * NB + 1) * SYNC_STRIDE * 4;
    unsigned* sync;
    CK(hipMalloc(&sync, sync_bytes));
    printf("nseq %d B %d T %d H %d: %d workgroups on %d CUs\n", nseq, B, T, H, grid, cus);
    run<0>("the kernel as shipped", p, sync, sync_bytes, grid);
    run<NO_WAIT>("no wait for h_{t-1}", p, sync, sync_bytes, grid);
    run<NO_WAIT | NO_SIGNAL>("no wait, no drain/barrier/arrival add", p, sync, sync_bytes, grid);
    run<NO_WAIT | NO_SIGNAL | NO_GSTORE>("... and no gates/c stores", p, sync, sync_bytes, grid);
    run<NO_WAIT | NO_SIGNAL | NO_GSTORE | NO_HLOAD>("... and no h loads", p, sync, sync_bytes, grid);
    run<NO_WAIT | NO_SIGNAL | NO_GSTORE | NO_HLOAD | NO_MFMA>("... and no MFMA (VALU stand-in)", p, sync, sync_bytes, grid);
    run<NO_MFMA>("full sync, no MFMA", p, sync, sync_bytes, grid);
    run<NO_GSTORE>("full sync, no gates/c stores", p, sync, sync_bytes, grid);
    run<NO_SLEEP>("poll without s_sleep", p, sync, sync_bytes, grid);
    run<H_FIRST>("h store first, drained alone, then gates/c", p, sync, sync_bytes, grid);
    run<WAVE_SIGNAL>("every storing wave signals for itself", p, sync, sync_bytes, grid);
    run<H_FIRST | WAVE_SIGNAL>("h first + per-wave signal", p, sync, sync_bytes, grid);
    run<H_FIRST | WAVE_SIGNAL | NO_SLEEP>("h first + per-wave signal + no sleep", p, sync, sync_bytes, grid);
    run<FAST_TANH>("tanh through __expf", p, sync, sync_bytes, grid);
    run<SENTINEL>("sentinel hand-off: no counters, re-load what has not arrived", p, sync, sync_bytes, grid);
    run<SENTINEL | NO_GSTORE>("sentinel hand-off, no gates/c stores", p, sync, sync_bytes, grid);
    run<SENTINEL | NO_MFMA>("sentinel hand-off, no MFMA", p, sync, sync_bytes, grid);
    return 0;
}
