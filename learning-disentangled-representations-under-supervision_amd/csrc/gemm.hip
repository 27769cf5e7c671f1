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

struct Axis {
    long stride, rpg, gs;
};

// indices and group sizes fit 32 bits (checked on the host): 32-bit division keeps the inlined code small
__device__ __forceinline__ long axis_off(const Axis& a, int i) {
    if (a.rpg > 0) {
        const int r = (int)a.rpg;
        const int g = i / r;
        return (long)g * a.gs + (long)(i - g * r) * a.stride;
    }
    return (long)i * a.stride;
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
    long ldc, ep_ld;
    Axis am, ak, bk, bn;
    int M, N, K;
    int k_per_split, splitk;
    int a_vec, b_vec;
    int a_transform, epilogue;
    int tiles_m, tiles_n;
    float alpha, beta, a_scale, ep_inv_tau, ep_ce_scale;
};

constexpr int KROUND = 32;  // split-K ranges are multiples of every BK in use
constexpr int NT = 256;

__device__ __forceinline__ float apply_epilogue(const GemmParams& p, int row, int col, float acc) {
    if (p.epilogue == CGN_EPI_SOFTMAX_BWD) {
        const float pr = expf(p.ep_logits[(long)row * p.ep_ld + col] * p.ep_inv_tau - p.ep_lse[row]);
        const float onehot = (p.ep_target[row] == (int64_t)col) ? 1.f : 0.f;
        return pr * (acc - p.ep_rowdot[row]) * p.ep_inv_tau + p.ep_ce_scale * (pr - onehot);
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
template <int ROWS, bool KC, int BK>
struct TileLoader {
    static constexpr int LDK = BK + 4;
    static constexpr int QUADS = ROWS * BK / 4;  // 4-k chunks per tile
    static constexpr int PER = QUADS / NT;       // chunks per thread
    static_assert(QUADS % NT == 0, "tile/threads mismatch");
    static constexpr int KGROUPS = NT / ROWS;    // !KC: thread groups along k
    static_assert(KC || (BK / KGROUPS == 4 * PER), "k coverage");

    f32x4 regs[PER];
    long roff[KC ? PER : 1];
    bool rok[KC ? PER : 1];
    float rparam[KC ? PER : 1];

    __device__ __forceinline__ void init(const Axis& ar, int r0, int rlimit, int transform, const float* row_param) {
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
                rparam[i] = (transform && rok[i]) ? row_param[row] : 0.f;
            }
        } else {
            const int row = r0 + tid % ROWS;
            rok[0] = row < rlimit;
            roff[0] = rok[0] ? axis_off(ar, row) : 0;
            rparam[0] = 0.f;
        }
    }

    // Issue the global loads of the tile whose first k is k0 (k valid in [k0, kend)).
    // Every load is UNCONDITIONAL on a clamped (always valid) address; nothing here consumes the loaded registers, so
    // the loads stay in flight under the MFMAs of the current tile. Invalid elements are zeroed (and the optional
    // exp transform applied) in store(), right before the LDS write. (A branch around a load, or a select right after
    // it, makes hipcc wait vmcnt(0) on the spot and serialises the L2 round trips.)
    __device__ __forceinline__ void load(const float* __restrict__ base, const Axis& ak, int k0, int kend, bool vec) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
            constexpr int QPR = BK / 4;
            const int q = tid % QPR;
            const int k = k0 + q * 4;
            const bool interior = k0 + BK <= kend;  // block-uniform: every k of this step is valid
            if (vec && interior) {
                const long koff = axis_off(ak, k);
#pragma unroll
                for (int i = 0; i < PER; ++i) regs[i] = *reinterpret_cast<const f32x4*>(base + roff[i] + koff);
            } else {
                long koff[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) koff[j] = axis_off(ak, min(k + j, kend - 1));
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = base[roff[i] + koff[j]];
                    regs[i] = v;
                }
            }
        } else {
            const int kb = k0 + (tid / ROWS) * (4 * PER);  // wave-uniform
            const float* rb = base + roff[0];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rb[axis_off(ak, min(kb + 4 * i + j, kend - 1))];
                regs[i] = v;
            }
        }
    }

    // Mask/transform the staged registers and write them into the LDS image dst[ROWS][LDK].
    __device__ __forceinline__ void store(float* __restrict__ dst, int k0, int kend, int transform, float scale) const {
        const int tid = threadIdx.x;
        if constexpr (KC) {
            constexpr int QPR = BK / 4;
            constexpr int RS = NT / QPR;
            const int q = tid % QPR;
            const int r = tid / QPR;
            const int k = k0 + q * 4;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v = regs[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = v[j];
                    if (transform == CGN_ATRANS_EXP) x = expf(x * scale - rparam[i]);
                    v[j] = (rok[i] && k + j < kend) ? x : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + (r + i * RS) * LDK + q * 4) = v;
            }
        } else {
            const int row = tid % ROWS;
            const int kl = (tid / ROWS) * (4 * PER);
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                f32x4 v = regs[i];
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

// BK: 16 for the 128-tile (3 blocks/CU by LDS and registers), 32 for the 64-tile (measured, tools/gemm_bench.py)
template <int BT, bool A_KC, bool B_KC, int ABL = 0, int BK = (BT == 128 ? 16 : 32)>
__global__ __launch_bounds__(NT) void gemm_kernel(const GemmParams p) {
    constexpr int TW = BT / 64;  // 32x32 accumulators per wave along each dim
    constexpr int LDK = BK + 4;  // LDS row stride (floats) of both operand images
    // One LDS array (both operands, two buffers): keeps every carve offset explicit and 16-byte aligned.
    constexpr int TILE = BT * LDK;  // floats per operand image
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE];

    const int ntiles = p.tiles_m * p.tiles_n;
    const int t = xcd_remap(blockIdx.x, ntiles);
    // m fastest: neighbouring tiles share the B (usually the weight) panel
    const int tile_m = t % p.tiles_m;
    const int tile_n = t / p.tiles_m;
    const int m0 = tile_m * BT, n0 = tile_n * BT;

    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    TileLoader<BT, A_KC, BK> la;
    TileLoader<BT, B_KC, BK> lb;
    la.init(p.am, m0, p.M, p.a_transform, p.a_row_param);
    lb.init(p.bn, n0, p.N, 0, nullptr);

    f32x16 acc[TW][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Pipeline (one barrier per K step, global loads a full step ahead, LDS double-buffered):
    //   step kt:  write tile kt+1 (registers, loaded during step kt-1) into buf[(kt+1)&1]
    //             issue the global loads of tile kt+2 into the staging registers
    //             MFMAs on buf[kt&1]
    //             barrier
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        la.load(p.A, p.ak, kbeg, kend, p.a_vec);
        lb.load(p.B, p.bk, kbeg, kend, p.b_vec);
        la.store(lds, kbeg, kend, p.a_transform, p.a_scale);
        lb.store(lds + TILE, kbeg, kend, 0, 0.f);
        if (nk > 1) {
            la.load(p.A, p.ak, kbeg + BK, kend, p.a_vec);
            lb.load(p.B, p.bk, kbeg + BK, kend, p.b_vec);
        }
    }
    __syncthreads();

    const int fr = lane & 31;  // fragment row/col
    const int fk = lane >> 5;  // k-slice within the 32x32x2 step
    for (int kt = 0; kt < nk; ++kt) {
        const float* As = lds + (kt & 1) * 2 * TILE;
        const float* Bs = As + TILE;
        if (ABL != 2 && kt + 1 < nk) {
            float* An = lds + ((kt + 1) & 1) * 2 * TILE;
            const int k1 = kbeg + (kt + 1) * BK;
            la.store(An, k1, kend, p.a_transform, p.a_scale);
            lb.store(An + TILE, k1, kend, 0, 0.f);
            if (kt + 2 < nk) {
                la.load(p.A, p.ak, k1 + BK, kend, p.a_vec);
                lb.load(p.B, p.bk, k1 + BK, kend, p.b_vec);
            }
        }
        // K order inside a step: lane half fk of block j holds k = 8j + 4fk + i for the i-th MFMA (any permutation of
        // k is legal as long as A and B use the same one), so a lane's 4 values are ONE 16-byte LDS read per tile.
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
            for (int e = 0; e < 4; ++e) {
                if (ABL == 1) {
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int j = 0; j < TW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)(e + i + lane), (float)(e - j + lane),
                                                                             acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int j = 0; j < TW; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur][i][e], b4[cur][j][e], acc[i][j], 0,
                                                                             0, 0);
                }
            }
        }
        __syncthreads();
    }

    // Epilogue. C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
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
                    p.ws[((long)blockIdx.y * p.M + row) * p.N + col] = acc[i][j][r];
                } else {
                    p.C[(long)row * p.ldc + col] = apply_epilogue(p, row, col, acc[i][j][r]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
    const long total = (long)p.M * p.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < p.splitk; ++z) s += p.ws[(long)z * total + idx];
        const int row = (int)(idx / p.N), col = (int)(idx % p.N);
        p.C[(long)row * p.ldc + col] = apply_epilogue(p, row, col, s);
    }
}

bool axis_vec_ok(const cgn_axis& contig, const cgn_axis& other) {
    if (contig.stride != 1) return false;
    if (contig.rpg != 0 && (contig.rpg % 4 != 0 || contig.gs % 4 != 0)) return false;
    if (other.stride % 4 != 0) return false;
    if (other.rpg != 0 && other.gs % 4 != 0) return false;
    return true;
}

int g_ablate = 0;  // diagnostic builds only (tools/gemm_bench.py): 1 = no LDS reads, 2 = no staging in the loop

template <int BT>
void launch(const GemmParams& p, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    if (a_kc && b_kc && g_ablate == 1)
        hipLaunchKernelGGL((gemm_kernel<BT, true, true, 1>), grid, dim3(NT), 0, s, p);
    else if (a_kc && b_kc && g_ablate == 2)
        hipLaunchKernelGGL((gemm_kernel<BT, true, true, 2>), grid, dim3(NT), 0, s, p);
    else if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, true, true>), grid, dim3(NT), 0, s, p);
    else if (a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, true, false>), grid, dim3(NT), 0, s, p);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_kernel<BT, false, true>), grid, dim3(NT), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_kernel<BT, false, false>), grid, dim3(NT), 0, s, p);
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
    if (splitk <= 0) {
        const int min_k = bt == 128 ? 512 : 256;  // keep at least this much K per split
        long want = (1024 + tiles - 1) / tiles;
        const long max_by_k = d->K / min_k;
        if (want > max_by_k) want = max_by_k;
        // a split costs a slab round trip plus a reduce launch (~5 us): skip it for products under ~0.25 GFLOP
        if ((double)d->M * d->N * d->K < 1.25e8) want = 1;
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

extern "C" int cgn_abi_version(void) { return 1; }

extern "C" void cgn_debug_set_ablation(int v) { g_ablate = v; }

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

extern "C" int cgn_gemm_f32(void* stream, const cgn_gemm_desc* d) {
    CGN_CHECK_ARG(d != nullptr);
    CGN_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0);
    if (d->M == 0 || d->N == 0) return CGN_OK;
    CGN_CHECK_ARG(d->A && d->B && d->C);
    CGN_CHECK_ARG(d->ldc >= d->N);
    CGN_CHECK_ARG(d->a_m.rpg < (1L << 31) && d->a_k.rpg < (1L << 31) && d->b_k.rpg < (1L << 31) &&
                  d->b_n.rpg < (1L << 31));
    CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE || (d->a_transform == CGN_ATRANS_EXP && d->a_row_param));
    CGN_CHECK_ARG(d->epilogue == CGN_EPI_LINEAR || d->epilogue == CGN_EPI_SOFTMAX_BWD);
    if (d->epilogue == CGN_EPI_SOFTMAX_BWD)
        CGN_CHECK_ARG(d->ep_logits && d->ep_lse && d->ep_rowdot && d->ep_target && d->ep_ld >= d->N);
    CGN_CHECK_ARG(d->force_tile == 0 || d->force_tile == 64 || d->force_tile == 128);
    // an operand is "k-contiguous" when its k stride is 1; otherwise lanes run along its row index
    const bool a_kc = d->a_k.stride == 1 || d->a_m.stride != 1;
    const bool b_kc = d->b_k.stride == 1 || d->b_n.stride != 1;
    CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE || a_kc);

    GemmParams p;
    p.A = d->A; p.B = d->B; p.C = d->C;
    p.bias0 = d->bias0; p.bias1 = d->bias1; p.a_row_param = d->a_row_param;
    p.ep_logits = d->ep_logits; p.ep_lse = d->ep_lse; p.ep_rowdot = d->ep_rowdot; p.ep_target = d->ep_target;
    p.ldc = d->ldc; p.ep_ld = d->ep_ld;
    p.am = {d->a_m.stride, d->a_m.rpg, d->a_m.gs};
    p.ak = {d->a_k.stride, d->a_k.rpg, d->a_k.gs};
    p.bk = {d->b_k.stride, d->b_k.rpg, d->b_k.gs};
    p.bn = {d->b_n.stride, d->b_n.rpg, d->b_n.gs};
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.a_transform = d->a_transform; p.epilogue = d->epilogue;
    p.alpha = d->alpha; p.beta = d->beta; p.a_scale = d->a_scale;
    p.ep_inv_tau = d->ep_inv_tau; p.ep_ce_scale = d->ep_ce_scale;
    p.a_vec = ((uintptr_t)d->A % 16 == 0) && (a_kc ? axis_vec_ok(d->a_k, d->a_m) : axis_vec_ok(d->a_m, d->a_k));
    p.b_vec = ((uintptr_t)d->B % 16 == 0) && (b_kc ? axis_vec_ok(d->b_k, d->b_n) : axis_vec_ok(d->b_n, d->b_k));

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
    p.ws = d->workspace;

    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)splitk);
    if (bt == 128)
        launch<128>(p, a_kc, b_kc, grid, s);
    else
        launch<64>(p, a_kc, b_kc, grid, s);
    CGN_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total = (long)d->M * d->N;
        int blocks = cgn_cdiv(total, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
        CGN_LAUNCH_CHECK();
    }
    return CGN_OK;
}
