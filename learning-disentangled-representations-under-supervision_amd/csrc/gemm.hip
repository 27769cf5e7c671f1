// MFMA GEMM for gfx950, host side: C = epilogue(alpha * A·B + beta*C + bias). Descriptor validation, the tile / split-K
// plan and the entry points cgn_gemm_f32 / cgn_gemm_group_f32 / cgn_gemm_plan. The device templates are in gemm_core.h;
// this file instantiates them for exact-fp32 operands (v_mfma_f32_32x32x2_f32), gemm_bf16.hip for bf16 operands.
#include <cstdlib>

#include "gemm_core.h"

using namespace cgn_gemm;

namespace cgn_gemm {
void launch_gemm_f32ops(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    launch_gemm<false>(p, bt, a_kc, b_kc, grid, s);
}
void launch_group_f32ops(const GemmGroup& g, int vec, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    launch_group<false>(g, vec, a_kc, b_kc, blocks, s);
}
}  // namespace cgn_gemm

namespace {

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

// Tile / split-K heuristic: MI355X has 256 CUs and this kernel fits ~3-4 blocks per CU; aim for >= ~512 blocks.
// x6: the product will run an X6 kernel (run_gemm / cgn_gemm_plan know after prepare_gemm)
int plan_gemm(const cgn_gemm_desc* d, int* bt_out, int* splitk_out, int* kps_out, bool x6 = false) {
    // X6 products take the 128-tile from 256 x 256 outputs on: its 64-tile kernel stages three planes per operand for a
    // quarter of the MFMAs and is bound by its LDS writes (same-box A/B, cfg2 step: 2.63 -> 2.52 ms with the mid-size
    // products - 160-280 tiles of K 825-2484 - on the 128-tile; cfg4 shapes 45.1 -> 43.9 ms). CGN_X6_TILE=64 restores
    // the fp32 kernels' tile choice for A/B runs. Splitting these products further along K (>= 256 or 128 per split
    // instead of 512) made them 4-8 % faster alone and the step 3-9 % slower: slab traffic and reduce launches beside
    // the other streams' work (profiles/r03_x6_tiles.txt).
    static const int x6_tile = getenv("CGN_X6_TILE") ? atoi(getenv("CGN_X6_TILE")) : 128;
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
        // Padding: the 128-wide tile grid may cover much more than M x N (the soft embedding's N = 300 pads to 384, its
        // M = 1344 to 1408: 0.75 of the MFMA work is real). The 64-tile's lower steady-state rate (0.86 against 0.92 of
        // peak, profiles/r02_gemm_table.txt) is the smaller loss then: 1344 x 320 covers 0.94. Measured on that product
        // (tools/gemm_bench.py, profiles/r03_softemb_tile.txt): 171 -> 146 us alone, -28 us per step.
        if (bt == 128 && !(d->epilogue == CGN_EPI_SOFTMAX_BWD) && !(d->storage != 0)) {
            const double a128 = (double)cgn_cdiv(d->M, 128) * 128.0 * (double)cgn_cdiv(d->N, 128) * 128.0;
            const double a64 = (double)cgn_cdiv(d->M, 64) * 64.0 * (double)cgn_cdiv(d->N, 64) * 64.0;
            if (0.86 / a64 > 1.05 * 0.92 / a128) bt = 64;
        }
        if (x6 && x6_tile == 128 && d->M >= 256 && d->N >= 256) bt = 128;
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
        // (X6 128-tile products with 64-96 tiles - the LSTM weight gradients, 2048 x 512 x 1344 - were tried with splits
        // down to K = 256 / 128: the step got 1.5 % / 3.5 % slower, profiles/r03_x6_tiles.txt)
        const int min_k = few ? 128 : (bt == 128 ? 512 : 256);
        long want = tiles >= 512 ? 1 : (1024 + tiles - 1) / tiles;
        // tall-K on the 64-tile (the soft embedding, 105 tiles x K = 20000): ~5 blocks per CU measured better than 4
        // (inside the step: split 12 or 16 -> 2.98 ms, split 10 -> 3.00 ms)
        if (bt == 64 && d->K >= 4096 && tiles < 512) want = (1280 + tiles - 1) / tiles;
        if (few && want > 16) want = 16;
        const long max_by_k = d->K / min_k;
        if (want > max_by_k) want = max_by_k;
        // a split costs a slab round trip plus a reduce launch (~5 us): skip it for small many-tile products
        if (!few && (double)d->M * d->N * d->K < 1.25e8) want = 1;
        if (want < 1) want = 1;
        splitk = (int)want;
    }
    if (splitk > 1) {
        const long ldw = (d->N + 3) / 4 * 4;  // slab rows are padded to whole quads
        const long need = (long)splitk * d->M * ldw * (long)sizeof(float);
        if (!d->workspace || d->workspace_bytes < need) {
            if (d->force_splitk > 0) return CGN_ERR_WORKSPACE;
            const long fit = d->workspace ? d->workspace_bytes / ((long)d->M * ldw * (long)sizeof(float)) : 0;
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

static int prepare_gemm(const cgn_gemm_desc* d, GemmParams& p, bool& a_kc, bool& b_kc);
static bool x6_variant_exists(const GemmParams& p, bool a_kc, bool b_kc);

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
    bool x6 = false;
    if (d->compute == CGN_COMPUTE_F32X6) {
        GemmParams p;
        bool a_kc = false, b_kc = false;
        const int rc0 = prepare_gemm(d, p, a_kc, b_kc);
        if (rc0) return rc0;
        x6 = d->M > 0 && d->N > 0 && x6_variant_exists(p, a_kc, b_kc);
    }
    const int rc = plan_gemm(d, &bt, &sk, &kps, x6);
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
    CGN_CHECK_ARG(d->epilogue == CGN_EPI_LINEAR || d->epilogue == CGN_EPI_SOFTMAX_BWD ||
                  d->epilogue == CGN_EPI_LINEAR_STATS);
    if (d->epilogue == CGN_EPI_SOFTMAX_BWD)
        CGN_CHECK_ARG(d->ep_logits && d->ep_lse && d->ep_rowdot && d->ep_target && d->ep_ld >= d->N);
    CGN_CHECK_ARG(d->force_tile == 0 || d->force_tile == 64 || d->force_tile == 128);
    CGN_CHECK_ARG(d->compute == CGN_COMPUTE_F32 || d->compute == CGN_COMPUTE_BF16 || d->compute == CGN_COMPUTE_F32X6);
    const bool a16 = (d->storage & CGN_STORE_A_BF16) != 0, c16 = (d->storage & CGN_STORE_C_BF16) != 0;
    const bool ep16 = (d->storage & CGN_STORE_EP_BF16) != 0;
    CGN_CHECK_ARG((d->storage & ~(CGN_STORE_A_BF16 | CGN_STORE_C_BF16 | CGN_STORE_EP_BF16)) == 0);
    CGN_CHECK_ARG(d->storage == 0 || d->compute == CGN_COMPUTE_BF16);
    if (a16)  // a plain k-contiguous bf16 matrix whose rows start on 16-byte boundaries
        CGN_CHECK_ARG(d->a_k.stride == 1 && d->a_k.rpg == 0 && d->a_m.rpg == 0 && d->a_m.stride % 8 == 0 &&
                      d->a_m.stride >= d->K && d->K % 8 == 0 && d->K > 0 && (uintptr_t)d->A % 16 == 0);
    if (c16) CGN_CHECK_ARG(d->beta == 0.f && (uintptr_t)d->C % 2 == 0);
    if (ep16) CGN_CHECK_ARG(d->epilogue == CGN_EPI_SOFTMAX_BWD);
    if (d->c_t) CGN_CHECK_ARG(c16 && d->epilogue == CGN_EPI_SOFTMAX_BWD && d->ld_ct >= d->M);
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
    p.a_transform = d->a_transform;
    p.epilogue = d->epilogue == CGN_EPI_LINEAR_STATS ? CGN_EPI_LINEAR : d->epilogue;  // (C itself is the linear epilogue's)
    p.stats = nullptr; p.ldstats = 0;
    p.alpha = d->alpha; p.beta = d->beta; p.a_scale = d->a_scale;
    p.ep_inv_tau = d->ep_inv_tau; p.ep_ce_scale = d->ep_ce_scale;
    p.a16 = a16; p.c16 = c16; p.ep16 = ep16; p.ct = d->c_t; p.ldct = d->ld_ct;
    // 16-byte loads run along the operand's contiguous axis: k for a k-contiguous operand, the row index otherwise
    p.a_vec = ((uintptr_t)d->A % 16 == 0) &&
              (a_kc ? axis_vec_ok(d->a_k, d->a_m, d->K) : axis_vec_ok(d->a_m, d->a_k, d->M));
    p.b_vec = ((uintptr_t)d->B % 16 == 0) &&
              (b_kc ? axis_vec_ok(d->b_k, d->b_n, d->K) : axis_vec_ok(d->b_n, d->b_k, d->N));
    if (a16) {
        // the kernel addresses a bf16 A in 4-byte words (gemm_core.h TileLoader SRC16): strides in words, k in pairs
        CGN_CHECK_ARG(p.b_vec);
        p.a_vec = 1;
        p.am.stride = d->a_m.stride / 2;
    }
    // fast K addressing: plain K axes, 16-byte loads on both operands, both operands inside 2 GB. An operand's bound is
    // its last addressed element rounded up to the quad that a 16-byte load may touch (the vector-load conditions above
    // already promise that this quad is memory the caller owns).
    p.fastk = 0; p.a_bytes = 0; p.b_bytes = 0;
    // mode 2: the dW products dG^T . h, whose B operand walks a [B, T+1, H] history buffer with k = (sentence, timestep):
    // neither operand k-contiguous, A's K axis plain, B's two-level with at least half a K step per group
    const int bk_max = d->compute == CGN_COMPUTE_BF16 ? 64 : 32;
    const bool grouped_b = d->b_k.rpg > 0 && d->a_k.rpg == 0 && !a_kc && !b_kc && 2 * d->b_k.rpg >= bk_max &&
                           d->a_transform == CGN_ATRANS_NONE && d->epilogue == CGN_EPI_LINEAR;
    if (p.a_vec && p.b_vec && d->a_k.rpg == 0 && (d->b_k.rpg == 0 || grouped_b) && d->K > 0) {
        auto last = [](const cgn_axis& ax, long n) {
            const long i = n - 1;
            return ax.rpg > 0 ? (i / ax.rpg) * ax.gs + (i % ax.rpg) * ax.stride : i * ax.stride;
        };
        // (bf16 A: the span in 4-byte words, rows are whole quads of words: K % 8 == 0)
        const long ea = a16 ? (long)(d->M - 1) * (d->a_m.stride / 2) + d->K / 2
                            : (last(d->a_m, d->M) + last(d->a_k, d->K) + 4) / 4 * 4;
        const long eb = (last(d->b_k, d->K) + last(d->b_n, d->N) + 4) / 4 * 4;
        const long lim = (1L << 31) / 4 - 64;
        if (ea < lim && eb < lim && d->a_k.stride < (1 << 24) && d->b_k.stride < (1 << 24) && d->b_k.gs < lim) {
            p.fastk = grouped_b ? 2 : 1;
            p.a_bytes = (unsigned)(ea * 4);
            p.b_bytes = (unsigned)(eb * 4);
        }
    }
    CGN_CHECK_ARG((uintptr_t)d->workspace % 16 == 0);
    p.ws = d->workspace;
    p.dyn = d->dyn_scalars;
    p.order = g_order;
    return CGN_OK;
}

// mirrors launch_x6_t (gemm_core.h): the layouts the X6 translation unit instantiates
static bool x6_variant_exists(const GemmParams& p, bool a_kc, bool b_kc) {
    if (!(p.a_vec && p.b_vec) || p.fastk == 0 || p.a16 || p.c16 || p.ep16) return false;
    if (p.fastk == 2) return p.a_transform == CGN_ATRANS_NONE;
    if (p.a_transform == CGN_ATRANS_EXP) return a_kc && !b_kc;
    return !(!a_kc && b_kc);
}

// CGN_EPI_LINEAR_STATS is served by one kernel: the unsplit 128-tile X6 product of two k-contiguous operands
static bool stats_servable(const cgn_gemm_desc* d, const GemmParams& p, bool a_kc, bool b_kc) {
    return d->compute == CGN_COMPUTE_F32X6 && x6_variant_exists(p, a_kc, b_kc) && p.fastk == 1 && a_kc && b_kc &&
           d->a_transform == CGN_ATRANS_NONE && d->beta == 0.f && d->M >= 256 && d->N >= 256 &&
           (d->force_tile == 0 || d->force_tile == 128) && d->force_splitk <= 1 && d->storage == 0;
}

extern "C" int cgn_gemm_stats_chunks(const cgn_gemm_desc* d, int32_t* chunks) {
    CGN_CHECK_ARG(d && chunks);
    *chunks = 0;
    GemmParams p;
    bool a_kc = false, b_kc = false;
    cgn_gemm_desc lin = *d;
    if (lin.epilogue == CGN_EPI_LINEAR_STATS) lin.epilogue = CGN_EPI_LINEAR;
    if (lin.epilogue != CGN_EPI_LINEAR) return CGN_OK;
    const int rc = prepare_gemm(&lin, p, a_kc, b_kc);
    if (rc) return rc;
    if (d->M > 0 && d->N > 0 && stats_servable(d, p, a_kc, b_kc)) *chunks = cgn_cdiv(d->N, 64);
    return CGN_OK;
}

static int run_gemm(void* stream, const cgn_gemm_desc* d, hipEvent_t ev_start, hipEvent_t ev_stop,
                    int32_t* vec_mode = nullptr) {
    GemmParams p;
    bool a_kc = false, b_kc = false;
    {
        const int rc = prepare_gemm(d, p, a_kc, b_kc);
        if (rc) return rc;
    }
    if (d->M == 0 || d->N == 0) return CGN_OK;
    const bool x6 = d->compute == CGN_COMPUTE_F32X6 && x6_variant_exists(p, a_kc, b_kc);
    int bt, splitk, kps;
    {
        const int rc = plan_gemm(d, &bt, &splitk, &kps, x6);
        if (rc) return rc;
    }
    if (d->epilogue == CGN_EPI_LINEAR_STATS) {
        CGN_CHECK_ARG(stats_servable(d, p, a_kc, b_kc) && d->ep_stats && d->ep_stats_ld >= cgn_cdiv(d->N, 64) &&
                      (uintptr_t)d->ep_stats % 8 == 0);
        bt = 128;
        splitk = 1;
        kps = cgn_cdiv(d->K > 0 ? d->K : 1, KROUND) * KROUND;
        p.stats = d->ep_stats;
        p.ldstats = d->ep_stats_ld;
    }
    if (p.c16 && splitk > 1) {  // a bf16 C is written by the product kernel itself: no slabs, no reduce pass
        splitk = 1;
        kps = cgn_cdiv(d->K > 0 ? d->K : 1, KROUND) * KROUND;
    }
    p.tiles_m = cgn_cdiv(d->M, bt);
    p.tiles_n = cgn_cdiv(d->N, bt);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    p.k_per_split = kps;
    p.splitk = splitk;
    p.ldw = (d->N + 3) / 4 * 4;

    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)tiles, (unsigned)splitk);
    if (vec_mode) *vec_mode = (p.a_vec ? 1 : 0) | (p.b_vec ? 2 : 0) | (p.fastk << 2) | (p.a16 ? 16 : 0) | (x6 ? 32 : 0) |
                              (p.stats ? 64 : 0);
    if (ev_start && hipEventRecord(ev_start, s) != hipSuccess) return CGN_ERR_HIP_BASE + (int)hipGetLastError();
    if (p.a16)
        launch_gemm_a16(p, bt, b_kc, grid, s);
    else if (d->compute == CGN_COMPUTE_BF16)
        launch_gemm_bf16ops(p, bt, a_kc, b_kc, grid, s);
    else if (x6)
        launch_gemm_x6(p, bt, a_kc, b_kc, grid, s);
    else
        launch_gemm_f32ops(p, bt, a_kc, b_kc, grid, s);
    CGN_LAUNCH_CHECK();
    if (ev_stop && hipEventRecord(ev_stop, s) != hipSuccess) return CGN_ERR_HIP_BASE + (int)hipGetLastError();
    if (splitk > 1) {
        int blocks = cgn_cdiv((long)d->M * p.ldw / 4, 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
        CGN_LAUNCH_CHECK();
    }
    return CGN_OK;
}

extern "C" int cgn_gemm_f32(void* stream, const cgn_gemm_desc* d) { return run_gemm(stream, d, nullptr, nullptr); }

extern "C" int cgn_gemm_timed_f32(void* stream, const cgn_gemm_desc* d, void* ev_start, void* ev_stop,
                                  int32_t* vec_mode) {
    CGN_CHECK_ARG(ev_start && ev_stop);
    return run_gemm(stream, d, (hipEvent_t)ev_start, (hipEvent_t)ev_stop, vec_mode);
}

extern "C" int cgn_gemm_group_f32(void* stream, const cgn_gemm_desc* descs, int32_t n) {
    CGN_CHECK_ARG(descs && n >= 1 && n <= GROUP_MAX);
    GemmGroup g;
    g.n = 0;
    bool a_kc0 = false, b_kc0 = false, va = true, vb = true, bf16 = false;
    // X6 group (128-tile): every member an X6 product of the same orientation and fast-K mode with >= 256 x 256 outputs
    bool x6g = true, big_x6 = false;
    int fk0 = -1;
    long tiles_total = 0, ws_floats = 0;
    float* ws = nullptr;
    for (int i = 0; i < n; ++i) {
        const cgn_gemm_desc* d = descs + i;
        GemmParams p;
        bool a_kc = false, b_kc = false;
        const int rc = prepare_gemm(d, p, a_kc, b_kc);
        if (rc) return rc;
        CGN_CHECK_ARG(d->a_transform == CGN_ATRANS_NONE && d->epilogue == CGN_EPI_LINEAR && d->storage == 0);
        if (i == 0) bf16 = d->compute == CGN_COMPUTE_BF16;
        CGN_CHECK_ARG((d->compute == CGN_COMPUTE_BF16) == bf16);  // one operand type per group
        if (d->M == 0 || d->N == 0) continue;
        // One kernel variant per group: the orientation flags of the first member are used for all. The scalar
        // loaders address any strides (orientation only decides which index runs along the lanes), so a member whose
        // own orientation differs is still computed correctly once 16-byte loads are off for that operand.
        if (g.n == 0) {
            a_kc0 = a_kc; b_kc0 = b_kc;
        }
        va = va && p.a_vec && a_kc == a_kc0;  // 16-byte loads only if every member allows them
        vb = vb && p.b_vec && b_kc == b_kc0;
        if (fk0 < 0) fk0 = p.fastk;
        x6g = x6g && d->compute == CGN_COMPUTE_F32X6 && x6_variant_exists(p, a_kc, b_kc) && a_kc == a_kc0 && b_kc == b_kc0 &&
              p.fastk == fk0 && d->M >= 256 && d->N >= 256 && !(a_kc0 == false && b_kc0 == true);
        big_x6 = big_x6 || (d->compute == CGN_COMPUTE_F32X6 && x6_variant_exists(p, a_kc, b_kc) && d->M >= 256 && d->N >= 256);
        p.tiles_m = cgn_cdiv(d->M, 64);  // (re-tiled below for an X6 group)
        p.tiles_n = cgn_cdiv(d->N, 64);
        p.splitk = 1;
        p.k_per_split = cgn_cdiv(d->K > 0 ? d->K : 1, KROUND) * KROUND;
        p.ldw = (d->N + 3) / 4 * 4;
        p.order = 2;
        tiles_total += (long)p.tiles_m * p.tiles_n;
        CGN_CHECK_ARG(tiles_total < (1L << 28));
        if (g.n == 0) {
            ws = d->workspace;
            ws_floats = d->workspace ? d->workspace_bytes / (long)sizeof(float) : 0;
        }
        g.p[g.n] = p;
        ++g.n;
    }
    if (g.n == 0) return CGN_OK;
    if (!x6g && !bf16 && big_x6) {
        // X6 products that do not share one kernel variant: the 64-tile fp32 group kernel would be the slower choice
        for (int i = 0; i < n; ++i) {
            const int rc = run_gemm(stream, descs + i, nullptr, nullptr);
            if (rc) return rc;
        }
        return CGN_OK;
    }
    if (x6g) {
        tiles_total = 0;
        for (int i = 0; i < g.n; ++i) {
            g.p[i].tiles_m = cgn_cdiv(g.p[i].M, 128);
            g.p[i].tiles_n = cgn_cdiv(g.p[i].N, 128);
            tiles_total += (long)g.p[i].tiles_m * g.p[i].tiles_n;
        }
    }
    // K split: a group with few tiles (the text-CNN: ~100 tiles of 30-50 K steps) is a latency chain on a fraction of
    // the CUs; split every member so that ~768 blocks exist, each keeping >= 4 K steps. Slabs are carved out of the
    // first member's workspace; without room the group runs unsplit.
    int want = tiles_total >= 384 ? 1 : (int)((768 + tiles_total - 1) / tiles_total);
    const int min_k = x6g ? 512 : 128;  // (X6 128-tile: 512 resident slots, splits below K = 512 lose in the step)
    if (x6g) want = tiles_total >= 256 ? 1 : (int)((512 + tiles_total - 1) / tiles_total);
    if (want > 64) want = 64;  // (each member is further limited to K / 128 splits below)
    long ws_off = 0;
    bool any_split = false;
    for (int pass = 0; pass < 2 && want > 1; ++pass) {  // pass 0: does it fit?  pass 1: commit
        ws_off = 0;
        for (int i = 0; i < g.n; ++i) {
            GemmParams& p = g.p[i];
            int sk = want;
            if (sk > p.K / min_k) sk = p.K / min_k;
            if (sk < 1) sk = 1;
            const int kps = cgn_cdiv(cgn_cdiv(p.K > 0 ? p.K : 1, sk), KROUND) * KROUND;
            sk = p.K > 0 ? cgn_cdiv(p.K, kps) : 1;
            const long need = sk > 1 ? ((long)sk * p.M * p.ldw + 63) / 64 * 64 : 0;
            if (pass == 1 && sk > 1) {
                p.splitk = sk;
                p.k_per_split = kps;
                p.ws = ws + ws_off;
                any_split = true;
            }
            ws_off += need;
        }
        if (pass == 0 && (ws == nullptr || ws_off > ws_floats)) want = 1;
    }
    long blocks = 0;
    int max_reduce_blocks = 1;
    for (int i = 0; i < g.n; ++i) {
        const GemmParams& p = g.p[i];
        blocks += (long)p.tiles_m * p.tiles_n * p.splitk;
        g.tile_end[i] = (int)blocks;
        if (p.splitk > 1) {
            const int rb = cgn_cdiv((long)p.M * p.ldw / 4, 256);
            if (rb > max_reduce_blocks) max_reduce_blocks = rb;
        }
    }
    CGN_CHECK_ARG(blocks < (1L << 30));
    hipStream_t s = (hipStream_t)stream;
    const int vec = (va ? 1 : 0) | (vb ? 2 : 0);
    if (bf16)
        launch_group_bf16ops(g, vec, a_kc0, b_kc0, (unsigned)blocks, s);
    else if (x6g)
        launch_group_x6(g, a_kc0, b_kc0, fk0, (unsigned)blocks, s);
    else
        launch_group_f32ops(g, vec, a_kc0, b_kc0, (unsigned)blocks, s);
    CGN_LAUNCH_CHECK();
    if (any_split) {
        if (max_reduce_blocks > 1024) max_reduce_blocks = 1024;
        hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(max_reduce_blocks, g.n), dim3(256), 0, s, g);
        CGN_LAUNCH_CHECK();
    }
    return CGN_OK;
}
