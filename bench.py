#!/usr/bin/env python3
"""Benchmark of the CGN wake-sleep training step on MI355X (BASELINE.json metric: sentences/sec).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one full wake-sleep step over one batch of synthetic sentences per GPU: discriminator step
(model/cgn.py:269-341 of the reference) + c inference (:134-172) + encoder/generator step (:1049-1154, calc_z_loss=False
as train_functions.py:90 passes) + the three Adam updates. The beam sampler is excluded; the "generated" batch is
synthetic. Workload = BASELINE.json configs[1]: vocab 20k, seq len 20, batch 64/GPU, emb 300,
hidden 512, z 64, fp32. Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel — the GEMM kernel SYMBOL
with the most time in the step, found live: every product of the step is timed with HIP events recorded directly around
its MFMA kernel on the stream it is launched on (cgn_gemm_timed_f32), aggregated per symbol — a `step_mfma` object (all
MFMA FLOPs of the step / step time / peak), and, at N=1, a `cpu_baseline` object (the CPU oracle restatement of the same
step on the host cores, bounded sample). --gemm-table PATH writes the per-shape table of the step's products.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(max_word_len=12, max_seq_len=20, word_vocab_size=20000, char_vocab_size=64, word_embed_size=300,
            encoder_rnn_size=512, latent_variable_size=64, decoder_rnn_size=512, decoder_num_layers=2,
            sentiment_kernel_size=(3, 4, 5), sentiment_kernel_num=100)
BATCH, SEQ = 64, 20
# BASELINE.json configs[3] shapes (large-vocab stress: vocab 50k, seq len 40, hidden 1024, batch 256/GPU): selected with
# --workload cfg4, never the default bench line; `--compute bf16` runs it in the mode the config is stated in
CFG4 = dict(CFG2, max_seq_len=40, word_vocab_size=50000, encoder_rnn_size=1024, decoder_rnn_size=1024)
WORKLOADS = {"cfg2": (CFG2, 64, 20), "cfg4": (CFG4, 256, 40)}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 peak (~2.5 PF)
# An X6 product (fp32 operands split exactly into three bf16 terms, six partial bf16 MFMA products, fp32 accumulation:
# include/cgn_hip.h CGN_COMPUTE_F32X6) executes 6 bf16 MFMA FLOPs per algorithmic fp32 FLOP: its MFMA roof in
# algorithmic FLOPs is the bf16 peak / 6
X6_MFMA_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


def make_config(Config, cfgd=None):
    kw = dict(cfgd or CFG2)
    pos = [kw.pop(k) for k in ("max_word_len", "max_seq_len", "word_vocab_size", "char_vocab_size")]
    return Config(*pos, 5e-5, 7.9, 0.1, 0.1, 1.0, **kw)


def device_data(synthetic, cfg, B, L, seed, dev, drop_p=0.3):
    d = synthetic.synthetic_batch(cfg, B, L, seed=seed)
    F_ = len(cfg.sentiment_kernel_size) * cfg.sentiment_kernel_num
    g = torch.Generator().manual_seed(seed + 1)
    keep = 1.0 - drop_p
    d["mask_labelled"] = (torch.rand(B, F_, generator=g) < keep).float() / keep
    d["mask_generated"] = (torch.rand(B, F_, generator=g) < keep).float() / keep
    return {k: v.to(dev).contiguous() for k, v in d.items()}, d


def cpu_baseline(cfgd, host_data, global_step, budget_s=25.0):
    """The oracle restatement of the same step on the host cores ("port"), bounded to ~budget_s of CPU work. The ONLY
    place in this file that touches oracle/ (test infrastructure): a reported CPU baseline, never the thing measured."""
    from oracle import cgn_oracle as O
    ocfg = O.OracleConfig(**cfgd)
    torch.manual_seed(1234)
    model = O.CGN(ocfg, O.synthetic_word_embed(ocfg))
    opts = O.make_optimizers(model)
    data = {k: v for k, v in host_data.items() if not k.startswith("mask_")}
    masks = dict(labelled=host_data["mask_labelled"], generated=host_data["mask_generated"])
    B = data["enc_word"].shape[0]
    n_warm = 3  # allocator, thread pool, first-touch of the 20k x 512 weight pages
    t0 = time.perf_counter()
    for _ in range(n_warm):
        O.wake_sleep_step(model, opts, data, global_step, calc_z_loss=False, masks=masks)
    warm = time.perf_counter() - t0
    times = []
    while len(times) < 10 and (sum(times) + warm) < budget_s:
        t0 = time.perf_counter()
        O.wake_sleep_step(model, opts, data, global_step, calc_z_loss=False, masks=masks)
        times.append(time.perf_counter() - t0)
    if not times:
        times = [warm / n_warm]
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "sentences/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": host_cpu_model(),
            "sample": "%d full wake-sleep steps of the same cfg2 batch (B=%d) after %d warm-up steps, median; CPU fp32 "
                      "torch restatement oracle/cgn_oracle.py, %d torch threads" % (len(times), B, n_warm,
                                                                                  torch.get_num_threads())}


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def gemm_symbol(sym):
    """Kernel symbol of a product from ops.GEMM_PROFILE's key (csrc/gemm_core.h gemm_kernel's template arguments)."""
    if sym[0] == "bb":  # the bf16 x bf16 kernel (csrc/bb_gemm.hip): 256x256 tile with 8 waves, or 128x128 with 4
        b = lambda v: "true" if v else "false"  # noqa: E731
        waves = {256: "4, 2, 2, 4", 192: "4, 2, 2, 2", 128: "2, 2, 2, 2", 384: "2, 3, 2, 4"}[sym[1]]
        return "cgn_bb::bb_kernel<%s, %s, %s>" % (waves, b(sym[3]), b(sym[4]))
    bt, a_kc, b_kc, vec, exp, smb, bf16 = sym
    b = lambda v: "true" if v else "false"  # noqa: E731
    # vec: bits 0/1 = 16-byte loads on A/B (the template's VEC), bits 2.. = fast K addressing mode (the template's FK)
    # bit 4 = A stored as bf16 (the template's A16), bit 5 = the X6 kernel ran (three-term bf16 split of fp32 operands)
    if is_x6(sym):  # (bit 6: the variant whose epilogue also leaves the log-softmax statistics, the template's LSE)
        return "cgn_gemm::gemm_kernel<%d, %s, %s, %d, %s, 2, %d, %s, true, %d, false, true, %s>" % (
            bt, b(a_kc), b(b_kc), vec & 3, b(exp), 16 if bt == 128 else 32, b(smb), (vec >> 2) & 3, b(vec & 64))
    kstep = (16 if bt == 128 else 32) * (2 if bf16 else 1)
    return "cgn_gemm::gemm_kernel<%d, %s, %s, %d, %s, 2, %d, %s, %s, %d, %s, false, false>" % (
        bt, b(a_kc), b(b_kc), vec & 3, b(exp), kstep, b(smb), b(bf16), (vec >> 2) & 3, b(vec & 16))


def is_x6(sym):
    return sym[0] != "bb" and bool(sym[3] & 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--z-loss", action="store_true", help="include the z-reconstruction pass (calc_z_loss=True)")
    ap.add_argument("--eager", action="store_true", help="issue the step kernel by kernel instead of replaying a hipGraph")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg2 = BASELINE configs[1] (the metric's workload); cfg4 = configs[3] shapes in fp32 (stress)")
    ap.add_argument("--compute", choices=("f32", "bf16"), default="f32",
                    help="MFMA operand type: f32 = the metric's parity mode; bf16 = north-star's optional mode (operands "
                         "rounded to bf16 inside the GEMM kernels, fp32 accumulate/storage; tests/test_step_gpu.py bf16 case)")
    ap.add_argument("--gemm-table", metavar="PATH", help="write the per-shape table of the step's products (shape, tile, "
                    "split, us, TFLOP/s, fraction of peak; live HIP-event timings) to PATH")
    ap.add_argument("--force-dp", action="store_true",
                    help="single-GPU rehearsal of the data-parallel schedule: a one-rank RCCL group, collectives issued")
    args = ap.parse_args()

    if os.environ.get("CGN_BENCH_FAULT_DUMP"):  # debugging aid: all threads' tracebacks to stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CGN_BENCH_FAULT_DUMP"]), exit=False)
    import ldrs_amd  # noqa: F401
    from ldrs_amd import parallel
    from ldrs_amd.hip import ops
    from ldrs_amd.model.cgn import Controlled_Generation_Sentence
    from ldrs_amd.utils.config import Config
    from ldrs_amd.utils.functional import kld_coef, temp_coef
    from ldrs_amd.utils import synthetic

    if args.force_dp and args.gpus == 1 and "RANK" not in os.environ:
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
        torch.cuda.set_device(0)
        torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    # CGN_DP_BACKEND / CGN_DP_DEVICE: rehearsals on a one-GPU box only (two ranks on one device over gloo; RCCL refuses that)
    rank, world, local = parallel.init_from_env(backend=os.environ.get("CGN_DP_BACKEND"))
    if "CGN_DP_DEVICE" in os.environ:
        local = int(os.environ["CGN_DP_DEVICE"])
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the CGN hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cfgd, BATCH, SEQ = WORKLOADS[args.workload]
    cfg = make_config(Config, cfgd)
    if args.workload != "cfg2":
        args.no_cpu_baseline = True  # the CPU oracle needs minutes per step at this size
    ops.set_compute(args.compute)
    torch.manual_seed(1234)  # identical replicas on every rank
    model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
    if world > 1 or args.force_dp:
        parallel.attach(model, force=args.force_dp)
    eng = model.step_engine()
    opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
    data, host_data = device_data(synthetic, cfg, BATCH, SEQ, 1234 + rank, dev)
    global_step = 5000
    kc, tau = kld_coef(global_step), temp_coef(global_step)

    # The step is captured once as one hipGraph per branch (hip/graph.py) and replayed; its schedules and Adam step counts
    # live in device memory, so every replay is a full, different training step. N > 1 (data parallel): the same graphs,
    # each stream's graph cut where that stream starts / waits for a gradient all-reduce; the RCCL calls are issued
    # eagerly between the halves. CGN_DP_EAGER=1 (or a failed capture) keeps the kernel-by-kernel step.
    dp = world > 1 or args.force_dp
    want_graph = not args.eager and not (dp and os.environ.get("CGN_DP_EAGER") == "1")

    def eager_step():
        return eng.wake_sleep(opts, data, kc, tau, calc_z_loss=args.z_loss)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(step, profile_inside):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(args.warmup):
            step()
        barrier()
        if profile_inside:
            ops.GEMM_PROFILE = {"variant": None, "records": []}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            el = tt.item()
        return el, out

    def summarise(eg_losses, prof, passes):
        """Host-side copies of everything emit() needs from the device (the watchdog path must not touch the GPU).
        prof: ops.GEMM_PROFILE after `passes` instrumented steps - one record per product launch, HIP events directly
        around its MFMA kernel. Aggregated per kernel symbol and per (symbol, split, shape)."""
        ce = eg_losses[0].item()
        if not (ce == ce) or ce <= 0:
            raise SystemExit("non-finite loss in the benchmark step")
        by_sym, by_shape = {}, {}
        for sym, sk, M, N, K, e0, e1 in prof["records"]:
            ms = e0.elapsed_time(e1)
            fl = 2.0 * M * N * K
            a = by_sym.setdefault(sym, [0, 0.0, 0.0])
            a[0] += 1; a[1] += ms; a[2] += fl
            a = by_shape.setdefault((sym, sk, M, N, K), [0, 0.0])
            a[0] += 1; a[1] += ms
        lstm_flops = 0.0
        He, Hg = cfgd["encoder_rnn_size"], cfgd["decoder_rnn_size"]
        mm = lambda H: 2.0 * BATCH * 4 * H * H  # noqa: E731 - one [B,H] x [H,4H] product of a recurrence step
        lstm_flops += 2 * SEQ * 2 * mm(He)                              # encoder: 2 directions x (fwd h.W_hh, bwd dg.W_hh)
        # the fused two-layer launches also carry layer 1's W_ih products; where the step does not fuse (H > 512:
        # hip/step.py's own eligibility test) those products are ordinary GEMMs and already counted in gemm_flops
        fused = 3 if (cfgd["decoder_num_layers"] == 2 and ops.lstm2_persistent_bytes(BATCH, SEQ + 1, Hg, dev) > 0) else 2
        lstm_flops += (SEQ + 1) * 2 * fused * mm(Hg)                    # generator: 2 layers, forward and backward
        return dict(by_sym=by_sym, by_shape=by_shape, passes=passes, group_flops=prof.get("group_flops", 0.0) / passes,
                    lstm_flops=lstm_flops)

    def emit(elapsed, summary, launch, graph_timing, extra=None):
        """rank 0: the ONE JSON line of the contract, for a timed region of exactly K steps that took `elapsed`."""
        passes = summary["passes"]
        peak = FP32_MFMA_PEAK_TFLOPS if args.compute == "f32" else BF16_MFMA_PEAK_TFLOPS
        # the roofline kernel = the symbol the step spends the most time in
        sym, (launches, ms, flops) = max(summary["by_sym"].items(), key=lambda kv: kv[1][1])
        name = gemm_symbol(sym)
        if args.compute == "f32" and is_x6(sym):
            peak_k, peak_note = X6_MFMA_PEAK_TFLOPS, ("dense bf16 MFMA peak / 6: the kernel forms an fp32-accurate product "
                                                      "from six partial bf16 MFMA products (X6)")
        else:
            peak_k, peak_note = peak, ("dense fp32 MFMA peak" if args.compute == "f32" else "dense bf16 MFMA peak")
        # HBM bytes per launch of that symbol: PMC counters cannot be read from inside the run, so the figure comes from
        # the committed rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py; FETCH_SIZE and WRITE_SIZE in
        # separate runs, corrected for gfx950 as MI355X_MICROARCH.md prescribes), averaged over the symbol's launches
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if os.path.exists(tp) and args.workload == "cfg2" and args.compute == "f32":
            with open(tp) as f:
                tj = json.load(f)
            tot_b = tot_n = 0
            for k, v in tj["kernels"].items():
                if k.startswith(name.replace("cgn_gemm::", "") + " ") or k.startswith(name.replace("cgn_gemm::", "") + "("):
                    tot_b += v["traffic"] * v["launches"]
                    tot_n += v["launches"]
            if tot_n:
                traffic = tot_b / tot_n
                traffic_src = "profiles/r03_pmc_traffic.json (rocprofv3 --pmc, %d launches of this symbol)" % tot_n
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        ms_step = 1e3 * elapsed / args.steps
        gemm_flops = sum(v[2] for v in summary["by_sym"].values()) / passes
        step_flops = gemm_flops + summary["group_flops"] + summary["lstm_flops"]
        shapes = sorted(((M, N, K, sk, cnt / passes, 1e3 * t / cnt) for (sy, sk, M, N, K), (cnt, t) in
                         summary["by_shape"].items() if sy == sym), key=lambda r: -r[4] * r[5])
        out = {
            "metric": "wake-sleep training sentences/sec", "value": world * BATCH * args.steps / elapsed,
            "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.compute, "data": "synthetic",
            "config": {"workload": "CGN full wake-sleep training step (disc step + c inference + enc/gen step + 3 Adam), "
                                   + ("BASELINE configs[1]: vocab 20k, seq len 20, batch 64/GPU, emb 300, hidden 512, "
                                      "z 64, " if args.workload == "cfg2" else
                                      "BASELINE configs[3] SHAPES (stress, not the metric's workload): vocab 50k, "
                                      "seq len 40, batch 256/GPU, hidden 1024, z 64, ")
                                   + (("fp32 (operands, accumulation, storage; products on the %s)" % (
                                       "bf16 matrix pipe as exact three-term operand splits, X6: fp32 accuracy, "
                                       "include/cgn_hip.h CGN_COMPUTE_F32X6" if ops.GEMM_X6 else "fp32 MFMA"))
                                      if args.compute == "f32" else "OPTIONAL bf16 mode (bf16 MFMA operands, fp32 "
                                      "accumulation; large products on operands cast to bf16 once (bf16 x bf16 kernel: "
                                      "%s); logits / dlogits %s; parameters, optimizer state, cell states and every "
                                      "other activation fp32 in memory; not the metric's fp32 configuration)" % (
                                          "on" if ops.BB_GEMM else "off", "held in bf16" if ops.BF16_STORAGE else "fp32"))
                                   + ", calc_z_loss=%s" % args.z_loss,
                       "global_batch": world * BATCH, "seq_len": SEQ, "parallelism": "dp%d" % world,
                       "launch": launch},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak_k, "unit": "TFLOP/s",
                         "frac": achieved / peak_k, "peak_is": peak_note,
                         "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS if args.compute == "f32" else None,
                         "traffic": traffic,
                         "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_src,
                         "kernel": name + " (the GEMM kernel symbol with the most time in the step: %.0f us of %.0f us "
                                          "per step)" % (1e3 * ms / passes, 1e3 * ms_step),
                         "launches_per_step": launches / passes,
                         "avg_launch_us": 1e3 * ms / max(launches, 1),
                         "algorithmic_gflop_per_launch": flops / max(launches, 1) / 1e9,
                         "shapes": [{"M": M, "N": N, "K": K, "splitk": sk, "per_step": n, "us": round(us, 1),
                                     "tflops": round(2.0 * M * N * K / us / 1e6, 1)} for M, N, K, sk, n, us in shapes[:6]],
                         "timing": "HIP events directly around each launch of the symbol on its launch stream, "
                                   + ("instrumented eager pass after the timed graph replays" if graph_timing else
                                      "inside the timed region")},
            # all MFMA work of one step (every GEMM product + the grouped convolution products + the LSTM recurrences)
            # against the step time: algorithmic fp32 FLOPs per second over the fp32 MFMA peak (with the X6 products on
            # the bf16 pipe this is a rate relative to what fp32 MFMAs could reach, not a pipe occupancy)
            "step_mfma": {"gflop_per_step": step_flops / 1e9, "gemm_gflop": gemm_flops / 1e9,
                          "grouped_gflop": summary["group_flops"] / 1e9, "lstm_gflop": summary["lstm_flops"] / 1e9,
                          "achieved_tflops": step_flops / (ms_step * 1e-3) / 1e12,
                          "utilisation": step_flops / (ms_step * 1e-3) / 1e12 / peak, "utilisation_peak_tflops": peak,
                          "gemm_kernel_ms_per_step": sum(v[1] for v in summary["by_sym"].values()) / passes},
        }
        if extra:
            out.update(extra)
        if args.gemm_table and rank == 0:
            write_gemm_table(args.gemm_table, summary, peak, ms_step)
        if world == 1 and not args.no_cpu_baseline and not dp:
            out["cpu_baseline"] = cpu_baseline(cfgd, host_data, global_step)
        print(json.dumps(out), flush=True)

    def write_gemm_table(path, summary, peak, ms_step):
        """Per-shape table + a four-factor model of each product's fraction of peak: the kernel's steady-state rate
        (measured at 4096^3: 0.92 of peak for the 128-tile, 0.86 for the 64-tile) x balance (tiles per CU / its ceiling:
        the CUs that get one tile more set the makespan) x pad (M*N / the tile grid's area) x K share (K steps per tile
        against a fixed per-tile cost worth 8 (128-tile) / 4 (64-tile) K steps: prologue loads + epilogue)."""
        import math
        import re
        passes = summary["passes"]
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        rows = sorted(summary["by_shape"].items(), key=lambda kv: -kv[1][1])
        with open(path, "w") as f:
            f.write("# per-shape table of the step's GEMM products (%s, %s): HIP events directly around each MFMA kernel\n"
                    "# launch (split-K reduce launches excluded), %d instrumented steps; step time %.3f ms\n"
                    "# frac = TFLOP/s over the kernel's MFMA roof: fp32 MFMA 157.3, X6 kernels (<..., true, FK, false, true, LSE>) bf16 peak / 6 = 416.7\n"
                    "# model = steady-state rate (0.92 | 0.86 of peak at 4096^3) x balance (tiles/CU over its ceiling, %d CUs)\n"
                    "#         x pad (M*N over the tile grid's area) x kshare (K steps per tile / (K steps + 8 | 4 of fixed cost))\n" %
                    (args.workload, args.compute, passes, ms_step, cus))
            f.write("%-62s %2s %6s %6s %6s %5s %8s %8s %6s | %6s %7s %5s %5s %6s %6s\n" % (
                "kernel symbol", "sk", "M", "N", "K", "n/st", "us", "TFLOP/s", "frac", "tiles", "per CU", "bal", "pad",
                "kshare", "model"))
            tot = 0.0
            for (sym, sk, M, N, K), (cnt, t) in rows:
                us = 1e3 * t / cnt
                tf = 2.0 * M * N * K / us / 1e6
                tot += t / passes
                name = gemm_symbol(sym).replace("cgn_gemm::", "")
                m = re.match(r"gemm_(?:group_)?kernel<(\d+)", name)
                bt = int(m.group(1)) if m else 64
                if sym[0] == "bb":
                    bt = 128 if sym[1] == 128 else 256  # (192 = the 256x128 tile: the table's tile columns assume 256^2)
                x6 = args.compute == "f32" and is_x6(sym)
                pk = X6_MFMA_PEAK_TFLOPS if x6 else peak
                bf = args.compute == "bf16" or x6  # (the model's constants are fp32-MFMA measurements: no model column then)
                bk = 32 if sym[0] == "bb" else (16 if bt == 128 else 32) * (2 if args.compute == "bf16" else 1)
                tm, tn = math.ceil(M / bt), math.ceil(N / bt)
                tiles = tm * tn * max(sk, 1)
                per_cu = tiles / cus
                bal = per_cu / math.ceil(per_cu)
                pad = M * N / float(tm * bt * tn * bt)
                ksteps = K / max(sk, 1) / bk
                kshare = ksteps / (ksteps + (8 if bt == 128 else 4))
                model = (0.92 if bt == 128 else 0.86) * bal * pad * kshare
                f.write("%-62s %2d %6d %6d %6d %5.1f %8.1f %8.1f %6.3f | %6d %7.2f %5.2f %5.2f %6.2f %6.2f\n" % (
                    name, sk, M, N, K, cnt / passes, us, tf, tf / pk, tiles, per_cu, bal, pad, kshare,
                    model if not bf else float("nan")))
            f.write("# per symbol\n")
            for sym, (cnt, t, fl) in sorted(summary["by_sym"].items(), key=lambda kv: -kv[1][1]):
                pk = X6_MFMA_PEAK_TFLOPS if (args.compute == "f32" and is_x6(sym)) else peak
                f.write("%-62s launches/step %5.1f  us/step %8.1f  TFLOP/s %7.1f  frac %5.3f\n" %
                        (gemm_symbol(sym).replace("cgn_gemm::", ""), cnt / passes, 1e3 * t / passes, fl / t / 1e9,
                         fl / t / 1e9 / pk))
            f.write("# all products: %.1f us per step\n" % (1e3 * tot))

    def check_lstm():
        """The persistent LSTM launches need their workgroups resident together; a bounded wait that runs out leaves a
        sticky flag (include/cgn_hip.h). A run with the flag set is not a measurement of a correct step."""
        if ops.lstm_sticky_timeouts():
            raise SystemExit("a persistent LSTM launch timed out (invalid results): rerun with CGN_LSTM_PERSISTENT=0")

    def replicas_identical():
        """Data parallel: every rank must hold bit-identical weights after any number of steps (same initial weights,
        same reduced gradients). A broken reduction shows up here at once, since the ranks train on different shards."""
        chk = torch.stack([a.p.double().sum() for a in eng.arenas.values()] +
                          [a.p.double().abs().sum() for a in eng.arenas.values()])
        if world == 1:
            return True
        both = torch.stack([chk, -chk])
        torch.distributed.all_reduce(both, op=torch.distributed.ReduceOp.MAX)  # max(x) == -max(-x)  <=>  all equal
        return bool(torch.equal(both[0], -both[1]))

    # Data-parallel runs time the eager step FIRST and keep that result: the segmented-graph replay has been run on
    # this project's one-GPU boxes only with a one-rank RCCL group (and two ranks over gloo), so if it does not come
    # back on a real multi-GPU node a watchdog reports the eager measurement instead of losing the run.
    eager_result = None
    watchdog = None
    lstm_note = ""
    if dp and want_graph:
        initial = {k: a.p.clone() for k, a in eng.arenas.items()}  # to start over if a launch times out (see below)
        eager_result = timed(eager_step, True)
        eager_prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        torch.cuda.synchronize()
        eager_summary = summarise(eager_result[1][2], eager_prof, args.steps)
        # A persistent LSTM launch that timed out on ANY rank (its workgroups could not become resident together, e.g.
        # beside a collective's kernels) invalidates the run: all ranks switch to the per-timestep kernels and repeat.
        bad = torch.tensor([float(ops.lstm_sticky_timeouts()) +
                            (1.0 if os.environ.get("CGN_DP_SIMULATE_LSTM_TIMEOUT") == "1" else 0.0)], device=dev)  # test hook
        if world > 1:
            torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if bad.item() > 0:
            print("rank %d: a persistent LSTM launch timed out; repeating with the per-timestep kernels" % rank,
                  file=sys.stderr, flush=True)
            ops.LSTM_PERSISTENT = False
            ops.clear_lstm_sticky_timeouts()
            for k, a in eng.arenas.items():  # the timed-out steps trained on invalid activations: start over
                a.p.copy_(initial[k])
            for st in opts.values():
                st.m.zero_()
                st.v.zero_()
                st.step = 0
                st.sync_device_step()
            lstm_note = ", per-timestep LSTM kernels (a persistent launch timed out beside the collectives)"
            eager_result = timed(eager_step, True)
            eager_prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
            torch.cuda.synchronize()
            eager_summary = summarise(eager_result[1][2], eager_prof, args.steps)
        if not replicas_identical():
            raise SystemExit("data-parallel replicas diverged in the eager schedule")
        check_lstm()
        limit = float(os.environ.get("CGN_DP_GRAPH_TIMEOUT", "120"))

        def bail():
            # The replay did not come back. The eager measurement is still a valid one and is printed, but marked, and
            # the process exits NON-ZERO: a hung replay must never be recorded as a clean run.
            if rank == 0:
                emit(eager_result[0], eager_summary,
                     "eager (the segmented-graph replay did not finish within %.0f s)" % limit + lstm_note, False,
                     extra={"replay_hung": True})
            sys.stdout.flush()
            os._exit(3)

        import threading
        watchdog = threading.Timer(limit, bail)
        watchdog.daemon = True
        watchdog.start()

    use_graph = want_graph
    prof_passes = args.steps
    if use_graph:
        from ldrs_amd.hip.graph import GraphedWakeSleep
        try:
            graphed = GraphedWakeSleep(eng, opts, data, calc_z_loss=args.z_loss, warmup_kld=kc, warmup_tau=tau)
        except Exception as e:  # noqa: BLE001 - report and fall back: the eager step is always available
            if not dp:
                raise
            print("rank %d: segmented graph capture failed (%s: %s); running the eager step" %
                  (rank, type(e).__name__, e), file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            use_graph = False
    if world > 1:
        # every rank must take the same path: one failed capture sends all ranks to the eager step
        flag = torch.tensor([1.0 if use_graph else 0.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        use_graph = bool(flag.item() > 0.5)

    if use_graph:
        if os.environ.get("CGN_DP_SIMULATE_HANG") == "1" and dp:  # test hook for the watchdog path
            time.sleep(1e6)
        elapsed, losses = timed(lambda: graphed(kc, tau, totals=False), False)
        # HIP events cannot bracket single kernels inside a graph replay: the roofline kernel's launches are timed in
        # an instrumented eager pass of the SAME step on the same data right after the timed region (its kernels run
        # alone on the stream, as they do inside the graph's forward chain); rocprofv3's kernel trace of this command
        # (profiles/) shows the in-graph durations.
        ops.GEMM_PROFILE = {"variant": None, "records": []}
        prof_passes = max(3, min(args.steps, 10))
        for _ in range(prof_passes):
            eager_step()
        torch.cuda.synchronize()
    elif eager_result is not None:
        (elapsed, losses), ops.GEMM_PROFILE = eager_result, eager_prof
        prof_passes = args.steps
    else:
        elapsed, losses = timed(eager_step, True)
    if watchdog is not None:
        watchdog.cancel()
    if dp and use_graph:
        torch.cuda.synchronize()
        bad = torch.tensor([float(ops.lstm_sticky_timeouts())], device=dev)
        if world > 1:
            torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if bad.item() > 0 or not replicas_identical():
            # never seen; if the replayed schedule ever lost a reduction or starved a persistent launch, the eager
            # measurement (checked above) stands
            print("rank %d: the segmented-graph replay left %s; reporting the eager schedule" %
                  (rank, "a timed-out LSTM launch" if bad.item() > 0 else "diverged replicas"), file=sys.stderr, flush=True)
            ops.clear_lstm_sticky_timeouts()
            (elapsed, losses), ops.GEMM_PROFILE = eager_result, eager_prof
            use_graph, prof_passes = False, args.steps
    if use_graph and eager_result is not None and eager_result[0] < elapsed:
        # both were measured over exactly K steps: report the faster schedule, and say which one it is
        (elapsed, losses), ops.GEMM_PROFILE = eager_result, eager_prof
        use_graph, prof_passes = False, args.steps
    prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
    torch.cuda.synchronize()
    check_lstm()
    if rank == 0:
        launch = (("hipGraph replay" if not dp else "per-branch hipGraphs + eager RCCL all-reduce on the producing streams") if use_graph else
                  ("eager" if eager_result is None or not want_graph else
                   "eager (faster here than the segmented-graph replay, which also ran)"))
        torch.cuda.synchronize()
        emit(elapsed, summarise(losses[2], prof, prof_passes), launch + lstm_note, use_graph)
    if world > 1:
        torch.distributed.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
