#!/usr/bin/env python
"""Headline benchmark: composed queries/sec of the encode -> fuse -> rank path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): FashionIQ ViT-B/16, one batch of 64 composed queries per GPU against a
46k-image fused gallery, fp32, synthetic data, random-init weights of the real architecture.

One STEP on one GPU = one batch of 64 composed queries, inputs already resident in HBM:
    encode_image(64 x 3x224x224)  ->  encode_text(64 x 77 tokens, one tower pass)
    -> ERN mode="test" fusion (2-layer BERT over 91 tokens, cross-attention, SR pooling, 3 Combiners)
    -> cosine top-50 against the replicated fused gallery [46000, 512].
The gallery is built before the timed region (every rank fuses its shard with mode="index", one RCCL all_gather).
N GPUs = N ranks (torchrun), query-data-parallel, weak scaling: value = N * 64 * K / (max-over-ranks time).

After the timed region the same step is run with libfern's HIP-event instrumentation (events on the launch stream
around every fp32-MFMA GEMM / attention / sweep / top-K launch) to fill `roofline`; on rank 0 at N=1 the CPU oracle
is timed on a bounded sample for `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fashionern_aaai2024_amd import distributed as fd  # noqa: E402
from fashionern_aaai2024_amd import synth  # noqa: E402
from fashionern_aaai2024_amd.clip_model import create_model  # noqa: E402
from fashionern_aaai2024_amd.model import ERN  # noqa: E402
from fashionern_aaai2024_amd.pipeline import ComposedQueryPipeline  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)
QUERY_BATCH, GALLERY, TOPK, D = 64, 46000, 50, 512


def cpu_baseline(clip_sd, fusion_sd, cfg, images, tokens, loc, gallery, sample, gpu_topk, repeats=2):
    """The CPU oracle (kind "port": this repo's restatement, pinned against the imported reference by tests/golden)
    on `sample` composed queries: encode image + text, fuse, rank against the same fused gallery."""
    from oracle import clip as oclip, fusion as ofusion, rank as orank
    # threads actually used: the host cores this process may run on, capped at 32 (torch's intra-op pool stops
    # scaling on these small per-query matrices well before that; an uncapped 256-thread pool is ~100x slower)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))
    torch.set_num_threads(threads)
    csd, fsd = ofusion.as_torch(clip_sd), ofusion.as_torch(fusion_sd)
    im, tk, lc, gal = images[:sample].cpu(), tokens[:sample].cpu(), loc[:sample].cpu(), gallery.cpu()

    def one():
        with torch.no_grad():
            rf = oclip.encode_image(csd, cfg, im)
            tg, ts = oclip.encode_text(csd, cfg, tk)
            q = ofusion.dvr_fuse(fsd, lc, ts, rf, tg)
            return orank.cosine_topk(q, gal, TOPK)

    warm = _timed(one)
    best = min([warm] + [_timed(one) for _ in range(repeats)]) if warm < 20 else warm      # keep the run bounded
    # the oracle is the checker as well: same queries, same gallery -> compare the HIP path's top-K with the CPU result
    o_s, o_i = one()
    g_s, g_i = gpu_topk[0][:sample].cpu(), gpu_topk[1][:sample].cpu()
    same_rows = int((g_i == o_i).all(dim=1).sum().item())
    return {"value": sample / best, "unit": "composed queries/sec", "cores": threads, "kind": "port",
            "sample": f"{sample} composed queries (ViT-B/16 image + text encode, fusion, top-{TOPK} of {gallery.shape[0]} rows), "
                      f"torch CPU fp32, best of {repeats}",
            "parity_vs_hip": {"queries": sample, "rows_with_identical_top%d_order" % TOPK: same_rows,
                              "positions_equal_frac": float((g_i == o_i).float().mean().item()),
                              "max_abs_cosine_diff": float((g_s - o_s).abs().max().item())}}


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    ap.add_argument("--gallery", type=int, default=GALLERY)
    ap.add_argument("--lanes", type=int, default=3, help="64-query batches kept in flight on separate HIP streams")
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp8"], default="fp32",
                    help="encoder operand precision of the TIMED path: fp32 = parity mode (the headline), bf16 = perf mode")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary legs (reduced-precision modes, lookup variant, 1M-row bf16 sweep, encode rate): the GEMM "
                         "dispatches of the run are then the headline step's, for comparing rocprofv3 averages with `roofline`")
    ap.add_argument("--pmc-mode", action="store_true",
                    help="for `rocprofv3 --pmc`: warm up, emit a marker dispatch, run exactly --steps steps, exit (no JSON)")
    args = ap.parse_args()

    rank, world, local = fd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("FERN_BENCH_SHARE_GPU"):      # debug only: several ranks on one GPU (with FERN_DIST_BACKEND=gloo)
        local = local % torch.cuda.device_count()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    cfg = synth.CLIP_CONFIGS["ViT-B-16"]
    n_gal = args.gallery

    # ---- weights (random init, real architecture) and synthetic inputs, resident in HBM --------------------
    clip_sd = synth.clip_state_dict(cfg, seed=0)
    fusion_sd = synth.fusion_state_dict(D, seed=0)
    clip = create_model(cfg, device=device)
    clip.load_state_dict(clip_sd)
    model = ERN(clip, D, device, engine=clip.engine).load_state_dict(fusion_sd)
    eng = model.engine
    seed = 42 + rank
    images = torch.from_numpy(synth.images(QUERY_BATCH, cfg, seed)).to(device)
    tokens = torch.from_numpy(synth.captions(QUERY_BATCH, cfg, seed)).to(device)
    loc = torch.from_numpy(synth.local_feats(QUERY_BATCH, D, seed)).to(device)
    g_raw = torch.from_numpy(synth.global_feats(n_gal, D, 7, "gallery")).to(device)
    g_loc = torch.from_numpy(synth.local_feats(n_gal, D, 7, "gallery-local")).to(device)

    # ---- gallery build (not in the step): shard -> mode="index" fuse -> RCCL all_gather ----------------------
    gallery = fd.build_gallery(eng, g_raw, g_loc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gallery = fd.build_gallery(eng, g_raw, g_loc)
    torch.cuda.synchronize()
    gallery_build_s = time.perf_counter() - t0
    del g_loc

    pipe = ComposedQueryPipeline(eng, lanes=args.lanes)
    pipe.set_precision(args.precision)

    def step():          # one batch of 64 composed queries; consecutive steps go to consecutive lanes (streams)
        return pipe.submit(images, tokens, loc, gallery, TOPK)

    def step_serial():   # the same work on the current stream (instrumented / PMC passes)
        rf = eng.encode_image(images)
        tg, ts = eng.encode_text(tokens)
        q = eng.dvr_fuse(rf, loc, tg, ts)
        return eng.sim_topk(q, gallery, TOPK)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, args.lanes)):      # every lane's workspace must exist before the timed region
        step()
    barrier()
    if args.pmc_mode:      # tools/pmc_traffic.py keys on this single-workgroup l2norm dispatch to find the measured steps
        eng.l2_normalize(torch.zeros(3, 64, device=device))
        for _ in range(args.steps):
            step_serial()
        barrier()
        return
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    value = world * QUERY_BATCH * args.steps / elapsed

    # ---- encoder perf mode (bf16 operands, fp32 accumulation) on the same workload: reported beside the fp32 headline,
    # never as `value` (north_star's parity bar -- scores within 1e-3, identical ordering -- is an fp32 statement)
    def reduced_precision_leg(prec, ref_scores, ref_idx):
        pipe.set_precision(prec)
        for _ in range(max(args.warmup, args.lanes)):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_b = step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        b_scores, b_idx = out_b.wait()
        eng.prof_enable(True)
        for _ in range(2):
            step_serial()
        sp = eng.prof_collect()
        eng.prof_enable(False)
        key = "gemm_fp8" if prec == "fp8" else "gemm_bf16"
        peak = BF16_MFMA_PEAK_TFLOPS      # the non-scaled fp8 MFMA (32x32x16) issues at the bf16 rate
        tfl = sp[key + "_flops"] / (sp[key + "_ms"] * 1e-3) / 1e12 if sp[key + "_ms"] > 0 else 0.0
        overlap = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ref_idx.cpu(), b_idx.cpu())) / ref_idx.numel()
        info = {"value": world * QUERY_BATCH * args.steps / el, "unit": "queries/sec", "ms_per_step": el / args.steps * 1e3,
                "dtype": ("bf16" if prec == "bf16" else "fp8 e4m3fn (per-token / per-channel scales)") +
                         " operands, f32 accumulate (encoder block GEMMs; attention and the fusion BERT blocks in bf16 operand form)",
                "gemm_tflops": tfl, "gemm_peak_tflops": peak, "gemm_frac": tfl / peak, "gemm_ms_per_step": sp[key + "_ms"] / 2,
                "gemm_f32_ms_per_step": sp["gemm_ms"] / 2, "attention_ms_per_step": sp["attn_ms"] / 2,
                "vs_fp32_top1_same": float((ref_idx[:, 0] == b_idx[:, 0]).float().mean().item()),
                "vs_fp32_top50_overlap": overlap,
                "vs_fp32_max_abs_top1_score_diff": float((ref_scores[:, 0] - b_scores[:, 0]).abs().max().item())}
        pipe.set_precision("fp32")
        return info

    bf16_info = fp8_info = None
    if args.precision == "fp32" and not args.headline_only:
        ref_scores, ref_idx = step().wait()
        bf16_info = reduced_precision_leg("bf16", ref_scores, ref_idx)
        fp8_info = reduced_precision_leg("fp8", ref_scores, ref_idx)

    # ---- roofline: instrumented passes (events around every kernel class), outside the timed region ----------
    step_serial()
    torch.cuda.synchronize()
    eng.prof_enable(True)
    prof_steps = max(2, min(5, args.steps))
    for _ in range(prof_steps):
        step_serial()
    st = eng.prof_collect()
    eng.prof_enable(False)
    gkey = {"fp32": "gemm", "bf16": "gemm_bf16", "fp8": "gemm_fp8"}[args.precision]      # the dominant GEMM family of this run
    gemm_tflops = st[gkey + "_flops"] / (st[gkey + "_ms"] * 1e-3) / 1e12 if st[gkey + "_ms"] > 0 else 0.0
    gemm_peak = F32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
    sweep_gbs = st["sweep_bytes"] / (st["sweep_ms"] * 1e-3) / 1e9 if st["sweep_ms"] > 0 else 0.0
    attn_tflops = st["attn_flops"] / (st["attn_ms"] * 1e-3) / 1e12 if st["attn_ms"] > 0 else 0.0
    enc_ips = lookup_qps = bf16_us = bf16_gbs = bf16_topk_us = None
    secondary = not args.headline_only
    # encoder throughput for the gallery side (bounded sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3 if secondary else 0):
        eng.encode_image(images)
    torch.cuda.synchronize()
    if secondary:
        enc_ips = 3 * QUERY_BATCH / (time.perf_counter() - t0)

    # reference-faithful query variant (run/test/test_fiq.py:104-107): the reference image feature is LOOKED UP in the raw
    # gallery index instead of being encoded per query -> text tower + fusion + rank only
    ref_rows = torch.arange(QUERY_BATCH, device=device) * 7 % n_gal

    def step_lookup():
        tg, ts = eng.encode_text(tokens)
        qf = eng.dvr_fuse(g_raw[ref_rows], loc, tg, ts)
        return eng.sim_topk(qf, gallery, TOPK)

    if secondary:
        for _ in range(3):
            step_lookup()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_lookup()
        torch.cuda.synchronize()
        lookup_qps = QUERY_BATCH * args.steps / (time.perf_counter() - t0)

        # HBM-bound variant of the sweep (BASELINE config 5 "bf16 similarity"): 1M-row bf16 gallery, same 64 queries
        big_n = 1_000_000
        gal_big = torch.nn.functional.normalize(torch.randn(big_n, D, device=device, generator=torch.Generator(device=device).manual_seed(3)), dim=-1)
        gal_bf16 = eng.gallery_to_bf16(gal_big)
        del gal_big
        q_unit = torch.nn.functional.normalize(torch.randn(QUERY_BATCH, D, device=device), dim=-1)
        for _ in range(2):
            eng.sim_topk_bf16(q_unit, gal_bf16, TOPK)
        eng.prof_enable(True)
        for _ in range(5):
            eng.sim_topk_bf16(q_unit, gal_bf16, TOPK)
        sb = eng.prof_collect()
        eng.prof_enable(False)
        bf16_us = sb["sweep_ms"] / max(1, sb["sweep_launches"]) * 1e3
        bf16_gbs = sb["sweep_bytes"] / max(1, sb["sweep_launches"]) / (bf16_us * 1e-6) / 1e9 if bf16_us > 0 else 0.0
        bf16_topk_us = sb["topk_ms"] / max(1, sb["topk_launches"]) * 1e3
        del gal_bf16

    result = None
    if rank == 0:
        result = {
            "metric": "composed queries/sec", "value": value, "unit": "queries/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16", "fp8": "fp8"}[args.precision], "data": "synthetic",
            "config": {"workload": "FashionIQ ViT-B/16 composed queries: 64-query batch per GPU vs 46k-image fused gallery "
                                   "(BASELINE.json configs[1])",
                       "query_batch_per_gpu": QUERY_BATCH, "gallery_rows": n_gal, "feature_dim": D, "top_k": TOPK,
                       "image": "3x224x224", "tokens": 77, "patch_feats": 13, "batches_in_flight": args.lanes,
                       "parallelism": f"dp{world} queries, gallery sharded for the build then all-gathered (RCCL)"},
            "roofline": {"bound": "mfma", "achieved": gemm_tflops, "peak": gemm_peak, "unit": "TFLOP/s",
                         "frac": gemm_tflops / gemm_peak, "traffic": None,
                         "kernel": {"fp32": "gemm_f32_glds_kernel / gemm_f32_kernel (fp32 MFMA GEMM, all tile variants)",
                                    "bf16": "gemm_bf16_glds_kernel (bf16 MFMA GEMM of the encoder blocks)",
                                    "fp8": "gemm_bf16_glds_kernel<FP8> (fp8 MFMA GEMM of the encoder blocks)"}[args.precision],
                         "gemm_ms_per_step": st[gkey + "_ms"] / prof_steps, "gemm_gflop_per_step": st[gkey + "_flops"] / prof_steps / 1e9,
                         "gemm_launches_per_step": st[gkey + "_launches"] / prof_steps},
            "roofline_sim_sweep": {"bound": "hbm", "achieved": sweep_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": sweep_gbs / HBM_PEAK_GBS, "traffic": None,
                                   "kernel": "gemm_f32_kernel as the cosine sweep inside fern_sim_topk",
                                   "sweep_us": st["sweep_ms"] / max(1, st["sweep_launches"]) * 1e3,
                                   "topk_us": st["topk_ms"] / max(1, st["topk_launches"]) * 1e3},
            "roofline_sim_sweep_bf16_1M": None if bf16_gbs is None else {"bound": "hbm", "achieved": bf16_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": bf16_gbs / HBM_PEAK_GBS, "traffic": None,
                                           "kernel": "sweep_bf16_kernel: 64 queries x 1M-row bf16 gallery (config 5's similarity mode)",
                                           "sweep_us": bf16_us, "topk_us": bf16_topk_us},
            "encoder_bf16": bf16_info,
            "encoder_fp8": fp8_info,
            "attention": {"achieved_tflops": attn_tflops, "ms_per_step": st["attn_ms"] / prof_steps},
            "lookup_variant": None if lookup_qps is None else {"value": lookup_qps * world, "unit": "queries/sec",
                               "note": "reference-faithful query path (test_fiq.py:104-107): reference features looked up in the index, "
                                       "no per-query image encode; one stream"},
            "gallery_build": {"index_fuse_all_gather_s": gallery_build_s, "rows_per_s": n_gal / gallery_build_s,
                              "encode_images_per_s_per_gpu": enc_ips},
        }
        traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(traffic_file) and args.precision == "fp32":      # HBM bytes per launch from a separate `rocprofv3 --pmc` pass of this command
            tr = json.load(open(traffic_file))
            result["roofline"]["traffic"] = tr.get("gemm", {}).get("hbm_bytes_per_launch")
            result["roofline"]["traffic_source"] = tr.get("source")
            result["roofline"]["algorithmic_bytes_per_launch"] = tr.get("gemm", {}).get("algorithmic_bytes_per_launch")
            result["roofline_sim_sweep"]["traffic"] = tr.get("sweep", {}).get("hbm_bytes_per_launch")
        if world == 1 and not args.no_cpu_baseline:
            gpu_topk = step_serial()
            torch.cuda.synchronize()
            result["cpu_baseline"] = cpu_baseline(clip_sd, fusion_sd, cfg, images, tokens, loc, gallery, args.cpu_sample, gpu_topk)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
