#!/usr/bin/env python
"""Lab: cProfile of the drop-in harness loops (bench.py's harness leg) in a process set up like bench.py's."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
sys.argv = ["bench.py"]
bench = importlib.util.module_from_spec(spec); sys.modules["bench"] = bench; spec.loader.exec_module(bench)
from fashionern_aaai2024_amd import synth
from fashionern_aaai2024_amd.clip_model import create_model
from fashionern_aaai2024_amd.model import ERN
from fashionern_aaai2024_amd.run.test_fiq import generate_fiq_val_predictions
from fashionern_aaai2024_amd.tokenizer import ClipBpeTokenizer, register_tokenizer
from fashionern_aaai2024_amd.utils import extract_index_features

def main():
    cfg = synth.CLIP_CONFIGS["ViT-B-16"]; D = 512; device = torch.device("cuda:0")
    clip = create_model(cfg, device=device); clip.load_state_dict(synth.clip_state_dict(cfg, seed=0))
    model = ERN(clip, D, device, engine=clip.engine).load_state_dict(synth.fusion_state_dict(D, seed=0))
    letters = "abcdefghijklmnopqrstuvwxyz-"
    merges = [(a, b) for a in "sleroncdbtfpmv" for b in "aeioulrt"][:96] + [(a, b + "</w>") for a in letters[:20] for b in "esdrnty"][:96]
    register_tokenizer("bench-clip-bpe", ClipBpeTokenizer(merges))
    n_gal = 46000
    g = torch.Generator(device=device).manual_seed(77)
    index_features = torch.randn((n_gal, D), generator=g, device=device)
    index_names = [f"img{i:06d}" for i in range(n_gal)]
    host_local = torch.from_numpy(synth.local_feats(256, D, 31, "bench-ql"))
    rel = bench._BenchRelativeDataset(2048, n_gal, host_local)
    pool_im = torch.from_numpy(synth.images(32, cfg, 9)); pool_lc = torch.from_numpy(synth.local_feats(32, D, 9, "bench-il"))
    ds = bench._BenchIndexDataset(1024, pool_im, pool_lc)


    def prof(label, fn):
        fn(); torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
        print(f"===== {label}: {dt:.3f} s"); print("\n".join(s.getvalue().splitlines()[4:32]), flush=True)


    for nw in (0, 4):
        prof(f"generate_predictions lanes nw={nw}", lambda: generate_fiq_val_predictions(clip, rel, model, index_names, index_features, device, D, 64, nw, "bench-clip-bpe"))
    for nw in (0, 4):
        prof(f"extract_index_features nw={nw}", lambda: extract_index_features(ds, clip, 13, device, D, 32, nw))


if __name__ == "__main__":      # (forkserver workers import this file as __mp_main__)
    main()
