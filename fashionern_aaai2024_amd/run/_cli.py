"""``python -m fashionern_aaai2024_amd.run.test_fiq ...``: the reference drivers' flags (test_fiq.py:127-136) on this path.

The datasets themselves are not available offline, so by default the driver evaluates a seeded synthetic dataset of the same
tuple format (``--synthetic-gallery N --synthetic-queries Q``) with random-init or user-supplied (``--clip-path`` /
``--fusion-model-path``) weights, and prints the reference's summary lines.  With ``--data-root DIR`` it reads the real
layouts instead through ``fashionern_aaai2024_amd.dataloader`` (fashion-iq/..., cirr_dataset/..., shoes: SURVEY.md 8f rank 3),
preprocessing images on the GPU (``preprocess.gpu_preprocess``) and tokenising with ``tokenizer.get_tokenizer``.

Multi-GPU: start it under torchrun (``python -m torch.distributed.run --nproc-per-node N -m fashionern_aaai2024_amd.run.test_fiq ...``);
every rank takes ``cuda:LOCAL_RANK``, the gallery encode (`distributed.extract_index_features_sharded`), the gallery fusion and the
query loop are sharded over the ranks (run/_common.py) and rank 0 prints the reference's summary lines -- the same numbers as one process."""
from __future__ import annotations

import os
import zlib
from argparse import ArgumentParser
from statistics import mean

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import distributed as fd
from .. import synth
from ..clip_model import create_model
from ..model import ERN
from ..tokenizer import register_tokenizer
from ..utils import setup_seed

_WORDS = ("red", "blue", "longer", "shorter", "sleeves", "striped", "floral", "darker", "brighter", "collar", "more", "less")


def hash_tokenizer(vocab_size):
    def tok(texts, context_length=77):
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [1 + zlib.crc32(w.encode()) % (vocab_size - 3) for w in t.lower().split()][: context_length - 2]
            out[i, 0] = vocab_size - 2
            out[i, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.long)
            out[i, 1 + len(ids)] = vocab_size - 1
        return out
    return tok


class _Classic(Dataset):
    def __init__(self, names, images, local):
        self.names, self.images, self.local = names, images, local

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        return self.names[i], self.images[i], self.local[i]


class _Relative(Dataset):
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def synthetic_split(kind, cfg, d, n, q, seed):
    r = np.random.default_rng(seed)
    images = torch.from_numpy(synth.images(n, cfg, seed))
    local = torch.from_numpy(synth.local_feats(n, d, seed, "cli-local"))
    dup = kind == "200k"
    names = [f"cap{int(i)}" for i in r.integers(0, max(2, n // 3), size=n)] if dup else [f"img{i:06d}" for i in range(n)]
    cap = lambda: " ".join(r.choice(_WORDS, size=int(r.integers(2, 6))))  # noqa: E731
    items = []
    for _ in range(q):
        ref, tgt = (int(v) for v in r.choice(n, size=2, replace=False))
        if kind in ("fiq", "val"):
            items.append((names[ref], names[tgt], [cap(), cap()], local[ref]))
        elif kind == "cirr":
            others = [int(v) for v in r.choice([i for i in range(n) if i not in (ref, tgt)], size=4, replace=False)]
            items.append((names[ref], names[tgt], cap(), local[ref], [names[i] for i in [ref, tgt] + others]))
        elif kind == "shoes":
            items.append((names[ref], names[tgt], cap(), local[ref], local[tgt]))
        else:
            items.append((images[ref], names[ref], cap(), names[tgt], 3, local[ref]))
    return _Classic(names, images, local), _Relative(items)


def file_splits(kind, args, clip_model):
    """[(label, classic_dataset, relative_dataset)] read from --data-root in the reference's directory layouts."""
    from ..dataloader import CIRRDataset, Fashion200kTestDataset, Fashion200kTestQueryDataset, FashionIQDataset, ShoesDataset
    from ..preprocess import gpu_preprocess
    pre = gpu_preprocess(clip_model.engine, args.target_ratio, args.input_dim)
    root = args.data_root
    if kind in ("fiq", "val"):
        local_dir = "fashioniq_13_vit_2b" if args.clip_model_name.startswith("ViT") else "fashion_local13"      # fashioniq.py:64,174
        return [(t, FashionIQDataset("val", [t], "classic", pre, base_path=root, local_dir=local_dir),
                 FashionIQDataset("val", [t], "relative", pre, base_path=root, local_dir=local_dir)) for t in ("dress", "toptee", "shirt")]
    if kind == "cirr":
        return [("cirr", CIRRDataset("val", "classic", pre, base_path=root), CIRRDataset("val", "relative", pre, base_path=root))]
    if kind == "shoes":
        sp = root if root.endswith("/") else root + "/"
        return [("shoes", ShoesDataset("test", "classic", pre, shoes_path=sp), ShoesDataset("test", "relative", pre, shoes_path=sp))]
    local_dir = "local_features" if args.clip_model_name.startswith("ViT") else "fashion200k_13_patch"       # fashion200k_patch.py:290,451
    return [("200k", Fashion200kTestDataset(root, "val", pre, local_dir=local_dir),
             Fashion200kTestQueryDataset(root, "val", pre, local_dir=local_dir))]


def main(kind: str) -> None:
    p = ArgumentParser()
    p.add_argument("--dataset", default={"fiq": "fashionIQ", "val": "fashionIQ", "cirr": "CIRR", "shoes": "shoes", "200k": "fashion200k"}[kind], type=str)
    p.add_argument("--input-dim", default=224, type=int, help="224 for ViT, 288 for RN50x4")
    p.add_argument("--feature-dim", default=512, type=int, help="512 for ViT, 640 for RN50x4")
    p.add_argument("--patch-num", default=13, type=int)
    p.add_argument("--num-workers", type=int, default=0)
    p.add_argument("--batch-size", default=32, type=int)
    p.add_argument("--target-ratio", default=1.25, type=float, help="TargetPad target ratio (preprocessing: not on this path)")
    p.add_argument("--clip-model-name", default="ViT-B-16", type=str)
    p.add_argument("--clip-path", type=str, help="checkpoint with key 'CLIP' (open_clip state dict)")
    p.add_argument("--fusion-model-path", type=str, help="ERN.state_dict() checkpoint")
    p.add_argument("--synthetic-gallery", default=2000, type=int)
    p.add_argument("--synthetic-queries", default=256, type=int)
    p.add_argument("--seed", default=42, type=int)
    p.add_argument("--data-root", type=str, default=None,
                   help="directory holding fashion-iq/ (fiq, val), cirr_dataset/ (cirr), the shoes files (shoes) or the Fashion200k root (200k); default: synthetic data")
    p.add_argument("--precision", default="fp32", choices=["fp32", "f32x3", "bf16", "fp8", "mx8"],
                   help="encoder operand precision: fp32 = the reference's arithmetic (bit-exact fma chains); f32x3 = fp32-accurate GEMMs from "
                        "three bf16 planes per operand (~1.4x faster); bf16 / fp8 / mx8 = perf modes (ViT / text towers)")
    args = p.parse_args()
    setup_seed(args.seed)
    rank, world, local = fd.init_from_env()                 # torchrun: one process per GPU; a lone process is (0, 1, 0)
    if world > 1 and os.environ.get("FERN_BENCH_SHARE_GPU"):  # debug only: several ranks on the one GPU of a dev box (gloo)
        local = local % torch.cuda.device_count()
    device = torch.device("cuda", local) if world > 1 else torch.device("cuda")
    if world > 1:
        torch.cuda.set_device(device)
    say = print if rank == 0 else (lambda *a, **k: None)
    clip_model = create_model(args.clip_model_name, device=device, seed=None if args.clip_path else args.seed, precision=args.precision)
    if args.clip_path:
        clip_model.load_state_dict(torch.load(args.clip_path, map_location="cpu")["CLIP"])
    cfg = clip_model.cfg
    if cfg.embed_dim != args.feature_dim or cfg.image_size != args.input_dim:
        raise SystemExit(f"--feature-dim/--input-dim do not match {cfg.name} ({cfg.embed_dim}/{cfg.image_size})")
    if not args.data_root:
        register_tokenizer(args.clip_model_name, hash_tokenizer(cfg.vocab_size))      # real data: open_clip / FERN_CLIP_BPE_VOCAB / registered
    model = ERN(clip_model, args.feature_dim, device, engine=clip_model.engine)      # one context: --precision reaches the fusion blocks too
    if args.fusion_model_path:
        model.load_state_dict(torch.load(args.fusion_model_path, map_location="cpu"))
    else:
        model.init_random(args.seed)
    from . import test_200k, test_cirr, test_fiq, test_shoes, test_val
    fn = {"fiq": test_fiq.compute_fiq_val_metrics, "val": test_val.compute_fiq_val_metrics, "cirr": test_cirr.compute_cirr_val_metrics,
          "shoes": test_shoes.compute_shoes_val_metrics, "200k": test_200k.compute_200k_val_metrics}[kind]
    splits = ["dress", "toptee", "shirt"] if kind in ("fiq", "val") else [kind]
    if args.data_root:
        triples = file_splits(kind, args, clip_model)
    else:
        triples = [(split,) + synthetic_split(kind, cfg, args.feature_dim, args.synthetic_gallery, args.synthetic_queries, args.seed + i)
                   for i, split in enumerate(splits)]
    results = []
    for split, classic, relative in triples:
        feats, names, local = fd.extract_index_features_sharded(classic, clip_model, args.patch_num, device, args.feature_dim,
                                                                num_workers=0 if args.data_root else args.num_workers)
        res = fn(relative, clip_model, feats, local, names, model, device, args.feature_dim, args.batch_size, args.num_workers,
                 args.clip_model_name)
        say(split, "recalls:", res)
        results.append(res)
    avg = [mean(r[j] for r in results) for j in range(len(results[0]))]
    if kind == "cirr":
        say("Average: ", (avg[4] + avg[0]) / 2)        # (R@5 + R_subset@1) / 2, test_cirr.py:198
    elif kind == "val":
        say("Average recalls: ", avg)
    else:
        say("R@10: ", avg[0])
        say("R@50: ", avg[1])
        say("Average: ", (avg[0] + avg[1]) / 2)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
