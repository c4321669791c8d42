"""File-backed dataset classes with the tuple formats the harness consumes.

Counterparts of /root/reference/dataloader/{fashioniq.py:10-110, cirr.py:7-102, shoes.py:11-61}: same constructor arguments,
same directory layouts and JSON schemas, same ``(mode, split)`` -> tuple table; the root directories are arguments here (the
reference hard-codes "./" and an absolute /mnt path).  Evaluation splits are covered (``val`` / ``test`` / ``test1`` and the
``classic`` gallery mode); ``train`` tuples are provided for completeness.  Like the reference, an item that cannot be read
yields ``None`` (after a warning) and ``utils.collate_fn`` drops it -- pass ``strict=True`` to raise instead.

``preprocess`` is a callable ``PIL.Image -> [3, dim, dim] float tensor`` (the reference passes ``targetpad_transform(...)``,
dataloader/dataset.py:73-87); ``preprocess.gpu_preprocess(engine, target_ratio, dim)`` builds the HIP one.  Local (13-patch)
features are ``torch.save``d ``[13, D]`` float tensors named ``<image name>.pth`` (utils/extract_fashioniq_patch.py:156-168).
"""
from __future__ import annotations

import json
import os
import warnings
from typing import Callable, Dict, List, Optional, Sequence

import torch
from torch.utils.data import Dataset


def _load_json(path: str):
    with open(path, "r") as f:
        return json.load(f)


def _load_feature(path: str) -> torch.Tensor:
    t = torch.load(path, map_location="cpu")
    return t.float() if isinstance(t, torch.Tensor) else torch.as_tensor(t, dtype=torch.float32)


class _FileDataset(Dataset):
    strict = False

    def _open(self, path: str):
        import PIL.Image
        return self.preprocess(PIL.Image.open(path))

    def _item(self, index):
        raise NotImplementedError

    def __getitem__(self, index):
        if self.strict:
            return self._item(index)
        try:
            return self._item(index)
        except Exception as e:      # the reference prints and returns None (fashioniq.py:99-100); collate_fn drops None items
            warnings.warn(f"{type(self).__name__}[{index}] unreadable: {e}")
            return None


class FashionIQDataset(_FileDataset):
    """fashion-iq/{captions/cap.<type>.<split>.json, image_splits/split.<type>.<split>.json, images/<name>.png, <local_dir>/<name>.pth}.

    classic: ``(image_name, image, local_feature)``; relative: train ``(ref_image, target_image, captions, ref_local, target_local)``,
    val ``(ref_name, target_name, captions, ref_local)``, test ``(ref_name, ref_image, captions)`` (fashioniq.py:58-92).
    ``local_dir`` is ``fashion_local13`` (RN50x4 features) or ``fashioniq_13_vit_2b`` (the ViT variant, fashioniq.py:174)."""

    def __init__(self, split: str, dress_types: Sequence[str], mode: str = "relative", preprocess: Optional[Callable] = None,
                 base_path: str = "./", local_dir: str = "fashion_local13", strict: bool = False):
        if mode not in ("relative", "classic"):
            raise ValueError("mode should be in ['relative', 'classic']")
        if split not in ("test", "train", "val"):
            raise ValueError("split should be in ['test', 'train', 'val']")
        for t in dress_types:
            if t not in ("dress", "shirt", "toptee"):
                raise ValueError("dress_type should be in ['dress', 'shirt', 'toptee']")
        self.mode, self.split, self.dress_types, self.preprocess, self.strict = mode, split, list(dress_types), preprocess, strict
        self.root = os.path.join(base_path, "fashion-iq")
        self.local_root = os.path.join(self.root, local_dir)
        self.triplets: List[dict] = []
        self.image_names: List[str] = []
        for t in self.dress_types:
            self.triplets.extend(_load_json(os.path.join(self.root, "captions", f"cap.{t}.{split}.json")))
            self.image_names.extend(_load_json(os.path.join(self.root, "image_splits", f"split.{t}.{split}.json")))

    def _image(self, name: str):
        return self._open(os.path.join(self.root, "images", f"{name}.png"))

    def _local(self, name: str) -> torch.Tensor:
        return _load_feature(os.path.join(self.local_root, f"{name}.pth"))

    def _item(self, index):
        if self.mode == "classic":
            name = self.image_names[index]
            return name, self._image(name), self._local(name)
        trip = self.triplets[index]
        captions, ref = trip["captions"], trip["candidate"]
        ref_local = self._local(ref)
        if self.split == "train":
            tgt = trip["target"]
            return self._image(ref), self._image(tgt), captions, ref_local, self._local(tgt)
        if self.split == "val":
            return ref, trip["target"], captions, ref_local
        return ref, self._image(ref), captions

    def __len__(self):
        return len(self.triplets) if self.mode == "relative" else len(self.image_names)


class CIRRDataset(_FileDataset):
    """cirr_dataset/{cirr/captions/cap.rc2.<split>.json, cirr/image_splits/split.rc2.<split>.json (name -> relative path),
    cirr_local_13/<name>.pth}.

    classic: ``(image_name, image, local_feature)``; relative: train ``(ref_image, target_image, caption, ref_local, target_local)``,
    val ``(ref_name, target_hard_name, caption, ref_local, group_members)``, test1 ``(pair_id, ref_name, caption, group_members)``
    (cirr.py:49-91)."""

    def __init__(self, split: str, mode: str, preprocess: Optional[Callable] = None, base_path: str = "./", strict: bool = False):
        if split not in ("test1", "train", "val"):
            raise ValueError("split should be in ['test1', 'train', 'val']")
        if mode not in ("relative", "classic"):
            raise ValueError("mode should be in ['relative', 'classic']")
        self.mode, self.split, self.preprocess, self.strict = mode, split, preprocess, strict
        self.root = os.path.join(base_path, "cirr_dataset")
        self.triplets: List[dict] = _load_json(os.path.join(self.root, "cirr", "captions", f"cap.rc2.{split}.json"))
        self.name_to_relpath: Dict[str, str] = _load_json(os.path.join(self.root, "cirr", "image_splits", f"split.rc2.{split}.json"))
        self._names = list(self.name_to_relpath.keys())

    def _image(self, name: str):
        return self._open(os.path.join(self.root, self.name_to_relpath[name]))

    def _local(self, name: str) -> torch.Tensor:
        return _load_feature(os.path.join(self.root, "cirr_local_13", f"{name}.pth"))

    def _item(self, index):
        if self.mode == "classic":
            name = self._names[index]
            return name, self._image(name), self._local(name)
        trip = self.triplets[index]
        members, ref, caption = trip["img_set"]["members"], trip["reference"], trip["caption"]
        if self.split == "test1":
            return trip["pairid"], ref, caption, members
        ref_local = self._local(ref)
        if self.split == "train":
            tgt = trip["target_hard"]
            return self._image(ref), self._image(tgt), caption, ref_local, self._local(tgt)
        return ref, trip["target_hard"], caption, ref_local, members

    def __len__(self):
        return len(self.triplets) if self.mode == "relative" else len(self.name_to_relpath)


class ShoesDataset(_FileDataset):
    """<shoes_path>/{split.<split>.json (list of relative image paths), triplet.<split>.json (ImageName, ReferenceImageName,
    RelativeCaption)} and <local_feature_path>/<image stem>.pth (shoes.py:11-61; image names are the file stems without ".jpg").

    classic: ``(image_name, image, local_feature)``; relative: train ``(ref_image, target_image, caption, ref_local, target_local)``,
    otherwise ``(ref_name, target_name, caption, ref_local, target_local)``."""

    def __init__(self, split: str, mode: str = "relative", preprocess: Optional[Callable] = None, shoes_path: str = "./shoes_dataset/",
                 local_feature_path: Optional[str] = None, strict: bool = False):
        if mode not in ("relative", "classic"):
            raise ValueError("mode should be in ['relative', 'classic']")
        self.mode, self.split, self.preprocess, self.strict = mode, split, preprocess, strict
        self.shoes_path = shoes_path
        self.local_feature_path = local_feature_path if local_feature_path is not None else os.path.join(shoes_path, "shoes_local_feature_13")
        self.image_id2name: List[str] = _load_json(os.path.join(shoes_path, f"split.{split}.json"))
        self.annotations: List[dict] = _load_json(os.path.join(shoes_path, f"triplet.{split}.json")) if mode == "relative" else []

    @staticmethod
    def _stem(path: str) -> str:
        return path.split("/")[-1].split(".jpg")[0]

    def _local(self, name: str) -> torch.Tensor:
        return _load_feature(os.path.join(self.local_feature_path, f"{name}.pth"))

    def _item(self, index):
        if self.mode == "classic":
            rel = self.image_id2name[index]
            name = self._stem(rel)
            return name, self._open(os.path.join(self.shoes_path, rel)), self._local(name)
        ann = self.annotations[index]
        ref_rel, tgt_rel = ann["ReferenceImageName"], ann["ImageName"]
        ref, tgt = self._stem(ref_rel), self._stem(tgt_rel)
        ref_local, tgt_local = self._local(ref), self._local(tgt)
        if self.split == "train":
            return (self._open(os.path.join(self.shoes_path, ref_rel)), self._open(os.path.join(self.shoes_path, tgt_rel)),
                    ann["RelativeCaption"], ref_local, tgt_local)
        return ref, tgt, ann["RelativeCaption"], ref_local, tgt_local

    def __len__(self):
        return len(self.annotations) if self.mode == "relative" else len(self.image_id2name)
