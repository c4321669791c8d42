"""File-backed dataset classes with the tuple formats the harness consumes.

Counterparts of /root/reference/dataloader/{fashioniq.py:10-110, cirr.py:7-102, shoes.py:11-61} and the evaluation classes of
fashion200k_patch.py (:238-405): same constructor arguments,
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


# ---- Fashion200k (evaluation side) ---------------------------------------------------------------------------------------
def caption_post_process(s: str) -> str:
    """fashion200k_patch.py:52-54: captions double as retrieval ids, so the marks that would not survive as names are spelled out."""
    return s.strip().replace(".", "dotmark").replace("?", "questionmark").replace("&", "andmark").replace("*", "starmark")


def get_different_word(source_caption: str, target_caption: str):
    """fashion200k_patch.py:39-49: the modification text of a test query, "replace <a> with <b>": <a> is the first source word
    missing from the target caption, <b> the first target word missing from the source caption -- and, as in the reference's
    loops, the LAST word of the respective caption when no such word exists."""
    source_words, target_words = source_caption.split(), target_caption.split()
    source_word = next((w for w in source_words if w not in target_words), source_words[-1])
    target_word = next((w for w in target_words if w not in source_words), target_words[-1])
    return source_word, target_word, "replace " + source_word + " with " + target_word


class _Fashion200kBase(_FileDataset):
    """<root>/labels/*_<split>_*.txt: tab-separated ``relative image path, <unused>, caption`` lines (fashion200k_patch.py:299-313);
    local features at the image path with "women" replaced by ``local_dir`` plus ".pth" (:290; ``fashion200k_13_patch`` for the
    RN50 variant, :451).  ``split="val"`` reads the test files, as in the reference (:243-244)."""

    def __init__(self, root_path: str, split: str = "test", img_transform: Optional[Callable] = None, text_transform: Optional[Callable] = None,
                 local_dir: str = "local_features", strict: bool = False):
        import glob
        self.root_path, self.split = root_path, ("test" if split == "val" else split)
        self.preprocess, self.text_transform, self.local_dir, self.strict = img_transform, text_transform, local_dir, strict
        self.imgs: List[dict] = []
        for label_file in sorted(glob.glob(os.path.join(root_path, "labels", "*_" + self.split + "_*.txt"))):
            with open(label_file, "r", encoding="utf8") as fd:
                for line in fd.readlines():
                    cols = line.split("\t")
                    self.imgs.append({"file_path": cols[0], "captions": [caption_post_process(cols[2])], "modifiable": False})

    def _local(self, img_path: str) -> torch.Tensor:
        return _load_feature(img_path.replace("women", self.local_dir) + ".pth")

    def _rgb(self, img_path: str):
        import PIL.Image
        with open(img_path, "rb") as f:
            img = PIL.Image.open(f).convert("RGB")
        return self.preprocess(img) if self.preprocess is not None else img


class Fashion200kTestDataset(_Fashion200kBase):
    """Gallery: ``(img_id, image, local_feature)`` with ``img_id`` = the post-processed first caption (fashion200k_patch.py:282-293):
    several images share an id, recall counts any of them (run/test/test_200k.py:54-60)."""

    def _item(self, idx):
        img = self.imgs[idx]
        path = os.path.join(self.root_path, img["file_path"])
        return caption_post_process(img["captions"][0]), self._rgb(path), self._local(path)

    def __len__(self):
        return len(self.imgs)


class Fashion200kTestQueryDataset(_Fashion200kBase):
    """Queries from <root>/test_queries.txt (``source_file target_file`` per line):
    ``(ref_image, ref_id, modifier, target_id, len(modifier), ref_local_feature)`` (fashion200k_patch.py:340-354, 371-405)."""

    def __init__(self, root_path: str, split: str = "test", img_transform: Optional[Callable] = None, text_transform: Optional[Callable] = None,
                 local_dir: str = "local_features", strict: bool = False):
        super().__init__(root_path, split, img_transform, text_transform, local_dir, strict)
        self.source_files: List[str] = []
        self.ref_ids: List[str] = []
        self.targ_ids: List[str] = []
        self.modify_texts: List[str] = []
        if self.split == "test":
            index = {img["file_path"]: i for i, img in enumerate(self.imgs)}
            with open(os.path.join(root_path, "test_queries.txt")) as f:
                for line in f.readlines():
                    source_file, target_file = line.split()
                    src_cap = self.imgs[index[source_file]]["captions"][0]
                    tgt_cap = self.imgs[index[target_file]]["captions"][0]
                    self.source_files.append(os.path.join(root_path, source_file))
                    self.ref_ids.append(src_cap)
                    self.targ_ids.append(tgt_cap)
                    self.modify_texts.append(get_different_word(src_cap, tgt_cap)[2])

    def _item(self, idx):
        path = self.source_files[idx]
        modifier = self.modify_texts[idx]
        modifier = self.text_transform(modifier) if self.text_transform else modifier
        return (self._rgb(path), caption_post_process(self.ref_ids[idx]), modifier, caption_post_process(self.targ_ids[idx]), len(modifier),
                self._local(path))

    def __len__(self):
        return len(self.modify_texts)
