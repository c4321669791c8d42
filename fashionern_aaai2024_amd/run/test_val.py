"""Counterpart of /root/reference/run/test/test_val.py (FashionIQ "VAL split": R@1,5,10,15,20,30,40,50; :18-67)."""
from . import _common
from ._cli import main as _main

KS = (1, 5, 10, 15, 20, 30, 40, 50)


def generate_fiq_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features, device, feature_dim,
                                 batch_size, num_workers, clip_model_name):
    out = _common.generate_predictions("val", clip_model, relative_val_dataset, model, index_names, index_features, device,
                                       feature_dim, batch_size, num_workers, clip_model_name)
    return out["predicted"], out["targets"]


def compute_fiq_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                            feature_dim, batch_size, num_workers, clip_model_name):
    predicted, target_names = generate_fiq_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features,
                                                           device, feature_dim, batch_size, num_workers, clip_model_name)
    index_fused = _common.fuse_index(model, index_features, index_local_features, prepared=True)
    return _common.recalls_unique(model, predicted, index_fused, index_names, target_names, KS)


if __name__ == "__main__":
    _main("val")
