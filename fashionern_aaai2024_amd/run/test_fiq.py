"""Counterpart of /root/reference/run/test/test_fiq.py (compute_fiq_val_metrics :18-64, generate_fiq_val_predictions)."""
from . import _common
from ._cli import main as _main


def generate_fiq_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features, device, feature_dim,
                                   batch_size, num_workers, clip_model_name):
    out = _common.generate_predictions("fiq", clip_model, relative_val_dataset, model, index_names, index_features, device,
                                       feature_dim, batch_size, num_workers, clip_model_name)
    return out["predicted"], out["targets"]


def compute_fiq_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                              feature_dim, batch_size, num_workers, clip_model_name):
    predicted, target_names = generate_fiq_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features,
                                                             device, feature_dim, batch_size, num_workers, clip_model_name)
    index_fused = _common.fuse_index(model, index_features, index_local_features, prepared=True)
    recall_at10, recall_at50 = _common.recalls_unique(model, predicted, index_fused, index_names, target_names, (10, 50))
    return recall_at10, recall_at50


if __name__ == "__main__":
    _main("fiq")
