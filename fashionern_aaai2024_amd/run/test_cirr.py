"""Counterpart of /root/reference/run/test/test_cirr.py (compute_cirr_val_metrics :18-80, generate_cirr_val_predictions :83-137)."""
from . import _common
from ._cli import main as _main


def generate_cirr_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features, device, feature_dim,
                                  batch_size, num_workers, clip_model_name):
    out = _common.generate_predictions("cirr", clip_model, relative_val_dataset, model, index_names, index_features, device,
                                       feature_dim, batch_size, num_workers, clip_model_name)
    return out["predicted"], out["references"], out["targets"], out["members"]


def compute_cirr_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                             feature_dim, batch_size, num_workers, clip_model_name):
    predicted, reference_names, target_names, group_members = generate_cirr_val_predictions(
        clip_model, relative_val_dataset, model, index_names, index_features, device, feature_dim, batch_size, num_workers,
        clip_model_name)
    index_fused = _common.fuse_index(model, index_features, index_local_features, prepared=True)
    return _common.recalls_cirr(model, predicted, index_fused, index_names, reference_names, target_names, group_members)


if __name__ == "__main__":
    _main("cirr")
