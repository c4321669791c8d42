"""Counterpart of /root/reference/run/test/test_200k.py (compute_200k_val_metrics :20-61, generate_200k_val_predictions :64-113).
The reference moves the gallery to the CPU and ranks there (:48-51); here the ranking stays on the GPU."""
from . import _common
from ._cli import main as _main


def generate_200k_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features, device, feature_dim,
                                  batch_size, num_workers, clip_model_name):
    out = _common.generate_predictions("200k", clip_model, relative_val_dataset, model, index_names, index_features, device,
                                       feature_dim, batch_size, num_workers, clip_model_name)
    return out["predicted"], out["targets"]


def compute_200k_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                             feature_dim, batch_size, num_workers, clip_model_name):
    predicted, target_names = generate_200k_val_predictions(clip_model, relative_val_dataset, model, index_names, index_features,
                                                            device, feature_dim, batch_size, num_workers, clip_model_name)
    index_fused = _common.fuse_index(model, index_features, index_local_features, prepared=True)
    return _common.recalls_anyhit(model, predicted, index_fused, index_names, target_names, (10, 50))


if __name__ == "__main__":
    _main("200k")
