"""In-training validation entry points of /root/reference/run/valid/validate_{fiq,cirr,shoes}.py: the same functions
with fixed ``batch_size=32``, no loader workers argument and the tokenizer name hard-coded to 'RN50x4'
(validate_fiq.py:11-47,50-107; validate_cirr.py:11-72; validate_shoes.py:11-48)."""
from . import test_cirr, test_fiq, test_shoes

_BS, _WORKERS, _TOK = 32, 0, "RN50x4"


def compute_fiq_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                            feature_dim):
    return test_fiq.compute_fiq_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names,
                                            model, device, feature_dim, _BS, _WORKERS, _TOK)


def compute_cirr_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                             feature_dim):
    return test_cirr.compute_cirr_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names,
                                              model, device, feature_dim, _BS, _WORKERS, _TOK)


def compute_shoes_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features, index_names, model, device,
                              feature_dim):
    return test_shoes.compute_shoes_val_metrics(relative_val_dataset, clip_model, index_features, index_local_features,
                                                index_names, model, device, feature_dim, _BS, _WORKERS, _TOK)
