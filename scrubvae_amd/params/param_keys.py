"""The four config sections and the keys the code reads (reference param_keys.py:1-34)."""
PARAM_KEYS = {
    "data": ["arena_size", "batch_size", "data_path", "dataset", "direction_process", "normalize", "remove_speed_outliers"],
    "disentangle": ["alpha", "balance_loss", "bandwidth", "features", "method", "polynomial", "var_mode"],
    "model": ["activation", "channel", "diag", "init_dilation", "kernel", "load_model", "prior", "start_epoch", "type",
              "window", "z_dim"],
    "train": ["beta_anneal", "lr", "num_epochs", "optimizer", "lr_schedule", "minimal_test"],
}
