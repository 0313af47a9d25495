"""scrubvae.get.model drop-in (reference: src/scrubvae/get/model.py:4-151): same signature,
same feature-dimension table, same disentangle wiring; returns the MI355X ResVAE."""
from __future__ import annotations

import torch

FEAT_DIMS = {"avg_speed": 1, "part_speed": 4, "avg_speed_3d": 3, "heading": 2, "heading_change": 1, "fluorescence": 1}


def model(model_config, load_model, epoch, disentangle_config, n_keypts, direction_process, loss_config=None,
          arena_size=None, kinematic_tree=None, bound=False, discrete_classes=None, device="cuda", verbose=1):
    from scrubvae_amd.model.disentangle import (AdvNetScrubber, GRScrubber, LinearProjection, MovingAverageFilter,
                                                MovingAvgLeastSquares, QuadraticDiscriminantFilter)
    from scrubvae_amd.model.residual import ResVAE

    feat_dim_dict = dict(FEAT_DIMS)
    feat_dim_dict["frame_speed"] = model_config["window"] - 1
    if discrete_classes is not None:
        feat_dim_dict.update({k: len(v) for k, v in discrete_classes.items()})

    in_channels = n_keypts * 6
    if direction_process in ["x360", "midfwd", None]:
        in_channels += 3

    methods = disentangle_config["method"] or {}
    if "conditional" in methods:
        conditional_dim = sum(feat_dim_dict[k] for k in methods["conditional"])
        conditional_keys = methods["conditional"]
    else:
        conditional_keys, conditional_dim = None, 0

    disentangle = {}
    if "linear" in methods:  # get/model.py:40-49; first, as in the reference (the other heads then read z_null)
        disentangle["linear"] = {feat: LinearProjection(model_config["z_dim"], feat_dim_dict[feat], bias=False)
                                 for feat in methods["linear"]}
    if "grad_reversal" in methods:
        disentangle["grad_reversal"] = {
            feat: GRScrubber(model_config["z_dim"], feat_dim_dict[feat], alpha=disentangle_config["alpha"], bound=bound)
            for feat in methods["grad_reversal"]}
    if "adversarial_net" in methods:
        disentangle["adversarial_net"] = {
            feat: AdvNetScrubber(model_config["z_dim"] + conditional_dim) for feat in methods["adversarial_net"]}

    if "moving_avg_lsq" in methods:  # get/model.py:73-84
        disentangle["moving_avg_lsq"] = {
            feat: MovingAvgLeastSquares(model_config["z_dim"], feat_dim_dict[feat], bias=loss_config[feat + "_mals"] < 0,
                                        polynomial_order=disentangle_config["polynomial"], l2_reg=disentangle_config["l2_reg"])
            for feat in methods["moving_avg_lsq"]}

    if "qda" in methods:  # get/model.py:86-94
        disentangle["qda"] = {feat: QuadraticDiscriminantFilter(model_config["z_dim"], discrete_classes[feat]) for feat in methods["qda"]}
    if "moving_avg" in methods:  # get/model.py:96-104
        disentangle["moving_avg"] = {feat: MovingAverageFilter(model_config["z_dim"], discrete_classes[feat])
                                     for feat in methods["moving_avg"]}

    if model_config["type"] != "rcnn":
        raise ValueError("only model.type == 'rcnn' exists (reference get/model.py:116)")
    vae = ResVAE(
        in_channels=in_channels, kernel=model_config["kernel"], z_dim=model_config["z_dim"],
        window=model_config["window"], activation=model_config.get("activation", "prelu") or "prelu",
        is_diag=model_config["diag"], conditional_dim=conditional_dim,
        init_dilation=model_config.get("init_dilation"), disentangle=disentangle,
        disentangle_keys=disentangle_config.get("features"), conditional_keys=conditional_keys,
        arena_size=arena_size, kinematic_tree=kinematic_tree, prior=model_config.get("prior", "gaussian") or "gaussian",
        ch=model_config["channel"], discrete_classes=discrete_classes, device=device)
    if verbose > 0:
        print(vae)
    if load_model is not None:
        load_path = "{}/weights/epoch_{}.pth".format(load_model, epoch)
        print("Loading Weights from:\n{}".format(load_path))
        state_dict = torch.load(load_path, map_location="cpu", weights_only=True)
        missing, unexpected = vae.load_state_dict(state_dict, strict=False)
        if verbose > 0:
            print("Missing Keys: {}".format(missing))
            print("Unexpected Keys: {}".format(unexpected))
    return vae
