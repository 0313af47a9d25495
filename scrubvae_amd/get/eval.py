"""Latent embedding of a dataset (reference: scrubvae/get/eval.py:8-70): eval-mode `encode` of every batch on
the HIP encoder, cached as <out_path>/latents/<split>_<epoch>.npy."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch


def latents(config, model=None, epoch=None, loader=None, device="cuda", train_val_test="test", overwrite=False):
    """NOT FOR TRAINING (puts the model in eval mode).  Returns mu [N, z] on the CPU; loads the cached file when it exists
    (and checks its length against the loader's dataset) unless `overwrite`."""
    if model is not None:
        model.eval()
    latent_path = Path("{}/latents/{}_{}.npy".format(config["out_path"], train_val_test, epoch))
    if not latent_path.exists() or overwrite:
        print("Latent projections not found - Embedding dataset ...")
        out = []
        with torch.no_grad():
            for data in loader:
                data = {k: v.to(device) for k, v in data.items() if k in ["x6d", "root"]}
                out += [model.encode(data)["mu"].detach().cpu()]
        lat = torch.cat(out, axis=0)
        latent_path.parent.mkdir(parents=True, exist_ok=True)  # the reference relies on params.read having made it
        np.save(latent_path, lat.numpy())
    else:
        print("Found existing latent projections - Loading ...")
        lat = np.load(latent_path)
        if loader is not None:
            assert lat.shape[0] == len(loader.dataset)
        lat = torch.tensor(lat)
    nonzero_std_z = torch.where(lat.std(dim=0) > 0.1)[0]
    print("Latent dimensions with variance over the dataset > 0.1 : {}".format(len(nonzero_std_z)))
    print(lat.std(dim=0))
    return lat
