"""read.config(path) (reference params/read.py:8-42): YAML -> dict, missing keys filled with
None, disentangle.method defaults to {}, disentangle.features to the union of the method
lists, out_path "current" -> the config's directory, output sub-folders created, config
re-dumped next to the outputs."""
from pathlib import Path

import yaml

from .param_keys import PARAM_KEYS


def config(path, make_dirs=True):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for section, keys in PARAM_KEYS.items():
        cfg.setdefault(section, {})
        if cfg[section] is None:
            cfg[section] = {}
        for k in keys:
            cfg[section].setdefault(k, None)
    if not cfg["disentangle"]["method"]:
        cfg["disentangle"]["method"] = {}
    feats = cfg["disentangle"]["features"]
    if feats is None or len(feats) < 1:
        allf = []
        for v in cfg["disentangle"]["method"].values():
            allf += v
        cfg["disentangle"]["features"] = list(dict.fromkeys(allf))
    if cfg.get("out_path") == "current":
        cfg["out_path"] = str(Path(path).parent) + "/"
    if make_dirs and cfg.get("out_path"):
        print("Saving folder: {}".format(cfg["out_path"]))
        for sub in ("weights/", "checkpoints/", "latents/"):
            Path(cfg["out_path"] + sub).mkdir(parents=True, exist_ok=True)
        with open(cfg["out_path"] + "/model_config.yaml", "w") as f:
            yaml.dump(cfg, f)
    return cfg
