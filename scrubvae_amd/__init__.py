"""scrubvae_amd -- MI355X-native (gfx950) implementation of scrubvae's SC-VAE training
hot path behind the reference's Python API (``import scrubvae_amd as scrubvae``).
See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"


def __getattr__(name):  # lazy sub-packages: importing the package alone needs neither torch nor a GPU
    if name in ("get", "train", "model", "params", "data", "parallel", "ops", "eval"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
