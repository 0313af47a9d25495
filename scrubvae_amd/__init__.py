"""scrubvae_amd -- MI355X-native (gfx950) implementation of scrubvae's SC-VAE training
hot path behind the reference's Python API.  See DESIGN.md."""
__version__ = "0.1.0"
