"""semantichuman_amd - MI355X-native spiral-convolution mesh-autoencoder training path.

Drop-in for the hot path of XiaokunSun/SemanticHuman (reference models.py / train_funcs.py):
hand-written HIP kernels for gfx950 behind a C ABI (include/sh_kernels.h), wrapped in the
reference's own nn.Module interface.  Importing the package works without a GPU; running a
model does not (no CPU fallback).
"""
from . import dataset, measure, optim  # noqa: F401
from .losses import FaceTables, edge_ratio_loss, eval_l1, l1_loss, recon_loss, vertex_l2_mm  # noqa: F401
from .models import SpiralAutoencoder, SpiralAutoencoder_multiz_partkps, SpiralConv  # noqa: F401

__version__ = "0.1.0"
