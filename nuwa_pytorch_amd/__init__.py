"""nuwa_pytorch_amd -- MI355X (gfx950) native NUWA video-decoder training path.

Mirrors the export list of the reference package (nuwa_pytorch/__init__.py:1-5) for the classes on
the accelerated path; the hot forward/backward of each runs in libamdnuwa (HIP kernels, C-ABI in
include/amdnuwa.h).  There is no CPU / eager fallback for the decoder path."""
from .nuwa_pytorch import (NUWA, NUWASketch, NUWAVideoAudio, Sparse3DNA, CrossModalityCrossAttention, Attention,
                           FeedForward, SandwichNorm, ShiftVideoTokens, StableLayerNorm, Transformer,
                           ReversibleTransformer)
from .vqgan_vae import VQGanVAE
from .kernels import set_precision, get_precision
from .optimizer import get_optimizer

__all__ = ['NUWA', 'NUWASketch', 'NUWAVideoAudio', 'Sparse3DNA', 'CrossModalityCrossAttention', 'VQGanVAE',
           'set_precision', 'get_precision', 'get_optimizer']
