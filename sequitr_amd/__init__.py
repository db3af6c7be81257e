"""sequitr_amd -- MI355X-native back end for the sequitr per-tile network hot path.

Drop-in scope: the U-Net / GAN leaf operators and the loss run
as hand-written HIP kernels (gfx950) behind a C-ABI (include/sequitr_hip.h); the
reference's job / operator / image-pipe interfaces are mirrored on the host side.
"""
__version__ = '0.1.0'
