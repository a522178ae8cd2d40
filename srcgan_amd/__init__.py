"""srcgan_amd -- MI355X (gfx950) native training hot path of huster-wgm/SRCGAN.

Exports the reference's names for the path (``from model import *`` at trainCas.py:12 resolves
``RDDBNet``; train.py:11 imports ``RDDBNetA, RDDBNetB, NLayerDiscriminator``; ``losses.L1Loss`` etc.).
Everything computes in libsrcgan_amd.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from ._native import set_default_dtype, LIB_PATH
from .model import RDDBNet, RDDBNetA, RDDBNetB, LegacyRDDBNet, ResDeconv, ESPCN, SRCNN, EDSR, SRDN, NLayerDiscriminator, ResidualDenseBlock_5, RRDB
from .losses import L1Loss, MSELoss, PSNRLoss, GANLoss

__all__ = ["RDDBNet", "RDDBNetA", "RDDBNetB", "LegacyRDDBNet", "ResDeconv", "ESPCN", "SRCNN", "EDSR", "SRDN", "NLayerDiscriminator", "ResidualDenseBlock_5", "RRDB",
           "L1Loss", "MSELoss", "PSNRLoss", "GANLoss", "set_default_dtype", "LIB_PATH"]
__version__ = "0.1.0"
