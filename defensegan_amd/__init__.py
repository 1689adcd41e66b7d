"""MI355X-native Defense-GAN latent projection (the reconstruct() hot path of kabkabm/defensegan).

Host-side mirror of the reference's model object (gan.py), its eval harness (gan_defense.py) and config surface
(config.py) over hand-written HIP kernels behind a C ABI (csrc/, include/defensegan_hip.h).  No CPU fallback.
"""
from .gan import (CelebADefenseGAN, DefenseGANBase, FmnistDefenseDefenseGAN, MnistDefenseGAN, dataset_gan_dict,  # noqa: F401
                  gan_from_config)

__all__ = ["DefenseGANBase", "MnistDefenseGAN", "FmnistDefenseDefenseGAN", "CelebADefenseGAN", "dataset_gan_dict",
           "gan_from_config"]
