"""MI355X-native (gfx950) pullback-metric engine: drop-in for the ``local_encoder_pullback`` path of
enkeejunior1/Diffusion-Pullback (power-iteration SVD of the U-Net latent->feature Jacobian + DDIM loop).

Product code only: nothing here imports ``oracle/``; the HIP library is mandatory (no CPU fallback)."""
from .lib import DpbError, build, load            # noqa: F401
from .pullback import PullbackUNet, UNetOutput, bind   # noqa: F401
