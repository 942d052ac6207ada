"""Drop-in alias of star_amd.models.VQSTAR under the reference's module path (train_completion.py:93 uses VQSTAR.vqstar)."""
from star_amd.models.VQSTAR import *  # noqa: F401,F403
from star_amd.models.VQSTAR import VQSTARViT, vq_amo_individual_star_patch8_dec256d4b, vqstar  # noqa: F401
