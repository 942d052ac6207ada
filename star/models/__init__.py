from star_amd.models import multiagent_mae  # noqa: F401
from . import VQSTAR  # noqa: F401
