from star_amd.models.multiagent_mae import *  # noqa: F401,F403
from star_amd.models.multiagent_mae import AmortizedIndivMMAEViT  # noqa: F401
