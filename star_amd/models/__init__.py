from . import multiagent_mae  # noqa: F401
