from star_amd.utils.metrics import Metrics  # noqa: F401
