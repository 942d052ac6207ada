from star_amd.utils.CoModule import CoModule  # noqa: F401
