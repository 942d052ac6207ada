from star_amd.utils.move_optim import optimizer_to  # noqa: F401
