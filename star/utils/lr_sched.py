from star_amd.utils.lr_sched import adjust_learning_rate  # noqa: F401
