from star_amd.utils.misc import NativeScalerWithGradNormCount, get_grad_norm_  # noqa: F401
