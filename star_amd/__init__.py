"""star_amd: an MI355X-native (gfx950) implementation of the STAR scene-completion hot path.

Host code mirrors the reference's `star.models` / `star.utils` interface; all arithmetic of the path runs
in hand-written HIP kernels behind the C ABI of libstar_hip.so (include/star_hip.h)."""
__version__ = "0.1.0"
