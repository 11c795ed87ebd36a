from . import c3d, dose_pyfer, oar_transeg  # noqa: F401
