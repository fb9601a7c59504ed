from .build import build_quantizer  # noqa: F401
