from .build import build_observer  # noqa: F401
