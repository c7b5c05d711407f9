from .spec import DataSpec  # noqa: F401  (reference src/mfp/mfp/data/__init__.py)
