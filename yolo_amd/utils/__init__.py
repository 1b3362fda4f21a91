"""Mirror of the reference's `utils` package, inference-path subset."""
from . import torch_utils  # noqa: F401
