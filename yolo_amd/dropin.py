"""Expose this package under the reference's import paths.

The reference's callers (detect.py, test.py, filter_pruning/*) and its pickled checkpoints import
`models.yolo`, `models.common`, `models.experimental`, `utils.utils`, `utils.torch_utils`.
install() registers the yolo_amd mirrors under exactly those names in sys.modules so that those
scripts and torch.load() of whole-module checkpoints resolve to the HIP-backed implementation.
"""
import importlib
import sys

_ALIASES = {
    'models': 'yolo_amd.models',
    'models.common': 'yolo_amd.models.common',
    'models.yolo': 'yolo_amd.models.yolo',
    'models.experimental': 'yolo_amd.models.experimental',
    'utils': 'yolo_amd.utils',
    'utils.utils': 'yolo_amd.utils.utils',
    'utils.torch_utils': 'yolo_amd.utils.torch_utils',
}


def install(force=False):
    """Alias the reference module paths to this package.  Existing foreign modules of the same
    name are left alone unless force=True."""
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force and not sys.modules[alias].__name__.startswith('yolo_amd'):
            continue
        sys.modules[alias] = importlib.import_module(target)
    return sorted(_ALIASES)
