"""`import dirt` -- the reference's package name (dirt/__init__.py:1) on the MI355X-native implementation.

The reference's samples/ and tests/ say `import dirt`, `dirt.rasterise(...)`, `import dirt.rasterise_ops`,
`dirt.matrices`, `dirt.lighting`, `dirt.projection`; this package re-exports `dirt_amd` under those names so
such scripts run unchanged apart from TensorFlow -> torch tensors.  It contains no code of its own."""
import sys as _sys

import dirt_amd as _impl
from dirt_amd.rasterise_ops import rasterise, rasterise_batch, rasterise_deferred, rasterise_batch_deferred  # noqa: F401

for _name in ('rasterise_ops', 'matrices', 'lighting', 'projection', 'texture'):
    _module = getattr(_impl, _name)
    globals()[_name] = _module
    _sys.modules[__name__ + '.' + _name] = _module   # `import dirt.lighting` finds the same module object
del _name, _module
