"""dirt_amd -- MI355X-native drop-in for the hot path of pmh47/dirt:
`rasterise`, `rasterise_batch`, `rasterise_deferred`, `rasterise_batch_deferred` (dirt/__init__.py:2)."""
from .rasterise_ops import rasterise, rasterise_batch, rasterise_deferred, rasterise_batch_deferred  # noqa: F401
from . import rasterise_ops  # noqa: F401
from . import matrices, lighting, projection  # noqa: F401  (dirt.matrices, dirt.lighting, dirt.projection)
from . import texture  # noqa: F401  (the texture helpers of samples/textured.py)
from .graphed import GraphedStep, backward  # noqa: F401  (a training step captured once as a HIP graph: the remedy for eager autograd's host cost)
