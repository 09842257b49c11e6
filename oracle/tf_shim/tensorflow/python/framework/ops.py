"""`tensorflow.python.framework.ops` of the numpy stand-in (oracle/tf_shim/tensorflow/__init__.py): name scopes only."""
import contextlib


@contextlib.contextmanager
def name_scope(name, default_name=None, values=None):
    yield name or default_name


def RegisterGradient(op_type):   # noqa: N802  (TensorFlow's name)
    """Registering a gradient function is a no-op here: nothing differentiates through the stand-in."""
    def decorator(fn):
        return fn
    return decorator
