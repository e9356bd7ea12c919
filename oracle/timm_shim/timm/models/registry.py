"""timm.models.registry.register_model -- used as a decorator at models/DeIT.py:66 ff."""
_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    return fn
