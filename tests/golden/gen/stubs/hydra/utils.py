import importlib


def instantiate(node, *args, **kwargs):
    target = node["_target_"] if isinstance(node, dict) else node._target_
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(*args, **kwargs)
