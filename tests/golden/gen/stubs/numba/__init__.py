"""Import shim (ours): numba 0.56.4 runs the reference's @jit function in object mode, i.e. with
interpreter semantics, so an identity decorator reproduces it."""


def jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]

    def deco(fn):
        return fn
    return deco


prange = range
float64 = int32 = boolean = object
