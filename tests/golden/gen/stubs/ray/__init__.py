"""Import shim (ours): actors become plain classes, remote objects plain values."""


def remote(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]

    def deco(obj):
        return obj
    return deco


def put(x):
    return x


def get(x):
    return x
