class SA:
    def __init__(self, *a, **k):
        raise NotImplementedError("sko is not installed; the env_n2n hot path does not use it")
