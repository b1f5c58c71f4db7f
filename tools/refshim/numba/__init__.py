"""Stand-in for numba: njit is the identity (the six jitted reference functions are plain NumPy)."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Sig) and not kwargs:
        return args[0]
    return lambda f: f


class _Sig:
    def __getitem__(self, item):
        return self

    def __call__(self, *a, **k):
        return self


float64 = _Sig()
