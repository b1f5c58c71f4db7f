def hashkey(*args, **kwargs):
    return args + tuple(sorted(kwargs.items()))
