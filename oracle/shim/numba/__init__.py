"""Identity stand-in for numba (not importable in this image).

TEST INFRASTRUCTURE ONLY.  The reference's @njit functions are plain
Python/NumPy source and numba is not given fastmath, so an identity decorator
preserves their semantics (loops just run at interpreter speed).
Used only by oracle/gen_golden.py when it imports /root/reference.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


jit = njit
