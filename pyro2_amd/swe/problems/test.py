"""uniform fluid at rest for unit tests (reference: swe/problems/test.py)"""
DEFAULT_INPUTS = None
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    del rp
    my_data.get_var("height")[:, :] = 1.0
    my_data.get_var("x-momentum")[:, :] = 0.0
    my_data.get_var("y-momentum")[:, :] = 0.0


def finalize():
    pass
