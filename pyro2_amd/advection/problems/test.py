"""uniform state used by unit tests (reference: advection/problems/test.py)"""
DEFAULT_INPUTS = None
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    del rp
    my_data.get_var("density")[:, :] = 1.0


def finalize():
    pass
