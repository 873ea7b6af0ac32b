"""uniform field for unit tests"""
DEFAULT_INPUTS = None
PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    del rp
    my_data.get_var("phi")[:, :] = 1.0


def finalize():
    pass
