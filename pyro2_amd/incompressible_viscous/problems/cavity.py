"""Lid-driven cavity (pyro/incompressible_viscous/problems/cavity.py:1-50): unit
square, fluid at rest, the upper wall moves to the right with unit velocity
("moving_lid" boundary), the other walls are no-slip.  Re = 1 / viscosity."""
from ...util import msg

DEFAULT_INPUTS = "inputs.cavity"

PROBLEM_PARAMS = {}


def init_data(my_data, rp):
    if rp.get_param("driver.verbose"):
        msg.bold("initializing the lid-driven cavity problem...")
    myg = my_data.grid
    if myg.xmin != 0 or myg.xmax != 1 or myg.ymin != 0 or myg.ymax != 1:
        msg.fail("ERROR: domain should be a unit square")
    my_data.get_var("x-velocity")[:, :] = 0
    my_data.get_var("y-velocity")[:, :] = 0


def finalize():
    """nothing to report"""
