"""The "moving_lid" boundary (pyro/incompressible_viscous/BC.py:9-50): fixed
tangential velocity 1 and normal velocity 0 in the ghost cells above the upper
y boundary.  The device ghost fill implements it (PYROHIP_BC_CONST,
csrc/ctx.hip k_fill_y); `user` is the host callback with the reference's
signature for data objects that are not device states."""
from ..util import msg


def lid_value(variable):
    """ghost-cell value of a variable at the lid (BC.py:33-42)"""
    if variable in ("x-velocity", "u"):
        return 1.0   # unit velocity
    if variable in ("y-velocity", "v"):
        return 0.0
    raise NotImplementedError("variable not defined")


def user(bc_name, bc_edge, variable, ccdata, ivars=None):
    myg = ccdata.grid
    if bc_name != "moving_lid":
        msg.fail(f"error: bc type {bc_name} not supported")
    if bc_edge != "yrb":
        msg.fail("error: moving_lid BC only implemented for 'yrb' (top boundary)")
    v = ccdata.get_var(variable)
    v[:, myg.jhi + 1:myg.jhi + myg.ng + 1] = lid_value(variable)
