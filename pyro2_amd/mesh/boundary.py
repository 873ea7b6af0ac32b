"""Boundary-condition bookkeeping, API of pyro/mesh/boundary.py:10-211.

BC objects only *describe* the boundaries; the ghost cells are filled on the
device by pyrohip_fill_bc (csrc/ctx.hip).  User-defined types registered with
define_bc keep working: their Python callbacks run on the host copy.
"""
from ..util import msg

# which boundary types are solid walls (no flux), boundary.py:10-17
bc_solid = {"outflow": False, "periodic": False, "reflect": True,
            "reflect-even": True, "reflect-odd": True, "dirichlet": True,
            "neumann": False}

# user supplied boundary routines {name: function}
ext_bcs = {}


# user boundary types that the device ghost fill implements itself
# {name: PYROHIP_BC_* code}; their Python callbacks are then not needed for the
# 4-variable compressible state
device_bcs = {}
# constant-value types (PYROHIP_BC_CONST): {name: function(variable) -> value}
const_bcs = {}


def define_bc(bc_type, function, is_solid=False, device_code=None, const_value=None):
    """register a solver-specific boundary type (boundary.py:19-32).
    device_code: the type has a kernel in csrc/ctx.hip (pyrohip.h BC codes);
    const_value(variable): ghost value of a constant-value type"""
    bc_solid[bc_type] = is_solid
    ext_bcs[bc_type] = function
    if device_code is not None:
        device_bcs[bc_type] = device_code
    if const_value is not None:
        const_bcs[bc_type] = const_value


class BCProp:
    """one property per boundary: xl, xr, yl, yr"""

    def __init__(self, xl_prop, xr_prop, yl_prop, yr_prop):
        self.xl, self.xr, self.yl, self.yr = xl_prop, xr_prop, yl_prop, yr_prop


def bc_is_solid(bc):
    return BCProp(*(int(bc_solid[b]) for b in (bc.xlb, bc.xrb, bc.ylb, bc.yrb)))


class BC:
    """boundary types of ONE variable on the four domain edges.

    'reflect' resolves to 'reflect-odd' along odd_reflect_dir and to
    'reflect-even' otherwise; inhomogeneous Dirichlet/Neumann data given as
    functions of the edge coordinate are evaluated once on `grid`
    (boundary.py:64-211)."""

    def __init__(self, *, xlb="outflow", xrb="outflow", ylb="outflow", yrb="outflow",
                 xl_func=None, xr_func=None, yl_func=None, yr_func=None, grid=None,
                 odd_reflect_dir=""):
        def resolve(name, value, direction):
            if value not in bc_solid:
                msg.fail(f"ERROR: {name} = {value} invalid BC")
            if value == "reflect":
                return "reflect-odd" if odd_reflect_dir == direction else "reflect-even"
            return value

        self.xlb = resolve("xlb", xlb, "x")
        self.xrb = resolve("xrb", xrb, "x")
        self.ylb = resolve("ylb", ylb, "y")
        self.yrb = resolve("yrb", yrb, "y")
        if (xlb == "periodic") != (xrb == "periodic"):
            msg.fail("ERROR: both xlb and xrb must be periodic")
        if (ylb == "periodic") != (yrb == "periodic"):
            msg.fail("ERROR: both ylb and yrb must be periodic")
        self.xl_value = xl_func(grid.y) if xl_func is not None else None
        self.xr_value = xr_func(grid.y) if xr_func is not None else None
        self.yl_value = yl_func(grid.x) if yl_func is not None else None
        self.yr_value = yr_func(grid.x) if yr_func is not None else None

    def sides(self):
        return (self.xlb, self.xrb, self.ylb, self.yrb)

    def values(self):
        return (self.xl_value, self.xr_value, self.yl_value, self.yr_value)

    def __str__(self):
        return f"BCs: -x: {self.xlb}  +x: {self.xrb}  -y: {self.ylb}  +y: {self.yrb}"
