"""A problem source with all four components, state and position dependent: the
`source_terms(myg, U, ivars, rp)` callback that oracle/gen_golden.py hands to the
REFERENCE's compressible solver (fixture comp_general_source) and the tests hand to
this package (host-evaluated source path).  Only + - * /: no libm in the loop."""


def source_terms(myg, U, ivars, rp):
    S = myg.scratch_array(nvar=ivars.nvar)
    d = U[:, :, ivars.idens]
    mx = U[:, :, ivars.ixmom]
    my = U[:, :, ivars.iymom]
    E = U[:, :, ivars.iener]
    k = 0.7
    S[:, :, ivars.idens] = 0.05 * d * myg.x2d                 # mass source
    S[:, :, ivars.ixmom] = 0.3 * d * myg.y2d - k * mx         # body force + drag
    S[:, :, ivars.iymom] = -k * my
    S[:, :, ivars.iener] = 0.2 * d + 0.05 * E * myg.x2d + 0.3 * mx * myg.y2d - k * (mx * mx + my * my) / d
    return S
