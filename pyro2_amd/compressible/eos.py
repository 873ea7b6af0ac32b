"""gamma-law gas, p = rho e (gamma - 1)  (pyro/compressible/eos.py)"""


def pres(gamma, rho, eint):
    return rho * eint * (gamma - 1.0)


def dens(gamma, p, eint):
    return p / (eint * (gamma - 1.0))


def rhoe(gamma, p):
    return p / (gamma - 1.0)
