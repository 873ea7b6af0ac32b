"""Cell-centred multigrid for (alpha - beta L) phi = f; `MG.CellCenterMG2d`
has the surface of pyro.multigrid.MG.CellCenterMG2d, the V-cycle runs in
csrc/multigrid.hip."""
