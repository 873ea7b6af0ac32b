"""Grid2d / CellCenterData2d / ArrayIndexer / BC with pyro.mesh's call
surface; cell data lives on the GPU (pyro2_amd.device)."""
