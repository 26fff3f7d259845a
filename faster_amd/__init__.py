"""faster_amd — MI355X-native batched trajectory-optimisation core for mit-acl/faster.

One hot path only: SolverGurobi::genNewTraj()/callOptimizer()/fillX()
(/root/reference/faster/src/solverGurobi.cpp:426-477, :549-657, :122-168) rebuilt as hand-written
HIP kernels for gfx950 behind the C ABI of include/fasterhip.h.

  faster_amd.abi      numpy/ctypes mirror of the C structs
  faster_amd.capi     ctypes binding of libfasterhip.so (fails loudly if the library is missing)
  faster_amd.corridor synthetic corridor/problem generator
  faster_amd/csrc     HIP kernels + C ABI implementation
  faster_amd/host     C++ `SolverHip` class with the SolverGurobi surface
"""
__version__ = "0.1.0"
