"""`trajopt` - the reference's import path (/root/reference/trajopt/__init__.py), served by trajopt_amd.

Code written against hanyas/trajopt keeps its imports for the path this repository rebuilds:

    from trajopt.ilqr import iLQR                      # /root/reference/trajopt/ilqr/__init__.py
    from trajopt.ilqr.core import backward_pass        # /root/reference/trajopt/ilqr/ilqr.py:7 (the pybind11 module)
    from trajopt.riccati import Riccati                # /root/reference/trajopt/riccati/__init__.py
    from trajopt.gps.core import backward_pass, ...    # /root/reference/trajopt/gps/mbgps.py:12-16
    from trajopt.bspilqr.core import backward_pass     # /root/reference/trajopt/bspilqr/bspilqr.py:11

Every name resolves to the MI355X implementation in `trajopt_amd` (HIP kernels behind the C ABI of
include/trajopt_amd.h; there is no CPU fallback).  The reference's gym registration (its trajopt/__init__.py) is not
reproduced - gym is not a dependency here; `trajopt.envs.make(name)` builds the same env parameter sets.  Solvers
outside SURVEY section 8 (MBGPS / MFGPS / rGPS / eLQR / bspiLQR drivers) are not part of this package.
"""
