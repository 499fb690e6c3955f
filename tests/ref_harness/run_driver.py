"""python run_driver.py <path to one of the reference's drivers> [its arguments...]

Runs the driver UNMODIFIED (runpy, from where it lies) in the import environment of pnode_amd_ref_plugin.py:
`petsc4py` -> compat shim, `pnode` -> shim package, device entry points -> CPU stand-in (this container has no GPU).
Seeds torch/numpy first so that the run is reproducible (the drivers draw their initial weights unseeded)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pnode_amd_ref_plugin  # noqa: F401,E402  (patches the import environment)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
script = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))      # as `python <script>` does: the driver's own directory first
sys.argv = [os.path.basename(script)] + sys.argv[2:]
runpy.run_path(script, run_name="__main__")
