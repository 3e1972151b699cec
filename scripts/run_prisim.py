#!/usr/bin/env python3
"""Entry point with the reference's name and flag: run_prisim.py -i parms.yaml (README.rst:93-99).
Multi-GPU: scripts/run_prisim.py -n N -i parms.yaml (starts its own N ranks, one per GPU -- no torch, no MPI;
`python -m prisim_amd.launch -n N scripts/run_prisim.py -i parms.yaml` is the same thing)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisim_amd import driver   # noqa: E402

if __name__ == '__main__':
    sys.exit(driver.main())
