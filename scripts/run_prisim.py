#!/usr/bin/env python3
"""Entry point with the reference's name and flag: run_prisim.py -i parms.yaml (README.rst:93-99).
Multi-GPU: python -m torch.distributed.run --nproc-per-node N scripts/run_prisim.py -i parms.yaml"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisim_amd import driver   # noqa: E402

if __name__ == '__main__':
    sys.exit(driver.main())
