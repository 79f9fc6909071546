#!/usr/bin/env python3
"""Same place and name as the reference script (see pyhillfit_amd/construct_hierarchical_cdfs.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhillfit_amd.construct_hierarchical_cdfs import main  # noqa: E402

if __name__ == "__main__":
    main()
