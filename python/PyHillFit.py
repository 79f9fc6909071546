#!/usr/bin/env python3
"""Same place and name as the reference script: `python PyHillFit.py --data-file ... -m 2 -a` (see pyhillfit_amd/PyHillFit.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhillfit_amd.PyHillFit import main  # noqa: E402

if __name__ == "__main__":
    main()
