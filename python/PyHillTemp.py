#!/usr/bin/env python3
"""Same place and name as the reference script: `python PyHillTemp.py --data-file ... -m 2 -d 0 -c 0` (see pyhillfit_amd/PyHillTemp.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyhillfit_amd.PyHillTemp import main  # noqa: E402

if __name__ == "__main__":
    main()
