"""pyhillfit_amd — MI355X-native Metropolis-Hastings engine for PyHillFit's sampling step.

Host side in Python on PyTorch-ROCm tensors (device memory, streams, torch.distributed only);
the sampler is hand-written HIP behind the C ABI of include/pyhillfit_amd.h."""
__version__ = "0.1.0"

import os as _os

# The hierarchical sampler runs one launch per group of pairs with equal numbers of experiments, each on its own HIP
# stream (Crumb: four groups).  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4,
# one of them taken by the null stream): a fifth stream shares a queue and its kernel waits for the other one
# (measured: 36 us per iteration for the four concurrent groups instead of 26).  Read when the HIP runtime initialises,
# i.e. at the first GPU call of the process; a value set by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
