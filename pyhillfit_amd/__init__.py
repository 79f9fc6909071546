"""pyhillfit_amd — MI355X-native Metropolis-Hastings engine for PyHillFit's sampling step.

Host side in Python on PyTorch-ROCm tensors (device memory, streams, torch.distributed only);
the sampler is hand-written HIP behind the C ABI of include/pyhillfit_amd.h."""
__version__ = "0.1.0"
