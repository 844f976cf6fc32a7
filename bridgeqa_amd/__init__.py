"""bridgeqa_amd -- MI355X-native implementation of BridgeQA's data-parallel hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); the
operators are hand-written HIP kernels for gfx950 behind the C ABI of include/bqhip.h
(bridgeqa_amd/lib/libbqhip.so).  Module and operator names mirror the reference
(matthewdm0816/BridgeQA) so the path drops into its scripts/train.py -- see INTEGRATION.md.
"""
__version__ = "0.1.0"
