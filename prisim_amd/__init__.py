"""prisim_amd -- MI355X-native drop-in for the PRISim per-baseline sky-sum
(InterferometerArray.observe, prisim/interferometry.py:5874-6410 of nithyanandan/PRISim).

Host code is Python (numpy + ctypes); all arithmetic of the hot path runs in hand-written HIP
kernels behind the C-ABI of include/prisim_hip.h.  There is no CPU fallback.
"""
__version__ = '0.1.0'
