"""libldpc_amd — MI355X-native LDPC belief-propagation decoder (drop-in for heat1q/libldpc's hot path).

The product is the C-ABI shared library libldpc_amd/libldpc.so (hand-written HIP kernels + C++17
host runtime, see include/ldpc_amd.h).  This package is the thin host-side mirror:

  * ``libldpc_amd.LDPC``       same class interface as the reference's pyLDPC.ldpc.LDPC, over our .so
  * ``libldpc_amd.HipDecoder`` batch interface (decode thousands of frames per launch; buffers may be
                               numpy arrays or torch CUDA tensors)

There is no CPU fallback: without the built library, or without a GPU, calls fail loudly.
"""
from .binding import LIB_PATH, Comm, HipDecoder, load_library  # noqa: F401
from .ldpc import LDPC  # noqa: F401
