"""km_amd — MI355X-native `km find_mutation` hot path (HIP kernels behind a C-ABI).

The HIP library is loaded lazily by :mod:`km_amd.lib`; importing the package
does not touch the GPU.
"""

__version__ = "0.1.0"
