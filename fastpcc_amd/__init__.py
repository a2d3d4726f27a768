"""fastpcc_amd -- MI355X-native encode/decode hot path of FastPCC.

The package holds the HIP/C++ sources (csrc/), their ctypes loaders and the host-side mirror of the reference's operator
interface for this path.  Device work goes through libfpcc_hip.so; there is no CPU fallback: calling a device op without
the library or without a GPU raises.
"""
__version__ = '0.1.0'
