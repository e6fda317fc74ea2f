"""hamilton_amd: MI355X-native equations-of-motion path of mstksg/hamilton.

Host-side mirror of `Numeric.Hamilton` (see hamilton_amd.api) over the C ABI
of libhamk.so (include/hamk.h); the HIP device library lives in csrc/.
"""
