"""soilmachine_amd -- MI355X-native particle-transport erosion hot path of SoilMachine.

Host-side Python mirror of the reference's interface for this path; the compute lives in the
hand-written HIP library ``soilmachine_amd/csrc`` behind the C-ABI of ``include/soilmx.h``.
"""
__version__ = "0.1.0"
