"""phylo_amd: MI355X-native Felsenstein-pruning likelihood + CSMC particle loop.

Python host code over a ctypes C ABI (include/phylo_hip.h -> phylo_amd/csrc/libphylo_hip.so);
see DESIGN.md.  No PyTorch anywhere in the package.
"""
__version__ = "0.1.0"
