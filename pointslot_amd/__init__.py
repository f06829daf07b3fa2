"""pointslot_amd — MI355X-native hot path (ORB front-end, Hamming matching, pose optimisation, object
bundle adjustment) behind the C-ABI of include/pointslot_hip.h.  No CPU fallback: importing the
operator modules requires libpointslot_hip.so."""
__version__ = "0.1.0"
