"""ctypes signatures of the backbone (MoE ConvNeXt) entry points of libsm3det_hip.so."""
import ctypes


def signatures():
    return {}
