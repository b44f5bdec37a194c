"""sm3det_amd -- MI355X-native hot path of SM3Det (grid-level sparse-MoE ConvNeXt backbone + rotated-detection
operators) behind the reference's own plug-in surface.  See DESIGN.md / INTEGRATION.md."""
__version__ = '0.1.0'
