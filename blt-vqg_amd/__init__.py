"""MI355X-native implementation of the BLT-VQG training hot path."""
__version__ = "0.1.0"
