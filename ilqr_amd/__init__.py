"""MI355X-native batched control-limited iLQR (drop-in for the solve path of ZhengYi0310/ilqr)."""
__version__ = "0.1.0"
