"""MI355X-native backend for the per-frame core of a stereo visual-odometry loop."""
__version__ = "0.1.0"
