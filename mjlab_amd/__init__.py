"""mjlab_amd: MI355X-native batched MuJoCo-style physics step behind mjlab's Simulation API."""

__version__ = "0.1.0"
