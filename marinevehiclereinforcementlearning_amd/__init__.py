"""MI355X-native vectorised marine-vehicle RL environments (BlueROV2 3/6-DoF, AuvEnv, turbulence field).

Heavy pieces (the HIP library) are loaded lazily on first use; importing the package is cheap.
"""
__version__ = "0.1.0"
