"""
pixel-nerf-yolo_amd: MI355X-native (gfx950) rendering hot path of pixelNeRF-YOLO.

Host-side mirror of the reference's Python call boundary over the C-ABI
library ``libpnyolo.so`` (include/pnyolo.h, sources in csrc/).  Sub-modules are imported
lazily so that ``synth`` (numpy only) can be used without torch or the HIP library.
"""
__version__ = "0.1.0"
