"""MI355X-native SuperPoint inference path (host side).

Sub-modules: `arch` (checkpoint table), `synth` (seeded inputs), `_lib` (ctypes
binding of the C-ABI in include/fpc.h), `inference` (mirror of the reference's
python/src/inferencewrapper.py / netutils.py interface), `dist` (frame-batch
sharding over one process per GPU).
"""
__all__ = ["arch", "synth"]
