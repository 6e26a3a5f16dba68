"""midastouch_amd - MI355X-native particle-filter hot path of MidasTouch (gfx950 HIP kernels behind a C ABI)."""
__version__ = "0.1.0"
