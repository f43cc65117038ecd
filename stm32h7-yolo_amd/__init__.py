"""MI355X-native int8 yoloface engine behind the reference's X-CUBE-AI `ai_network_*` C-ABI.

The directory name carries a hyphen (it mirrors the reference repository's name), so import it with
    import importlib; yf = importlib.import_module("stm32h7-yolo_amd")
"""
from .binding import (Network, NetworkError, build, load, LIB_PATH, DET_DTYPE, YF_DECODE_PY, YF_DECODE_FW, YF_DECODE_FW_HOST,
                      YF_ROUND_TFLITE_REF, YF_ROUND_TIES_UP, YF_ROUND_TIES_UP_ALL, YF_ROUND_SINGLE, YF_ROUND_GENERIC_KERNELS,  # noqa: F401
                      format_uart,
                      IN_BYTES, OUT_BYTES)
