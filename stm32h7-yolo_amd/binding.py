"""ctypes binding of libyf_network.so (the C-ABI declared in include/yf_network.h).

Host-side mirror of the reference's call sequence (stm32/X-CUBE-AI/App/yoloface.c:188-240):
    aiInit : ai_network_create -> ai_network_init(AI_NETWORK_PARAMS_INIT(weights, activations))
    aiRun  : ai_network_run(network, &ai_input, &ai_output)
The library has no CPU compute path: without a gfx950 GPU `init()` raises.
"""
import ctypes
import os
import sys
import subprocess
import weakref

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libyf_network.so")
if os.environ.get("YF_LIB_PATH"):        # developer override: A/B of differently compiled libraries (tools/)
    LIB_PATH = os.environ["YF_LIB_PATH"]

AI_BUFFER_FORMAT_U8 = 0x00040440
AI_BUFFER_FORMAT_S8 = 0x00840440
AI_BUFFER_FMT_FLAG_CONST = 1 << 30
AI_MAGIC_MARKER = 0xA1FACADE
IN_H, IN_W, IN_C = 56, 56, 3
OUT_H, OUT_W, OUT_C = 7, 7, 18
IN_BYTES, OUT_BYTES = IN_H * IN_W * IN_C, OUT_H * OUT_W * OUT_C
WEIGHTS_BYTES, ACTIVATIONS_BYTES = 11304, 29784
YF_DECODE_PY, YF_DECODE_FW, YF_DECODE_FW_HOST = 0, 1, 2
# rounding of the requantisation step (include/yf_network.h, yf_network_set_requant_rounding)
YF_ROUND_TFLITE_REF, YF_ROUND_TIES_UP, YF_ROUND_TIES_UP_ALL, YF_ROUND_SINGLE = 0, 1, 2, 3
YF_ROUND_GENERIC_KERNELS = 0x100      # or-ed into a rounding: keep the reference rounding's four-instruction kernels (A/B against the sign-free kernel set)


class AiError(ctypes.Structure):
    _fields_ = [("type", ctypes.c_uint32, 8), ("code", ctypes.c_uint32, 24)]


class AiBuffer(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int32), ("n_batches", ctypes.c_uint16), ("height", ctypes.c_uint16),
                ("width", ctypes.c_uint16), ("channels", ctypes.c_uint32), ("data", ctypes.c_void_p),
                ("meta_info", ctypes.c_void_p)]


class AiBufferArray(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_uint16), ("size", ctypes.c_uint16), ("buffer", ctypes.POINTER(AiBuffer))]


class _ParamsLegacy(ctypes.Structure):
    _fields_ = [("params", AiBuffer), ("activations", AiBuffer)]


class _ParamsMap(ctypes.Structure):
    _fields_ = [("map_signature", ctypes.c_uint32), ("map_weights", AiBufferArray), ("map_activations", AiBufferArray)]


class AiNetworkParams(ctypes.Union):
    _anonymous_ = ("legacy", "map")
    _fields_ = [("legacy", _ParamsLegacy), ("map", _ParamsMap)]


class AiPlatformVersion(ctypes.Structure):
    _fields_ = [("major", ctypes.c_uint8), ("minor", ctypes.c_uint8), ("micro", ctypes.c_uint8), ("reserved", ctypes.c_uint8)]


class AiNetworkReport(ctypes.Structure):
    _anonymous_ = ("p",)
    _fields_ = [("model_name", ctypes.c_char_p), ("model_signature", ctypes.c_char_p), ("model_datetime", ctypes.c_char_p),
                ("compile_datetime", ctypes.c_char_p), ("runtime_revision", ctypes.c_char_p),
                ("runtime_version", AiPlatformVersion), ("tool_revision", ctypes.c_char_p),
                ("tool_version", AiPlatformVersion), ("tool_api_version", AiPlatformVersion),
                ("api_version", AiPlatformVersion), ("interface_api_version", AiPlatformVersion),
                ("n_macc", ctypes.c_uint32), ("n_inputs", ctypes.c_uint16), ("n_outputs", ctypes.c_uint16),
                ("inputs", ctypes.POINTER(AiBuffer)), ("outputs", ctypes.POINTER(AiBuffer)),
                ("p", AiNetworkParams), ("n_nodes", ctypes.c_uint32), ("signature", ctypes.c_uint32)]


class YfScratchStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_ulonglong) for n in ("events_recorded", "events_skipped", "event_waits", "device_syncs", "acquire_waits", "regions")]


class YfDet(ctypes.Structure):
    _fields_ = [("frame", ctypes.c_int32), ("anchor", ctypes.c_uint8), ("row", ctypes.c_uint8), ("col", ctypes.c_uint8),
                ("q_conf", ctypes.c_int8), ("conf", ctypes.c_float), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32),
                ("x2", ctypes.c_int32), ("y2", ctypes.c_int32)]


DET_DTYPE = np.dtype([("frame", "<i4"), ("anchor", "u1"), ("row", "u1"), ("col", "u1"), ("q_conf", "i1"),
                      ("conf", "<f4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4")])
assert DET_DTYPE.itemsize == ctypes.sizeof(YfDet) == 28

EXPORTS = ["ai_network_create", "ai_network_init", "ai_network_run", "ai_network_forward", "ai_network_get_error",
           "ai_network_destroy", "ai_network_get_info", "ai_network_get_report", "ai_network_data_weights_get",
           "ai_network_data_params_get", "ai_platform_bind_network_params", "yf_network_set_device",
           "yf_network_configure", "yf_network_run_device", "yf_network_run_device_dump", "yf_network_dump_bytes", "yf_network_run_device_hw",
           "yf_network_decode_device", "yf_network_run_decode_device", "yf_network_pack_detections_device", "yf_network_unpack_detections_device", "yf_network_prepare_rgb565_device", "yf_network_run_camera_device", "yf_network_time_device",
           "yf_network_time_stages", "yf_network_format_uart", "yf_network_shard_range", "yf_network_table_plan", "yf_network_all_gather_device", "yf_network_fp16_init", "yf_network_fp16_run_device", "yf_network_release_stream", "yf_network_scratch_bytes", "yf_network_scratch_stats", "yf_network_set_requant_rounding", "yf_network_get_requant_rounding", "yf_network_last_error_text",
           "yf_network_kernel_name", "yf_network_kernel_name_for", "yf_network_build_id", "yf_network_host_id",
           "ai_platform_observer_node_info", "ai_platform_observer_register", "ai_platform_observer_register_s",
           "ai_platform_observer_unregister", "ai_platform_observer_unregister_s",
           # runtime-level boundary (csrc/platform_abi.c): what the reference's generated network.c references
           "ai_platform_context_acquire", "ai_platform_network_create", "ai_platform_network_destroy",
           "ai_platform_network_get_error", "ai_platform_network_init", "ai_platform_network_post_init",
           "ai_platform_network_process", "ai_platform_get_weights_map", "ai_platform_get_activations_map",
           "ai_platform_api_get_network_report", "ai_platform_runtime_get_revision", "ai_platform_runtime_get_version",
           "ai_platform_api_get_version", "ai_platform_interface_api_get_version", "forward_conv2d_integer_SSSA_ch",
           "forward_mp_integer_INT8", "forward_eltwise_integer_INT8", "forward_concat", "nl_func_array_integer",
           "ai_sum_f32", "ai_sum_buffer_INT8"]


def expected_build_id(extra_hipflags="", extra_fp16flags=""):
    """The id csrc/Makefile bakes into the library (yf_network_build_id): sha256 over the device sources, flags.mk and the extra flags."""
    import hashlib
    csrc = os.path.join(_PKG, "csrc")
    flags = open(os.path.join(csrc, "flags.mk"), "rb").read()
    srcs = [ln.split(b"=", 1)[1].split() for ln in flags.splitlines() if ln.startswith(b"DEVICE_SRCS")][0]
    h = hashlib.sha256()
    for f in srcs:
        h.update(open(os.path.join(csrc, f.decode()), "rb").read())
    h.update(flags)
    h.update(f"{extra_hipflags}|{extra_fp16flags}\n".encode())
    return h.hexdigest()[:16]


def expected_host_id():
    """The id of the C host layer csrc/Makefile bakes into the library (yf_network_host_id): sha256 over HOST_SRCS and the C flags."""
    import hashlib
    import re
    csrc = os.path.join(_PKG, "csrc")
    flags = open(os.path.join(csrc, "flags.mk")).read()
    srcs = re.search(r"^HOST_SRCS\s*=\s*(.*)$", flags, re.M).group(1).split()
    cflags = re.search(r"^CFLAGS\s*=\s*(.*)$", open(os.path.join(csrc, "Makefile")).read(), re.M).group(1).strip()
    h = hashlib.sha256()
    for f in srcs:
        h.update(open(os.path.join(csrc, f), "rb").read())
    h.update((cflags + "\n").encode())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile the library in-tree for gfx950 (hipcc cross-compiles without a GPU).  One build at a time: the processes that share a checkout
    (the two ranks of bench.py's self-launch, parallel test workers, profiler-wrapped tools) serialise on a lock file beside the Makefile -- not
    in the output directory, which `make clean` empties while a forced build holds the lock."""
    import fcntl
    out = os.path.join(_PKG, "lib")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(_PKG, "csrc", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if force:
            subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "clean"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(_PKG, "csrc"), "-j4", "all", "../lib/libyf_hostprep.so"],
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


def library_is_current():
    """True when the in-tree library can be loaded WITHOUT running make: it exists, the ids make stamped beside it (lib/build_id.stamp,
    lib/host_id.stamp) are the ids of the sources as they stand, and it is newer than every one of them.  load() then starts no child process --
    under rocprofv3 every child of a GPU-holding process is instrumented by the profiler's preloaded tool (round 5: make, sh, cut and sha256sum
    in the middle of a counter pass).  The ids baked into the library are still checked after it is loaded."""
    import re
    csrc, lib = os.path.join(_PKG, "csrc"), os.path.join(_PKG, "lib")
    try:
        if open(os.path.join(lib, "build_id.stamp")).read().strip() != expected_build_id():
            return False
        if open(os.path.join(lib, "host_id.stamp")).read().strip() != expected_host_id():
            return False
        built = os.path.getmtime(LIB_PATH)
        flags = open(os.path.join(csrc, "flags.mk")).read()
        srcs = ["Makefile", "flags.mk"]
        for var in ("DEVICE_SRCS", "HOST_SRCS"):
            srcs += re.search(r"^%s\s*=\s*(.*)$" % var, flags, re.M).group(1).split()
        return all(os.path.getmtime(os.path.join(csrc, f)) <= built for f in srcs)
    except (OSError, AttributeError):
        return False


_lib = None


def _one_hip_runtime():
    """A process can drive the GPU through ONE HIP/HSA runtime only.  PyTorch-ROCm ships its own copy (torch/lib/libamdhip64.so) and
    loads it by path; libyf_network.so asks for `libamdhip64.so.7` by name.  If PyTorch comes first, that name resolves to PyTorch's copy
    and all is well; if this library came first, /opt/rocm's copy would be loaded, PyTorch would add its own, and whichever initialises
    second finds no device ("no ROCm-capable device is detected" / "No HIP GPUs are available").  So when PyTorch is installed and not
    yet imported, its runtime is pre-loaded here; a C caller, or a Python process without PyTorch, uses /opt/rocm's."""
    import importlib.util
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen the library (building it first if the .so is missing) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    check_id = False
    if os.environ.get("YF_LIB_PATH"):
        pass
    elif os.environ.get("YF_NO_BUILD") == "1" or library_is_current():
        # no child process at all (profiler runs set YF_NO_BUILD=1: tools/profile_*.sh); a library built from other sources is refused below
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"YF_NO_BUILD=1 and {LIB_PATH} does not exist: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        check_id = True
    else:
        try:
            build()              # `make`: a stale .so is never loaded under fresh sources
        except (OSError, subprocess.CalledProcessError) as e:
            # no make / no hipcc on this box (or the build failed): an existing library is used if -- and only if -- it was built from these
            # sources with these flags (its baked-in id against the id computed here)
            if not os.path.exists(LIB_PATH):
                raise
            import warnings
            warnings.warn(f"stm32h7-yolo_amd: could not run the build ({e}); loading the existing library after checking its device and host build ids")
            check_id = True
    _one_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    if check_id:
        lib.yf_network_build_id.restype = ctypes.c_char_p
        have, want = (lib.yf_network_build_id() or b"").decode(), expected_build_id()
        if have != want:
            raise RuntimeError(f"{LIB_PATH} was built from other sources or flags (build id {have}, expected {want}) and is not being rebuilt here (no make / YF_NO_BUILD=1)")
        lib.yf_network_host_id.restype = ctypes.c_char_p
        have, want = (lib.yf_network_host_id() or b"").decode(), expected_host_id()
        if have != want:
            raise RuntimeError(f"{LIB_PATH}: its C host layer was built from other sources (host id {have}, expected {want}) and is not being rebuilt here (no make / YF_NO_BUILD=1)")
    vp, cl = ctypes.c_void_p, ctypes.c_long
    lib.ai_network_create.restype = AiError
    lib.ai_network_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(AiBuffer)]
    lib.ai_network_init.restype = ctypes.c_bool
    lib.ai_network_init.argtypes = [vp, ctypes.POINTER(AiNetworkParams)]
    lib.ai_network_run.restype = ctypes.c_int32
    lib.ai_network_run.argtypes = [vp, ctypes.POINTER(AiBuffer), ctypes.POINTER(AiBuffer)]
    lib.ai_network_forward.restype = ctypes.c_int32
    lib.ai_network_forward.argtypes = [vp, ctypes.POINTER(AiBuffer)]
    lib.ai_network_get_error.restype = AiError
    lib.ai_network_get_error.argtypes = [vp]
    lib.ai_network_destroy.restype = vp
    lib.ai_network_destroy.argtypes = [vp]
    for f in (lib.ai_network_get_info, lib.ai_network_get_report):
        f.restype = ctypes.c_bool
        f.argtypes = [vp, ctypes.POINTER(AiNetworkReport)]
    lib.ai_network_data_weights_get.restype = vp
    lib.ai_network_data_weights_get.argtypes = []
    lib.ai_network_data_params_get.restype = ctypes.c_bool
    lib.ai_network_data_params_get.argtypes = [vp, ctypes.POINTER(AiNetworkParams)]
    lib.ai_platform_bind_network_params.restype = ctypes.c_bool
    lib.ai_platform_bind_network_params.argtypes = [vp, ctypes.POINTER(AiNetworkParams), ctypes.POINTER(AiBufferArray),
                                                    ctypes.POINTER(AiBufferArray)]
    lib.yf_network_set_device.argtypes = [vp, ctypes.c_int]
    lib.yf_network_configure.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.yf_network_run_device.restype = cl
    lib.yf_network_run_device.argtypes = [vp, vp, vp, cl, vp]
    lib.yf_network_run_device_dump.restype = cl
    lib.yf_network_run_device_dump.argtypes = [vp, vp, vp, vp, cl, vp]
    lib.yf_network_dump_bytes.restype = cl
    lib.yf_network_run_device_hw.restype = cl
    lib.yf_network_run_device_hw.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp, cl, vp]
    lib.yf_network_decode_device.restype = cl
    lib.yf_network_decode_device.argtypes = [vp, vp, cl, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp, vp, ctypes.c_int, vp]
    lib.yf_network_pack_detections_device.restype = cl
    lib.yf_network_pack_detections_device.argtypes = [vp, vp, vp, vp, vp, cl, ctypes.c_int, vp]
    lib.yf_network_unpack_detections_device.restype = cl
    lib.yf_network_unpack_detections_device.argtypes = [vp, vp, vp, vp, cl, ctypes.c_int, vp]
    lib.yf_network_run_decode_device.restype = cl
    lib.yf_network_run_decode_device.argtypes = [vp, vp, vp, cl, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp, vp, ctypes.c_int, vp]
    lib.yf_network_prepare_rgb565_device.restype = cl
    lib.yf_network_prepare_rgb565_device.argtypes = [vp, vp, vp, cl, vp]
    lib.yf_network_run_camera_device.restype = cl
    lib.yf_network_run_camera_device.argtypes = [vp, vp, vp, cl, ctypes.c_int, ctypes.c_float, ctypes.c_float, vp, vp, ctypes.c_int, vp]
    lib.yf_network_shard_range.restype = None
    lib.yf_network_shard_range.argtypes = [cl, ctypes.c_int, ctypes.c_int, ctypes.POINTER(cl), ctypes.POINTER(cl)]
    lib.yf_network_all_gather_device.restype = cl
    lib.yf_network_all_gather_device.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.yf_network_format_uart.restype = cl
    lib.yf_network_format_uart.argtypes = [ctypes.c_uint, vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.yf_network_time_device.restype = cl
    lib.yf_network_time_device.argtypes = [vp, vp, vp, cl, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
    lib.yf_network_time_stages.restype = cl
    lib.yf_network_time_stages.argtypes = [vp, vp, vp, cl, ctypes.c_int, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float)]
    lib.yf_network_fp16_init.argtypes = [vp, vp, ctypes.c_size_t]
    lib.yf_network_fp16_run_device.restype = cl
    lib.yf_network_fp16_run_device.argtypes = [vp, vp, vp, cl, vp]
    if hasattr(lib, "yf_network_release_stream"):          # (an older library loaded through YF_LIB_PATH for an A/B run does without)
        lib.yf_network_release_stream.restype = ctypes.c_int
        lib.yf_network_release_stream.argtypes = [vp, vp]
        lib.yf_network_scratch_bytes.restype = ctypes.c_size_t
        lib.yf_network_scratch_bytes.argtypes = [vp]
    if hasattr(lib, "yf_network_set_requant_rounding"):
        lib.yf_network_set_requant_rounding.argtypes = [vp, ctypes.c_int]
        lib.yf_network_get_requant_rounding.argtypes = [vp]
        lib.yf_network_scratch_stats.argtypes = [vp, ctypes.POINTER(YfScratchStats)]
    lib.yf_network_last_error_text.restype = ctypes.c_char_p
    lib.yf_network_last_error_text.argtypes = [vp]
    lib.yf_network_kernel_name.restype = ctypes.c_char_p
    lib.yf_network_kernel_name.argtypes = [vp]
    lib.yf_network_kernel_name_for.restype = ctypes.c_char_p
    lib.yf_network_kernel_name_for.argtypes = [vp, ctypes.c_long]
    lib.yf_network_build_id.restype = ctypes.c_char_p
    lib.yf_network_build_id.argtypes = []
    _lib = lib
    return lib


def format_uart(frame_no, dets, count=None):
    """The firmware's UART text of one frame (stm32/User/main.c:46,53; yoloface.c:148) through the library's C
    formatter.  dets: DET_DTYPE records of the frame (firmware mode), count: the frame's candidate count."""
    d = np.ascontiguousarray(dets, dtype=DET_DTYPE).reshape(-1)
    count = d.shape[0] if count is None else int(count)
    lib = load()
    need = lib.yf_network_format_uart(frame_no, d.ctypes.data if d.shape[0] else None, count, d.shape[0], None, 0)
    buf = ctypes.create_string_buffer(need + 1)
    lib.yf_network_format_uart(frame_no, d.ctypes.data if d.shape[0] else None, count, d.shape[0], buf, need + 1)
    return buf.raw[:need]


class NetworkError(RuntimeError):
    def __init__(self, what, etype, code, text=""):
        super().__init__(f"{what}: ai_error type=0x{etype:02x} code=0x{code:04x} {text}".strip())
        self.type, self.code, self.text = etype, code, text


def make_buffer(fmt, h, w, ch, n_batches=1, data=None):
    """AI_BUFFER_OBJ_INIT (reference ai_platform.h:322-330)."""
    return AiBuffer(fmt, n_batches, h, w, ch, data, None)


_live_owner = None      # weakref to the Network object that currently owns the library's single context


class Network:
    """The one network instance of the library (the reference has a static singleton too, network.c:2929).
    ai_network_create re-creates that singleton, so constructing a second Network (or an Interpreter) SUPERSEDES the first:
    the older object is marked dead and its calls raise instead of failing later with a puzzling INVALID_STATE."""

    def __init__(self, device=0, frames_per_wg=0, waves_per_wg=0):
        global _live_owner
        self.lib = load()
        self._device, self._cfg = device, (frames_per_wg, waves_per_wg)
        self._take_ownership()
        self.handle = ctypes.c_void_p()
        err = self.lib.ai_network_create(ctypes.byref(self.handle), None)          # yoloface.c:192
        if err.type != 0:
            raise NetworkError("ai_network_create", err.type, err.code)
        self.lib.yf_network_set_device(self.handle, device)
        if frames_per_wg or waves_per_wg:
            self.lib.yf_network_configure(self.handle, frames_per_wg, waves_per_wg)
        self._activations = (ctypes.c_uint8 * ACTIVATIONS_BYTES)()                 # yoloface.c:180-181
        self.ready = False

    def _take_ownership(self):
        global _live_owner
        prev = _live_owner() if _live_owner is not None else None
        if prev is not None and prev is not self:
            prev.ready, prev.superseded, prev.handle = False, True, ctypes.c_void_p()    # NULL handle: every call fails, _raise says why
        _live_owner = weakref.ref(self)
        self.superseded = False

    def reclaim(self):
        """Make THIS object the owner of the library's single instance again (ai_network_create anew); call init() next."""
        self._take_ownership()
        self.handle = ctypes.c_void_p()
        err = self.lib.ai_network_create(ctypes.byref(self.handle), None)
        if err.type != 0:
            raise NetworkError("ai_network_create", err.type, err.code)
        self.lib.yf_network_set_device(self.handle, self._device)
        if any(self._cfg):
            self.lib.yf_network_configure(self.handle, *self._cfg)
        self.ready = False
        return self

    def _raise(self, what):
        if self.superseded:
            raise NetworkError(what + ": this Network object was superseded by a newer Network/Interpreter (the library has ONE "
                               "network instance, like the reference: network.c:2929)", 0x11, 0x14)
        err = self.lib.ai_network_get_error(self.handle)
        text = (self.lib.yf_network_last_error_text(self.handle) or b"").decode()
        raise NetworkError(what, err.type, err.code, text)

    def init(self, weights=None):
        """ai_network_init with AI_NETWORK_PARAMS_INIT(AI_NETWORK_DATA_WEIGHTS(...), AI_NETWORK_DATA_ACTIVATIONS(...))
        (yoloface.c:199-204).  weights: None = the library's blob via ai_network_data_weights_get()."""
        p = AiNetworkParams()
        if weights is None:
            wptr = self.lib.ai_network_data_weights_get()
        else:
            self._w = np.ascontiguousarray(weights, dtype=np.uint8)
            wptr = self._w.ctypes.data
        p.params = make_buffer(AI_BUFFER_FORMAT_U8 | AI_BUFFER_FMT_FLAG_CONST, 1, 1, WEIGHTS_BYTES, 1, wptr)
        p.activations = make_buffer(AI_BUFFER_FORMAT_U8, 1, 1, ACTIVATIONS_BYTES, 1, ctypes.addressof(self._activations))
        if not self.lib.ai_network_init(self.handle, ctypes.byref(p)):
            self._raise("ai_network_init")
        self.ready = True
        return self

    def run(self, frames, out=None):
        """ai_network_run on host memory: int8 [n,56,56,3] -> int8 [n,7,7,18] (yoloface.c:216-240, n_batches = n).  `out`: the caller's
        own result array (the firmware passes the same static out_data every call), else a fresh one."""
        x = np.ascontiguousarray(frames, dtype=np.int8).reshape(-1, IN_H, IN_W, IN_C)
        if out is None:
            out = np.empty((x.shape[0], OUT_H, OUT_W, OUT_C), np.int8)
        elif out.dtype != np.int8 or not out.flags.c_contiguous or out.size != x.shape[0] * OUT_H * OUT_W * OUT_C:
            raise ValueError("out must be a C-contiguous int8 array of n x 7 x 7 x 18")
        done = 0
        while done < x.shape[0]:                     # n_batches is 16 bit (ai_platform.h:519)
            n = min(65535, x.shape[0] - done)
            bi = make_buffer(AI_BUFFER_FORMAT_S8, IN_H, IN_W, IN_C, n, x[done:].ctypes.data)
            bo = make_buffer(AI_BUFFER_FORMAT_S8, OUT_H, OUT_W, OUT_C, n, out[done:].ctypes.data)
            if self.lib.ai_network_run(self.handle, ctypes.byref(bi), ctypes.byref(bo)) != n:
                self._raise("ai_network_run")
            done += n
        return out

    def run_device(self, d_in, d_out, n, stream=None, d_dump=None):
        if d_dump is None:
            rc = self.lib.yf_network_run_device(self.handle, d_in, d_out, n, stream)
        else:
            rc = self.lib.yf_network_run_device_dump(self.handle, d_in, d_out, d_dump, n, stream)
        if rc != n:
            self._raise("yf_network_run_device")

    def run_device_hw(self, h, w, d_in, d_out, n, stream=None):
        if self.lib.yf_network_run_device_hw(self.handle, h, w, d_in, d_out, n, stream) != n:
            self._raise("yf_network_run_device_hw")

    def fp16_init(self, yfw_path=None):
        path = yfw_path or os.path.join(_PKG, "model", "yoloface_fp32.yfw")
        self._yfw = open(path, "rb").read()
        if self.lib.yf_network_fp16_init(self.handle, self._yfw, len(self._yfw)) != 0:
            self._raise("yf_network_fp16_init")

    def release_stream(self, stream):
        """Hand the scratch regions of a stream back before destroying it (bounded per-stream scratch, include/yf_network.h)."""
        if self.lib.yf_network_release_stream(self.handle, stream) != 0:
            self._raise("yf_network_release_stream")

    def scratch_bytes(self):
        return int(self.lib.yf_network_scratch_bytes(self.handle))

    def scratch_stats(self):
        """events recorded / skipped and the waits of the all-busy path, as a dict (include/yf_network.h, yf_scratch_stats)"""
        st = YfScratchStats()
        if self.lib.yf_network_scratch_stats(self.handle, ctypes.byref(st)) != 0:
            self._raise("yf_network_scratch_stats")
        return {n: int(getattr(st, n)) for n, _ in YfScratchStats._fields_}

    def set_requant_rounding(self, rounding):
        """Which published rounding of TFLite's requantisation the network computes (YF_ROUND_*; default: the builtin reference kernels).
        Other constants for the same kernels, or -- the roundings without a sign term, by default -- for the kernel set with the three-instruction
        dense epilogue; before or after init()."""
        if self.lib.yf_network_set_requant_rounding(self.handle, int(rounding)) != 0:
            self._raise("yf_network_set_requant_rounding")
        return self

    @property
    def requant_rounding(self):
        return int(self.lib.yf_network_get_requant_rounding(self.handle))

    def fp16_run_device(self, d_in_f16, d_out_f32, n, stream=None):
        if self.lib.yf_network_fp16_run_device(self.handle, d_in_f16, d_out_f32, n, stream) != n:
            self._raise("yf_network_fp16_run_device")

    def decode_device(self, d_heads, n, d_dets, d_counts, cap, mode=YF_DECODE_PY, w_scale=1.0, h_scale=1.0, stream=None):
        if self.lib.yf_network_decode_device(self.handle, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream) != n:
            self._raise("yf_network_decode_device")

    def pack_detections_device(self, d_dets, d_counts, d_heads, d_wire, n, cap, stream=None):
        """yf_det records -> 12-byte wire records (one launch; include/yf_network.h)."""
        if self.lib.yf_network_pack_detections_device(self.handle, d_dets, d_counts, d_heads, d_wire, n, cap, stream) != n:
            self._raise("yf_network_pack_detections_device")

    def unpack_detections_device(self, d_wire, d_counts, d_heads, n, cap, stream=None):
        """wire records + counts -> the sparse int8 heads they stand for (decode_device on them gives the sender's records)."""
        if self.lib.yf_network_unpack_detections_device(self.handle, d_wire, d_counts, d_heads, n, cap, stream) != n:
            self._raise("yf_network_unpack_detections_device")

    def run_decode_device(self, d_in, d_heads, n, d_dets, d_counts, cap, mode=YF_DECODE_PY, w_scale=1.0, h_scale=1.0, stream=None):
        """Network + box decode in one launch (heads are decoded while still in LDS)."""
        if self.lib.yf_network_run_decode_device(self.handle, d_in, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream) != n:
            self._raise("yf_network_run_decode_device")

    def run_camera_device(self, d_rgb565, d_heads, n, d_dets=None, d_counts=None, cap=0, mode=YF_DECODE_FW, w_scale=1.0, h_scale=1.0, stream=None):
        """Camera frames (112x112 big-endian RGB565) -> heads (+ detection records) in one launch."""
        if self.lib.yf_network_run_camera_device(self.handle, d_rgb565, d_heads, n, mode, w_scale, h_scale, d_dets, d_counts, cap, stream) != n:
            self._raise("yf_network_run_camera_device")

    def prepare_rgb565_device(self, d_rgb, d_out, n, stream=None):
        if self.lib.yf_network_prepare_rgb565_device(self.handle, d_rgb, d_out, n, stream) != n:
            self._raise("yf_network_prepare_rgb565_device")

    def time_device(self, d_in, d_out, n, iters, stream=None):
        ms = ctypes.c_float()
        if self.lib.yf_network_time_device(self.handle, d_in, d_out, n, iters, stream, ctypes.byref(ms)) != n:
            self._raise("yf_network_time_device")
        return ms.value

    def time_stages(self, d_in, d_out, n, iters, stop_stage, stream=None):
        ms = ctypes.c_float()
        if self.lib.yf_network_time_stages(self.handle, d_in, d_out, n, iters, stop_stage, stream, ctypes.byref(ms)) != n:
            self._raise("yf_network_time_stages")
        return ms.value

    def configure(self, frames_per_wg, waves_per_wg):
        if self.lib.yf_network_configure(self.handle, frames_per_wg, waves_per_wg) != 0:
            self._raise("yf_network_configure")

    @property
    def kernel_name(self):
        return (self.lib.yf_network_kernel_name(self.handle) or b"").decode()

    def kernel_name_for(self, n):
        """the kernel shape a batch of n frames runs (automatic choice unless configure() fixed one)"""
        return (self.lib.yf_network_kernel_name_for(self.handle, n) or b"").decode()

    @property
    def build_id(self):
        return (self.lib.yf_network_build_id() or b"").decode()

    def dump_bytes(self):
        return self.lib.yf_network_dump_bytes()

    def report(self):
        r = AiNetworkReport()
        if not self.lib.ai_network_get_report(self.handle, ctypes.byref(r)):
            self._raise("ai_network_get_report")
        return r

    def get_error(self):
        e = self.lib.ai_network_get_error(self.handle)
        return e.type, e.code

    def destroy(self):
        if self.handle:
            self.lib.ai_network_destroy(self.handle)
            self.handle = ctypes.c_void_p()
            self.ready = False
