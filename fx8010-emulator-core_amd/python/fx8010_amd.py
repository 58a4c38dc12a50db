"""ctypes binding of libfx8010_amd.so (the C ABI in include/fx8010_amd.h).

Used by tests/, bench.py and __graft_entry__.py.  It only marshals arguments: every computation
happens in the HIP kernel behind the C ABI.  There is no fallback — if the library is missing
``load()`` raises, and if no GPU is usable ``Batch(...)`` raises with the library's own message.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FX8010_AMD_LIB: another build of the same library - the sanitizer build of `make -C csrc asan`, tests/test_host_sanitizers.py)
LIB_PATH = os.environ.get("FX8010_AMD_LIB") or os.path.join(_HERE, "..", "libfx8010_amd.so")

_f32p = C.POINTER(C.c_float)
_lib = None

OPT_TRAM_DANE, OPT_TRAM_ADDR_SHIFT, OPT_TRAM_INTERP = 1, 2, 4  # FX_OPT_* of include/fx8010_amd.h

# selectors of fxb_info / fxp_lower_info
INFO = {
    "num_instructions": 0, "num_registers": 1, "num_lane_regs": 2, "num_uniform_regs": 3, "lds_bytes_per_wg": 4,
    "waves_per_wg": 5, "num_microops": 6, "itram_slots": 7, "xtram_slots": 8, "tram_ops": 9, "multipass": 10,
    "num_shadowed": 11, "num_ccr_live": 12, "device": 13, "grid": 14, "inst_per_lane": 15, "kernel": 16, "num_rows": 17,
    "xlate_code_bytes": 18, "xlate_inlined": 19, "xlate_called": 20, "xlate_unsaturated": 21, "xlate_valu": 22, "xlate_valu_slow": 23, "xlate_valu_clocks": 24, "xlate_vgpr_constants": 25, "xlate_builds": 26, "code_cache_hits": 27, "code_cached": 28, "xlate_background_builds": 29, "xlate_code_hash": 30, "stage_trials": 31, "control_rows": 32,
}

# every symbol include/fx8010_amd.h declares (tests check that the library exports them all)
SYMBOLS = [
    "fx_create", "fx_destroy", "fx_load_file", "fx_process", "fx_process_block", "fx_set_register", "fx_get_register",
    "fx_instruction_counter", "fx_error_count", "fx_error_desc", "fx_error_row", "fx_control_count", "fx_control_at",
    "fx_meta_get", "fx_set_option", "fxb_set_option", "fxp_set_option", "fx_set_channels", "fx_get_channels", "fx_ready", "fx_last_error", "fx_last_create_error",
    "fxb_create", "fxb_create_sharded", "fxb_create_on_devices", "fxb_shard_count", "fxb_shard_info", "fxb_shard_kernel_ms", "fxb_shard_plan", "fxb_process_block_dev_shards", "fxb_destroy", "fxb_load_file", "fxb_load_text", "fxb_set_register", "fxb_set_register_i",
    "fxb_get_register_i", "fxb_set_register_track", "fxb_set_register_array", "fxb_get_register_array", "fxb_seed_noise_i", "fxb_prepare", "fxb_state_size", "fxb_save_state", "fxb_load_state", "fxb_get_tram_i", "fxb_get_cursors_i", "fxb_process_block", "fxb_process_block_dev", "fxb_sync",
    "fxb_instruction_counter", "fxb_instruction_counter_i", "fxb_ood_flags", "fxb_error_count", "fxb_error_desc",
    "fxb_error_row", "fxb_control_count", "fxb_control_at", "fxb_meta_get", "fxb_ready", "fxb_last_error", "fxb_tier_note",
    "fxb_last_kernel_ms", "fxb_info", "fxb_device_count", "fxb_version", "fxb_host_alloc", "fxb_host_free",
    "fxp_create", "fxp_destroy", "fxp_load_file", "fxp_load_text", "fxp_num_registers", "fxp_register_name",
    "fxp_register_type", "fxp_register_ioindex", "fxp_register_value", "fxp_num_instructions", "fxp_instruction",
    "fxp_itram_size", "fxp_xtram_size", "fxp_error_count", "fxp_error_desc", "fxp_error_row", "fxp_control_count",
    "fxp_control_at", "fxp_meta_get", "fxp_ready", "fxp_lut", "fxp_lower", "fxp_lower_info", "fxp_translate", "fxp_track_register", "fxp_translate_staged", "fxp_code_hash", "fxp_last_error",
]


def load():
    """dlopen the in-tree library and declare the prototypes; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C fx8010-emulator-core_amd/csrc`" % path)
    lib = C.CDLL(path)
    vp, cp, i32, i64, f32 = C.c_void_p, C.c_char_p, C.c_int, C.c_int64, C.c_float

    def sig(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("fx_create", vp, i32); sig("fx_destroy", None, vp); sig("fx_load_file", i32, vp, cp)
    sig("fx_process", i32, vp, _f32p, _f32p); sig("fx_process_block", i32, vp, _f32p, _f32p, i32)
    sig("fx_set_register", i32, vp, cp, f32); sig("fx_get_register", f32, vp, cp)
    sig("fx_instruction_counter", i64, vp)
    for pfx in ("fx_", "fxb_", "fxp_"):
        sig(pfx + "set_option", i32, vp, C.c_uint, i32)
    sig("fx_set_channels", None, vp, i32); sig("fx_get_channels", i32, vp); sig("fx_ready", i32, vp)
    sig("fx_last_error", cp, vp); sig("fx_last_create_error", cp)
    sig("fxb_create", vp, i64, i32, i32); sig("fxb_destroy", None, vp)
    sig("fxb_create_sharded", vp, i64, i32, C.c_uint64); sig("fxb_create_on_devices", vp, i64, i32, C.POINTER(C.c_int), i32)
    sig("fxb_shard_count", i32, vp); sig("fxb_shard_info", i32, vp, i32, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64))
    sig("fxb_shard_plan", i32, i64, i32, C.POINTER(i64), C.POINTER(i64))
    sig("fxb_shard_kernel_ms", f32, vp, i32)
    sig("fxb_prepare", i32, vp, i32, i32); sig("fxb_state_size", i64, vp); sig("fxb_save_state", i32, vp, vp, i64); sig("fxb_load_state", i32, vp, vp, i64)
    sig("fxb_get_tram_i", i32, vp, i32, i64, vp, i32); sig("fxb_get_cursors_i", i32, vp, i64, C.POINTER(C.c_int32))
    sig("fxb_process_block_dev_shards", i32, vp, C.POINTER(vp), C.POINTER(vp), i32)
    sig("fxb_load_file", i32, vp, cp); sig("fxb_load_text", i32, vp, cp)
    sig("fxb_set_register", i32, vp, cp, f32); sig("fxb_set_register_i", i32, vp, cp, i64, f32)
    sig("fxb_set_register_track", i32, vp, cp, vp, i32, i32, i32)
    sig("fxb_set_register_array", i32, vp, cp, vp); sig("fxb_get_register_array", i32, vp, cp, vp)
    sig("fxb_get_register_i", f32, vp, cp, i64); sig("fxb_seed_noise_i", i32, vp, i64, C.c_int32, C.c_int32)
    sig("fxb_process_block", i32, vp, _f32p, _f32p, i32)
    sig("fxb_process_block_dev", i32, vp, vp, vp, i32, vp); sig("fxb_sync", i32, vp)
    sig("fxb_instruction_counter", i64, vp); sig("fxb_instruction_counter_i", i64, vp, i64)
    sig("fxb_ood_flags", C.c_uint32, vp); sig("fxb_ready", i32, vp); sig("fxb_last_error", cp, vp); sig("fxb_tier_note", i32, vp, C.c_char_p, i32)
    sig("fxb_last_kernel_ms", f32, vp); sig("fxb_info", i64, vp, i32)
    sig("fxb_device_count", i32); sig("fxb_version", cp)
    sig("fxb_host_alloc", vp, i64); sig("fxb_host_free", None, vp)
    for pfx in ("fx_", "fxb_", "fxp_"):
        sig(pfx + "error_count", i32, vp); sig(pfx + "error_desc", cp, vp, i32); sig(pfx + "error_row", i32, vp, i32)
        sig(pfx + "control_count", i32, vp); sig(pfx + "control_at", cp, vp, i32)
        sig(pfx + "meta_get", i32, vp, cp, cp, i32)
    sig("fxp_create", vp, i32); sig("fxp_destroy", None, vp); sig("fxp_load_file", i32, vp, cp); sig("fxp_load_text", i32, vp, cp)
    sig("fxp_num_registers", i32, vp); sig("fxp_register_name", cp, vp, i32); sig("fxp_register_type", i32, vp, i32)
    sig("fxp_register_ioindex", i32, vp, i32); sig("fxp_register_value", f32, vp, i32)
    sig("fxp_num_instructions", i32, vp); sig("fxp_instruction", None, vp, i32, C.POINTER(C.c_int))
    sig("fxp_itram_size", i32, vp); sig("fxp_xtram_size", i32, vp); sig("fxp_ready", i32, vp)
    sig("fxp_lut", C.POINTER(C.c_double), i32, i32); sig("fxp_lower", i32, vp); sig("fxp_lower_info", i64, vp, i32)
    sig("fxp_last_error", cp, vp)
    sig("fxp_translate", i64, vp, i32, i32, vp, i64, C.c_char_p, i64)
    sig("fxp_track_register", i32, vp, cp)
    sig("fxp_translate_staged", i64, vp, i32, i32, i32, i32, vp, i64, C.c_char_p, i64, C.POINTER(C.c_int), C.POINTER(C.c_int), i32)
    sig("fxp_code_hash", i64, vp, i32, i32, C.c_uint)
    _lib = lib
    return lib


def device_count():
    return int(load().fxb_device_count())


class HostBuffer:
    """float32 numpy array in pinned, device-visible host memory (fxb_host_alloc): blocks on such buffers are processed in place"""

    def __init__(self, shape):
        self._lib = load()
        self.shape = tuple(int(v) for v in shape)
        n = int(np.prod(self.shape))
        self._p = self._lib.fxb_host_alloc(n * 4)
        if not self._p:
            raise RuntimeError(self._lib.fx_last_create_error().decode("latin-1"))
        self.array = np.ctypeslib.as_array(C.cast(self._p, _f32p), shape=(n,)).reshape(self.shape)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            self._lib.fxb_host_free(self._p)
            self._p = None

    __del__ = close


class _Reports:
    """error list / control list / metadata accessors shared by the three handle kinds"""
    _pfx = ""

    def _call(self, name, *a):
        return getattr(self._lib, self._pfx + name)(self._h, *a)

    def set_option(self, option, on=True):
        """FX_OPT_* (OPT_TRAM_DANE, OPT_TRAM_ADDR_SHIFT, OPT_TRAM_INTERP): behaviour beyond the reference; before loading"""
        rc = self._call("set_option", option, 1 if on else 0)
        if rc != 0:
            raise RuntimeError("set_option(%d) failed: %d" % (option, rc))

    def errors(self):
        return [(self._call("error_desc", i).decode("latin-1"), self._call("error_row", i)) for i in range(self._call("error_count"))]

    def controls(self):
        return [self._call("control_at", i).decode("latin-1") for i in range(self._call("control_count"))]

    def meta(self):
        out = {}
        buf = C.create_string_buffer(1024)
        for k in ("name", "copyright", "created", "engine", "comment", "guid"):
            if self._call("meta_get", k.encode(), buf, 1024):
                out[k] = buf.value.decode("latin-1")
        return out

    def ready(self):
        return bool(self._call("ready"))


class FrontEnd(_Reports):
    """Host-only loader + lowering (fxp_*): no GPU needed."""
    _pfx = "fxp_"

    def __init__(self, channels=1):
        self._lib = load()
        self._h = self._lib.fxp_create(channels)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fxp_destroy(self._h)
            self._h = None

    __del__ = close

    def load_text(self, text):
        return bool(self._lib.fxp_load_text(self._h, text.encode() if isinstance(text, str) else text))

    def load_file(self, path):
        return bool(self._lib.fxp_load_file(self._h, path.encode()))

    def registers(self):
        n = self._lib.fxp_num_registers(self._h)
        return [(self._lib.fxp_register_name(self._h, i).decode("latin-1"), self._lib.fxp_register_type(self._h, i),
                 self._lib.fxp_register_ioindex(self._h, i),
                 int(np.float32(self._lib.fxp_register_value(self._h, i)).view(np.uint32))) for i in range(n)]

    def instructions(self):
        buf = (C.c_int * 8)()
        out = []
        for i in range(self._lib.fxp_num_instructions(self._h)):
            self._lib.fxp_instruction(self._h, i, buf)
            out.append(tuple(buf))
        return out

    def tram_sizes(self):
        return self._lib.fxp_itram_size(self._h), self._lib.fxp_xtram_size(self._h)

    def lower(self):
        return int(self._lib.fxp_lower(self._h))

    def lower_info(self, what):
        return int(self._lib.fxp_lower_info(self._h, INFO[what]))

    def track_register(self, key):
        """translate() from now on generates the code of a batch in which `key` can have a control track"""
        rc = int(self._lib.fxp_track_register(self._h, key.encode()))
        if rc < 0:
            raise RuntimeError("fxp_track_register: %d %s" % (rc, self.last_error()))
        return rc

    def translate(self, vgprs=0, stream=0):
        """gfx950 machine code of the program as the batch path generates it: (code bytes, assembler listing).
        stream: 0 steady fast, 1 steady exact, 2 last-sample fast, 3 last-sample exact."""
        cap, tcap = 1 << 20, 1 << 23
        code = C.create_string_buffer(cap)
        text = C.create_string_buffer(tcap)
        n = int(self._lib.fxp_translate(self._h, int(vgprs), int(stream), code, cap, text, tcap))
        if n < 0:
            raise RuntimeError("fxp_translate: %d %s" % (n, self.last_error()))
        return code.raw[:n], text.value.decode("ascii")

    def translate_staged(self, stages, stage=0, stream=0, vgprs=0):
        """the program cut into at most `stages` pipeline stages (fx_xlate.hpp StageInfo): (code, listing, actual stages, info)
        of one stream of one stage; info = [cut record, rows handed over] per cut + [LDS bytes]"""
        cap, tcap = 1 << 20, 1 << 23
        code = C.create_string_buffer(cap)
        text = C.create_string_buffer(tcap)
        actual = C.c_int(0)
        info = (C.c_int * 512)()
        n = int(self._lib.fxp_translate_staged(self._h, int(vgprs), int(stages), int(stage), int(stream), code, cap, text, tcap, C.byref(actual), info, 512))
        if n < 0:
            raise RuntimeError("fxp_translate_staged: %d %s" % (n, self.last_error()))
        k = actual.value
        head = 2 * max(k - 1, 0) + (1 if k > 1 else 0)
        self.stage_store = [int(v) for v in info[head + 1: head + 1 + int(info[head])]] if k > 1 else []   # per row: the stage that stores it
        return code.raw[:n], text.value.decode("ascii"), k, [int(v) for v in info[:head]]

    def code_hash(self, vgprs, stages=1, tram_streaming=False, priority_slices=False):
        """fingerprint of the code object a batch would load (fxb_info xlate_code_hash of a batch in that situation);
        priority_slices: what a batch generates when its launch fills the build's wave slots once, two or more per SIMD"""
        v = int(self._lib.fxp_code_hash(self._h, int(vgprs), int(stages), (1 if tram_streaming else 0) | (2 if priority_slices else 0)))
        if v < 0:
            raise RuntimeError("fxp_code_hash: %d %s" % (v, self.last_error()))
        return v

    def last_error(self):
        return self._lib.fxp_last_error(self._h).decode("latin-1")

    @staticmethod
    def lut(kind, exponent):
        p = load().fxp_lut(kind, exponent)
        return np.ctypeslib.as_array(p, shape=(64,)).copy()


def shard_plan(n_instances, n_shards):
    """[(first_instance, count)] of the partition fxb_create_sharded / fxb_create_on_devices would use; None when a shard
    would be empty.  No device needed."""
    lib = load()
    first, count = (C.c_int64 * max(n_shards, 1))(), (C.c_int64 * max(n_shards, 1))()
    if lib.fxb_shard_plan(int(n_instances), int(n_shards), first, count) != 0:
        return None
    return [(int(first[k]), int(count[k])) for k in range(n_shards)]


class Batch(_Reports):
    """N instances of one program on one GPU (fxb_*)."""
    _pfx = "fxb_"

    def __init__(self, n_instances, channels=1, device=-1, devices=None, device_mask=None):
        """device: one HIP ordinal (-1: current).  devices=[...]: one shard per entry (fxb_create_on_devices; an ordinal may
        repeat).  device_mask: one shard per set bit (fxb_create_sharded)."""
        self._lib = load()
        self.n = int(n_instances)
        self.channels = channels
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            self._h = self._lib.fxb_create_on_devices(self.n, channels, arr, len(devices))
        elif device_mask is not None:
            self._h = self._lib.fxb_create_sharded(self.n, channels, C.c_uint64(device_mask))
        else:
            self._h = self._lib.fxb_create(self.n, channels, device)
        if not self._h:
            raise RuntimeError("fxb_create failed: " + self._lib.fx_last_create_error().decode("latin-1"))

    def shards(self):
        """[(device, first_instance, n_instances)] of the batch's shards"""
        out = []
        for k in range(self._lib.fxb_shard_count(self._h)):
            dev, first, cnt = C.c_int(), C.c_int64(), C.c_int64()
            self._check(self._lib.fxb_shard_info(self._h, k, C.byref(dev), C.byref(first), C.byref(cnt)), "shard_info")
            out.append((dev.value, first.value, cnt.value))
        return out

    def shard_kernel_ms(self):
        """HIP-event time of every shard's most recent launch, ms"""
        return [float(self._lib.fxb_shard_kernel_ms(self._h, k)) for k in range(self._lib.fxb_shard_count(self._h))]

    def process_block_dev_shards(self, d_in, d_out, n_samples):
        """d_in / d_out: one device pointer (int) per shard; asynchronous."""
        k = len(d_in)
        a = (C.c_void_p * k)(*[C.c_void_p(p) for p in d_in])
        b = (C.c_void_p * k)(*[C.c_void_p(p) for p in d_out])
        return self._check(self._lib.fxb_process_block_dev_shards(self._h, a, b, n_samples), "process_block_dev_shards")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fxb_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self._lib.fxb_last_error(self._h).decode("latin-1")))
        return rc

    def load_text(self, text):
        return bool(self._lib.fxb_load_text(self._h, text.encode() if isinstance(text, str) else text))

    def load_file(self, path):
        return bool(self._lib.fxb_load_file(self._h, path.encode()))

    def set_register(self, key, v):
        return self._check(self._lib.fxb_set_register(self._h, key.encode(), C.c_float(v)), "set_register")

    def set_register_i(self, key, inst, v):
        return self._check(self._lib.fxb_set_register_i(self._h, key.encode(), inst, C.c_float(v)), "set_register_i")

    def set_register_array(self, key, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        assert v.shape == (self.n,)
        return self._check(self._lib.fxb_set_register_array(self._h, key.encode(), v.ctypes.data_as(C.c_void_p)), "set_register_array")

    def set_register_track(self, key, values, period):
        """values: [steps] (one value for all instances) or [steps, N] (per instance); applied by the next process call at
        samples 0, period, 2*period, ..."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        per = v.ndim == 2
        assert v.ndim == 1 or v.shape[1] == self.n
        return self._check(self._lib.fxb_set_register_track(self._h, key.encode(), v.ctypes.data_as(C.c_void_p), int(v.shape[0]), int(period), 1 if per else 0),
                           "set_register_track")

    def get_register_array(self, key):
        v = np.empty(self.n, dtype=np.float32)
        self._check(self._lib.fxb_get_register_array(self._h, key.encode(), v.ctypes.data_as(C.c_void_p)), "get_register_array")
        return v

    def get_register_i(self, key, inst):
        return float(self._lib.fxb_get_register_i(self._h, key.encode(), inst))

    def get_register_bits_i(self, key, inst):
        return int(np.float32(self.get_register_i(key, inst)).view(np.uint32))

    def seed_noise_i(self, inst, x1, x2):
        return self._check(self._lib.fxb_seed_noise_i(self._h, inst, x1, x2), "seed_noise_i")

    def prepare(self, n_samples, wait=True):
        """generate the code for blocks of n_samples samples now (and wait for the builder thread's follow-ups)"""
        return self._check(self._lib.fxb_prepare(self._h, int(n_samples), 1 if wait else 0), "prepare")

    def save_state(self):
        """the whole batch's state as one image (numpy uint8): registers, latches, delay memory, positions, LFSR, counters"""
        n = int(self._lib.fxb_state_size(self._h))
        if n < 0:
            raise RuntimeError("state_size failed (%d): %s" % (n, self.last_error()))
        buf = np.empty(n, dtype=np.uint8)
        self._check(self._lib.fxb_save_state(self._h, buf.ctypes.data_as(C.c_void_p), n), "save_state")
        return buf

    def load_state(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        return self._check(self._lib.fxb_load_state(self._h, image.ctypes.data_as(C.c_void_p), image.size), "load_state")

    def get_tram_i(self, which, inst, n_slots):
        out = np.empty(n_slots, dtype=np.float32)
        self._check(self._lib.fxb_get_tram_i(self._h, int(which), int(inst), out.ctypes.data_as(C.c_void_p), int(n_slots)), "get_tram_i")
        return out

    def get_cursors_i(self, inst):
        buf = (C.c_int32 * 4)()
        self._check(self._lib.fxb_get_cursors_i(self._h, int(inst), buf), "get_cursors_i")
        return list(buf)

    def process_block(self, x, out=None):
        """x: float32 [S, N] (mono) or [S, channels, N]; returns the same shape (into `out` when given: e.g. a view of pinned
        memory - large blocks from pinned buffers are copied at DMA rate and overlap with the kernel)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        S = x.shape[0]
        assert x.size == S * self.channels * self.n, "input must be [S, channels, N]"
        if out is None:
            out = np.empty_like(x)
        assert out.dtype == np.float32 and out.flags["C_CONTIGUOUS"] and out.size == x.size
        self._check(self._lib.fxb_process_block(self._h, x.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p), S), "process_block")
        return out

    def process_block_dev(self, d_in, d_out, n_samples, stream=None):
        """device pointers (ints); asynchronous."""
        return self._check(self._lib.fxb_process_block_dev(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n_samples, C.c_void_p(stream or 0)), "process_block_dev")

    def sync(self):
        return self._check(self._lib.fxb_sync(self._h), "sync")

    def instruction_counter(self):
        return int(self._lib.fxb_instruction_counter(self._h))

    def instruction_counter_i(self, inst):
        return int(self._lib.fxb_instruction_counter_i(self._h, inst))

    def ood_flags(self):
        return int(self._lib.fxb_ood_flags(self._h))

    def tier_note(self):
        """which tier runs the program as it stands, and why not a faster one"""
        buf = C.create_string_buffer(512)
        self._check(min(self._lib.fxb_tier_note(self._h, buf, 512), 0), "tier_note")
        return buf.value.decode("latin-1")

    def last_kernel_ms(self):
        return float(self._lib.fxb_last_kernel_ms(self._h))

    def info(self, what):
        return int(self._lib.fxb_info(self._h, INFO[what]))

    def last_error(self):
        return self._lib.fxb_last_error(self._h).decode("latin-1")


class Single(_Reports):
    """One emulated DSP with the reference's call-per-sample surface (fx_*)."""
    _pfx = "fx_"

    def __init__(self, channels=1):
        self._lib = load()
        self.channels = channels
        self._h = self._lib.fx_create(channels)
        if not self._h:
            raise RuntimeError("fx_create failed: " + self._lib.fx_last_create_error().decode("latin-1"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fx_destroy(self._h)
            self._h = None

    __del__ = close

    def load_file(self, path):
        return bool(self._lib.fx_load_file(self._h, path.encode()))

    def process(self, sample):
        x = np.ascontiguousarray(sample, dtype=np.float32).reshape(self.channels)
        out = np.empty_like(x)
        rc = self._lib.fx_process(self._h, x.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p))
        if rc < 0:
            raise RuntimeError("fx_process failed: " + self._lib.fx_last_error(self._h).decode("latin-1"))
        return out

    def process_block(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        rc = self._lib.fx_process_block(self._h, x.ctypes.data_as(_f32p), out.ctypes.data_as(_f32p), x.size // self.channels)
        if rc < 0:
            raise RuntimeError("fx_process_block failed: " + self._lib.fx_last_error(self._h).decode("latin-1"))
        return out

    def set_register(self, key, v):
        return int(self._lib.fx_set_register(self._h, key.encode(), C.c_float(v)))

    def get_register(self, key):
        return float(self._lib.fx_get_register(self._h, key.encode()))

    def instruction_counter(self):
        return int(self._lib.fx_instruction_counter(self._h))
