"""oracle/pyoracle.py — TEST INFRASTRUCTURE ONLY.

ctypes bindings for the two CPU checkers:

* ``Oracle``    — oracle/libfxoracle.so, this repo's C restatement of the reference
                  interpreter (oracle/fx8010_oracle.c).  Always available after
                  ``make -C oracle port``.
* ``Reference`` — oracle/_ref/libfxref.so, the UNMODIFIED reference compiled from
                  /root/reference by oracle/Makefile (``make ref``); present in the build
                  container and, as a prebuilt .so, on the GPU box.  ``Reference.available()``
                  says whether it can be used.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FXORACLE_SO selects another build of the same restatement (oracle/Makefile `variants`: -O0, clang, ASan+UBSan) for
# tests/test_oracle_variants.py; everything else uses the pinned gcc -O2 build
_PORT_SO = os.environ.get("FXORACLE_SO") or os.path.join(_HERE, "libfxoracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libfxref.so")

_f32p = C.POINTER(C.c_float)


def _as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


class _Base:
    """Shared Python surface; subclasses bind the symbol prefix."""

    _lib = None
    _pfx = ""

    def _fn(self, name):
        return getattr(self._lib, self._pfx + name)

    def __init__(self, channels=1):
        self.channels = channels
        self._h = self._fn("create")(channels)
        if not self._h:
            raise RuntimeError("create failed")
        self._tmp = []

    def close(self):
        if getattr(self, "_h", None):
            self._fn("destroy")(self._h)
            self._h = None
        for p in getattr(self, "_tmp", []):
            try:
                os.unlink(p)
            except OSError:
                pass
        self._tmp = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_file(self, path):
        return bool(self._fn("load_file")(self._h, path.encode()))

    def load_text(self, text):
        # written byte-exact: the loader is sensitive to the final newline
        fd, path = tempfile.mkstemp(suffix=".da")
        with os.fdopen(fd, "wb") as fh:
            fh.write(text.encode() if isinstance(text, str) else text)
        self._tmp.append(path)
        return self.load_file(path)

    def process_block(self, x):
        """x: [S] (mono) or [S, channels] float32 -> same shape."""
        x = np.asarray(x, dtype=np.float32)
        mono = x.ndim == 1
        xi = x.reshape(-1, self.channels)
        xi, pin = _as_f32(xi)
        out = np.empty_like(xi)
        self._fn("process_block")(self._h, pin, out.ctypes.data_as(_f32p), xi.shape[0])
        return out.reshape(-1) if mono else out

    def set_register(self, key, v):
        return self._fn("set_register")(self._h, key.encode(), C.c_float(v))

    def get_register(self, key):
        return float(self._fn("get_register")(self._h, key.encode()))

    def get_register_bits(self, key):
        return int(np.float32(self._fn("get_register")(self._h, key.encode())).view(np.uint32))

    def instruction_counter(self):
        return int(self._fn("instruction_counter")(self._h))

    def errors(self):
        n = self._fn("error_count")(self._h)
        return [(self._fn("error_desc")(self._h, i).decode("latin-1"), self._fn("error_row")(self._h, i)) for i in range(n)]

    def controls(self):
        n = self._fn("control_count")(self._h)
        return [self._fn("control_at")(self._h, i).decode("latin-1") for i in range(n)]

    def meta(self):
        out = {}
        buf = C.create_string_buffer(1024)
        for k in ("name", "copyright", "created", "engine", "comment", "guid"):
            if self._fn("meta_get")(self._h, k.encode(), buf, 1024):
                out[k] = buf.value.decode("latin-1")
        return out

    def ready(self):
        return bool(self._fn("ready")(self._h))


def _bind_common(lib, pfx, counter_t):
    g = lambda n: getattr(lib, pfx + n)
    g("create").restype = C.c_void_p
    g("create").argtypes = [C.c_int]
    g("destroy").argtypes = [C.c_void_p]
    g("destroy").restype = None
    g("load_file").argtypes = [C.c_void_p, C.c_char_p]
    g("load_file").restype = C.c_int
    g("process").argtypes = [C.c_void_p, _f32p, _f32p]
    g("process").restype = None
    g("process_block").argtypes = [C.c_void_p, _f32p, _f32p, C.c_int]
    g("process_block").restype = None
    g("set_register").argtypes = [C.c_void_p, C.c_char_p, C.c_float]
    g("set_register").restype = C.c_int
    g("get_register").argtypes = [C.c_void_p, C.c_char_p]
    g("get_register").restype = C.c_float
    g("instruction_counter").argtypes = [C.c_void_p]
    g("instruction_counter").restype = counter_t
    g("ready").argtypes = [C.c_void_p]
    g("ready").restype = C.c_int
    g("error_count").argtypes = [C.c_void_p]
    g("error_count").restype = C.c_int
    g("error_desc").argtypes = [C.c_void_p, C.c_int]
    g("error_desc").restype = C.c_char_p
    g("error_row").argtypes = [C.c_void_p, C.c_int]
    g("error_row").restype = C.c_int
    g("control_count").argtypes = [C.c_void_p]
    g("control_count").restype = C.c_int
    g("control_at").argtypes = [C.c_void_p, C.c_int]
    g("control_at").restype = C.c_char_p
    g("meta_get").argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    g("meta_get").restype = C.c_int
    g("bench").argtypes = [C.c_char_p, C.c_long, C.c_int, _f32p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
    g("bench").restype = C.c_double


def _bench(lib, pfx, path, samples, threads, stim):
    stim, pst = _as_f32(stim)
    instr = C.c_longlong(0)
    cs = C.c_double(0.0)
    secs = getattr(lib, pfx + "bench")(path.encode(), int(samples), int(threads), pst, stim.shape[0], C.byref(instr), C.byref(cs))
    return float(secs), int(instr.value), float(cs.value)


class Oracle(_Base):
    _pfx = "fxo_"

    @classmethod
    def _load(cls):
        if cls._lib is None:
            if not os.path.exists(_PORT_SO):
                raise RuntimeError("oracle/libfxoracle.so missing: run `make -C oracle port`")
            lib = C.CDLL(_PORT_SO)
            _bind_common(lib, "fxo_", C.c_int64)
            lib.fxo_load_text.argtypes = [C.c_void_p, C.c_char_p]
            lib.fxo_load_text.restype = C.c_int
            lib.fxo_ood_flags.argtypes = [C.c_void_p]
            lib.fxo_ood_flags.restype = C.c_uint
            lib.fxo_seed_noise.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
            lib.fxo_set_option.argtypes = [C.c_void_p, C.c_uint, C.c_int]
            lib.fxo_set_option.restype = None
            lib.fxo_num_registers.argtypes = [C.c_void_p]
            lib.fxo_register_name.argtypes = [C.c_void_p, C.c_int]
            lib.fxo_register_name.restype = C.c_char_p
            lib.fxo_register_type.argtypes = [C.c_void_p, C.c_int]
            lib.fxo_register_ioindex.argtypes = [C.c_void_p, C.c_int]
            lib.fxo_register_value.argtypes = [C.c_void_p, C.c_int]
            lib.fxo_register_value.restype = C.c_float
            lib.fxo_num_instructions.argtypes = [C.c_void_p]
            lib.fxo_instruction.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
            lib.fxo_instruction.restype = None
            lib.fxo_itram_size.argtypes = [C.c_void_p]
            lib.fxo_xtram_size.argtypes = [C.c_void_p]
            lib.fxo_lut.argtypes = [C.c_void_p, C.c_int, C.c_int]
            lib.fxo_lut.restype = C.POINTER(C.c_double)
            lib.fxo_tram.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            lib.fxo_tram.restype = C.c_int
            lib.fxo_cursors.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            lib.fxo_cursors.restype = None
            lib.fxo_lfsr.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
            lib.fxo_lfsr.restype = None
            cls._lib = lib
        return cls._lib

    def __init__(self, channels=1):
        self._load()
        super().__init__(channels)

    def load_text(self, text):
        b = text.encode() if isinstance(text, str) else text
        return bool(self._lib.fxo_load_text(self._h, b))

    def ood_flags(self):
        return int(self._lib.fxo_ood_flags(self._h))

    def set_option(self, option, on=True):
        """FXO_OPT_* (1 = DANE delay-line model, 2 = DANE address shift, 4 = interpolated reads): behaviour beyond the reference; before loading"""
        self._lib.fxo_set_option(self._h, C.c_uint(option), 1 if on else 0)

    def seed_noise(self, x1, x2):
        self._lib.fxo_seed_noise(self._h, C.c_int32(x1), C.c_int32(x2))

    def tram(self, which, n):
        """first n words of the delay memory (0: smallDelayBuffer, 1: largeDelayBuffer) as float32"""
        out = np.zeros(n, dtype=np.float32)
        self._lib.fxo_tram.restype = C.c_int
        got = self._lib.fxo_tram(self._h, int(which), out.ctypes.data_as(C.c_void_p), int(n))
        return out[:got]

    def cursors(self):
        """[iTRAM write, iTRAM read, xTRAM write, xTRAM read] positions"""
        buf = (C.c_int * 4)()
        self._lib.fxo_cursors(self._h, buf)
        return list(buf)

    def lfsr(self):
        buf = (C.c_int32 * 2)()
        self._lib.fxo_lfsr(self._h, buf)
        return [int(buf[0]), int(buf[1])]

    def registers(self):
        n = self._lib.fxo_num_registers(self._h)
        return [
            (
                self._lib.fxo_register_name(self._h, i).decode("latin-1"),
                self._lib.fxo_register_type(self._h, i),
                self._lib.fxo_register_ioindex(self._h, i),
                int(np.float32(self._lib.fxo_register_value(self._h, i)).view(np.uint32)),
            )
            for i in range(n)
        ]

    def instructions(self):
        n = self._lib.fxo_num_instructions(self._h)
        out = []
        buf = (C.c_int * 8)()
        for i in range(n):
            self._lib.fxo_instruction(self._h, i, buf)
            out.append(tuple(buf))
        return out

    def tram_sizes(self):
        return self._lib.fxo_itram_size(self._h), self._lib.fxo_xtram_size(self._h)

    def lut(self, kind, exponent):
        p = self._lib.fxo_lut(self._h, kind, exponent)
        return np.ctypeslib.as_array(p, shape=(64,)).copy()

    @classmethod
    def bench(cls, path, samples, threads, stim):
        return _bench(cls._load(), "fxo_", path, samples, threads, stim)


class Reference(_Base):
    _pfx = "ref_"

    @staticmethod
    def available():
        return os.path.exists(_REF_SO)

    @classmethod
    def _load(cls):
        if cls._lib is None:
            if not os.path.exists(_REF_SO):
                raise RuntimeError("oracle/_ref/libfxref.so missing (needs /root/reference: `make -C oracle ref`)")
            lib = C.CDLL(_REF_SO)
            _bind_common(lib, "ref_", C.c_int)
            cls._lib = lib
        return cls._lib

    def __init__(self, channels=1):
        self._load()
        super().__init__(channels)

    @classmethod
    def bench(cls, path, samples, threads, stim):
        return _bench(cls._load(), "ref_", path, samples, threads, stim)
