"""Host-side mirror of the frame level of include/x264hip.h (ctypes).

Thin: it owns no arithmetic.  Device buffers other than pictures are
DeviceArray objects backed by the library's own malloc/memcpy entry points
(torch's bundled HIP runtime and the system one this library links cannot
both drive the GPU from one process, so torch.cuda is never touched here).
"""
import ctypes as C

import numpy as np

PADH = PADV = 32


class Dims(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("mb_w", C.c_int), ("mb_h", C.c_int),
                ("stride_y", C.c_int), ("stride_c", C.c_int), ("lines_y", C.c_int), ("lines_c", C.c_int),
                ("batch", C.c_int)]


class Picture(C.Structure):
    _fields_ = [("plane", C.c_void_p * 3), ("filtered", C.c_void_p * 4), ("lowres", C.c_void_p * 4),
                ("integral", C.c_void_p), ("stride_lowres", C.c_int), ("width_lowres", C.c_int),
                ("lines_lowres", C.c_int)]


class MeParams(C.Structure):
    _fields_ = [("range", C.c_int), ("cost_mv", C.c_void_p), ("cost_mv_range", C.c_int),
                ("centers", C.c_void_p), ("mvp", C.c_void_p), ("sad_surface", C.c_void_p),
                ("mv_range", C.c_int)]


class Me16Params(C.Structure):
    _fields_ = [("me_method", C.c_int), ("me_range", C.c_int), ("subme", C.c_int), ("chroma_me", C.c_int),
                ("mv_range", C.c_int), ("cost_mv", C.c_void_p), ("cost_mv_range", C.c_int),
                ("mvp", C.c_void_p), ("mvc", C.c_void_p), ("n_mvc", C.c_void_p), ("ref_cost", C.c_int * 8)]


class ResidualParams(C.Structure):
    _fields_ = [("qp", C.c_int), ("qp_chroma", C.c_int), ("transform8x8", C.c_int), ("b_interlaced", C.c_int),
                ("quant4_mf", C.c_void_p), ("quant4_bias", C.c_void_p),
                ("quant8_mf", C.c_void_p), ("quant8_bias", C.c_void_p),
                ("dequant4_mf", C.c_void_p), ("dequant8_mf", C.c_void_p),
                ("mv4x4_out", C.c_void_p), ("ref_out", C.c_void_p)]


class DeblockParams(C.Structure):
    _fields_ = [("mb_type", C.c_void_p), ("qp", C.c_void_p), ("nnz", C.c_void_p), ("transform8x8", C.c_void_p),
                ("mv", C.c_void_p), ("ref", C.c_void_p),
                ("alpha_c0_offset", C.c_int), ("beta_offset", C.c_int), ("chroma_qp_offset", C.c_int),
                ("state_layout", C.c_int), ("sub8x8", C.c_int)]


class DeviceArray:
    """A device buffer with a numpy shape/dtype, moved with the C ABI's memcpy helpers."""

    def __init__(self, lib, shape, dtype, init=None):
        self.lib, self.shape, self.dtype = lib, tuple(np.atleast_1d(shape)), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        lib.x264hip_malloc.restype = C.c_void_p
        self.ptr = lib.x264hip_malloc(C.c_size_t(self.nbytes))
        if not self.ptr:
            raise MemoryError("x264hip_malloc(%d) failed" % self.nbytes)
        if init is not None:
            self.set(init)

    @property
    def p(self):
        return C.c_void_p(self.ptr)

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes == self.nbytes
        assert self.lib.x264hip_memcpy_h2d(self.p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes)) == 0

    def get(self):
        out = np.zeros(self.shape, self.dtype)
        assert self.lib.x264hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.p, C.c_size_t(self.nbytes)) == 0
        return out

    def free(self):
        if self.ptr:
            self.lib.x264hip_free(self.p)
            self.ptr = None


PLANE_IDS = {"y": 0, "u": 1, "v": 2, "h": 3, "vv": 4, "c": 5, "l0": 6, "lh": 7, "lv": 8, "lc": 9}


class FrameCtx:
    """x264hip_frame_ctx + helpers to move numpy images in and out."""

    def __init__(self, lib, width, height, stream=None, batch=1):
        self.lib = lib
        lib.x264hip_frame_ctx_new.restype = C.c_void_p
        lib.x264hip_frame_ctx_stream.restype = C.c_void_p
        self.batch = batch
        self.dims = Dims(width=width, height=height, batch=batch)
        self.h = lib.x264hip_frame_ctx_new(C.byref(self.dims), C.c_void_p(stream))
        if not self.h:
            raise RuntimeError("x264hip_frame_ctx_new failed: %s" % lib.x264hip_last_error().decode())
        self.h = C.c_void_p(self.h)
        self.pictures = []

    @property
    def stream(self):
        return self.lib.x264hip_frame_ctx_stream(self.h)

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.x264hip_last_error().decode()))

    def new_picture(self, source_only=False):
        pic = Picture()
        if source_only:                 # Y, U, V only: an input frame
            self.check(self.lib.x264hip_picture_alloc_source(self.h, C.byref(pic)), "picture_alloc_source")
        else:
            self.check(self.lib.x264hip_picture_alloc(self.h, C.byref(pic)), "picture_alloc")
        self.pictures.append(pic)
        return pic

    def copy_element(self, dst, dst_b, src, src_b):
        self.check(self.lib.x264hip_picture_copy_element(self.h, C.byref(dst), C.c_int(dst_b), C.byref(src), C.c_int(src_b)), "picture_copy_element")

    def synth(self, pic, t0, t_stride=0):
        """Element b of `pic` becomes frame t0 + b * t_stride of the synthetic clip (x264hip_picture_synth; synth.frame on the device)."""
        self.check(self.lib.x264hip_picture_synth(self.h, C.byref(pic), C.c_int(t0), C.c_int(t_stride)), "picture_synth")

    def select(self, b):
        """Choose the batch element that upload / download / x264hip_ssd_frame address."""
        self.check(self.lib.x264hip_frame_ctx_select(self.h, b), "frame_ctx_select")

    def upload(self, pic, y, u, v, b=None):
        if b is not None:
            self.select(b)
        y, u, v = (np.ascontiguousarray(a) for a in (y, u, v))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.check(self.lib.x264hip_picture_upload(self.h, C.byref(pic), p(y), y.shape[1], p(u), u.shape[1],
                                                   p(v), v.shape[1]), "picture_upload")

    def geometry(self, pic, name):
        d = self.dims
        i = PLANE_IDS[name]
        if i == 0 or 3 <= i < 6:
            return d.stride_y, d.mb_w * 16, d.lines_y, PADH, PADV
        if i < 3:
            return d.stride_c, d.mb_w * 8, d.lines_c, PADH // 2, PADV // 2
        return pic.stride_lowres, pic.width_lowres, pic.lines_lowres, PADH, PADV

    def download(self, pic, name, padded=True, b=None):
        if b is not None:
            self.select(b)
        stride, w, h, padh, padv = self.geometry(pic, name)
        shape = (h + 2 * padv, stride) if padded else (h, w)
        out = np.zeros(shape, np.uint8)
        self.check(self.lib.x264hip_picture_download(self.h, C.byref(pic), PLANE_IDS[name],
                                                     out.ctypes.data_as(C.c_void_p), shape[1], int(padded)),
                   "picture_download")
        return out

    def sync(self):
        self.check(self.lib.x264hip_sync(self.h), "sync")

    def close(self):
        for pic in self.pictures:
            self.lib.x264hip_picture_free(self.h, C.byref(pic))
        self.pictures = []
        if self.h:
            self.lib.x264hip_frame_ctx_delete(self.h)
            self.h = None


def chroma_qp(qp, offset=0):
    """h->mb.i_chroma_qp: the H.264 chroma QP mapping (table 8-15; R/common/macroblock.h:241-251)."""
    q = min(max(qp + offset, 0), 51)
    return q if q < 30 else (29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39)[q - 30]


class CqmDevice:
    """Flat-matrix quantiser tables (x264_cqm_init's output) resident in HBM."""

    def __init__(self, lib, cqm):
        self.bufs = {
            "quant4_mf": DeviceArray(lib, cqm["quant4_mf"].shape, np.uint16, cqm["quant4_mf"]),
            "quant4_bias": DeviceArray(lib, cqm["quant4_bias"].shape, np.uint16, cqm["quant4_bias"]),
            "quant8_mf": DeviceArray(lib, cqm["quant8_mf"].shape, np.uint16, cqm["quant8_mf"]),
            "quant8_bias": DeviceArray(lib, cqm["quant8_bias"].shape, np.uint16, cqm["quant8_bias"]),
            "dequant4_mf": DeviceArray(lib, cqm["dequant4_mf"].shape, np.int32, cqm["dequant4_mf"]),
            "dequant8_mf": DeviceArray(lib, cqm["dequant8_mf"].shape, np.int32, cqm["dequant8_mf"]),
        }

    def params(self, qp, transform8x8=0, interlaced=0, chroma_offset=0):
        b = self.bufs
        return ResidualParams(qp=qp, qp_chroma=chroma_qp(qp, chroma_offset), transform8x8=transform8x8,
                              b_interlaced=interlaced, quant4_mf=b["quant4_mf"].ptr, quant4_bias=b["quant4_bias"].ptr,
                              quant8_mf=b["quant8_mf"].ptr, quant8_bias=b["quant8_bias"].ptr,
                              dequant4_mf=b["dequant4_mf"].ptr, dequant8_mf=b["dequant8_mf"].ptr)

    def free(self):
        for v in self.bufs.values():
            v.free()


class CqmTables(C.Structure):
    """x264hip_cqm_tables (include/x264hip.h)."""
    _fields_ = [("quant4_mf", C.c_uint16 * (4 * 52 * 16)), ("quant4_bias", C.c_uint16 * (4 * 52 * 16)),
                ("quant8_mf", C.c_uint16 * (2 * 52 * 64)), ("quant8_bias", C.c_uint16 * (2 * 52 * 64)),
                ("dequant4_mf", C.c_int32 * (4 * 6 * 16)), ("dequant8_mf", C.c_int32 * (2 * 6 * 64)),
                ("unquant4_mf", C.c_int32 * (4 * 52 * 16)), ("unquant8_mf", C.c_int32 * (2 * 52 * 64))]


CQM_SHAPES = {"quant4_mf": (4, 52, 16), "quant4_bias": (4, 52, 16), "quant8_mf": (2, 52, 64), "quant8_bias": (2, 52, 64),
              "dequant4_mf": (4, 6, 16), "dequant8_mf": (2, 6, 64), "unquant4_mf": (4, 52, 16), "unquant8_mf": (2, 52, 64)}


# the default scaling lists of ITU-T H.264 tables 7-3 / 7-4 in raster order (what --cqm jvt selects: x264_cqm_jvt, R/common/set.c): data of the standard
JVT4I = [6, 13, 20, 28, 13, 20, 28, 32, 20, 28, 32, 37, 28, 32, 37, 42]
JVT4P = [10, 14, 20, 24, 14, 20, 24, 27, 20, 24, 27, 30, 24, 27, 30, 34]
JVT8I = [6, 10, 13, 16, 18, 23, 25, 27, 10, 11, 16, 18, 23, 25, 27, 29, 13, 16, 18, 23, 25, 27, 29, 31, 16, 18, 23, 25, 27, 29, 31, 33,
         18, 23, 25, 27, 29, 31, 33, 36, 23, 25, 27, 29, 31, 33, 36, 38, 25, 27, 29, 31, 33, 36, 38, 40, 27, 29, 31, 33, 36, 38, 40, 42]
JVT8P = [9, 13, 15, 17, 19, 21, 22, 24, 13, 13, 17, 19, 21, 22, 24, 25, 15, 17, 19, 21, 22, 24, 25, 27, 17, 19, 21, 22, 24, 25, 27, 28,
         19, 21, 22, 24, 25, 27, 28, 30, 21, 22, 24, 25, 27, 28, 30, 32, 22, 24, 25, 27, 28, 30, 32, 33, 24, 25, 27, 28, 30, 32, 33, 35]
JVT_LISTS = [JVT4I, JVT4P, JVT4I, JVT4P, JVT8I, JVT8P]          # CQM_4IY, 4PY, 4IC, 4PC, 8IY, 8PY


def cqm_init(lib, scaling_lists=None, luma_deadzone=None, qp_min=0):
    """The quantiser tables from the library's x264hip_cqm_init (x264_cqm_init, R/common/set.c:68-168): a dict of numpy arrays with the
    layout of h->quant4_mf ... h->unquant8_mf.  scaling_lists: six raster-order lists (4x4 intra Y, inter Y, intra C, inter C, 8x8
    intra Y, inter Y) or None for flat matrices.  Needs no GPU."""
    t = CqmTables()
    lists = None
    keep = []
    if scaling_lists is not None:
        arr = (C.POINTER(C.c_uint8) * 6)()
        for i, l in enumerate(scaling_lists):
            a = np.ascontiguousarray(l, dtype=np.uint8)
            assert a.size == (16 if i < 4 else 64)
            keep.append(a)
            arr[i] = a.ctypes.data_as(C.POINTER(C.c_uint8))
        lists = arr
    dz = (C.c_int * 2)(*luma_deadzone) if luma_deadzone is not None else None
    if lib.x264hip_cqm_init(lists, dz, C.c_int(qp_min), C.byref(t)) != 0:
        raise RuntimeError("x264hip_cqm_init failed: %s" % lib.x264hip_last_error().decode())
    return {k: np.ctypeslib.as_array(getattr(t, k)).reshape(shape).copy() for k, shape in CQM_SHAPES.items()}


def host_plane(stride, lines, padh, padv):
    """Zeroed host image with the device layout; returns (full, offset_of_pixel00)."""
    full = np.zeros((lines + 2 * padv, stride), np.uint8)
    return full, padv * stride + padh


def cost_mv_table(lam, span):
    """p_cost_mv for one lambda (R/encoder/analyse.c:182-198): index i + span
    holds the cost of a qpel delta of i.  The reference's macro is
    log2f(x) = ((float)log((double)x)) / log(2.0) (analyse.c:40), i.e. the
    natural log is rounded to float and everything after it is double; the
    result is truncated to int16."""
    i = np.arange(0, span + 1, dtype=np.float64)
    log2 = np.log(i + 1.0).astype(np.float32).astype(np.float64) / np.log(2.0)
    v = (lam * (log2 * 2 + float(np.float32(0.718)) + (i != 0)) + float(np.float32(0.5)))
    v = v.astype(np.int64).astype(np.int16).view(np.uint16)
    tab = np.zeros(2 * span + 1, np.uint16)
    tab[span:] = v
    tab[:span] = v[1:][::-1]
    return tab
