"""ctypes mirrors of include/x264hip_tables.h -- the six x264 DSP tables.

The layouts restate R/common/pixel.h:63-103, dct.h:89-124, quant.h:26-44,
mc.h:31-77, predict.h:27-29 and frame.h:94-108 (R/ =
x264-snapshot-20090216-2245/).  Any shared library that fills these structs
(this package's HIP back-end, the CPU oracle, or the reference's own C build)
can be driven through :class:`TableSet` with identical calling code, which is
what makes the parity tests read like the reference's checkasm.
"""
import ctypes as C

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int)
i8p = C.POINTER(C.c_int8)

CMP = C.CFUNCTYPE(C.c_int, u8p, C.c_int, u8p, C.c_int)
CMP_X3 = C.CFUNCTYPE(None, u8p, u8p, u8p, u8p, C.c_int, i32p)
CMP_X4 = C.CFUNCTYPE(None, u8p, u8p, u8p, u8p, u8p, C.c_int, i32p)
VAR = C.CFUNCTYPE(C.c_int, u8p, C.c_int)
HAC = C.CFUNCTYPE(C.c_uint64, u8p, C.c_int)
SSIM_CORE = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, i32p)
SSIM_END4 = C.CFUNCTYPE(C.c_float, i32p, i32p, C.c_int)
ADS = C.CFUNCTYPE(C.c_int, i32p, u16p, C.c_int, u16p, i16p, C.c_int, C.c_int)
INTRA_X3 = C.CFUNCTYPE(None, u8p, u8p, i32p)


class PixelTable(C.Structure):
    _fields_ = [
        ("sad", CMP * 7), ("ssd", CMP * 7), ("satd", CMP * 7), ("ssim", CMP * 7),
        ("sa8d", CMP * 4), ("mbcmp", CMP * 7), ("mbcmp_unaligned", CMP * 7),
        ("fpelcmp", CMP * 7), ("fpelcmp_x3", CMP_X3 * 7), ("fpelcmp_x4", CMP_X4 * 7),
        ("sad_aligned", CMP * 7),
        ("var", VAR * 4), ("hadamard_ac", HAC * 4),
        ("ssim_4x4x2_core", SSIM_CORE), ("ssim_end4", SSIM_END4),
        ("sad_x3", CMP_X3 * 7), ("sad_x4", CMP_X4 * 7),
        ("satd_x3", CMP_X3 * 7), ("satd_x4", CMP_X4 * 7),
        ("ads", ADS * 7),
        ("intra_mbcmp_x3_16x16", INTRA_X3), ("intra_satd_x3_16x16", INTRA_X3),
        ("intra_sad_x3_16x16", INTRA_X3), ("intra_satd_x3_8x8c", INTRA_X3),
        ("intra_satd_x3_4x4", INTRA_X3), ("intra_sa8d_x3_8x8", INTRA_X3),
    ]


SUB_DCT = C.CFUNCTYPE(None, i16p, u8p, u8p)
ADD_IDCT = C.CFUNCTYPE(None, u8p, i16p)
DC_FN = C.CFUNCTYPE(None, i16p)


class DctTable(C.Structure):
    _fields_ = [
        ("sub4x4_dct", SUB_DCT), ("add4x4_idct", ADD_IDCT),
        ("sub8x8_dct", SUB_DCT), ("add8x8_idct", ADD_IDCT), ("add8x8_idct_dc", ADD_IDCT),
        ("sub16x16_dct", SUB_DCT), ("add16x16_idct", ADD_IDCT), ("add16x16_idct_dc", ADD_IDCT),
        ("sub8x8_dct8", SUB_DCT), ("add8x8_idct8", ADD_IDCT),
        ("sub16x16_dct8", SUB_DCT), ("add16x16_idct8", ADD_IDCT),
        ("dct4x4dc", DC_FN), ("idct4x4dc", DC_FN),
    ]


SCAN = C.CFUNCTYPE(None, i16p, i16p)
ZSUB = C.CFUNCTYPE(None, i16p, u8p, u8p)
INTERLEAVE = C.CFUNCTYPE(None, i16p, i16p, u8p)


class ZigzagTable(C.Structure):
    _fields_ = [("scan_8x8", SCAN), ("scan_4x4", SCAN), ("sub_8x8", ZSUB), ("sub_4x4", ZSUB),
                ("interleave_8x8_cavlc", INTERLEAVE)]


class RunLevel(C.Structure):
    _fields_ = [("last", C.c_int), ("level", C.c_int16 * 16), ("run", C.c_uint8 * 16)]


QUANT = C.CFUNCTYPE(C.c_int, i16p, u16p, u16p)
QUANT_DC = C.CFUNCTYPE(C.c_int, i16p, C.c_int, C.c_int)
DEQUANT = C.CFUNCTYPE(None, i16p, i32p, C.c_int)
DENOISE = C.CFUNCTYPE(None, i16p, u32p, u16p, C.c_int)
COEF_INT = C.CFUNCTYPE(C.c_int, i16p)
LEVEL_RUN = C.CFUNCTYPE(C.c_int, i16p, C.POINTER(RunLevel))


class QuantTable(C.Structure):
    _fields_ = [
        ("quant_8x8", QUANT), ("quant_4x4", QUANT), ("quant_4x4_dc", QUANT_DC), ("quant_2x2_dc", QUANT_DC),
        ("dequant_8x8", DEQUANT), ("dequant_4x4", DEQUANT), ("dequant_4x4_dc", DEQUANT),
        ("denoise_dct", DENOISE),
        ("decimate_score15", COEF_INT), ("decimate_score16", COEF_INT), ("decimate_score64", COEF_INT),
        ("coeff_last", COEF_INT * 6), ("coeff_level_run", LEVEL_RUN * 5),
    ]


MC_LUMA = C.CFUNCTYPE(None, u8p, C.c_int, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
GET_REF = C.CFUNCTYPE(C.c_void_p, u8p, i32p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
MC_CHROMA = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
AVG = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int)
COPY = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, C.c_int)
PLANE_COPY = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int)
HPEL = C.CFUNCTYPE(None, u8p, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, i16p)
PREFETCH_FENC = C.CFUNCTYPE(None, u8p, C.c_int, u8p, C.c_int, C.c_int)
PREFETCH_REF = C.CFUNCTYPE(None, u8p, C.c_int, C.c_int)
MEMCPY = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
MEMZERO = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
INTEGRAL_H = C.CFUNCTYPE(None, u16p, u8p, C.c_int)
INTEGRAL_4V = C.CFUNCTYPE(None, u16p, u16p, C.c_int)
INTEGRAL_8V = C.CFUNCTYPE(None, u16p, C.c_int)
LOWRES = C.CFUNCTYPE(None, u8p, u8p, u8p, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int)


class McTable(C.Structure):
    _fields_ = [
        ("mc_luma", MC_LUMA), ("get_ref", GET_REF), ("mc_chroma", MC_CHROMA),
        ("avg", AVG * 10), ("copy", COPY * 7), ("copy_16x16_unaligned", COPY),
        ("plane_copy", PLANE_COPY), ("hpel_filter", HPEL),
        ("prefetch_fenc", PREFETCH_FENC), ("prefetch_ref", PREFETCH_REF),
        ("memcpy_aligned", MEMCPY), ("memzero_aligned", MEMZERO),
        ("integral_init4h", INTEGRAL_H), ("integral_init8h", INTEGRAL_H),
        ("integral_init4v", INTEGRAL_4V), ("integral_init8v", INTEGRAL_8V),
        ("frame_init_lowres_core", LOWRES),
    ]


PREDICT = C.CFUNCTYPE(None, u8p)
PREDICT8 = C.CFUNCTYPE(None, u8p, u8p)
PREDICT8_FILTER = C.CFUNCTYPE(None, u8p, u8p, C.c_int, C.c_int)

DEBLOCK_INTER = C.CFUNCTYPE(None, u8p, C.c_int, C.c_int, C.c_int, i8p)
DEBLOCK_INTRA = C.CFUNCTYPE(None, u8p, C.c_int, C.c_int, C.c_int)


class DeblockTable(C.Structure):
    _fields_ = [
        ("deblock_v_luma", DEBLOCK_INTER), ("deblock_h_luma", DEBLOCK_INTER),
        ("deblock_v_chroma", DEBLOCK_INTER), ("deblock_h_chroma", DEBLOCK_INTER),
        ("deblock_v_luma_intra", DEBLOCK_INTRA), ("deblock_h_luma_intra", DEBLOCK_INTRA),
        ("deblock_v_chroma_intra", DEBLOCK_INTRA), ("deblock_h_chroma_intra", DEBLOCK_INTRA),
    ]


PIXEL_W = (16, 16, 8, 8, 8, 4, 4, 4, 2, 2)
PIXEL_H = (16, 8, 16, 8, 4, 8, 4, 2, 4, 2)
FENC_STRIDE = 16
FDEC_STRIDE = 32


class TableSet:
    """All six tables filled by one back-end.

    flavor:
      "hip"    -- this package's HIP library  (x264_*_init_hip)
      "oracle" -- oracle/liboracle.so          (x264o_*_init); tests only
      "ref"    -- oracle/_ref/libx264ref.so    (x264_*_init(cpu=0, ...)); tests only
    """

    def __init__(self, lib, flavor, interlaced=0):
        self.lib, self.flavor, self.interlaced = lib, flavor, interlaced
        self.pixel, self.dct, self.zigzag = PixelTable(), DctTable(), ZigzagTable()
        self.quant, self.mc, self.deblock = QuantTable(), McTable(), DeblockTable()
        self.predict_16x16 = (PREDICT * 7)()
        self.predict_8x8c = (PREDICT * 7)()
        self.predict_4x4 = (PREDICT * 12)()
        self.predict_8x8 = (PREDICT8 * 12)()
        self.predict_8x8_filter = PREDICT8_FILTER()
        R = C.byref
        if flavor == "ref":
            lib.x264_pixel_init(0, R(self.pixel))
            lib.x264_dct_init(0, R(self.dct))
            lib.x264_zigzag_init(0, R(self.zigzag), interlaced)
            lib.x264_quant_init(None, 0, R(self.quant))
            lib.x264_mc_init(0, R(self.mc))
            lib.x264_predict_16x16_init(0, self.predict_16x16)
            lib.x264_predict_8x8c_init(0, self.predict_8x8c)
            lib.x264_predict_4x4_init(0, self.predict_4x4)
            lib.x264_predict_8x8_init(0, self.predict_8x8, R(self.predict_8x8_filter))
            lib.x264_deblock_init(0, R(self.deblock))
        else:
            n = (lambda s: "x264o_%s_init" % s) if flavor == "oracle" else (lambda s: "x264_%s_init_hip" % s)
            getattr(lib, n("pixel"))(R(self.pixel))
            getattr(lib, n("dct"))(R(self.dct))
            getattr(lib, n("zigzag"))(R(self.zigzag), interlaced)
            getattr(lib, n("quant"))(R(self.quant))
            getattr(lib, n("mc"))(R(self.mc))
            getattr(lib, n("predict_16x16"))(self.predict_16x16)
            getattr(lib, n("predict_8x8c"))(self.predict_8x8c)
            getattr(lib, n("predict_4x4"))(self.predict_4x4)
            getattr(lib, n("predict_8x8"))(self.predict_8x8, R(self.predict_8x8_filter))
            getattr(lib, n("deblock"))(R(self.deblock))
