import sys,re
def patch(path, pairs):
    s=open(path).read()
    for a,b in pairs:
        n=s.count(a)
        if n!=1:
            print("MISMATCH",n,path,a[:80]); sys.exit(1)
        s=s.replace(a,b)
    open(path,'w').write(s)

patch('/root/repo/x264_vs2008_amd/slice.py', [
('''               [("progress", C.c_void_p), ("poc", C.c_int), ("n_ref0", C.c_int), ("inv_ref_poc", C.c_int * 8)]
''','''               [("progress", C.c_void_p), ("poc", C.c_int), ("n_ref0", C.c_int), ("inv_ref_poc", C.c_int * 8), ("mvd", C.c_void_p)]


class SliceRd(C.Structure):
    """x264hip_slice_rd: the raster-order variant of the sweep (RD levels, trellis, adaptive quantisation, the entropy coder in the loop)."""
    _fields_ = [("trellis", C.c_int), ("psy_rd", C.c_int), ("write", C.c_int), ("cabac_init_idc", C.c_int), ("i_frame", C.c_int),
                ("qp_min", C.c_int), ("qp_max", C.c_int), ("f_qpm", C.c_float), ("aq_offset", C.c_void_p), ("cost_mv_all", C.c_void_p),
                ("unquant4_mf", C.c_void_p), ("unquant8_mf", C.c_void_p), ("payload", C.c_void_p), ("payload_cap", C.c_int),
                ("payload_len", C.c_void_p), ("mb_bits", C.c_void_p)]


PAYLOAD_LEAD = 64
'''),
('''("noise_reduction", C.c_int), ("nr", C.c_void_p), ("lossless", C.c_int)]''','''("noise_reduction", C.c_int), ("nr", C.c_void_p), ("lossless", C.c_int),
                ("rd", C.c_void_p)]'''),
('''                 chroma_qp_offset=0, keyint=0, mixed_refs=0, noise_reduction=0, mv_range=0):
        self.lib = lib''','''                 chroma_qp_offset=0, keyint=0, mixed_refs=0, noise_reduction=0, mv_range=0,
                 trellis=0, psy_rd=0.0, aq_mode=0, aq_strength=1.0, write=0, cabac_init_idc=0, qp_min=0, qp_max=51, payload_cap=0, raster=None):
        self.lib = lib
        # x264_validate_parameters (R/encoder/encoder.c:493-522): what the RD-side options do to each other
        trellis = min(max(trellis, 0), 2) if cabac else 0
        psy_rd = 0.0 if subme < 6 else min(max(float(psy_rd), 0.0), 10.0)
        self.psy_rd_fix = int(np.float32(psy_rd) * 256 + 0.5)             # FIX8
        if self.psy_rd_fix:
            chroma_qp_offset = min(max(chroma_qp_offset - (1 if psy_rd < 0.25 else 2), -12), 12)
        aq_strength = min(max(float(aq_strength), 0.0), 3.0)
        aq_mode = 0 if aq_strength == 0 else min(max(aq_mode, 0), 1)
        # the raster-order variant of the sweep: needed by the RD levels, trellis, adaptive quantisation, or simply to get the payload
        self.raster = bool(subme >= 6 or trellis or aq_mode or write) if raster is None else bool(raster)
        self.rd_opt = dict(trellis=trellis, aq_mode=aq_mode, aq_strength=aq_strength, write=int(bool(write or subme >= 6 or trellis)),
                           cabac_init_idc=cabac_init_idc, qp_min=qp_min, qp_max=qp_max)'''),
('''        self.cqm = CqmDevice(lib, cqm)
        self.cost = {}''','''        self.cqm = CqmDevice(lib, cqm)
        self.cost = {}
        self.rd_bufs = None
        if self.raster:
            d, B = self.ctx.dims, batch
            n = d.mb_w * d.mb_h
            cap = payload_cap or (n * 800 + 4096 + PAYLOAD_LEAD)
            rb = dict(payload=DeviceArray(lib, (B, cap), np.uint8), payload_len=DeviceArray(lib, (B,), np.int32),
                      mb_bits=DeviceArray(lib, (B, n), np.int32))
            # p_cost_mv of every QP and the unquant tables, built by the library's host C (x264hip_cost_mv_table / _unquant_table)
            tabs = np.zeros((52, 2 * COST_SPAN + 1), np.int16)
            for q in range(52):
                lib.x264hip_cost_mv_table(C.c_int(LAMBDA_TAB[q]), C.c_int(COST_SPAN), tabs[q].ctypes.data_as(C.c_void_p))
            rb["cost_mv_all"] = DeviceArray(lib, tabs.shape, np.int16, tabs)
            q4 = np.ascontiguousarray(cqm["quant4_mf"][:, 6:12, :].astype(np.int32))       # the shift is zero at qp 6..11 (4x4) / 0..5 (8x8)
            q8 = np.ascontiguousarray(cqm["quant8_mf"][:, 0:6, :].astype(np.int32))
            u4, u8 = np.zeros((4, 52, 16), np.int32), np.zeros((2, 52, 64), np.int32)
            lib.x264hip_unquant_table(q4.ctypes.data_as(C.c_void_p), C.c_int(4), C.c_int(16), u4.ctypes.data_as(C.c_void_p))
            lib.x264hip_unquant_table(q8.ctypes.data_as(C.c_void_p), C.c_int(2), C.c_int(64), u8.ctypes.data_as(C.c_void_p))
            rb["unquant4_mf"] = DeviceArray(lib, u4.shape, np.int32, u4)
            rb["unquant8_mf"] = DeviceArray(lib, u8.shape, np.int32, u8)
            if aq_mode:
                rb["aq_energy"] = DeviceArray(lib, (B, n), np.int32)
                rb["aq_offset"] = DeviceArray(lib, (B, n), np.float32)
            self.rd_bufs = rb
            self.payload_cap = cap
        self.i_frame = 0'''),
('''                        noise_reduction=o["noise_reduction"], nr=C.addressof(self.nr) if self.nr else None, lossless=self.lossless)''',
 '''                        noise_reduction=o["noise_reduction"], nr=C.addressof(self.nr) if self.nr else None, lossless=self.lossless)
        if self.raster:
            rb, ro = self.rd_bufs, self.rd_opt
            if ro["aq_mode"]:                  # x264_adaptive_quant_frame on the source (R/encoder/encoder.c:1421)
                L.x264hip_adaptive_quant_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
                c.check(L.x264hip_adaptive_quant_frame(c.h, C.byref(fenc), C.c_float(ro["aq_strength"]), rb["aq_energy"].p, rb["aq_offset"].p), "adaptive_quant_frame")
            self.rd = SliceRd(trellis=ro["trellis"], psy_rd=self.psy_rd_fix, write=ro["write"], cabac_init_idc=ro["cabac_init_idc"], i_frame=self.i_frame,
                              qp_min=ro["qp_min"], qp_max=ro["qp_max"], f_qpm=float(qp), aq_offset=rb["aq_offset"].ptr if ro["aq_mode"] else None,
                              cost_mv_all=rb["cost_mv_all"].ptr, unquant4_mf=rb["unquant4_mf"].ptr, unquant8_mf=rb["unquant8_mf"].ptr,
                              payload=rb["payload"].ptr, payload_cap=self.payload_cap, payload_len=rb["payload_len"].ptr, mb_bits=rb["mb_bits"].ptr)
            p.rd = C.addressof(self.rd)'''),
('''        self.refs.insert(0, (recon, state, 2 * (self.t - self.last_idr)))
        del self.refs[o["n_refs"]:]
        self.t += 1''','''        self.refs.insert(0, (recon, state, 2 * (self.t - self.last_idr)))
        del self.refs[o["n_refs"]:]
        self.t += 1
        self.i_frame += 1

    def payloads(self):
        """slice_data() of the last frame of every chain (valid after ctx.sync()): list of bytes objects."""
        rb = self.rd_bufs
        n = rb["payload_len"].get()
        raw = rb["payload"].get()
        return [bytes(raw[b, PAYLOAD_LEAD:PAYLOAD_LEAD + n[b]]) for b in range(len(n))]'''),
('''        for d in self.cost.values():
            d.free()''','''        for d in self.cost.values():
            d.free()
        for d in (self.rd_bufs or {}).values():
            d.free()'''),
])
print('ok')
