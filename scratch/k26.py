import sys
p='/root/repo/x264_vs2008_amd/slice.py'
s=open(p).read()
def rep(a,b):
    global s
    if s.count(a)!=1: print("MISMATCH",s.count(a),a[:100]); sys.exit(1)
    s=s.replace(a,b)
rep('''def iframe_qp(qp, ip_factor=1.4):''','''def bframe_qp(qp, pb_factor=1.3):
    """rc->qp_constant[SLICE_TYPE_B] (R/encoder/ratecontrol.c:369-372)."""
    return min(max(int(qp + 6.0 * math.log(float(np.float32(pb_factor))) / math.log(2.0) + 0.5), 0), 51)


def coding_order(n_frames, keyint, bframes):
    """[(display index, slice type)] in coding order for a fixed pattern of `bframes` disposable B frames (x264_slicetype_decide
    without b-adapt, then x264_encoder_encode's reordering): an anchor every bframes + 1 frames after an IDR, the last frame before
    the next IDR / the end of the clip is an anchor too, every anchor is coded before the B frames it closes."""
    out, t = [], 0
    while t < n_frames:
        if (t % keyint == 0) if keyint > 0 else t == 0:
            out.append((t, SLICE_I))
            t += 1
            continue
        lim = min((t // keyint + 1) * keyint if keyint > 0 else n_frames, n_frames)
        anchor = min(t + bframes, lim - 1)
        out.append((anchor, SLICE_P))
        out += [(b, SLICE_B) for b in range(t, anchor)]
        t = anchor + 1
    return out


def iframe_qp(qp, ip_factor=1.4):''')
rep('''                 trellis=0, psy_rd=0.0, aq_mode=0, aq_strength=1.0, write=0, cabac_init_idc=0, qp_min=0, qp_max=51, payload_cap=0, raster=None):''',
'''                 trellis=0, psy_rd=0.0, aq_mode=0, aq_strength=1.0, write=0, cabac_init_idc=0, qp_min=0, qp_max=51, payload_cap=0, raster=None,
                 bframes=0, weightb=0, direct_pred=1):''')
rep('''        self.i_frame, self.i_frame_stride = 0, 0      # shard.py sets both when the chains are the GOPs of one stream
        self.fenc = self.ctx.new_picture()
        self.pool = [self.ctx.new_picture() for _ in range(n_refs + 1)]
        self.states = [DeviceState(self.ctx) for _ in range(n_refs + 1)]''','''        self.i_frame, self.i_frame_stride = 0, 0      # shard.py sets both when the chains are the GOPs of one stream
        self.fenc = self.ctx.new_picture()
        # B frames (disposable, one list-1 picture): encode_frame(src, stype, disp) in coding_order(); the DPB then holds
        # max(n_refs, 2) pictures (sps->vui.i_max_dec_frame_buffering, R/encoder/set.c:196-200)
        self.bopt = dict(bframes=bframes, weightb=int(bool(weightb)), direct_spatial=int(direct_pred != 2))
        self.dpb = max(n_refs, 2 if bframes else 1)
        self.pool = [self.ctx.new_picture() for _ in range(self.dpb + 1)]
        self.states = [DeviceState(self.ctx) for _ in range(self.dpb + 1)]''')
rep('''    def encode_frame(self, src=None):
        """The macroblock sweep for the frame held by `src` (default: the picture upload() fills) in every
        batch element.  Returns (slice_type, qp, state) -- the state's arrays are valid after ctx.sync()."""
        L, c, o = self.lib, self.ctx, self.opt
        fenc = self.fenc if src is None else src
        idr = (self.t % o["keyint"] == 0) if o["keyint"] > 0 else self.t == 0
        if idr:
            self.refs, self.last_idr = [], self.t
        used = [r[0] for r in self.refs]
        pic_i = next(i for i, p in enumerate(self.pool) if not any(p is q for q in used))
        recon, state = self.pool[pic_i], self.states[pic_i]
        refs = self.refs[:o["n_refs"]]
        stype = SLICE_I if idr else SLICE_P
        qp = iframe_qp(o["qp"]) if idr else o["qp"]
        poc = 2 * (self.t - self.last_idr)''','''    def encode_frame(self, src=None, stype=None, disp=None):
        """The macroblock sweep for the frame held by `src` (default: the picture upload() fills) in every
        batch element.  Returns (slice_type, qp, state) -- the state's arrays are valid after ctx.sync().
        Without stype: I / P chains in display order (an IDR every keyint frames).  With stype / disp (see coding_order): the
        frame's slice type and display index, frames arriving in coding order -- the way B frames are coded."""
        L, c, o = self.lib, self.ctx, self.opt
        fenc = self.fenc if src is None else src
        if stype is None:
            idr = (self.t % o["keyint"] == 0) if o["keyint"] > 0 else self.t == 0
            stype, disp = (SLICE_I if idr else SLICE_P), self.t
        idr, is_b = stype == SLICE_I, stype == SLICE_B
        if idr:
            self.refs, self.last_idr = [], disp
        poc = 2 * (disp - self.last_idr)
        used = [r[0] for r in self.refs]
        pic_i = next(i for i, p in enumerate(self.pool) if not any(p is q for q in used))
        recon, state = self.pool[pic_i], self.states[pic_i]
        # x264_reference_build_list (R/encoder/encoder.c:911-981): list 0 = earlier pictures, nearest first; list 1 = later ones
        refs = sorted([r for r in self.refs if r[2] < poc], key=lambda r: -r[2])[:o["n_refs"]]
        refs1 = sorted([r for r in self.refs if r[2] > poc], key=lambda r: r[2])[:1] if is_b else []
        qp = iframe_qp(o["qp"]) if idr else bframe_qp(o["qp"]) if is_b else o["qp"]
        self.last_is_b, self.last_poc = is_b, poc''')
rep('''            p.rd = C.addressof(self.rd)
        for i, r in enumerate(refs):''','''            p.rd = C.addressof(self.rd)
        if is_b:
            self.sb = SliceB(fref1=C.addressof(refs1[0][0]), l1_state=C.addressof(refs1[0][1].st), ref1_poc=refs1[0][2],
                             weightb=self.bopt["weightb"], direct_spatial=self.bopt["direct_spatial"])
            p.b = C.addressof(self.sb)
        for i, r in enumerate(refs):''')
rep('''        L, c, o = self.lib, self.ctx, self.opt
        recon, state = self.last
        if o["deblock"]:''','''        L, c, o = self.lib, self.ctx, self.opt
        recon, state = self.last
        if getattr(self, "last_is_b", False):          # a disposable B frame: neither filtered nor kept (R/encoder/encoder.c:986-1024,1060-1068)
            self.t += 1
            self.i_frame += 1
            return
        if o["deblock"]:''')
rep('''        self.refs.insert(0, (recon, state, 2 * (self.t - self.last_idr)))
        del self.refs[o["n_refs"]:]''','''        self.refs.insert(0, (recon, state, getattr(self, "last_poc", 2 * (self.t - self.last_idr))))
        del self.refs[self.dpb:]''')
open(p,'w').write(s)
print('ok')
