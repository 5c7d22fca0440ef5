// slice_b_flow.h -- the B-slice macroblock of the raster sweep: textually included inside k_slice_sweep's macroblock loop (the BS
// instantiation), where the I / P flow stands for the other slice types.  It is the B branch of x264_macroblock_analyse
// (R/encoder/analyse.c:2467-2733) with the RD mode decision (subme 7: a->i_mbrd = 1 in a B slice) -- direct prediction
// (spatial, R/common/macroblock.c:226-309), x264_mb_analyse_inter_direct / _b16x16 / _b8x8 / _b16x8 / _b8x16 (:1521-1933),
// x264_mb_analyse_b_rd (:2007-2076), x264_refine_bidir (:2078-2107) with x264_me_refine_bidir_satd (R/encoder/me.c:790-928),
// x264_analyse_update_cache's B cases (:2848-2914) -- followed by the same single call of the encoder as the I / P flow.
// oracle/b_oracle.c is the CPU statement of the same steps (pinned against the reference); comments there are not repeated here.
// The motion caches (h->mb.cache.ref / mv of both lists, x264_scan8 layout) ARE the LDS arrays the entropy coder reads:
// sr.cref / sr.cmv (list 0), sb.cref1 / sb.cmv1 (list 1).
{
    const size_t cb4 = 4 * cb, cb32 = 32 * cb;
    auto CREF = [&](int l, int k) -> int { return UNI(l ? sb.cref1[k] : sr.cref[k]); };
    auto CMVX = [&](int l, int k) -> int { return UNI(l ? sb.cmv1[k][0] : sr.cmv[k][0]); };
    auto CMVY = [&](int l, int k) -> int { return UNI(l ? sb.cmv1[k][1] : sr.cmv[k][1]); };
    // x264_macroblock_cache_ref / _mv / _mvd / _skip on a w x h run of 4x4 blocks at (x, y): every lane looks after its own entry
    auto cache_set_b = [&](int l, int x, int y, int w, int h, int r, int vx, int vy, int set_ref, int set_mv) {
        const int k = lane - 12, i = k & 7, j = k >> 3;
        if (k >= 0 && i < 4 && j < 4 && i >= x && i < x + w && j >= y && j < y + h) {
            if (l) { if (set_ref) sb.cref1[lane] = (signed char)r; if (set_mv) { sb.cmv1[lane][0] = (i16)vx; sb.cmv1[lane][1] = (i16)vy; } }
            else { if (set_ref) sr.cref[lane] = (signed char)r; if (set_mv) { sr.cmv[lane][0] = (i16)vx; sr.cmv[lane][1] = (i16)vy; } }
        }
    };
    auto cache_mvd0_b = [&](int l, int x, int y, int w, int h) {
        const int k = lane - 12, i = k & 7, j = k >> 3;
        if (k >= 0 && i < 4 && j < 4 && i >= x && i < x + w && j >= y && j < y + h) {
            if (l) { sb.cmvd1[lane][0] = 0; sb.cmvd1[lane][1] = 0; } else { sr.cmvd[lane][0] = 0; sr.cmvd[lane][1] = 0; }
        }
    };
    auto cache_skip_b = [&](int x, int y, int w, int h, int v) {
        const int k = lane - 12, i = k & 7, j = k >> 3;
        if (k >= 0 && i < 4 && j < 4 && i >= x && i < x + w && j >= y && j < y + h) sb.cskip[lane] = (signed char)v;
    };
    // ---- x264_macroblock_cache_load: the neighbours of both lists, list 1's mvd, the skip flags of direct blocks ----
    if (lane < 48) {
        sr.cref[lane] = -2; sr.cmv[lane][0] = 0; sr.cmv[lane][1] = 0; sb.cref1[lane] = -2; sb.cmv1[lane][0] = 0; sb.cmv1[lane][1] = 0;
        sb.cskip[lane] = 0; sb.cmvd1[lane][0] = 0; sb.cmvd1[lane][1] = 0;
    }
    if (TD && lane == 30) {         // ... except the entry of block 12, which the reference's cache keeps from the previous macroblock (SwRd::stale)
        sr.cref[30] = (signed char)sb.stale[0]; sr.cmv[30][0] = sb.stale[1]; sr.cmv[30][1] = sb.stale[2];
        sb.cref1[30] = (signed char)sb.stale[3]; sb.cmv1[30][0] = sb.stale[4]; sb.cmv1[30][1] = sb.stale[5];
    }
    WAVE_SYNC();
    {
        const signed char *r0 = a.ref, *r1 = rd.ref1 + cb4;
        const i16 *m0 = a.mv, *m1 = rd.mv1 + cb32;
        int o = -1, blk = 0;
        if ((nb & NB_TOP) && lane >= 4 && lane < 8) { o = mb - a.mb_w; blk = 12 + lane - 4; }
        if ((nb & NB_TOPLEFT) && lane == 3) { o = mb - a.mb_w - 1; blk = 15; }
        if ((nb & NB_TOPRIGHT) && lane == 8) { o = mb - a.mb_w + 1; blk = 12; }
        if (o >= 0) {
            const int b8 = (blk >> 3) * 2 + ((blk & 3) >> 1);
            sr.cref[lane] = r0[o * 4 + b8]; sr.cmv[lane][0] = m0[(o * 16 + blk) * 2]; sr.cmv[lane][1] = m0[(o * 16 + blk) * 2 + 1];
            sb.cref1[lane] = r1[o * 4 + b8]; sb.cmv1[lane][0] = m1[(o * 16 + blk) * 2]; sb.cmv1[lane][1] = m1[(o * 16 + blk) * 2 + 1];
        }
        if ((nb & NB_LEFT) && lane >= 11 && lane < 36 && ((lane - 11) & 7) == 0) {
            const int i = (lane - 11) >> 3;
            sr.cref[lane] = s.left_r8[i >> 1]; sr.cmv[lane][0] = s.left_mv4[i][0]; sr.cmv[lane][1] = s.left_mv4[i][1];
            sb.cref1[lane] = sb.left_r8_1[i >> 1]; sb.cmv1[lane][0] = sb.left_mv4_1[i][0]; sb.cmv1[lane][1] = sb.left_mv4_1[i][1];
            sb.cmvd1[lane][0] = sb.left_mvd1[i][0]; sb.cmvd1[lane][1] = sb.left_mvd1[i][1];
        }
        if ((nb & NB_TOP) && lane >= 40 && lane < 44) {
            const i16 *mvd = rd.mvd1 + ((cb + (mb - a.mb_w)) * 16 + 12 + (lane - 40)) * 2;
            sb.cmvd1[4 + lane - 40][0] = mvd[0]; sb.cmvd1[4 + lane - 40][1] = mvd[1];
        }
        if (lane == 44) {
            const int st = (nb & NB_TOP) ? (rd.skipbp + cb)[mb - a.mb_w] : 0, sl = (nb & NB_LEFT) ? sb.left_skipbp : 0;
            sb.cskip[4] = (signed char)(st & 4); sb.cskip[6] = (signed char)(st & 8); sb.cskip[11] = (signed char)(sl & 2); sb.cskip[27] = (signed char)(sl & 8);
        }
    }
    WAVE_SYNC();

    // x264_mb_predict_mv_16x16 from the cache (R/common/macroblock.c:90-128)
    auto predict16_b = [&](int l, int i_ref, int &px, int &py) {
        int ra = CREF(l, 11), rb = CREF(l, 4), rc = CREF(l, 8), kc = 8;
        if (rc == -2) { kc = 3; rc = CREF(l, 3); }
        const int ax = CMVX(l, 11), ay = CMVY(l, 11), bx = CMVX(l, 4), byv = CMVY(l, 4), cx = CMVX(l, kc), cy = CMVY(l, kc);
        const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
        if (cnt > 1) { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
        else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx; py = byv; } else { px = cx; py = cy; } }
        else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
        else { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
    };
    // x264_mb_predict_mv (:28-88) from the cache; cur_part = h->mb.i_partition
    auto predict_blk_b = [&](int l, int cur_part, int idx, int width, int &px, int &py) {
        const int i8 = sw_scan8(idx), i_ref = CREF(l, i8);
        int ra = CREF(l, i8 - 1), rb = CREF(l, i8 - 8), kc = i8 - 8 + width, rc = CREF(l, kc);
        if ((idx & 3) == 3 || (width == 2 && (idx & 3) == 2) || rc == -2) { kc = i8 - 8 - 1; rc = CREF(l, kc); }
        const int ax = CMVX(l, i8 - 1), ay = CMVY(l, i8 - 1), bx = CMVX(l, i8 - 8), byv = CMVY(l, i8 - 8), cx = CMVX(l, kc), cy = CMVY(l, kc);
        if (cur_part == 14) {
            if (idx == 0 && rb == i_ref) { px = bx; py = byv; return; }
            if (idx != 0 && ra == i_ref) { px = ax; py = ay; return; }
        } else if (cur_part == 15) {
            if (idx == 0 && ra == i_ref) { px = ax; py = ay; return; }
            if (idx != 0 && rc == i_ref) { px = cx; py = cy; return; }
        }
        const int cnt = (ra == i_ref) + (rb == i_ref) + (rc == i_ref);
        if (cnt > 1) { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
        else if (cnt == 1) { if (ra == i_ref) { px = ax; py = ay; } else if (rb == i_ref) { px = bx; py = byv; } else { px = cx; py = cy; } }
        else if (rb == -2 && rc == -2 && ra != -2) { px = ax; py = ay; }
        else { px = sw_median(ax, bx, cx); py = sw_median(ay, byv, cy); }
    };
    // quarter-sample luma prediction of four pixels of row r at x (macroblock coordinates) from list l's picture ri (mc_luma / get_ref)
    auto luma4 = [&](int l, int ri, int vx, int vy, int r, int x, int out[4]) {
        const int qx = vx & 3, qy = vy & 3, idx = qy * 4 + qx;
        const ptrdiff_t base = oy + (ptrdiff_t)((vy >> 2) + r) * a.sy + (vx >> 2) + x + (ptrdiff_t)by_;
        const u8 *pa = (l ? refs.y1[c_qpel_a[idx]] : refs.y[ri][c_qpel_a[idx]]) + base + (qy == 3) * a.sy;
        const u8 *pb = (l ? refs.y1[c_qpel_b[idx]] : refs.y[ri][c_qpel_b[idx]]) + base + (qx == 3);
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = (idx & 5) ? (((int)pa[i] + (int)pb[i] + 1) >> 1) : (int)pa[i];
    };
    auto chroma1 = [&](int l, int ri, int pl, int vx, int vy, int cx, int cy) -> int {
        const int dx = vx & 7, dyy = vy & 7;
        const int ca = (8 - dx) * (8 - dyy), cbv = dx * (8 - dyy), cc = (8 - dx) * dyy, cd = dx * dyy;
        const ptrdiff_t cbase = oc + (ptrdiff_t)((vy >> 3) + cy) * a.sc + (vx >> 3) + cx + (ptrdiff_t)bc_;
        const u8 *p = (l ? (pl ? refs.v1 : refs.u1) : (pl ? refs.v[ri] : refs.u[ri])) + cbase;
        return (ca * p[0] + cbv * p[1] + cc * p[a.sc] + cd * p[a.sc + 1] + 32) >> 6;
    };
    // x264_mb_mc for the B types: every pixel from the final vectors / references of its 4x4 / 8x8 block (s.mv4, s.ref8 | sb.mv4_1, sb.ref8_1)
    auto mc_b = [&]() {
        const int mnx = 4 * (-16 * mbx - 24), mxx = 4 * (16 * (a.mb_w - mbx - 1) + 24), mny = 4 * (-16 * mby - 24), mxy = 4 * (16 * (a.mb_h - mby - 1) + 24);
        {
            const int r = lane >> 2, x = (lane & 3) * 4, blk = (r >> 2) * 4 + (x >> 2), b8 = (r >> 3) * 2 + (x >> 3);
            const int r0 = s.ref8[b8], r1 = sb.ref8_1[b8];
            int p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
            if (r0 >= 0) luma4(0, r0, clip3((int)s.mv4[blk][0], mnx, mxx), clip3((int)s.mv4[blk][1], mny, mxy), r, x, p0);
            if (r1 >= 0) luma4(1, r1, clip3((int)sb.mv4_1[blk][0], mnx, mxx), clip3((int)sb.mv4_1[blk][1], mny, mxy), r, x, p1);
            const int w = r0 >= 0 ? refs.biw[r0] : 32;
#pragma unroll
            for (int i = 0; i < 4; i++)
                s.fd[FDY + r * FD + x + i] = (u8)(r0 >= 0 && r1 >= 0 ? clip_u8((p0[i] * w + p1[i] * (64 - w) + 32) >> 6) : r0 >= 0 ? p0[i] : p1[i]);
        }
        {
            const int cx = lane & 7, cy = lane >> 3, blk = (cy >> 1) * 4 + (cx >> 1), b8 = (cy >> 2) * 2 + (cx >> 2);
            const int r0 = s.ref8[b8], r1 = sb.ref8_1[b8], w = r0 >= 0 ? refs.biw[r0] : 32;
            const int vx0 = clip3((int)s.mv4[blk][0], mnx, mxx), vy0 = clip3((int)s.mv4[blk][1], mny, mxy);
            const int vx1 = clip3((int)sb.mv4_1[blk][0], mnx, mxx), vy1 = clip3((int)sb.mv4_1[blk][1], mny, mxy);
#pragma unroll
            for (int pl = 0; pl < 2; pl++) {
                const int q0 = r0 >= 0 ? chroma1(0, r0, pl, vx0, vy0, cx, cy) : 0, q1 = r1 >= 0 ? chroma1(1, r1, pl, vx1, vy1, cx, cy) : 0;
                s.fd[(pl ? FDV : FDU) + cy * FD + cx] = (u8)(r0 >= 0 && r1 >= 0 ? clip_u8((q0 * w + q1 * (64 - w) + 32) >> 6) : r0 >= 0 ? q0 : q1);
            }
        }
        WAVE_SYNC();
    };
    // SATD (mbcmp) of a region of the source against a 16-wide buffer with the macroblock's geometry (p: row stride st)
    auto satd_region = [&](const u8 *p, int st, int bx, int by, int w, int h) -> int {
        int v = 0;
        if (lane < 32) {
            const int blk = lane >> 2, r = lane & 3, x8 = (blk & 1) * 8, y = (blk >> 1) * 4 + r;
            if (x8 >= bx && x8 < bx + w && y >= by && y < by + h) {
                v = sw_satd_row8(s.fe + y * 16 + x8, p + y * st + x8, lane);
                if (r) v = 0;
            }
        }
        return wave_sum(v);
    };
    // the weighted average of the two lists' predictions of a region into the temporary (16-wide, macroblock geometry): what
    // h->mc.avg of two get_ref results gives
    u8 *tmp = s.patch;
    auto bi_to_tmp = [&](int bx, int by, int w, int h, int r0, int vx0, int vy0, int vx1, int vy1) {
        const int r = lane >> 2, x = (lane & 3) * 4;
        if (x >= bx && x < bx + w && r >= by && r < by + h) {
            int p0[4], p1[4];
            // get_ref takes the block's own plane pointers: the vector is relative to the block, like everywhere
            luma4(0, r0, vx0, vy0, r, x, p0); luma4(1, 0, vx1, vy1, r, x, p1);
            const int wgt = refs.biw[r0];
#pragma unroll
            for (int i = 0; i < 4; i++) tmp[r * 16 + x + i] = (u8)clip_u8((p0[i] * wgt + p1[i] * (64 - wgt) + 32) >> 6);
        }
        WAVE_SYNC();
    };
    // the analysis records (x264_me_t's mv / cost / cost_mv / mvp of a->l0 / a->l1): slot 0 me16x16, 1-4 me8x8, 5-6 me16x8, 7-8 me8x16
    auto ME = [&](int l, int slot, int f) -> int { return UNI(sb.me[l][slot][f]); };
    auto me_put = [&](int l, int slot, int vx, int vy, int cost, int cost_mv, int px, int py) {
        if (lane < 6) sb.me[l][slot][lane] = lane == 0 ? vx : lane == 1 ? vy : lane == 2 ? cost : lane == 3 ? cost_mv : lane == 4 ? px : py;
        WAVE_SYNC();
    };
    const MeLimits L = me_limits(mbx, mby, a.mb_w, a.mb_h, a.mv_range);
    MxCtx c;
    c.fe = (MX_LDS(u32))s.fe; c.fe_u = (MX_LDS(u8))(s.fe + 256); c.fe_v = (MX_LDS(u8))(s.fe + 320); c.sy = a.sy; c.sc = a.sc; c.lane = lane; c.set_block(16, 16, 0, 0);
    c.cost_g = (MX_GLB(i16))cost_g; c.cost_l = (MX_LDS(i16))s.costl; c.has_cost_l = true; c.patch = (MX_LDS(u8))s.patch; c.has_patch = true; c.patch_on = false;
    MeOpts mo_b = mo;
    mo_b.chroma_me = 0;                                   // h->mb.b_chroma_me is for P slices (analyse.c:234)
    auto aim_b = [&](int l, int r, int w, int h, int bx, int by) {
#pragma unroll
        for (int k = 0; k < 4; k++) c.pl[k] = (MX_GLB(u8))((l ? refs.y1[k] : refs.y[r][k]) + by_ + oy + (ptrdiff_t)by * a.sy + bx);
        c.cu = (MX_GLB(u8))((l ? refs.u1 : refs.u[r]) + bc_ + oc + (ptrdiff_t)(by >> 1) * a.sc + (bx >> 1));
        c.cv = (MX_GLB(u8))((l ? refs.v1 : refs.v[r]) + bc_ + oc + (ptrdiff_t)(by >> 1) * a.sc + (bx >> 1));
        c.set_block(w, h, bx, by);
    };
    auto REFC = [&](int l, int r) -> int { return l ? 0 : Q.lambda * refs.ref_bits[r]; };     // one list-1 picture: no bits
    int l_ref0 = 0, l_ref1 = 0;                           // a->l0.i_ref / a->l1.i_ref
#define LREF(l_) ((l_) ? l_ref1 : l_ref0)
    int cost16bi = MX_COST_MAX, cost16direct = MX_COST_MAX, cost8bi = MX_COST_MAX, cost16x8bi = MX_COST_MAX, cost8x16bi = MX_COST_MAX;
    int part16x8_0 = 3, part16x8_1 = 3, part8x16_0 = 3, part8x16_1 = 3, type16x8 = T_B_L0_L0, type8x16 = T_B_L0_L0;
    int rd_direct = MX_COST_MAX, rd_l0 = MX_COST_MAX, rd_l1 = MX_COST_MAX, rd_bi = MX_COST_MAX, rd_8 = MX_COST_MAX, rd_168 = MX_COST_MAX, rd_816 = MX_COST_MAX;
    auto SUB = [&](int i) -> int { return UNI(sb.sub[i]); };
    auto sub_uses = [&](int sub, int l) -> bool { return sub == 12 ? false : l ? (sub >= 4 && sub <= 11) : (sub <= 3 || (sub >= 8 && sub <= 11)); };

    // ---- x264_mb_load_mv_direct8x8 / CACHE_MV_BI / x264_analyse_update_cache (B types), then the final vectors the motion compensation reads ----
    auto load_direct8x8 = [&](int idx) {
        const int k = lane - 12, i = k & 7, j = k >> 3;
        if (k >= 0 && i < 4 && j < 4 && (j >> 1) * 2 + (i >> 1) == idx) {
            sr.cref[lane] = sb.dref[0][idx]; sb.cref1[lane] = sb.dref[1][idx];
            sr.cmv[lane][0] = sb.dmv[0][j * 4 + i][0]; sr.cmv[lane][1] = sb.dmv[0][j * 4 + i][1];
            sb.cmv1[lane][0] = sb.dmv[1][j * 4 + i][0]; sb.cmv1[lane][1] = sb.dmv[1][j * 4 + i][1];
        }
    };
    auto cache_mv_bi = [&](int x, int y, int w, int h, int slot, int ptype, int b_mvd) {
        for (int l = 0; l < 2; l++) {
            if (sub_uses(ptype, l)) cache_set_b(l, x, y, w, h, LREF(l), ME(l, slot, 0), ME(l, slot, 1), 1, 1);
            else { cache_set_b(l, x, y, w, h, -1, 0, 0, 1, 1); if (b_mvd) cache_mvd0_b(l, x, y, w, h); }
        }
    };
    auto cache_mv_b8x8 = [&](int i, int b_mvd) {
        const int x = 2 * (i & 1), y = 2 * (i >> 1), st = SUB(i);
        if (st == 12) {
            load_direct8x8(i);
            if (b_mvd) { cache_mvd0_b(0, x, y, 2, 2); cache_mvd0_b(1, x, y, 2, 2); cache_skip_b(x, y, 2, 2, 1); }
        } else
            cache_mv_bi(x, y, 2, 2, 1 + i, st, b_mvd);
    };
    auto finals_from_cache = [&]() {
        WAVE_SYNC();
        if (lane < 16) {
            const int k = 12 + (lane & 3) + 8 * (lane >> 2);
            s.mv4[lane][0] = sr.cmv[k][0]; s.mv4[lane][1] = sr.cmv[k][1]; sb.mv4_1[lane][0] = sb.cmv1[k][0]; sb.mv4_1[lane][1] = sb.cmv1[k][1];
        }
        if (lane < 4) { const int k = 12 + 2 * (lane & 1) + 16 * (lane >> 1); s.ref8[lane] = sr.cref[k]; sb.ref8_1[lane] = sb.cref1[k]; }
        WAVE_SYNC();
    };
    auto update_cache_b = [&]() {
        if (type == T_B_SKIP || type == T_B_DIRECT) { for (int i = 0; i < 4; i++) load_direct8x8(i); }
        else if (type == T_B_8x8) { for (int i = 0; i < 4; i++) cache_mv_b8x8(i, 1); }
        else if (part == 16) cache_mv_bi(0, 0, 4, 4, 0, type == T_B_L0_L0 ? 3 : type == T_B_L1_L1 ? 7 : 11, 1);
        else if (part == 14) { cache_mv_bi(0, 0, 4, 2, 5, part16x8_0, 1); cache_mv_bi(0, 2, 4, 2, 6, part16x8_1, 1); }
        else { cache_mv_bi(0, 0, 2, 4, 7, part8x16_0, 1); cache_mv_bi(2, 0, 2, 4, 8, part8x16_1, 1); }
        finals_from_cache();
    };

    // ---- x264_me_refine_bidir_satd (R/encoder/me.c:790-928) on one bi-predicted block ----
    u8 *visited = (u8 *)s.t8;                             // x264_me_refine_bidir's visited[8][8][8]: the 8x8 transform's scratch is idle while analysing
    auto refine_bidir_satd = [&](int slot, int bx, int by, int w, int h) {
        const int r0 = l_ref0;
        int bm0x = ME(0, slot, 0), bm0y = ME(0, slot, 1), bm1x = ME(1, slot, 0), bm1y = ME(1, slot, 1);
        if (bm0y > L.smax1 - 8 || bm1y > L.smax1 - 8) return;
        // all four cost tables are centred on predictors clipped to the HORIZONTAL range, as the reference has it
        const int c0x = clip3(ME(0, slot, 4), L.smin0, L.smax0), c0y = clip3(ME(0, slot, 5), L.smin0, L.smax0);
        const int c1x = clip3(ME(1, slot, 4), L.smin0, L.smax0), c1y = clip3(ME(1, slot, 5), L.smin0, L.smax0);
        for (int k = lane; k < 512; k += 64) visited[k] = 0;
        WAVE_SYNC();
        int om0x = bm0x, om0y = bm0y, om1x = bm1x, om1y = bm1y, bcost = MX_COST_MAX;
#pragma nounroll
        for (int pass = -1; pass < 8; pass++) {             // pass -1: the starting point alone (CHECK_BIDIR(0,0,0,0) ahead of the loop)
            const int ncand = pass < 0 ? 1 : 32;
#pragma nounroll
            for (int k = 0; k < ncand; k++) {
                const u32 d = pass < 0 ? 0x5555u : c_bidir_dirs[k];        // four 2-bit fields, value - 1 = the offset
                const int x0 = om0x + (int)(d & 3) - 1, y0 = om0y + (int)((d >> 2) & 3) - 1, x1 = om1x + (int)((d >> 4) & 3) - 1, y1 = om1y + (int)((d >> 6) & 3) - 1;
                const int vi = ((x0 & 7) * 8 + (y0 & 7)) * 8 + (x1 & 7), vb = 1 << (y1 & 7);
                const int seen = UNI(visited[vi]);
                if (pass > 0 && (seen & vb)) continue;
                if (lane == 0) visited[vi] = (u8)(seen | vb);
                bi_to_tmp(bx, by, w, h, r0, x0, y0, x1, y1);
                const int cost = satd_region(tmp, 16, bx, by, w, h) + UNI(cost_g[x0 - c0x]) + UNI(cost_g[y0 - c0y]) + UNI(cost_g[x1 - c1x]) + UNI(cost_g[y1 - c1y]);
                if (cost < bcost) { bcost = cost; bm0x = x0; bm0y = y0; bm1x = x1; bm1y = y1; }
            }
            if (pass >= 0) {
                if (om0x == bm0x && om0y == bm0y && om1x == bm1x && om1y == bm1y) break;
                om0x = bm0x; om0y = bm0y; om1x = bm1x; om1y = bm1y;
            }
        }
        WAVE_SYNC();
        if (lane < 2) { sb.me[0][slot][lane] = lane ? bm0y : bm0x; sb.me[1][slot][lane] = lane ? bm1y : bm1x; }
        WAVE_SYNC();
    };

    // ================================================================== the macroblock ====
    enum { BS_PRE, BS_CAND, BS_AFTER_EARLY, BS_AN2, BS_SELECT, BS_T8, BS_I16, BS_I4, BS_I8, BS_FINAL };
    int bstep = BS_PRE, kcand = 0, pass = 0, bthresh = 0, bskip_cost = MX_COST_MAX;
    bool direct_ok = true;          // x264_mb_predict_mv_direct16x16's return value: temporal prediction fails when a co-located reference is not in list 0
    int i_type_b = T_B_L0_L0, i_part_b = 16, i_cost_b = MX_COST_MAX, i_satd_inter_b = 0;
#pragma nounroll
    for (;;) {
        bool fin = false;
        if (bstep == BS_PRE) {
            cache_fenc_satd();
            type = T_B_SKIP;
            bool frame_temporal = false, dauto = false;
            if constexpr (TD) { frame_temporal = rd.direct_temporal != 0; dauto = rd.direct_score != nullptr; }
            bool b_skip = false, prev_ok = false;
            // --direct auto (h->mb.b_direct_auto_write, R/encoder/analyse.c:2476-2496): BOTH direct modes are predicted in every macroblock -- the other one
            // first, then the frame's -- and each is credited with the skip it would give; the frame's stays in the caches and in fdec
            for (int it = dauto ? 0 : 1; it < 2; it++) {
            const bool temporal = it == 0 ? !frame_temporal : frame_temporal;
            direct_ok = true;
            if (temporal) {
                // ---- x264_mb_predict_mv_direct16x16, temporal (R/common/macroblock.c:155-224; direct_8x8_inference: the corner blocks) ----
                const int type_col = UNI((rd.col_type + cb)[mb]);
                cache_set_b(1, 0, 0, 4, 4, 0, 0, 0, 1, 0);
                if (IS_INTRA_T(type_col)) { cache_set_b(0, 0, 0, 4, 4, 0, 0, 0, 1, 1); cache_set_b(1, 0, 0, 4, 4, 0, 0, 0, 0, 1); }
                else
                    for (int i8 = 0; i8 < 4; i8++) {
                        const int x8 = i8 & 1, y8 = i8 >> 1, cr = UNI((rd.col_ref + cb4)[mb * 4 + i8]);
                        const int i_ref = cr < 0 ? cr : refs.map_col[cr];
                        if (i_ref < 0) { direct_ok = false; break; }           // (what was written so far stays, as in the reference)
                        const i16 *mvcol = rd.col_mv + cb32 + ((size_t)mb * 16 + 3 * x8 + 12 * y8) * 2;
                        const int cx0 = UNI(mvcol[0]), cy0 = UNI(mvcol[1]), dsf = refs.dsf[i_ref];
                        const int l0x = (dsf * cx0 + 128) >> 8, l0y = (dsf * cy0 + 128) >> 8;
                        cache_set_b(0, 2 * x8, 2 * y8, 2, 2, i_ref, (i16)l0x, (i16)l0y, 1, 1);
                        cache_set_b(1, 2 * x8, 2 * y8, 2, 2, 0, (i16)(l0x - cx0), (i16)(l0y - cy0), 0, 1);
                    }
            } else {
            // ---- x264_mb_predict_mv_direct16x16, spatial (R/common/macroblock.c:226-309) ----
            auto dref_of = [&](int l) -> int {
                const int ra = CREF(l, 11), rb = CREF(l, 4);
                int rc = CREF(l, 8);
                if (rc == -2) rc = CREF(l, 3);
                int r = ra;
                if (r < 0 || (rb < r && rb >= 0)) r = rb;
                if (r < 0 || (rc < r && rc >= 0)) r = rc;
                return r < 0 ? -1 : r;
            };
            const int dr0 = dref_of(0), dr1 = dref_of(1);
            int dvx0 = 0, dvy0 = 0, dvx1 = 0, dvy1 = 0;
            if (dr0 < 0 && dr1 < 0) { cache_set_b(0, 0, 0, 4, 4, 0, 0, 0, 1, 1); cache_set_b(1, 0, 0, 4, 4, 0, 0, 0, 1, 1); }
            else {
                if (dr0 >= 0) predict16_b(0, dr0, dvx0, dvy0);
                if (dr1 >= 0) predict16_b(1, dr1, dvx1, dvy1);
                cache_set_b(0, 0, 0, 4, 4, dr0, dvx0, dvy0, 1, 1); cache_set_b(1, 0, 0, 4, 4, dr1, dvx1, dvy1, 1, 1);
                const int type_col = UNI((rd.col_type + cb)[mb]);
                if (!(IS_INTRA_T(type_col) || (dr0 && dr1)))
                    for (int i8 = 0; i8 < 4; i8++) {            // col_zero_flag
                        const int x8 = i8 & 1, y8 = i8 >> 1;
                        if (UNI((rd.col_ref + cb4)[mb * 4 + i8]) == 0) {
                            const i16 *mvcol = rd.col_mv + cb32 + ((size_t)mb * 16 + 3 * x8 + 12 * y8) * 2;
                            const int cx0 = UNI(mvcol[0]), cy0 = UNI(mvcol[1]);
                            if (iabs(cx0) <= 1 && iabs(cy0) <= 1) {
                                if (dr0 == 0) cache_set_b(0, 2 * x8, 2 * y8, 2, 2, 0, 0, 0, 0, 1);
                                if (dr1 == 0) cache_set_b(1, 2 * x8, 2 * y8, 2, 2, 0, 0, 0, 0, 1);
                            }
                        }
                    }
            }
            }
            WAVE_SYNC();
            // x264_mb_predict_mv_direct16x16's b_changed (R/common/macroblock.c:323-343): asked for the second prediction when the first was available
            bool changed = true;
            if (dauto && it == 1 && prev_ok && direct_ok) {
                const int type_col = UNI((rd.col_type + cb)[mb]);
                int diff = 0;
                if (IS_INTRA_T(type_col) || type_col == T_P_SKIP) {
                    if (lane == 0) diff = sb.dref[0][0] != sr.cref[12] || sb.dref[1][0] != sb.cref1[12] || sb.dmv[0][0][0] != sr.cmv[12][0] || sb.dmv[0][0][1] != sr.cmv[12][1] ||
                                          sb.dmv[1][0][0] != sb.cmv1[12][0] || sb.dmv[1][0][1] != sb.cmv1[12][1];
                } else {
                    if (lane < 16) {
                        const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                        diff = sb.dmv[0][lane][0] != sr.cmv[k][0] || sb.dmv[0][lane][1] != sr.cmv[k][1] || sb.dmv[1][lane][0] != sb.cmv1[k][0] || sb.dmv[1][lane][1] != sb.cmv1[k][1];
                    }
                    if (lane < 4) { const int k = 12 + 2 * (lane & 1) + 16 * (lane >> 1); diff |= sb.dref[0][lane] != sr.cref[k] || sb.dref[1][lane] != sb.cref1[k]; }
                }
                changed = __builtin_amdgcn_ballot_w64(diff != 0) != 0;
                WAVE_SYNC();
            }
            if (direct_ok && changed && lane < 16) {
                const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                sb.dmv[0][lane][0] = sr.cmv[k][0]; sb.dmv[0][lane][1] = sr.cmv[k][1]; sb.dmv[1][lane][0] = sb.cmv1[k][0]; sb.dmv[1][lane][1] = sb.cmv1[k][1];
            }
            if (direct_ok && changed && lane < 4) { const int k = 12 + 2 * (lane & 1) + 16 * (lane >> 1); sb.dref[0][lane] = sr.cref[k]; sb.dref[1][lane] = sb.cref1[k]; }
            if (dauto) {
                if (direct_ok) {
                    if (changed) {
                        WAVE_SYNC();
                        finals_from_cache();
                        mc_b();
                        b_skip = sw_probe_pskip(s, refs, a, Q, 0, 0, mbx, mby, oy, oc, by_, bc_, lane, true) != 0;
                    }
                    if (temporal) dscore0 += b_skip; else dscore1 += b_skip;
                } else
                    b_skip = false;
                prev_ok = direct_ok;
            }
            }                                       // (the two predictions of --direct auto)
            if (direct_ok) {
                if (!dauto) { finals_from_cache(); mc_b(); }
                if (mbrd) { bskip_cost = ssd_mb(); b_skip = bskip_cost <= ((6 * Q.lambda2 + 128) >> 8); }      // "6 = minimum cavlc cost of a non-skipped MB"
                else if (!dauto) b_skip = sw_probe_pskip(s, refs, a, Q, 0, 0, mbx, mby, oy, oc, by_, bc_, lane, true) != 0;   // x264_macroblock_probe_bskip
            }
            if (b_skip) { skip_mc = 1; fin = true; }
            else {
                skip_mc = 0;
                // ---- x264_mb_analyse_inter_direct: the direct prediction is in fdec ----
                if (direct_ok) {
                    cost16direct = Q.lambda * 1;
                    for (int i = 0; i < 4; i++) {
                        const int c8d = satd_region(s.fd + FDY, FD, 8 * (i & 1), 8 * (i >> 1), 8, 8);
                        cost16direct += c8d;
                        if (lane == 0) sb.cost8direct[i] = c8d + Q.lambda * 1;
                    }
                } else if (lane < 4) sb.cost8direct[lane] = MX_COST_MAX;
                // ---- x264_mb_analyse_inter_b16x16 ----
                for (int l = 0; l < 2; l++) {
                    const int n = l ? 1 : a.n_refs;
                    int thresh = 0x7fffffff, best = 0x7fffffff;
                    for (int r = 0; r < n; r++) {
                        int px, py, n_mvc = 0;
                        predict16_b(l, r, px, py);
                        {
                            const i16 *mvr = l ? rd.mvr1 + 2 * cb : a.mvr + (size_t)r * nmb * 2;
                            const int top = mb - a.mb_w;
                            WAVE_SYNC();
#define SETC(vx_, vy_) do { s.mvc[n_mvc][0] = (i16)(vx_); s.mvc[n_mvc][1] = (i16)(vy_); n_mvc++; } while (0)
                            if (CREF(l, 30) == r) SETC(CMVX(l, 30), CMVY(l, 30));                  // b_direct
                            if (TD && r == 0 && (l ? a.lowres1 : a.lowres0)) {     // the lookahead's vector, twice (0x7fff in the chain's first component: none)
                                const i16 *lw = (l ? a.lowres1 : a.lowres0) + 2 * cb;
                                if (UNI(lw[0]) != 0x7fff) SETC((u16)(UNI(lw[2 * mb]) << 1), (u16)(UNI(lw[2 * mb + 1]) << 1));
                            }
                            if ((nb & NB_LEFT) && !IS_SKIP_T(left_type)) { if (l) SETC(sb.left_mvr1[0], sb.left_mvr1[1]); else SETC(s.left_mvr[r][0], s.left_mvr[r][1]); }
                            if (nb & NB_TOP) {
                                if (!IS_SKIP_T(type_top)) SETC(mvr[2 * top], mvr[2 * top + 1]);
                                if ((nb & NB_TOPLEFT) && !IS_SKIP_T(type_topleft)) SETC(mvr[2 * (top - 1)], mvr[2 * (top - 1) + 1]);
                                if (mbx < a.mb_w - 1 && !IS_SKIP_T(type_topright)) SETC(mvr[2 * (top + 1)], mvr[2 * (top + 1) + 1]);
                            }
                            if (a.l0_n_ref0 > 0)
                                for (int k = 0; k < 3; k++) {
                                    const int dx = k == 1, dy = k == 2;
                                    if ((dx && mbx >= a.mb_w - 1) || (dy && mby >= a.mb_h - 1)) continue;
                                    const int o = mb + dx + dy * a.mb_w, ref_col = a.l0_ref[o * 4];
                                    if (ref_col >= 0) {
                                        const int scale = refs.poc_delta[r] * refs.l0_inv_ref_poc[ref_col];
                                        SETC((a.l0_mv[o * 32] * scale + 128) >> 8, (a.l0_mv[o * 32 + 1] * scale + 128) >> 8);
                                    }
                                }
#undef SETC
                            WAVE_SYNC();
                        }
                        aim_b(l, r, 16, 16, 0, 0);
                        c.mvpx = px; c.mvpy = py;
                        int smx, smy, cost_mv;
                        LAUNDER(); c.lane = lane;
                        int cost = me_search_ref16(c, L, mo_b, &s.mvc[0][0], n_mvc, &thresh, smx, smy, cost_mv) + REFC(l, r);
                        if (cost < best) { best = cost; if (l) l_ref1 = r; else l_ref0 = r; me_put(l, 0, smx, smy, cost, cost_mv, px, py); }
                        if (lane == 0) {
                            if (l) { (rd.mvr1 + 2 * cb)[mb * 2] = (i16)smx; (rd.mvr1 + 2 * cb)[mb * 2 + 1] = (i16)smy; sb.left_mvr1[0] = (i16)smx; sb.left_mvr1[1] = (i16)smy; }
                            else { a.mvr[((size_t)r * nmb + mb) * 2] = (i16)smx; a.mvr[((size_t)r * nmb + mb) * 2 + 1] = (i16)smy; s.left_mvr[r][0] = (i16)smx; s.left_mvr[r][1] = (i16)smy; }
                        }
                    }
                    if (lane == 0) sb.me[l][0][2] = best - REFC(l, LREF(l));
                    WAVE_SYNC();
                }
                cache_set_b(0, 0, 0, 4, 4, l_ref0, 0, 0, 1, 0); cache_set_b(1, 0, 0, 4, 4, l_ref1, 0, 0, 1, 0);
                WAVE_SYNC();
                bi_to_tmp(0, 0, 16, 16, l_ref0, ME(0, 0, 0), ME(0, 0, 1), ME(1, 0, 0), ME(1, 0, 1));
                cost16bi = satd_region(tmp, 16, 0, 0, 16, 16) + REFC(0, l_ref0) + ME(0, 0, 3) + ME(1, 0, 3) + Q.lambda * 5;
                if (lane == 0) { sb.me[0][0][2] += Q.lambda * 3; sb.me[1][0][2] += Q.lambda * 3; }
                WAVE_SYNC();
                i_type_b = T_B_L0_L0; i_part_b = 16; i_cost_b = ME(0, 0, 2);
                if (ME(1, 0, 2) < i_cost_b) { i_cost_b = ME(1, 0, 2); i_type_b = T_B_L1_L1; }
                if (cost16bi < i_cost_b) { i_cost_b = cost16bi; i_type_b = T_B_BI_BI; }
                if (cost16direct < i_cost_b) { i_cost_b = cost16direct; i_type_b = T_B_DIRECT; }
                if (mbrd && cost16direct <= i_cost_b * 33 / 32) { pass = 0; bthresh = i_cost_b * (17 + (rd.psy_rd != 0)) / 16; kcand = 0; bstep = BS_CAND; }
                else bstep = BS_AN2;
                continue;
            }
        } else if (bstep == BS_CAND) {                          // x264_mb_analyse_b_rd: the next candidate within the threshold
            for (; kcand < 7; kcand++) {
                const bool ok = kcand == 0 ? (direct_ok && rd_direct == MX_COST_MAX)
                              : kcand == 1 ? (ME(0, 0, 2) <= bthresh && rd_l0 == MX_COST_MAX) : kcand == 2 ? (ME(1, 0, 2) <= bthresh && rd_l1 == MX_COST_MAX)
                              : kcand == 3 ? (cost16bi <= bthresh && rd_bi == MX_COST_MAX) : kcand == 4 ? (cost8bi <= bthresh && rd_8 == MX_COST_MAX)
                              : kcand == 5 ? (cost16x8bi <= bthresh && rd_168 == MX_COST_MAX) : (cost8x16bi <= bthresh && rd_816 == MX_COST_MAX);
                if (ok) break;
            }
            if (kcand == 7) { bstep = pass == 0 ? BS_AFTER_EARLY : BS_SELECT; continue; }
            skip_mc = kcand == 0;                                   // "Assumes direct/skip MC is still in fdec"
            if (kcand == 0) type = T_B_DIRECT;
            else if (kcand < 4) { type = kcand == 1 ? T_B_L0_L0 : kcand == 2 ? T_B_L1_L1 : T_B_BI_BI; part = 16; }
            else if (kcand == 4) { type = T_B_8x8; part = 13; }
            else if (kcand == 5) { type = type16x8; part = 14; }
            else { type = type8x16; part = 15; }
        } else if (bstep == BS_AFTER_EARLY) {
            if (bskip_cost < rd_direct && bskip_cost < rd_bi && bskip_cost < rd_l0 && bskip_cost < rd_l1) { type = T_B_SKIP; skip_mc = 0; fin = true; }
            else { bstep = BS_AN2; continue; }
        } else if (bstep == BS_AN2) {
            if (a.flags_inter & 0x100) {                         // X264_ANALYSE_BSUB16x16
                // ---- x264_mb_analyse_inter_b8x8 ----
                cost8bi = 0;
                for (int i = 0; i < 4; i++) {
                    const int x8 = i & 1, y8 = i >> 1;
                    int part_cost_bi = 0;
                    for (int l = 0; l < 2; l++) {
                        int px, py, vx, vy, cm;
                        predict_blk_b(l, 13, 4 * i, 2, px, py);
                        WAVE_SYNC();
                        if (lane < 2) s.mvc[0][lane] = (i16)sb.me[l][0][lane];            // (lane-dependent index: a plain LDS read, not ME())
                        WAVE_SYNC();
                        aim_b(l, LREF(l), 8, 8, 8 * x8, 8 * y8);
                        c.mvpx = px; c.mvpy = py;
                        LAUNDER(); c.lane = lane;
                        const int cost = me_search_ref16(c, L, mo_b, &s.mvc[0][0], 1, nullptr, vx, vy, cm);
                        me_put(l, 1 + i, vx, vy, cost, cm, px, py);
                        cache_set_b(l, 2 * x8, 2 * y8, 2, 2, 0, vx, vy, 0, 1);
                        part_cost_bi += cm;
                    }
                    WAVE_SYNC();
                    bi_to_tmp(8 * x8, 8 * y8, 8, 8, l_ref0, ME(0, 1 + i, 0), ME(0, 1 + i, 1), ME(1, 1 + i, 0), ME(1, 1 + i, 1));
                    part_cost_bi += satd_region(tmp, 16, 8 * x8, 8 * y8, 8, 8) + Q.lambda * 5;
                    if (lane == 0) { sb.me[0][1 + i][2] += Q.lambda * 3; sb.me[1][1 + i][2] += Q.lambda * 3; }
                    WAVE_SYNC();
                    int part_cost = ME(0, 1 + i, 2), st = 3;
                    if (ME(1, 1 + i, 2) < part_cost) { part_cost = ME(1, 1 + i, 2); st = 7; }
                    if (part_cost_bi < part_cost) { part_cost = part_cost_bi; st = 11; }
                    { const int c8d = UNI(sb.cost8direct[i]); if (c8d < part_cost) { part_cost = c8d; st = 12; } }
                    cost8bi += part_cost;
                    if (lane == 0) sb.sub[i] = (signed char)st;
                    WAVE_SYNC();
                    cache_mv_b8x8(i, 0);
                    WAVE_SYNC();
                }
                cost8bi += Q.lambda * 9;
                if (cost8bi < i_cost_b) {
                    i_type_b = T_B_8x8; i_part_b = 13; i_cost_b = cost8bi;
#pragma nounroll
                    for (int dir = 0; dir < 2; dir++) {          // 0: x264_mb_analyse_inter_b16x8, 1: _b8x16
                        if (dir == 0 ? !(SUB(0) == SUB(1) || SUB(2) == SUB(3)) : !(SUB(0) == SUB(2) || SUB(1) == SUB(3))) continue;
                        int total = 0, pt0 = 3, pt1 = 3;
                        for (int i = 0; i < 2; i++) {
                            const int bx = dir ? 8 * i : 0, by = dir ? 0 : 8 * i, w = dir ? 8 : 16, h = dir ? 16 : 8, slot = (dir ? 7 : 5) + i;
                            int part_cost_bi = 0;
                            for (int l = 0; l < 2; l++) {
                                int px, py, vx, vy, cm;
                                const int sa = dir ? 1 + i : 1 + 2 * i, sb2 = dir ? 3 + i : 2 + 2 * i;
                                predict_blk_b(l, dir ? 15 : 14, dir ? 4 * i : 8 * i, 2, px, py);           // width 2 for both shapes, as the reference has it
                                WAVE_SYNC();
                                if (lane < 4) s.mvc[lane >> 1][lane & 1] = (i16)sb.me[l][lane >> 1 ? sb2 : sa][lane & 1];
                                WAVE_SYNC();
                                aim_b(l, LREF(l), w, h, bx, by);
                                c.mvpx = px; c.mvpy = py;
                                LAUNDER(); c.lane = lane;
                                const int cost = me_search_ref16(c, L, mo_b, &s.mvc[0][0], 2, nullptr, vx, vy, cm);
                                me_put(l, slot, vx, vy, cost, cm, px, py);
                                part_cost_bi += cm;
                            }
                            bi_to_tmp(bx, by, w, h, l_ref0, ME(0, slot, 0), ME(0, slot, 1), ME(1, slot, 0), ME(1, slot, 1));
                            part_cost_bi += satd_region(tmp, 16, bx, by, w, h);
                            int part_cost = ME(0, slot, 2), pti = 3;
                            if (ME(1, slot, 2) < part_cost) { part_cost = ME(1, slot, 2); pti = 7; }
                            if (part_cost_bi + Q.lambda * 1 < part_cost) { part_cost = part_cost_bi; pti = 11; }
                            total += part_cost;
                            if (i) pt1 = pti; else pt0 = pti;
                            if (dir) cache_mv_bi(2 * i, 0, 2, 4, slot, pti, 0); else cache_mv_bi(0, 2 * i, 4, 2, slot, pti, 0);
                            WAVE_SYNC();
                        }
                        const int ty = T_B_L0_L0 + (pt0 >> 2) * 3 + (pt1 >> 2);
                        total += Q.lambda * (int)((0x999757775ull >> (4 * (ty - T_B_L0_L0))) & 15);     // i_mb_b16x8_cost_table[B_L0_L0 ..]: 5 7 7 7 5 7 9 9 9
                        if (dir) { type8x16 = ty; cost8x16bi = total; part8x16_0 = pt0; part8x16_1 = pt1; }
                        else { type16x8 = ty; cost16x8bi = total; part16x8_0 = pt0; part16x8_1 = pt1; }
                        if (total < i_cost_b) { i_cost_b = total; i_type_b = ty; i_part_b = dir ? 15 : 14; }
                    }
                }
            }
            PROF(2);
            LAUNDER();
            if (!mbrd) {
                // ---- x264_me_refine_qpel on the winning partition (analyse.c:2586-2655): one loop over (block, list) so that the refinement
                // exists once; a block's sub-partition type cost leaves its cost while it is refined ----
#pragma nounroll
                for (int j = 0; j < 8; j++) {
                    const int l = j & 1, i = j >> 1;
                    int slot, w, h, bx, by, ptype, tc = 0;
                    if (i_part_b == 16) {
                        if (i) continue;
                        slot = 0; w = 16; h = 16; bx = 0; by = 0; tc = Q.lambda * 3;
                        ptype = i_type_b == T_B_L0_L0 ? 3 : i_type_b == T_B_L1_L1 ? 7 : i_type_b == T_B_BI_BI ? 11 : 12;
                    } else if (i_part_b == 14) { if (i > 1) continue; slot = 5 + i; w = 16; h = 8; bx = 0; by = 8 * i; ptype = i ? part16x8_1 : part16x8_0; }
                    else if (i_part_b == 15) { if (i > 1) continue; slot = 7 + i; w = 8; h = 16; bx = 8 * i; by = 0; ptype = i ? part8x16_1 : part8x16_0; }
                    else { slot = 1 + i; w = 8; h = 8; bx = 8 * (i & 1); by = 8 * (i >> 1); ptype = SUB(i); tc = Q.lambda * 3; }
                    if (!sub_uses(ptype, l)) continue;
                    int vx = ME(l, slot, 0), vy = ME(l, slot, 1);
                    const int old = ME(l, slot, 2);
                    aim_b(l, LREF(l), w, h, bx, by);
                    c.mvpx = ME(l, slot, 4); c.mvpy = ME(l, slot, 5);
                    LAUNDER(); c.lane = lane;
                    const int nc = me_refine_qpel16(c, L, mo_b, old - tc, vx, vy);
                    WAVE_SYNC();
                    if (lane < 3) sb.me[l][slot][lane] = lane == 0 ? vx : lane == 1 ? vy : nc;
                    WAVE_SYNC();
                    if (i_part_b == 16 && ptype != 11) i_cost_b = nc + tc;
                    if (i_part_b == 13 && ptype != 11) cost8bi += nc + tc - old;
                }
                bstep = BS_SELECT;
                continue;
            }
            i_satd_inter_b = i_cost_b;
            pass = 1; bthresh = i_satd_inter_b * (17 + (rd.psy_rd != 0)) / 16; kcand = 0; bstep = BS_CAND;
            continue;
        } else if (bstep == BS_SELECT) {
            if (mbrd) {
                i_type_b = T_B_SKIP; i_cost_b = bskip_cost; i_part_b = 16;
                if (rd_l0 < i_cost_b) { i_cost_b = rd_l0; i_type_b = T_B_L0_L0; }
                if (rd_l1 < i_cost_b) { i_cost_b = rd_l1; i_type_b = T_B_L1_L1; }
                if (rd_bi < i_cost_b) { i_cost_b = rd_bi; i_type_b = T_B_BI_BI; }
                if (rd_direct < i_cost_b) { i_cost_b = rd_direct; i_type_b = T_B_DIRECT; }
                if (rd_168 < i_cost_b) { i_cost_b = rd_168; i_type_b = type16x8; i_part_b = 14; }
                if (rd_816 < i_cost_b) { i_cost_b = rd_816; i_type_b = type8x16; i_part_b = 15; }
                if (rd_8 < i_cost_b) { i_cost_b = rd_8; i_type_b = T_B_8x8; i_part_b = 13; }
                type = i_type_b; part = i_part_b;
            }
            analyse_intra(i_satd_inter_b);                      // without the RD levels the reference passes 0 here (its i_satd_inter is only set for them): only I_16x16 gets a cost
            PROF(6);
            if (!mbrd) { bstep = BS_FINAL; continue; }
            // x264_mb_analyse_transform_rd: every B type but B_SKIP may use the 8x8 transform (direct_8x8_inference is on)
            if (a.transform8x8 && type != T_B_SKIP) { t8 = !t8; skip_mc = 0; bstep = BS_T8; }
            else { bstep = BS_I16; continue; }
        } else if (bstep == BS_I16) {                           // x264_intra_rd with i_satd_inter * 17 / 16
            if (!(satd_i16 <= i_satd_inter_b * 17 / 16)) { satd_i16 = MX_COST_MAX; bstep = BS_I4; continue; }
            type = T_I_16x16;
        } else if (bstep == BS_I4) {
            if (!(satd_i4 <= i_satd_inter_b * 17 / 16 && satd_i4 < MX_COST_MAX)) { satd_i4 = MX_COST_MAX; bstep = BS_I8; continue; }
            type = T_I_4x4;
        } else if (bstep == BS_I8) {
            if (!(satd_i8 <= i_satd_inter_b * 17 / 16 && satd_i8 < MX_COST_MAX)) { satd_i8 = MX_COST_MAX; bstep = BS_FINAL; continue; }
            type = T_I_8x8;
        } else {                                                // BS_FINAL
            fin = true;
            if (satd_i16 < i_cost_b) { i_cost_b = satd_i16; i_type_b = T_I_16x16; }
            if (satd_i8 < i_cost_b) { i_cost_b = satd_i8; i_type_b = T_I_8x8; }
            if (satd_i4 < i_cost_b) { i_cost_b = satd_i4; i_type_b = T_I_4x4; }
            if (satd_pcm < i_cost_b) { i_cost_b = satd_pcm; i_type_b = T_I_PCM; }
            type = i_type_b; part = i_part_b;
            skip_mc = 0;
            // x264_refine_bidir (subme >= 5): the bi-predicted blocks of the chosen partition
            if (!IS_INTRA_T(type) && a.subme >= 5) {
                if (part == 16) { if (type == T_B_BI_BI) refine_bidir_satd(0, 0, 0, 16, 16); }
                else if (part == 14) { if (part16x8_0 == 11) refine_bidir_satd(5, 0, 0, 16, 8); if (part16x8_1 == 11) refine_bidir_satd(6, 0, 8, 16, 8); }
                else if (part == 15) { if (part8x16_0 == 11) refine_bidir_satd(7, 0, 0, 8, 16); if (part8x16_1 == 11) refine_bidir_satd(8, 8, 0, 8, 16); }
                else if (type == T_B_8x8) for (int i = 0; i < 4; i++) if (SUB(i) == 11) refine_bidir_satd(1 + i, 8 * (i & 1), 8 * (i >> 1), 8, 8);
            }
        }
        if (fin) {                                              // analyse.c:2768-2773
            tq.on = rd.trellis != 0;
            if (rd.trellis == 1 || a.nr) skip_intra = 0;
        }
        // x264_analyse_update_cache, then the encoder: a trial of x264_rd_cost_mb or the real thing
        if (!IS_INTRA_T(type)) { update_cache_b(); if (!skip_mc) mc_b(); }
        const int t8_bak = t8;
        PROF(6);
        if (!(fin && type == T_I_PCM)) encode_mb(fin ? 1 : 0);
        if (fin) { encoded = true; break; }
        PROF(0);
        int cst = ssd_mb();
        if (type == T_B_SKIP) cst += (Q.lambda2 + 128) >> 8;
        else {
            syn_prepare();
            for (int k = lane; k < 460; k += 64) sr.cabac_tmp[k] = sr.cabac[k];
            const MbSynDev y0 = make_syn();
            WAVE_SYNC();
            if (lane == 0) {
                DCabac tcb = {0, 0x1FE, -1, 0, nullptr, 0};
                MbSynDev y = y0;
                cw_macroblock(tcb, sr.cabac_tmp, 1, y, s.fe, 0);
                sr.tmp_i[0] = tcb.f8;
            }
            WAVE_SYNC();
            const int f8 = UNI(sr.tmp_i[0]);
            cst += (int)(((unsigned long long)(u32)f8 * (u32)Q.lambda2 + 32768) >> 16);
        }
        t8 = t8_bak;
        PROF(7);
        if (bstep == BS_CAND) {
            if (kcand == 0) rd_direct = cst; else if (kcand == 1) rd_l0 = cst; else if (kcand == 2) rd_l1 = cst; else if (kcand == 3) rd_bi = cst;
            else if (kcand == 4) { rd_8 = cst; cache_skip_b(0, 0, 4, 4, 0); WAVE_SYNC(); }
            else if (kcand == 5) rd_168 = cst; else rd_816 = cst;
            skip_mc = 0;
            kcand++;
        } else if (bstep == BS_T8) {
            if (i_cost_b >= cst) {
                if (i_cost_b > 0) i_satd_inter_b = (int)((long long)i_satd_inter_b * cst / i_cost_b);
                if (i_satd_inter_b == 0) i_satd_inter_b = 1;
                i_cost_b = cst;
            } else
                t8 = !t8;
            bstep = BS_I16;
        } else if (bstep == BS_I16) { satd_i16 = cst; bstep = BS_I4; }
        else if (bstep == BS_I4) { satd_i4 = cst; bstep = BS_I8; }
        else { satd_i8 = cst; bstep = BS_FINAL; }
    }
    // the cache entry of block 12 as this macroblock leaves it: the next one inherits it (SwRd::stale)
    if (TD && lane == 30) {         // (lane 30 is the only writer of entry 30: cache_set_b)
        sb.stale[0] = sr.cref[30]; sb.stale[1] = sr.cmv[30][0]; sb.stale[2] = sr.cmv[30][1];
        sb.stale[3] = sb.cref1[30]; sb.stale[4] = sb.cmv1[30][0]; sb.stale[5] = sb.cmv1[30][1];
    }
}
