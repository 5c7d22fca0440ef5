// slice_refine.h -- textually included by slice_kernel.h inside the candidate loop of k_slice_sweep<.., RF = true>, at step 11:
// the RD refinement of subme 8-9 (a->i_mbrd >= 2) for I and P slices,
//   x264_intra_rd_refine            R/encoder/analyse.c:876-1056   (I_16x16 modes, chroma mode, I_4x4 / I_8x8 modes block by block)
//   x264_me_refine_qpel_rd          R/encoder/me.c:961-1047        (every partition of the winner, 16x16 .. 8x8)
//   x264_rd_cost_part / _i8x8 / _i4x4 / _i8x8_chroma   R/encoder/rdo.c:202-315
//   x264_macroblock_encode_p8x8     R/encoder/macroblock.c:917-1042
// A trial that prices the WHOLE macroblock (a 16x16 vector, an I_16x16 mode: x264_rd_cost_mb) leaves this block with rf_emit set and
// runs through the loop's one tail (encoder, distortion, bit counter); the loop comes back here with its cost.  Everything else
// -- the partial encodes -- is priced in place.  Sub-8x8 partitions stay refused with the RD levels: the reference's partial bit
// counts then read cache entries the PREVIOUS macroblock left (oracle/slice_oracle.c: carry_nnz), state this kernel does not carry.
{
    bool rf_emit = false;
    // ---- the bit counters of the partial costs: lane 0 on a copy of the live contexts (COPY_CABAC, rdo.c:62) ----
    auto rf_bits = [&](int kind, int p0, int p1) __attribute__((always_inline)) -> int {
        syn_prepare();
        for (int k = lane; k < 460; k += 64) sr.cabac_tmp[k] = sr.cabac[k];
        const MbSynDev y0 = make_syn();
        WAVE_SYNC();
        if (lane == 0) {
            DCabac tcb = {0, 0x1FE, -1, 0, nullptr, 0};
            MbSynDev y = y0;
            if (kind == 0) cw_partition_size(tcb, sr.cabac_tmp, y, p0, p1);
            else if (kind == 2) cw_partition_i8x8_size(tcb, sr.cabac_tmp, y, p0, p1);
            else if (kind == 3) cw_partition_i4x4_size(tcb, sr.cabac_tmp, y, p0, p1);
            else cw_i8x8_chroma_size(tcb, sr.cabac_tmp, y);
            sr.tmp_i[0] = tcb.f8;
        }
        WAVE_SYNC();
        return UNI(sr.tmp_i[0]);
    };
    // (ssd << 8) + bits: the partition costs carry 8 more bits than x264_rd_cost_mb's
    auto rf_cost64 = [&](int ssd, int f8, int lambda2) -> unsigned long long {
        return ((unsigned long long)(u32)ssd << 8) + (((unsigned long long)(u32)f8 * (u32)lambda2 + 128) >> 8);
    };
    // psy-RD's term of ssd_plane for luma (rdo.c:106-130): |complexity of the reconstruction - complexity of the source|
    auto rf_psy = [&](int v) -> int { return (v * rd.psy_rd * Q.lambda + 128) >> 8; };

    for (;;) {
        if (rf_kind == 1) {
            // ---- x264_intra_rd_refine, I_16x16: every other mode whose SATD cost is within 9/8 of the winner's, by x264_rd_cost_mb ----
            while (rf_i < rf_n) {
                const int m = (int)((rf_list >> (4 * rf_i)) & 15);
                rf_i++;
                if (m == rf_old16 || UNI(sf.i16dir[m]) > rf_thr) continue;
                pred16 = m; rf_emit = true;
                break;
            }
            if (rf_emit) break;
            pred16 = rf_best16;
            rf_kind = 2;
        } else if (rf_kind == 2) {
            // ---- RD selection for chroma prediction, analyse.c:907-951 ----
            {
                int n;
                const u32 list = sw_modes8c(nb, n);
                if (n > 1) {
                    const int thresh = satd_chroma * 5 / 4;
                    u32 flist = 0;
                    int fn = 0;
                    for (int i = 0; i < n; i++) {
                        const int m = (int)((list >> (4 * i)) & 15);
                        if (UNI(sf.cdir[i]) < thresh && m != predc) { flist |= (u32)m << (4 * fn); fn++; }
                    }
                    if (fn > 0) {
                        const int lam = d_lambda2_tab[Q.qpc];
                        int cbp_best = cbp_chroma, predc_best = predc;
                        unsigned long long best = 0;
                        for (int i = -1; i < fn; i++) {              // -1: the current mode, whose pixels and levels are still around (no transform)
                            const int m = i < 0 ? predc : (int)((flist >> (4 * i)) & 15);
                            bool b_dct = false;
                            if (i >= 0) {
                                sw_pred8c(s, m, lane, 0);
                                b_dct = cbp_chroma != 0;             // "if we've already found a mode that needs no residual ..." (the LAST trial's cbp)
                            }
                            if (b_dct) cbp_chroma = sw_encode_chroma(s, a, Q, tq, 0, lane);
                            int acc;
                            {
                                const int cx = lane & 7, cy = lane >> 3;
                                const int du = (int)s.fe[256 + cy * 8 + cx] - (int)s.fd[FDU + cy * FD + cx], dv = (int)s.fe[320 + cy * 8 + cx] - (int)s.fd[FDV + cy * FD + cx];
                                acc = du * du + dv * dv;
                            }
                            const int ssd = wave_sum(acc);
                            predc = m;                               // h->mb.i_chroma_pred_mode = i_mode
                            const unsigned long long c64 = rf_cost64(ssd, rf_bits(4, 0, 0), lam);
                            if (i < 0) best = c64;
                            else if (c64 < best) { best = c64; predc_best = m; cbp_best = cbp_chroma; }
                        }
                        predc = predc_best; cbp_chroma = cbp_best;
                    }
                }
            }
            if (type == T_I_4x4) {
                // ---- every 4x4 block's mode by x264_rd_cost_i4x4, analyse.c:953-999 ----
                for (int idx = 0; idx < 16; idx++) {
                    int bx, by, n;
                    sw_blk_xy(idx, bx, by);
                    const int nb4 = sw_nb4(idx, nb);
                    const unsigned long long list = sw_modes4(nb4, n);
                    u8 *dst = s.fd + FDY + by * FD + bx;
                    if ((nb4 & (NB_TOPRIGHT | NB_TOP)) == NB_TOP && lane < 4) dst[4 - FD + lane] = dst[3 - FD];
                    WAVE_SYNC();
                    sw_pred4_table(s, dst, lane);
                    WAVE_SYNC();
                    unsigned long long best = ~0ull;
                    int best_mode = 0, best_nnz = 0;
                    for (int i = 0; i < n; i++) {
                        const int mode = (int)((list >> (4 * i)) & 15);
                        if (lane < 16) dst[(lane >> 2) * FD + (lane & 3)] = s.pt4[(s.p4lut[mode * 4 + (lane >> 2)] >> (8 * (lane & 3))) & 255];
                        WAVE_SYNC();
                        sw_encode_i4x4(s, a, Q, tq, idx, cbp_luma, lane);
                        int d2 = 0, pix = 0;
                        if (lane < 16) { pix = dst[(lane >> 2) * FD + (lane & 3)]; const int d = (int)s.fe[(by + (lane >> 2)) * 16 + bx + (lane & 3)] - pix; d2 = d * d; }
                        int ssd = UNI(row_sum16(d2));
                        if (rd.psy_rd) {                             // size > PIXEL_8x8: |SATD of the block against zero - its DC / 2 - the source's|
                            const int dc = UNI(row_sum16(pix)) >> 1;
                            int sat = 0;
                            if (lane == 0) sat = satd_4x4(dst, FD, sr.zero16, 0);
                            sat = UNI(sat);
                            ssd += rf_psy(iabs(sat - dc - UNI(sr.fenc_satd[(by >> 2) * 4 + (bx >> 2)])));
                        }
                        const unsigned long long c64 = rf_cost64(ssd, rf_bits(3, idx, mode), Q.lambda2);
                        if (best > c64) {
                            best = c64; best_mode = mode; best_nnz = UNI(s.nnz[idx]);
                            if (lane < 16) sf.pels[lane] = dst[(lane >> 2) * FD + (lane & 3)];
                        }
                        WAVE_SYNC();
                    }
                    if (lane < 16) dst[(lane >> 2) * FD + (lane & 3)] = sf.pels[lane];
                    if (lane == 0) { s.nnz[idx] = (u8)best_nnz; s.pred4[idx] = (signed char)best_mode; s.i4c[sw_scan8(idx)] = (signed char)best_mode; }
                    WAVE_SYNC();
                }
            } else if (type == T_I_8x8) {
                // ---- every 8x8 block's mode by x264_rd_cost_i8x8, analyse.c:1000-1055 ----
                for (int idx = 0; idx < 4; idx++) {
                    const int bx = 8 * (idx & 1), by = 8 * (idx >> 1), nb8 = sw_nb8(idx, nb);
                    int n;
                    const unsigned long long list = sw_modes4(nb8, n);
                    const int thresh = UNI(sf.i8dir[UNI((int)s.pred8[idx])][idx]) * 11 / 8;
                    u8 *dst = s.fd + FDY + by * FD + bx;
                    sw_pred8_filter_all(s.edge8, dst, nb8, lane);
                    WAVE_SYNC();
                    sw_pred8_table(s, lane);
                    WAVE_SYNC();
                    unsigned long long best = ~0ull;
                    int best_mode = UNI((int)s.pred8[idx]), cbp_new = 0;
                    u32 best_nnz = 0;
                    for (int i = 0; i < n; i++) {
                        const int mode = (int)((list >> (4 * i)) & 15);
                        if (UNI(sf.i8dir[mode][idx]) > thresh) continue;
                        {
                            const int v = s.pt8[(s.p8lut[(mode * 8 + (lane >> 3)) * 2 + ((lane >> 2) & 1)] >> (8 * (lane & 3))) & 255];
                            WAVE_SYNC();
                            dst[(lane >> 3) * FD + (lane & 7)] = (u8)v;
                            WAVE_SYNC();
                        }
                        cbp_luma = i8_cbp_rd & ~(1 << idx);          // h->mb.i_cbp_luma = a->i_cbp_i8x8_luma, then x264_rd_cost_i8x8 clears the block's bit
                        t8 = 1;
                        sw_encode_i8x8(s, a, Q, tq, idx, cbp_luma, lane);
                        int ssd;
                        { const int d = (int)s.fe[(by + (lane >> 3)) * 16 + bx + (lane & 7)] - (int)dst[(lane >> 3) * FD + (lane & 7)]; ssd = wave_sum(d * d); }
                        if (rd.psy_rd) {
                            unsigned long long h = 0;
                            if (lane == 0) h = hadamard_ac_8x8(dst, FD);
                            const u32 lo = (u32)UNI((int)(u32)h), hi = (u32)UNI((int)(u32)(h >> 32));
                            const unsigned long long sum = ((unsigned long long)hi << 32) + lo;        // x264_pixel_hadamard_ac_8x8: ((sum >> 34) << 32) + ((u32)sum >> 1)
                            const int s4 = (int)((u32)sum >> 1), s8 = (int)(sum >> 34);
                            const int k0 = (by >> 2) * 4 + (bx >> 2);
                            const int f4 = UNI(sr.fenc_satd[k0]) + UNI(sr.fenc_satd[k0 + 1]) + UNI(sr.fenc_satd[k0 + 4]) + UNI(sr.fenc_satd[k0 + 5]);
                            ssd += rf_psy((iabs(s4 - f4) + iabs(s8 - UNI(sr.fenc_sa8d[idx]))) >> 1);
                        }
                        const unsigned long long c64 = rf_cost64(ssd, rf_bits(2, idx, mode), Q.lambda2);
                        if (best > c64) {
                            best = c64; best_mode = mode; cbp_new = cbp_luma;
                            // pels_h: the block's last row; pels_v: its last column (rows 0..6) for the blocks on the left
                            if (lane < 8) sf.pels[lane] = dst[7 * FD + lane];
                            else if (lane < 15) sf.pels[lane] = dst[(lane - 8) * FD + 7];
                            best_nnz = *(const u32 *)(s.nnz + 4 * idx);
                        }
                        WAVE_SYNC();
                    }
                    i8_cbp_rd = cbp_new;
                    if (lane < 8) dst[7 * FD + lane] = sf.pels[lane];
                    else if (lane < 15 && !(idx & 1)) dst[(lane - 8) * FD + 7] = sf.pels[lane];
                    if (lane == 0) { *(u32 *)(s.nnz + 4 * idx) = best_nnz; s.pred8[idx] = (signed char)best_mode; }
                    if (lane < 4) s.i4c[sw_scan8(4 * idx) + (lane & 1) + 8 * (lane >> 1)] = (signed char)best_mode;
                    WAVE_SYNC();
                }
            }
            rf_kind = 9;
        } else if (rf_kind == 3) {
            // ---- x264_me_refine_qpel_rd on every partition of the winner, analyse.c:2412-2462 ----
            const int rf_np = part == 16 ? 1 : part == 13 ? 4 : 2;
            if (q_st == -1) {
                // the references of the partitions into the motion cache (x264_macroblock_cache_ref; a P_8x8 macroblock: x264_analyse_update_cache)
                if (part == 16) cache_set(0, 0, 4, 4, me16r, 0, 0, 0);
                else if (part == 14) { cache_set(0, 0, 4, 2, pme(4, 4), 0, 0, 0); cache_set(0, 2, 4, 2, pme(5, 4), 0, 0, 0); }
                else if (part == 15) { cache_set(0, 0, 2, 4, pme(6, 4), 0, 0, 0); cache_set(2, 0, 2, 4, pme(7, 4), 0, 0, 0); }
                else for (int i = 0; i < 4; i++) cache_set(2 * (i & 1), 2 * (i >> 1), 2, 2, pme(i, 4), pme(i, 0), pme(i, 1), 1);
                update_cache_p();
                q_st = 0;
            }
            // one round of four SATD candidates per trip around (q_omx, q_omy): lane group g = candidate base + g of the offset lists
            auto rf_satd = [&](int n, u32 dxs, u32 dys, int base, bool avoid) __attribute__((always_inline)) {
                LAUNDER(); c.lane = lane;
                MxCtx cu = mx_uniform(c);
                cu.patch_on = false;
                if (cu.has_patch) mx_load_patch(cu, q_omx, q_omy);
                for (int t = 0; t < n; t += 4) {
                    const int j = t + (lane >> 4);
                    const bool in = j < n;
                    const int x = q_omx + (in ? mx_nib(dxs, base + j) : 0), y = q_omy + (in ? mx_nib(dys, base + j) : 0);
                    const int cost = subpel_sum16_lane(cu, x, y, 1, 0) + cu.lane_cost(x, y);
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if (t + k < n) {
                            const int xk = __builtin_amdgcn_readlane(x, 16 * k), yk = __builtin_amdgcn_readlane(y, 16 * k);
                            const bool skip = avoid && xk == q_pmx && yk == q_pmy;
                            const int v = skip ? MX_COST_MAX : __builtin_amdgcn_readlane(cost, 16 * k);
                            if (lane == t + k) q_satds = v;
                            if (!skip && (u32)v < q_bsatd) q_bsatd = (u32)v;
                        }
                }
            };
            bool have = false;                                       // a candidate (q_cx, q_cy, q_tag) is to be priced
            while (!have && rf_kind == 3) {
                if (q_st == 0) {                                     // a partition starts: COST_MV_SATD( bmx, bmy, bsatd, 0 ); COST_MV_RD( bmx, bmy, 0, 0, 0 )
                    q_slot = part == 13 ? rf_i : part == 14 ? 4 + rf_i : part == 15 ? 6 + rf_i : -1;
                    q_pix = part == 16 ? 0 : part == 14 ? 1 : part == 15 ? 2 : 3;
                    q_w = part == 16 || part == 14 ? 16 : 8; q_h = part == 16 || part == 15 ? 16 : 8;
                    q_bx = part == 13 ? 8 * (rf_i & 1) : part == 15 ? 8 * rf_i : 0; q_by = part == 13 ? 8 * (rf_i >> 1) : part == 14 ? 8 * rf_i : 0;
                    q_i4 = part == 13 ? 4 * rf_i : part == 14 ? 8 * rf_i : part == 15 ? 4 * rf_i : 0;
                    if (q_slot < 0) { q_bmx = me16x; q_bmy = me16y; q_ref = me16r; q_mvpx = bmvpx; q_mvpy = bmvpy; rf_best = (unsigned long long)(u32)rd16; }
                    else { q_bmx = pme(q_slot, 0); q_bmy = pme(q_slot, 1); q_ref = pme(q_slot, 4); q_mvpx = pme(q_slot, 6); q_mvpy = pme(q_slot, 7); rf_best = ~0ull >> 4; }
                    if (q_pix != 0 && q_i4 != 0) predict_blk(part, q_i4, q_w >> 2, q_mvpx, q_mvpy);
                    q_m0x = q_bmx; q_m0y = q_bmy; q_pmx = q_mvpx; q_pmy = q_mvpy;
                    aim(q_ref, q_w, q_h, q_bx, q_by);
                    c.mvpx = q_mvpx; c.mvpy = q_mvpy;
                    q_omx = q_bmx; q_omy = q_bmy; q_bsatd = 0xffffffffu;
                    rf_satd(1, 0, 0, 0, false);
                    q_cx = q_bmx; q_cy = q_bmy; q_tag = -3; q_dir = -2;
                    q_st = 1; have = true;
                } else if (q_st == 1) {                              // "check the predicted mv"
                    q_st = 2; q_after_pm = 0;
                    if ((q_bmx != q_pmx || q_bmy != q_pmy) && q_pmx >= L.smin0 && q_pmx <= L.smax0 && q_pmy >= L.smin1 && q_pmy <= L.smax1) {
                        q_omx = q_pmx; q_omy = q_pmy;
                        rf_satd(1, 0, 0, 0, false);
                        if ((u32)__builtin_amdgcn_readlane(q_satds, 0) <= q_bsatd * 17 / 16) { q_cx = q_pmx; q_cy = q_pmy; q_tag = -3; q_after_pm = 1; have = true; }
                    }
                } else if (q_st == 2) {                              // the hexagon's first six
                    // "if pmv is chosen, set the MV to avoid checking to bmv instead"
                    if (q_after_pm && q_bmx == q_pmx && q_bmy == q_pmy) { q_pmx = q_m0x; q_pmy = q_m0y; }
                    q_dir = -2; q_omx = q_bmx; q_omy = q_bmy;
                    rf_satd(6, MX_HEX2_DX, MX_HEX2_DY, 1, true);
                    q_j = 0; q_st = 3;
                } else if (q_st == 3) {
                    while (q_j < 6 && !((u32)__builtin_amdgcn_readlane(q_satds, q_j) <= q_bsatd * 17 / 16)) q_j++;
                    if (q_j < 6) { q_cx = q_omx + mx_nib(MX_HEX2_DX, q_j + 1); q_cy = q_omy + mx_nib(MX_HEX2_DY, q_j + 1); q_tag = q_j; q_j++; have = true; }
                    else if (q_dir != -2) { q_it = 1; q_st = 4; }
                    else q_st = 6;
                } else if (q_st == 4) {                              // "half hexagon, not overlapping the previous iteration"
                    if (q_it >= 10 || q_bmy > L.smax1 - 2 || q_bmy < L.smin1 - 2) q_st = 6;
                    else {
                        q_odir = (q_dir + 6) % 6;                   // mod6m1[dir + 1], dir = -1 .. 6
                        q_dir = -2; q_omx = q_bmx; q_omy = q_bmy;
                        rf_satd(3, MX_HEX2_DX, MX_HEX2_DY, q_odir, true);
                        q_j = 0; q_st = 5;
                    }
                } else if (q_st == 5) {
                    while (q_j < 3 && !((u32)__builtin_amdgcn_readlane(q_satds, q_j) <= q_bsatd * 17 / 16)) q_j++;
                    if (q_j < 3) { q_cx = q_omx + mx_nib(MX_HEX2_DX, q_odir + q_j); q_cy = q_omy + mx_nib(MX_HEX2_DY, q_odir + q_j); q_tag = q_odir - 1 + q_j; q_j++; have = true; }
                    else if (q_dir == -2) q_st = 6;
                    else { q_it++; q_st = 4; }
                } else if (q_st == 6) {                              // "square refine"
                    q_omx = q_bmx; q_omy = q_bmy;
                    rf_satd(8, MX_NIB8(0, 0, -1, 1, -1, 1, -1, 1), MX_NIB8(-1, 1, 0, 0, -1, 1, 1, -1), 0, true);
                    q_j = 0; q_st = 7;
                } else if (q_st == 7) {
                    while (q_j < 8 && !((u32)__builtin_amdgcn_readlane(q_satds, q_j) <= q_bsatd * 17 / 16)) q_j++;
                    if (q_j < 8) { q_cx = q_omx + mx_nib(MX_NIB8(0, 0, -1, 1, -1, 1, -1, 1), q_j); q_cy = q_omy + mx_nib(MX_NIB8(-1, 1, 0, 0, -1, 1, 1, -1), q_j); q_tag = -3; q_j++; have = true; }
                    else q_st = 8;
                } else {                                             // the partition is done (me.c:1041-1046)
                    q_bmy = clip3(q_bmy, L.smin1, L.smax1);
                    if (q_slot < 0) { me16x = q_bmx; me16y = q_bmy; }
                    else { if (lane == q_slot * 8) pme_v = q_bmx; if (lane == q_slot * 8 + 1) pme_v = q_bmy; }
                    cache_set(q_bx >> 2, q_by >> 2, q_w >> 2, q_h >> 2, q_ref, q_bmx, q_bmy, 1);
                    update_cache_p();
                    if (lane < 16) {                                 // x264_macroblock_cache_mvd
                        const int x4 = (lane & 3) * 4, y4 = (lane >> 2) * 4;
                        if (x4 >= q_bx && x4 < q_bx + q_w && y4 >= q_by && y4 < q_by + q_h) {
                            const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                            sr.cmvd[k][0] = (i16)(q_bmx - q_mvpx); sr.cmvd[k][1] = (i16)(q_bmy - q_mvpy);
                        }
                    }
                    WAVE_SYNC();
                    rf_i++; q_st = 0;
                    if (rf_i >= rf_np) rf_kind = 9;
                }
            }
            if (have) {
                // the candidate into the record the encoders read (the cache's two entries of me.c:953-954: here the whole partition, which
                // nothing else reads while it is being tried)
                if (q_slot < 0) { me16x = q_cx; me16y = q_cy; rf_emit = true; break; }     // x264_rd_cost_part( .., PIXEL_16x16 ) = x264_rd_cost_mb: the loop's tail
                if (lane == q_slot * 8) pme_v = q_cx;
                if (lane == q_slot * 8 + 1) pme_v = q_cy;
                update_cache_p();
                // ---- x264_rd_cost_part, rdo.c:202-242: the partition's 8x8 blocks encoded, their distortion, x264_partition_size_cabac ----
                const int i8 = q_i4 >> 2, n8 = q_pix == 3 ? 1 : 2, st8 = q_pix == 2 ? 2 : 1;
                cbp_luma = 0;
                for (int k = 0; k < n8; k++) {
                    // x264_macroblock_encode_p8x8 (macroblock.c:917-1042): the block's prediction ...
                    const int b8 = i8 + k * st8, x8 = 8 * (b8 & 1), y8 = 8 * (b8 >> 1);
                    {
                        const int lox = 4 * (-16 * mbx - 24), hix = 4 * (16 * (a.mb_w - mbx - 1) + 24), loy = 4 * (-16 * mby - 24), hiy = 4 * (16 * (a.mb_h - mby - 1) + 24);
                        const int vx = clip3(UNI(s.mv4[(y8 >> 2) * 4 + (x8 >> 2)][0]), lox, hix), vy = clip3(UNI(s.mv4[(y8 >> 2) * 4 + (x8 >> 2)][1]), loy, hiy), ri = UNI((int)s.ref8[b8]);
                        if (lane < 16) {
                            const int r = y8 + (lane >> 1), x = x8 + (lane & 1) * 4;
                            const int qx = vx & 3, qy = vy & 3, idx = qy * 4 + qx;
                            const ptrdiff_t base = oy + (ptrdiff_t)((vy >> 2) + r) * a.sy + (vx >> 2) + x + (ptrdiff_t)by_;
                            const u8 *pa = refs.y[ri][c_qpel_a[idx]] + base + (qy == 3) * a.sy, *pb = refs.y[ri][c_qpel_b[idx]] + base + (qx == 3);
#pragma unroll
                            for (int i = 0; i < 4; i++) s.fd[FDY + r * FD + x + i] = (idx & 5) ? (u8)(((int)pa[i] + (int)pb[i] + 1) >> 1) : pa[i];
                        } else if (lane < 48) {
                            const int l = lane - 16, pl = l >> 4, cy = (y8 >> 1) + ((l >> 2) & 3), cx = (x8 >> 1) + (l & 3);
                            const int dx = vx & 7, dyy = vy & 7;
                            const int ca = (8 - dx) * (8 - dyy), cb_ = dx * (8 - dyy), cc = (8 - dx) * dyy, cd = dx * dyy;
                            const ptrdiff_t cbase = oc + (ptrdiff_t)((vy >> 3) + cy) * a.sc + (vx >> 3) + cx + (ptrdiff_t)bc_;
                            const u8 *pp = (pl ? refs.v[ri] : refs.u[ri]) + cbase;
                            s.fd[(pl ? FDV : FDU) + cy * FD + cx] = (u8)((ca * pp[0] + cb_ * pp[1] + cc * pp[a.sc] + cd * pp[a.sc + 1] + 32) >> 6);
                        }
                        WAVE_SYNC();
                    }
                    // ... its luma residual ...
                    int nnz8 = 0;
                    if (t8) {
                        sw_luma8x8_fwd(s, Q, tq, 1, 1 << b8, lane);
                        const int sc = UNI(s.score[b8]);
                        nnz8 = (sc >> 8) & 1;
                        if (nnz8 && a.dct_decimate && !tq.on) nnz8 = (sc & 255) >= 4;
                        if (lane < 4) s.nnz[4 * b8 + lane] = (u8)nnz8;
                        WAVE_SYNC();
                        if (nnz8) sw_luma8x8_add(s, 1, Q.qp, 1 << b8, lane);
                    } else {
                        sw_luma4x4_fwd(s, a, Q, tq, 1, false, lane, nullptr, 0, 1 << b8);
                        int dec = 0;
#pragma unroll
                        for (int i4 = 0; i4 < 4; i4++) {
                            const int sc = UNI(s.score[4 * b8 + i4]);
                            if (sc >> 8) { nnz8 = 1; if (a.dct_decimate) dec += sc & 255; }
                            if (lane == i4) s.nnz[4 * b8 + i4] = (u8)(sc >> 8);
                        }
                        if (a.dct_decimate && dec < 4) nnz8 = 0;
                        WAVE_SYNC();
                        if (nnz8) sw_luma4x4_add(s, lane, 1 << b8);
                        else { if (lane < 4) s.nnz[4 * b8 + lane] = 0; WAVE_SYNC(); }
                    }
                    cbp_luma |= nnz8 << b8;
                    // ... and the chroma block of either plane: AC only ("doesn't transform chroma dc")
                    {
                        i16 cc[16], lv[16];
                        const int ch = lane, cblk = 4 * ch + b8;
                        if (lane < 2) {
                            int r[16];
                            const u8 *fe = s.fe + 256 + 64 * ch + (y8 >> 1) * 8 + (x8 >> 1), *pr = s.fd + (ch ? FDV : FDU) + (y8 >> 1) * FD + (x8 >> 1);
#pragma unroll
                            for (int j = 0; j < 4; j++)
#pragma unroll
                                for (int i = 0; i < 4; i++) r[4 * j + i] = (int)fe[j * 8 + i] - (int)pr[j * FD + i];
                            fwd4x4(cc, r);
                            cc[0] = 0;
                            if (tq.on) {
#pragma unroll
                                for (int i = 0; i < 16; i++) s.ccoef[cblk][i] = cc[i];
                            }
                        }
                        if (tq.on) {                                 // x264_quant_4x4_trellis( .., CQM_4PC, i_qp, DCT_CHROMA_AC, 0, 0 ), macroblock.c:1020
                            WAVE_SYNC();
                            td_trellis_wave(tq.r->tw, (u32 *)s.patch, &s.ccoef[4 * ((lane >> 4) & 1) + b8][0], lane < 32, s.qmf[3], tq.r->unq4[3], tq.r->w4z, tq.r->zz4, tq.r->cabac, 4,
                                            d_trellis_lambda2[0][Q.qpc], 1, 0, 16, lane);
                            WAVE_SYNC();
                        }
                        if (lane < 2) {
                            int nz = 0;
                            if (tq.on) {
#pragma unroll
                                for (int i = 0; i < 16; i++) { cc[i] = s.ccoef[cblk][i]; nz |= cc[i]; }
                            } else {
#pragma unroll
                                for (int i = 0; i < 16; i++) { const int q = quant_one(cc[i], s.qmf[3][i], s.qbias[3][i]); cc[i] = (i16)q; nz |= q; }
                            }
                            s.nnz[16 + b8 + 4 * ch] = (u8)(nz != 0);
                            if (nz) {
                                SCAN4_FRAME(lv, cc);
                                int res[16];
                                i16 dq[16];
#pragma unroll
                                for (int i = 0; i < 16; i++) { s.lv_cac[16 * cblk + i] = lv[i]; dq[i] = (i16)dequant_one(cc[i], s.qdq[3][i], Q.qpc / 6 - 4); }
                                inv4x4(res, dq);
                                u8 *pr = s.fd + (ch ? FDV : FDU) + (y8 >> 1) * FD + (x8 >> 1);
#pragma unroll
                                for (int j = 0; j < 4; j++)
#pragma unroll
                                    for (int i = 0; i < 4; i++) { u8 *p = pr + j * FD + i; *p = (u8)clip_u8((int)*p + res[4 * j + i]); }
                            }
                        }
                        WAVE_SYNC();
                    }
                }
                cbp_chroma = 2;
                // ssd_plane of the partition's luma and of both chroma blocks (rdo.c:223-225)
                int ssd;
                {
                    int acc = 0;
                    const int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
                    if (r >= q_by && r < q_by + q_h && x >= q_bx && x < q_bx + q_w) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { const int d = (int)s.fe[r * 16 + x + i] - (int)s.fd[FDY + r * FD + x + i]; acc += d * d; }
                    }
                    if (cy >= (q_by >> 1) && cy < ((q_by + q_h) >> 1) && cx >= (q_bx >> 1) && cx < ((q_bx + q_w) >> 1)) {
                        const int du = (int)s.fe[256 + cy * 8 + cx] - (int)s.fd[FDU + cy * FD + cx], dv = (int)s.fe[320 + cy * 8 + cx] - (int)s.fd[FDV + cy * FD + cx];
                        acc += du * du + dv * dv;
                    }
                    ssd = wave_sum(acc);
                }
                if (rd.psy_rd) {                                     // size <= PIXEL_8x8: hadamard_ac of the partition against the cached source complexity
                    unsigned long long h = 0;
                    const bool in = lane < 4 && 8 * (lane & 1) >= q_bx && 8 * (lane & 1) < q_bx + q_w && 8 * (lane >> 1) >= q_by && 8 * (lane >> 1) < q_by + q_h;
                    if (in) h = hadamard_ac_8x8(s.fd + FDY + (lane >> 1) * 8 * FD + (lane & 1) * 8, FD);
                    const u32 lo = (u32)h, hi = (u32)(h >> 32);
                    unsigned long long sum = 0;
                    int f4 = 0, f8 = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        sum += ((unsigned long long)(u32)__builtin_amdgcn_readlane((int)hi, k) << 32) + (u32)__builtin_amdgcn_readlane((int)lo, k);
                        if (8 * (k & 1) >= q_bx && 8 * (k & 1) < q_bx + q_w && 8 * (k >> 1) >= q_by && 8 * (k >> 1) < q_by + q_h) {
                            const int k0 = (k >> 1) * 8 + (k & 1) * 2;
                            f4 += UNI(sr.fenc_satd[k0]) + UNI(sr.fenc_satd[k0 + 1]) + UNI(sr.fenc_satd[k0 + 4]) + UNI(sr.fenc_satd[k0 + 5]);
                            f8 += UNI(sr.fenc_sa8d[k]);
                        }
                    }
                    const int s4 = (int)((u32)sum >> 1), s8 = (int)(sum >> 34);
                    ssd += rf_psy((iabs(s4 - f4) + iabs(s8 - f8)) >> 1);
                }
                const unsigned long long c64 = rf_cost64(ssd, rf_bits(0, i8, q_pix), Q.lambda2);
                if (c64 < rf_best) { rf_best = c64; q_bmx = q_cx; q_bmy = q_cy; if (q_tag != -3) q_dir = q_tag; }
            }
        } else
            break;
    }
    if (!rf_emit) continue;                                          // the refinement is complete: on to step 12, the final encode
}
