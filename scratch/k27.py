import sys
def patch(p, pairs):
    s=open(p).read()
    for a,b in pairs:
        if s.count(a)!=1: print("MISMATCH",p,s.count(a),a[:100]); sys.exit(1)
        s=s.replace(a,b)
    open(p,'w').write(s)
C='/root/repo/x264_vs2008_amd/csrc/'
patch(C+'slice_kernel.h',[
('''__device__ __forceinline__ int sw_probe_pskip(SwLds &s, const SwRefs &refs, const SwArgs &a, const SwQp &Q, int pmx, int pmy, int mbx, int mby,
                                              ptrdiff_t oy, ptrdiff_t oc, size_t by_, size_t bc_, int lane)
{
    const int vx = clip3(pmx, 4 * (-16 * mbx - 24), 4 * (16 * (a.mb_w - mbx - 1) + 24));
    const int vy = clip3(pmy, 4 * (-16 * mby - 24), 4 * (16 * (a.mb_h - mby - 1) + 24));
    sw_mc16(s, refs, a, 0, vx, vy, oy, oc, by_, bc_, lane, true);
    WAVE_SYNC();''','''__device__ __forceinline__ int sw_probe_pskip(SwLds &s, const SwRefs &refs, const SwArgs &a, const SwQp &Q, int pmx, int pmy, int mbx, int mby,
                                              ptrdiff_t oy, ptrdiff_t oc, size_t by_, size_t bc_, int lane, bool b_bidir = false)
{
    if (!b_bidir) {                 // x264_macroblock_probe_bskip: the (direct) prediction is in fdec already
        const int vx = clip3(pmx, 4 * (-16 * mbx - 24), 4 * (16 * (a.mb_w - mbx - 1) + 24));
        const int vy = clip3(pmy, 4 * (-16 * mby - 24), 4 * (16 * (a.mb_h - mby - 1) + 24));
        sw_mc16(s, refs, a, 0, vx, vy, oy, oc, by_, bc_, lane, true);
        WAVE_SYNC();
    }''')])
patch(C+'slice_b_flow.h',[
('''            bskip_cost = ssd_mb();
            if (bskip_cost <= ((6 * Q.lambda2 + 128) >> 8)) { skip_mc = 1; fin = true; }      // "6 = minimum cavlc cost of a non-skipped MB"
            else {''','''            bool b_skip;
            if (mbrd) { bskip_cost = ssd_mb(); b_skip = bskip_cost <= ((6 * Q.lambda2 + 128) >> 8); }      // "6 = minimum cavlc cost of a non-skipped MB"
            else b_skip = sw_probe_pskip(s, refs, a, Q, 0, 0, mbx, mby, oy, oc, by_, bc_, lane, true) != 0;   // x264_macroblock_probe_bskip
            if (b_skip) { skip_mc = 1; fin = true; }
            else {'''),
('''                if (cost16direct <= i_cost_b * 33 / 32) { pass = 0;''','''                if (mbrd && cost16direct <= i_cost_b * 33 / 32) { pass = 0;'''),
('''            PROF(2);
            LAUNDER();
            i_satd_inter_b = i_cost_b;
            pass = 1; bthresh = i_satd_inter_b * (17 + (rd.psy_rd != 0)) / 16; kcand = 0; bstep = BS_CAND;
            continue;''','''            PROF(2);
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
            continue;'''),
('''        } else if (bstep == BS_SELECT) {
            i_type_b = T_B_SKIP; i_cost_b = bskip_cost; i_part_b = 16;
            if (rd_l0 < i_cost_b) { i_cost_b = rd_l0; i_type_b = T_B_L0_L0; }
            if (rd_l1 < i_cost_b) { i_cost_b = rd_l1; i_type_b = T_B_L1_L1; }
            if (rd_bi < i_cost_b) { i_cost_b = rd_bi; i_type_b = T_B_BI_BI; }
            if (rd_direct < i_cost_b) { i_cost_b = rd_direct; i_type_b = T_B_DIRECT; }
            if (rd_168 < i_cost_b) { i_cost_b = rd_168; i_type_b = type16x8; i_part_b = 14; }
            if (rd_816 < i_cost_b) { i_cost_b = rd_816; i_type_b = type8x16; i_part_b = 15; }
            if (rd_8 < i_cost_b) { i_cost_b = rd_8; i_type_b = T_B_8x8; i_part_b = 13; }
            type = i_type_b; part = i_part_b;
            analyse_intra(i_satd_inter_b);
            PROF(6);''','''        } else if (bstep == BS_SELECT) {
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
            if (!mbrd) { bstep = BS_FINAL; continue; }'''),
('''            // x264_refine_bidir (subme >= 5): the bi-predicted blocks of the chosen partition
            if (!IS_INTRA_T(type)) {''','''            // x264_refine_bidir (subme >= 5): the bi-predicted blocks of the chosen partition
            if (!IS_INTRA_T(type) && a.subme >= 5) {'''),
])
s=open(C+'slice_b_flow.h').read()
# the satd flag: mbcmp is SAD below subme 2 -- restrict in the host instead; nothing to do here
open(C+'slice_b_flow.h','w').write(s)
patch(C+'frame_slice.hip',[
('''        if (p->subme != 7 || !p->rd->write || !p->cabac) { set_error("slice_sweep: B slices are built for subme 7 (mode-decision RD) with the CABAC writer in the loop"); return -1; }''',
'''        if (p->subme < 2 || p->subme > 7 || !p->rd->write || !p->cabac) { set_error("slice_sweep: B slices are built for subme 2..7 with the CABAC writer in the loop"); return -1; }'''),
])
print('ok')
