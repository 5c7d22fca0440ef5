import sys,re
p='/root/repo/x264_vs2008_amd/csrc/frame_slice.hip'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    n=s.count(a)
    if n<1 or (cnt and n!=cnt):
        print("MISMATCH count",n,"for:",a[:100]); sys.exit(1)
    s=s.replace(a,b)

# --- MbSynDev + zero buffer in SwLdsRd
rep('''    TrellisScratch ts;
};
struct SwLdsNone { int unused; };''','''    u8 zero16[16];                  // sixteen zeros (SATD / SA8D of the source against nothing)
    int tmp_i[4];                   // lane 0 -> wave: bit count / QP after the writer
    TrellisScratch ts;
};
struct SwLdsNone { int unused; };
// the record cabac_dev.h's writer walks (same member names as MbSyn): scalars in registers, arrays where the kernel keeps them in LDS
struct MbSynDev {
    int slice_type, type, partition, i16mode, chroma_mode, cbp_luma, cbp_chroma, t8, qp, n_ref, pps_t8, t8_allowed;
    int type_left, type_top, cbp_left, cbp_top, cpm_left, cpm_top, nb_t8, last_qp, last_dqp, prev_coded;
    signed char *sub, *i4c, *cref;
    i16 (*cmv)[2], (*cmvd)[2];
    u8 *nnz, *nz_l, *nz_t;
    u8 (*nz_lc)[2], (*nz_tc)[2];
    i16 (*lv4)[16], (*lv8)[64], *lv_dc, (*lv_cdc)[4], (*lv_cac)[16];
};''')

# --- RD lambdas after encode_mb lambda
rep('''
        if (!is_p) {
            analyse_intra(MX_COST_MAX);''','''
        // ---- the RD levels: x264_mb_cache_fenc_satd, ssd_mb, x264_macroblock_size_cabac, x264_rd_cost_mb ----
        int fenc_satd_sum = 0, fenc_sa8d_sum = 0;
        auto cache_fenc_satd = [&]() {     // R/encoder/analyse.c:509-537 (the 16x16 sums; sub-partition RD is not built)
            if (!rd.psy_rd) return;
            int v4 = 0, v8 = 0;
            if (lane < 16) {
                const u8 *fe = s.fe + (lane >> 2) * 64 + (lane & 3) * 4;
                int sad = 0;
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) sad += fe[j * 16 + i];
                v4 = satd_4x4(sr.zero16, 0, fe, 16) - (sad >> 1);
            } else if (lane < 20) {
                const int b = lane - 16;
                const u8 *fe = s.fe + (b >> 1) * 128 + (b & 1) * 8;
                int sad = 0;
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int i = 0; i < 8; i++) sad += fe[j * 16 + i];
                v8 = ((sa8d_8x8_raw(sr.zero16, 0, fe, 16) + 2) >> 2) - (sad >> 2);
            }
            fenc_satd_sum = wave_sum(v4); fenc_sa8d_sum = wave_sum(v8);
        };
        auto ssd_mb = [&]() -> int {       // ssd_mb / ssd_plane, R/encoder/rdo.c:106-137
            int acc = 0;
            {
                const int r = lane >> 2, x = (lane & 3) * 4, cx = lane & 7, cy = lane >> 3;
#pragma unroll
                for (int i = 0; i < 4; i++) { const int d = (int)s.fe[r * 16 + x + i] - (int)s.fd[FDY + r * FD + x + i]; acc += d * d; }
                const int du = (int)s.fe[256 + cy * 8 + cx] - (int)s.fd[FDU + cy * FD + cx], dv = (int)s.fe[320 + cy * 8 + cx] - (int)s.fd[FDV + cy * FD + cx];
                acc += du * du + dv * dv;
            }
            int ssd = wave_sum(acc);
            if (rd.psy_rd) {
                unsigned long long h = 0;
                if (lane < 4) h = hadamard_ac_8x8(s.fd + FDY + (lane >> 1) * 8 * FD + (lane & 1) * 8, FD);
                const u32 lo = (u32)h, hi = (u32)(h >> 32);
                unsigned long long sum = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) sum += ((unsigned long long)(u32)__builtin_amdgcn_readlane((int)hi, k) << 32) + (u32)__builtin_amdgcn_readlane((int)lo, k);
                const int s4 = (int)((u32)sum >> 1), s8 = (int)(sum >> 34);
                const int satd = (iabs(s4 - fenc_satd_sum) + iabs(s8 - fenc_sa8d_sum)) >> 1;
                ssd += (satd * rd.psy_rd * Q.lambda + 128) >> 8;
            }
            return ssd;
        };
        // what the entropy coder reads of this macroblock: the interior of the motion cache from s.mv4 / s.ref8 (all lanes) ...
        auto syn_prepare = [&]() {
            if (is_p && lane < 16) {
                const int k = 12 + (lane & 3) + 8 * (lane >> 2);
                sr.cref[k] = s.ref8[(lane >> 3) * 2 + ((lane & 3) >> 1)]; sr.cmv[k][0] = s.mv4[lane][0]; sr.cmv[k][1] = s.mv4[lane][1];
            }
            if (lane < 4) sr.sub[lane] = (signed char)sub_t_mb;
            WAVE_SYNC();
        };
        // ... and the record the writer walks (scalars: wave-uniform registers)
        auto make_syn = [&]() -> MbSynDev {
            MbSynDev y;
            y.slice_type = a.slice_type; y.type = type; y.partition = part; y.i16mode = pred16; y.chroma_mode = predc;
            y.cbp_luma = cbp_luma; y.cbp_chroma = cbp_chroma; y.t8 = t8; y.qp = Q.qp; y.n_ref = a.n_refs; y.pps_t8 = a.transform8x8;
            y.t8_allowed = a.transform8x8 && (type == T_P_L0 || (type == T_P_8x8 && __ballot(lane < 4 && sub_t_mb != 3) == 0));
            y.type_left = left_type; y.type_top = type_top; y.cbp_left = left_cbp; y.cbp_top = cbp_top; y.cpm_left = left_cpm; y.cpm_top = cpm_top;
            y.nb_t8 = (left_type >= 0 && left_t8) + (type_top >= 0 && t8_top);
            y.last_qp = last_qp; y.last_dqp = last_dqp; y.prev_coded = prev_coded;
            y.sub = sr.sub; y.i4c = s.i4c; y.cref = sr.cref; y.cmv = sr.cmv; y.cmvd = sr.cmvd;
            y.nnz = s.nnz; y.nz_l = sr.nz_l; y.nz_t = sr.nz_t; y.nz_lc = sr.nz_lc; y.nz_tc = sr.nz_tc;
            y.lv4 = (i16 (*)[16])s.lv_y; y.lv8 = (i16 (*)[64])s.lv_y8; y.lv_dc = s.lv_dc; y.lv_cdc = (i16 (*)[4])s.lv_cdc; y.lv_cac = (i16 (*)[16])s.lv_cac;
            return y;
        };
        // x264_rd_cost_mb (R/encoder/rdo.c:139-171): trial encode, distortion, the syntax priced against a copy of the live contexts.
        // Like the reference it leaves `type` as the encode left it (P_SKIP when nothing was left to code on the skip vector).
        auto rd_cost_mb = [&]() -> int {
            const int t8_bak = t8;
            encode_mb(0);
            int cost = ssd_mb();
            if (type == T_P_SKIP) cost += (Q.lambda2 + 128) >> 8;
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
                cost += (int)(((unsigned long long)(u32)f8 * (u32)Q.lambda2 + 32768) >> 16);
            }
            t8 = t8_bak;
            return cost;
        };
        (void)cache_fenc_satd; (void)rd_cost_mb;
        // a->i_satd_pcm, analyse.c:246
        const int satd_pcm = RD && !rd.psy_rd && mbrd ? (int)(((unsigned long long)(386 * 8) * (u32)Q.lambda2 + 128) >> 8) : MX_COST_MAX;

        if (!is_p) {
          if constexpr (RD) {
            // x264_macroblock_analyse, I slice (analyse.c:2169-2186), the RD candidates and the final encode through ONE copy of the encoder
            if (mbrd) cache_fenc_satd();
            analyse_intra(MX_COST_MAX);
#pragma nounroll
            for (int step = mbrd ? 0 : 3; step < 4; step++) {
                if (step == 0) { if (!(satd_i16 <= MX_COST_MAX)) continue; type = T_I_16x16; }                                  // x264_intra_rd, :845-874
                else if (step == 1) { if (!(satd_i4 < MX_COST_MAX)) { satd_i4 = MX_COST_MAX; continue; } type = T_I_4x4; }
                else if (step == 2) { if (!(satd_i8 < MX_COST_MAX)) { satd_i8 = MX_COST_MAX; continue; } type = T_I_8x8; }
                else {
                    type = T_I_16x16;
                    int i_cost = satd_i16;
                    if (satd_i4 < i_cost) { i_cost = satd_i4; type = T_I_4x4; }
                    if (satd_i8 < i_cost) { i_cost = satd_i8; type = T_I_8x8; }
                    if (satd_pcm < i_cost) type = T_I_PCM;
                    tq.on = rd.trellis != 0;                                      // analyse.c:2768-2773
                    if (rd.trellis == 1 || a.nr) skip_intra = 0;
                    if (type != T_I_PCM) encode_mb(1);
                    break;
                }
                const int c = rd_cost_mb();
                if (step == 0) satd_i16 = c; else if (step == 1) satd_i4 = c; else satd_i8 = c;
            }
          } else {
            analyse_intra(MX_COST_MAX);''')
rep('''            if (satd_i8 < i_cost) { i_cost = satd_i8; type = T_I_8x8; }
        } else {
            // ---- motion neighbours: what cache_load puts around the block''','''            if (satd_i8 < i_cost) { i_cost = satd_i8; type = T_I_8x8; }
          }
        } else {
            // ---- motion neighbours: what cache_load puts around the block''')
open(p,'w').write(s)
print("ok")
