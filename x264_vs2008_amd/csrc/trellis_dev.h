// trellis_dev.h -- trellis quantisation against the live CABAC contexts (quant_trellis_cabac, R/encoder/rdo.c:411-628) for the
// slice kernel's final encode (--trellis 1) and RD trial encodes (--trellis 2).  A dynamic programme over the coefficients of one
// block in reverse scan order: up to 8 survivors (one per "node" = state of the level-coding contexts), two candidate levels per
// coefficient.  First version: one lane walks it, scratch in LDS.  Plain C++, also compiled for the host (see cabac_dev.h).
#pragma once
#include "cabac_dev.h"

#define TD_INF ((long long)1 << 50)
struct TrellisNode { long long score; int lv; u8 st[10]; };
struct TrellisScratch {
    TrellisNode nodes[2][8];
    TrellisNode tmp;                                 // the candidate being priced (kept here, not in private memory: its context array is indexed dynamically)
    u16 abs_c[64];
    u8 st_sig[64], st_last[64];
    u16 lvl_abs[64 * 8 * 2], lvl_next[64 * 8 * 2];   // the level tree: (level, previous entry) per accepted candidate
};

static __device__ const int d_trellis_lambda2[2][52] = {        // lambda2_tab, rdo.c:362-383: [0] inter, [1] intra
    {46, 58, 73, 92, 117, 147, 185, 233, 294, 370, 466, 587, 740, 932, 1174, 1480, 1864, 2349, 2959, 3728, 4697, 5918, 7457, 9395,
     11837, 14914, 18790, 23674, 29828, 37581, 47349, 59656, 75163, 94699, 119313, 150326, 189399, 238627, 300652, 378798,
     477255, 601304, 757596, 954511, 1202608, 1515192, 1909022, 2405217, 3030384, 3818045, 4810435, 6060769},
    {27, 34, 43, 54, 68, 86, 108, 136, 172, 216, 273, 343, 433, 545, 687, 865, 1090, 1374, 1731, 2180, 2747, 3461, 4361, 5494,
     6922, 8721, 10988, 13844, 17442, 21976, 27688, 34885, 43953, 55377, 69771, 87906, 110755, 139543, 175813, 221511,
     279087, 351627, 443023, 558174, 703255, 886046, 1116348, 1406511, 1772093, 2232697, 2813022, 3544186}};

// dct: the block in raster order, quantised in place; mf / unq: the quantiser multipliers and their inverses (h->quant4_mf[cat][qp],
// h->unquant4_mf[cat][qp]) in raster order; weight: x264_dct4/8_weight2_zigzag (scan order; unused for DC); zz: scan position ->
// raster index; st: the 460 context states (read only); cat: block category as in cabac_dev.h; dc: DC block (one multiplier);
// b_ac: scan position 0 is not part of the block.  Returns "some level is not zero".
template <class TS, class DCT, class MF, class UNQ, class WT, class ZZ, class ST>
CD_FN int td_trellis_quant(TS &t, DCT dct, MF mf, UNQ unq, WT weight, ZZ zz, ST st, int cat, int lambda2, int b_ac, int dc, int n_coef)
{
    const int f = 1 << 15;
    int i, j, n_lvl = 1, cur = 0;
    unsigned long long neg = 0;                      // sign of the coefficient at scan position i
    for (i = n_coef - 1; i >= b_ac; i--)
        if ((unsigned)((int)dct[zz[i]] * (dc ? (int)mf[0] >> 1 : (int)mf[zz[i]]) + f - 1) >= 2u * f) break;
    if (i < b_ac) { for (j = 0; j < n_coef; j++) dct[j] = 0; return 0; }
    const int last_nnz = i;
    for (; i >= b_ac; i--) { const int c = dct[zz[i]]; t.abs_c[i] = (u16)cd_abs(c); if (c < 0) neg |= 1ull << i; }

    for (j = 1; j < 8; j++) t.nodes[0][j].score = TD_INF;
    t.nodes[0][0].score = 0; t.nodes[0][0].lv = 0;
    t.lvl_abs[0] = 0; t.lvl_next[0] = 0;
    if (n_coef == 64)
#pragma nounroll
        for (i = 0; i < 63; i++) { t.st_sig[i] = st[CD_SIG_OFF(5) + CD_SIG8(i)]; t.st_last[i] = st[CD_LAST_OFF(5) + CD_LAST8(i)]; }
    else {
        const int k = (!dc || cat != 3) ? 15 : 3, so = CD_SIG_OFF(cat), lo = CD_LAST_OFF(cat);
        for (i = 0; i < k; i++) { t.st_sig[i] = st[so + i]; t.st_last[i] = st[lo + i]; }
    }
    { const int lo = CD_LEVEL_OFF(cat); for (i = 0; i < 10; i++) t.nodes[0][0].st[i] = st[lo + i]; }

    for (i = last_nnz; i >= b_ac; i--) {
        const int coef = t.abs_c[i], q = (f + coef * (dc ? (int)mf[0] >> 1 : (int)mf[zz[i]])) >> 16;
        if (q == 0) {                                // only the "not significant" flag to pay, for every live node but 0
            const u32 c0 = (u32)((unsigned long long)CD_ENT(t.st_sig[i], 0) * (unsigned)lambda2 >> 4);
            for (j = 1; j < 8; j++)
                if (t.nodes[cur][j].score != TD_INF) {
                    t.lvl_abs[n_lvl] = 0; t.lvl_next[n_lvl] = (u16)t.nodes[cur][j].lv; t.nodes[cur][j].lv = n_lvl++;
                    t.nodes[cur][j].score += c0;
                }
            continue;
        }
        cur ^= 1;
        const int prv = cur ^ 1;
        for (j = 0; j < 8; j++) t.nodes[cur][j].score = TD_INF;
        int cost_sig0 = 0, cost_sig1 = 0, cost_last0 = 0, cost_last1 = 0;
        if (i < n_coef - 1) {
            cost_sig0 = CD_ENT(t.st_sig[i], 0); cost_sig1 = CD_ENT(t.st_sig[i], 1);
            cost_last0 = CD_ENT(t.st_last[i], 0); cost_last1 = CD_ENT(t.st_last[i], 1);
        }
#pragma nounroll
        for (int lvl = q; lvl >= q - 1; lvl--) {
            const int unq_lvl = ((dc ? (int)unq[0] << 1 : (int)unq[zz[i]]) * lvl + 128) >> 8, d = coef - unq_lvl;
            const long long ssd = (long long)d * d * (dc ? 256 : (int)weight[i]);
#pragma nounroll
            for (j = 0; j < 8; j++) {
                if (t.nodes[prv][j].score == TD_INF) continue;
                int node = j;
#ifdef TD_PRIVATE_NODE
                TrellisNode n = t.nodes[prv][j];
#else
                TrellisNode &n = t.tmp;
                n = t.nodes[prv][j];
#endif
                if (lvl || node) {
                    unsigned bits = lvl ? cost_sig1 : cost_sig0;
                    if (lvl) {
                        const int prefix = lvl - 1 < 14 ? lvl - 1 : 14, c1 = CD_LVL1_CTX(node);
                        bits += node == 0 ? cost_last1 : cost_last0;
                        bits += CD_ENT(n.st[c1], prefix > 0); n.st[c1] = (u8)CD_TRANS(n.st[c1], prefix > 0);
                        if (prefix > 0) {
                            bits += cd_unary(&n.st[0], CD_LVLGT1_CTX(node), prefix);
                            if (lvl >= 15) bits += cd_ue_size((unsigned)(lvl - 15)) << 8;
                            node = CD_NODE_NEXT1(node);
                        } else {
                            bits += 256;
                            node = CD_NODE_NEXT0(node);
                        }
                    }
                    n.score += (long long)((unsigned long long)bits * (unsigned)lambda2 >> 4);
                }
                n.score += ssd;
                if (n.score < t.nodes[cur][node].score) {
                    t.lvl_abs[n_lvl] = (u16)lvl; t.lvl_next[n_lvl] = (u16)n.lv; n.lv = n_lvl++;
                    t.nodes[cur][node] = n;
                }
            }
        }
    }
    int b = 0;
    for (j = 1; j < 8; j++) if (t.nodes[cur][j].score < t.nodes[cur][b].score) b = j;
    int nz = 0;
    j = t.nodes[cur][b].lv;
    for (i = b_ac; i < n_coef; i++) {
        const int a = t.lvl_abs[j];
        dct[zz[i]] = (i16)((neg >> i) & 1 ? -a : a);
        nz |= a;
        j = t.lvl_next[j];
    }
    return nz != 0;
}
