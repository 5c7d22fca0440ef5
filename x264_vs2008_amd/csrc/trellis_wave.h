// trellis_wave.h -- the lane-parallel form of trellis_dev.h's td_trellis_quant (quant_trellis_cabac, R/encoder/rdo.c:411-628).
//
// The dynamic programme keeps up to 8 survivors ("nodes") per coefficient and prices two candidate levels from each: 16
// candidates per coefficient, independent of one another.  Here a group of 16 lanes walks one block -- lane c prices candidate
// (level q - (c >> 3), node c & 7) -- and the four groups of the wavefront walk four blocks at once (the blocks of a macroblock are
// quantised against the same, read-only, context states).  What the serial form decides by visiting the candidates in order
// ("the first candidate that is strictly cheaper takes the node") is decided here by rank: candidate c wins its target node when
// no candidate of the same target is cheaper, or as cheap with a lower index.  The survivors, the candidates and the level lists
// live in LDS; nothing crosses lanes except through them, so groups may leave the coefficient loop at different times.
// Results are bit-identical to td_trellis_quant (tests/test_gpu_slice_rd.py: the trellis chains and the hash fixtures).
#pragma once
#include "trellis_dev.h"

struct TdWave {                          // per-wavefront scratch, [group] first
    long long nscore[4][2][8];           // score of node j of generation 0 / 1
    u32 nst[4][2][8][3];                 // its ten level-coding context states, packed
    u16 nlv[4][2][8];                    // head of its level list
    long long cscore[4][16];             // the candidates of the current coefficient
    u8 ctgt[4][16];                      // their target nodes (0xff: none)
    u16 abs_c[64];                       // |coefficient| by scan position: group g at [16 g] for 16-coefficient blocks, one group for 64
    u8 st_sig[64], st_last[64];          // the states of the significance / last flags by scan position, same layout
};
#define TDW_TREE_STRIDE 132              // level-list entries per group for blocks of up to 16 coefficients (1 + 16 * 8, padded)
#define TDW_TREE_ENTRIES 516             // ... and what one 64-coefficient block needs (1 + 64 * 8, padded): the area holds max(4 * 132, 516) words

__device__ __forceinline__ int tdw_st_get(const u32 w[3], int i) { const u32 v = i < 4 ? w[0] : i < 8 ? w[1] : w[2]; return (int)((v >> (8 * (i & 3))) & 255u); }
__device__ __forceinline__ void tdw_st_set(u32 w[3], int i, int s)
{
    const u32 m = 255u << (8 * (i & 3)), v = (u32)s << (8 * (i & 3));
    if (i < 4) w[0] = (w[0] & ~m) | v; else if (i < 8) w[1] = (w[1] & ~m) | v; else w[2] = (w[2] & ~m) | v;
}
#define TDW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

// Every lane of the wavefront calls this.  Group g = lane >> 4 quantises the block at `dct` (its own pointer) when `active`; mf, unq,
// weight, zz, st, cat, lambda2, b_ac, dc, n_coef as td_trellis_quant and the same for all groups.  n_coef == 64: only group 0 may be
// active.  tree: TDW_TREE_ENTRIES words of LDS.  Returns "some level of this group's block is not zero" (0 for an inactive group).
template <class DCT, class MF, class UNQ, class WT, class ZZ, class ST>
__device__ __forceinline__ int td_trellis_wave(TdWave &w, u32 *tree_all, DCT dct, bool active, MF mf, UNQ unq, WT weight, ZZ zz, ST st,
                                               int cat, int lambda2, int b_ac, int dc, int n_coef, int lane)
{
    const int f = 1 << 15, g = lane >> 4, c = lane & 15, j = c & 7, lvsel = c >> 3;
    const int cb = n_coef == 64 ? 0 : 16 * g;                      // this group's base in abs_c / st_sig / st_last
    u32 *tree = tree_all + (n_coef == 64 ? 0 : TDW_TREE_STRIDE * g);
    // ---- the last coefficient that does not quantise to zero ----
    int my_last = -1;
    if (active)
        for (int i = c; i < n_coef; i += 16)
            if (i >= b_ac && (unsigned)((int)dct[zz[i]] * (dc ? (int)mf[0] >> 1 : (int)mf[zz[i]]) + f - 1) >= 2u * f) my_last = i;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) { const int o = __shfl_xor(my_last, m, 16); my_last = o > my_last ? o : my_last; }
    const int last_nnz = my_last;
    int nz = 0;
    if (active && last_nnz < b_ac) { for (int i = c; i < n_coef; i += 16) dct[i] = 0; }
    if (active && last_nnz >= b_ac) {
        for (int i = c; i < n_coef; i += 16) {
            if (i >= b_ac && i <= last_nnz) w.abs_c[cb + i] = (u16)cd_abs((int)dct[zz[i]]);
            if (n_coef == 64) { if (i < 63) { w.st_sig[i] = st[CD_SIG_OFF(5) + CD_SIG8(i)]; w.st_last[i] = st[CD_LAST_OFF(5) + CD_LAST8(i)]; } }
            else if (i < ((!dc || cat != 3) ? 15 : 3)) { w.st_sig[cb + i] = st[CD_SIG_OFF(cat) + i]; w.st_last[cb + i] = st[CD_LAST_OFF(cat) + i]; }
        }
        if (lvsel == 0) {
            w.nscore[g][0][j] = j == 0 ? 0 : TD_INF;
            if (j == 0) {
                const int lo = CD_LEVEL_OFF(cat);
                u32 s3[3] = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 10; k++) tdw_st_set(s3, k, st[lo + k]);
                w.nst[g][0][0][0] = s3[0]; w.nst[g][0][0][1] = s3[1]; w.nst[g][0][0][2] = s3[2];
                w.nlv[g][0][0] = 0;
                tree[0] = 0;                                        // the list's end: level 0, next = itself
            }
        }
        TDW_SYNC();
        int cur = 0, slot = 1;
#pragma nounroll
        for (int i = last_nnz; i >= b_ac; i--, slot += 8) {
            const int coef = w.abs_c[cb + i], q = (f + coef * (dc ? (int)mf[0] >> 1 : (int)mf[zz[i]])) >> 16;
            if (q == 0) {                                          // only "not significant" to pay, for every live node but 0
                if (lvsel == 0 && j > 0 && w.nscore[g][cur][j] != TD_INF) {
                    const u32 c0 = (u32)((unsigned long long)CD_ENT(w.st_sig[cb + i], 0) * (unsigned)lambda2 >> 4);
                    tree[slot + j] = (u32)w.nlv[g][cur][j] << 16;
                    w.nlv[g][cur][j] = (u16)(slot + j);
                    w.nscore[g][cur][j] += c0;
                }
                TDW_SYNC();
                continue;
            }
            cur ^= 1;
            const int prv = cur ^ 1;
            if (lvsel == 0) w.nscore[g][cur][j] = TD_INF;
            // ---- this lane's candidate ----
            int cost_sig0 = 0, cost_sig1 = 0, cost_last0 = 0, cost_last1 = 0;
            if (i < n_coef - 1) {
                cost_sig0 = CD_ENT(w.st_sig[cb + i], 0); cost_sig1 = CD_ENT(w.st_sig[cb + i], 1);
                cost_last0 = CD_ENT(w.st_last[cb + i], 0); cost_last1 = CD_ENT(w.st_last[cb + i], 1);
            }
            const int lvl = q - lvsel;
            long long score = w.nscore[g][prv][j];
            const bool live = score != TD_INF;
            int node = j;
            u32 s3[3] = {w.nst[g][prv][j][0], w.nst[g][prv][j][1], w.nst[g][prv][j][2]};
            const int lv_prev = w.nlv[g][prv][j];
            if (live) {
                const int unq_lvl = ((dc ? (int)unq[0] << 1 : (int)unq[zz[i]]) * lvl + 128) >> 8, d = coef - unq_lvl;
                const long long ssd = (long long)d * d * (dc ? 256 : (int)weight[i]);
                if (lvl || node) {
                    unsigned bits = lvl ? cost_sig1 : cost_sig0;
                    if (lvl) {
                        const int prefix = lvl - 1 < 14 ? lvl - 1 : 14, c1 = CD_LVL1_CTX(node);
                        bits += node == 0 ? cost_last1 : cost_last0;
                        const int s1 = tdw_st_get(s3, c1);
                        bits += CD_ENT(s1, prefix > 0); tdw_st_set(s3, c1, CD_TRANS(s1, prefix > 0));
                        if (prefix > 0) {
                            const int cg = CD_LVLGT1_CTX(node);
                            int sb = 0, sg = tdw_st_get(s3, cg);       // cd_unary on the packed states
                            for (int k = 1; k < prefix; k++) { sb += CD_ENT(sg, 1); sg = CD_TRANS(sg, 1); }
                            if (prefix < 14) { sb += CD_ENT(sg, 0); sg = CD_TRANS(sg, 0); }
                            tdw_st_set(s3, cg, sg);
                            bits += sb + 256;
                            if (lvl >= 15) bits += cd_ue_size((unsigned)(lvl - 15)) << 8;
                            node = CD_NODE_NEXT1(node);
                        } else {
                            bits += 256;
                            node = CD_NODE_NEXT0(node);
                        }
                    }
                    score += (long long)((unsigned long long)bits * (unsigned)lambda2 >> 4);
                }
                score += ssd;
            }
            w.cscore[g][c] = score; w.ctgt[g][c] = (u8)(live ? node : 0xff);
            TDW_SYNC();
            // ---- the first cheapest candidate of a node takes it ----
            if (live) {
                bool win = true;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const long long so = w.cscore[g][k];
                    if (w.ctgt[g][k] == node && (so < score || (so == score && k < c))) win = false;
                }
                if (win) {
                    w.nscore[g][cur][node] = score;
                    w.nst[g][cur][node][0] = s3[0]; w.nst[g][cur][node][1] = s3[1]; w.nst[g][cur][node][2] = s3[2];
                    tree[slot + node] = (u32)lvl | (u32)lv_prev << 16;
                    w.nlv[g][cur][node] = (u16)(slot + node);
                }
            }
            TDW_SYNC();
        }
        // ---- the cheapest survivor's levels go back into the block ----
        if (c == 0) {
            int b = 0;
            for (int k = 1; k < 8; k++) if (w.nscore[g][cur][k] < w.nscore[g][cur][b]) b = k;
            int e = w.nlv[g][cur][b];
            for (int i = b_ac; i < n_coef; i++) {
                const u32 t = tree[e];
                const int a = (int)(t & 0xffffu);
                dct[zz[i]] = (i16)((int)dct[zz[i]] < 0 ? -a : a);
                nz |= a;
                e = (int)(t >> 16);
            }
        }
    }
    TDW_SYNC();
    return __shfl(nz, 0, 16) != 0;
}
