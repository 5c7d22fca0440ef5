"""Developer tool: patch sub-phase timers into the motion search (slots of the sweep's profile array are re-purposed).
Apply, build, run scratch/perf_sweep.py, then `git checkout x264_vs2008_amd/csrc scratch/perf_sweep.py`."""
import re
p = 'x264_vs2008_amd/csrc/me_exact.h'
s = open(p).read()
s = s.replace('    int mvpx, mvpy;           // the predictor the costs are relative to', '    long long *pacc, *ptime;\n    int mvpx, mvpy;           // the predictor the costs are relative to')
s = s.replace('#define MX_PS 28 ', '#define MXP(k_) do { if (c.pacc) { long long n_ = (long long)wall_clock64(); c.pacc[k_] += n_ - *c.ptime; *c.ptime = n_; } } while (0)\n#define MX_PS 28 ')
s = s.replace('    const int lane = c.lane, g16 = lane >> 4, g8 = lane >> 3;\n', '    const int lane = c.lane, g16 = lane >> 4, g8 = lane >> 3;\n    MXP(0);\n', 1)
s = s.replace('    bool do_hex = o.method == 1;', '    MXP(2);\n    bool do_hex = o.method == 1;', 1)
s = s.replace('    int mvx, mvy, mcost;\n    if (bpcost < bcost)', '    MXP(3);\n    int mvx, mvy, mcost;\n    if (bpcost < bcost)', 1)
s = s.replace('        if (c.has_patch) mx_load_patch(c, bx, by);\n        if (hpel && o.subme < 3)', '        if (c.has_patch) mx_load_patch(c, bx, by);\n        MXP(4);\n        if (hpel && o.subme < 3)', 1)
s = s.replace('        if (by > L.smax1) by = L.smax1;\n        bc = __builtin_amdgcn_readlane(subpel_sum16_lane', '        MXP(5);\n        if (by > L.smax1) by = L.smax1;\n        bc = __builtin_amdgcn_readlane(subpel_sum16_lane', 1)
s = s.replace('        if (thresh) {\n            const int th = MX_UNI(*thresh);', '        MXP(6);\n        if (thresh) {\n            const int th = MX_UNI(*thresh);', 1)
s = s.replace('#undef INRANGE\n    out_mvx = mvx; out_mvy = mvy;', '#undef INRANGE\n    MXP(7);\n    out_mvx = mvx; out_mvy = mvy;', 1)
s = s.replace('    const int lane = c.lane, g16 = lane >> 4;\n    const int hpel = c_subpel_iters[o.subme][0]', '    const int lane = c.lane, g16 = lane >> 4;\n    MXP(0);\n    const int hpel = c_subpel_iters[o.subme][0]', 1)
s = s.replace('    mvx = bx; mvy = by;\n    return bc;', '    MXP(1);\n    mvx = bx; mvy = by;\n    return bc;', 1)
open(p, 'w').write(s)
p = 'x264_vs2008_amd/csrc/frame_slice.hip'
s = open(p).read()
s = s.replace('c.has_patch = true; c.patch_on = false;', 'c.has_patch = true; c.patch_on = false; c.pacc = a.prof ? pacc : nullptr; c.ptime = &ptime;')
s = s.replace('pacc[k_] += now_ - ptime; ptime = now_; } } while (0)', 'pacc[0] += now_ - ptime; ptime = now_; } } while (0)')
open(p, 'w').write(s)
p = 'x264_vs2008_amd/csrc/frame_me_exact.hip'
s = open(p).read()
s = s.replace('c.has_patch = false; c.patch_on = false;', 'c.has_patch = false; c.patch_on = false; c.pacc = nullptr; c.ptime = nullptr;')
open(p, 'w').write(s)
p = 'scratch/perf_sweep.py'
s = open(p).read()
s = s.replace('            print("   us/MB (mean over rows): wait', '            print("   SUB us/MB: all-else %.1f refine_qpel %.1f predictors+fpel %.1f walk %.1f patch %.1f hpel %.1f satd-at-best %.1f qpel %.1f" % tuple(pr[:, k].mean() for k in range(8)))\n            print("   us/MB (mean over rows): wait')
open(p, 'w').write(s)
