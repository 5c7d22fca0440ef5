"""Developer aid: random stream configurations over this round's additions (--direct auto, the post-encode scene cut with and without the lookahead running
ahead, b-adapt 0 / 1 / 2, CQP / CRF, subme 2-8) through the StreamEncoder against the reference's encoder.  python scratch/fuzz_stream_new.py <first> <count>"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import look_cases as K
import test_gpu_stream as T
from x264_vs2008_amd import lib as L

hip = L.load()
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    cs, pipe = T.mixed_config(seed)
    c = cs[0]
    what = "seed %d pipe %d %s" % (seed, pipe, {k: c[k] for k in ("w", "h", "frames", "subme", "n_refs", "bframes", "b_adapt", "crf", "trellis", "direct_pred", "aq", "inter", "pre_scenecut", "scenecut_threshold", "cut", "keyint")})
    try:
        got = T.run_stream(hip, cs, pipeline=pipe)
        gave = 0
        for i, ck in enumerate(cs):
            a = K.reference_records(ck)
            T.check(got[i], a, ck, what + " chain %d" % i)
            gave += int(a["stat"][:ck["frames"], 3].sum())
            for f in range(ck["frames"]):
                if int(a["frame_info"][f][0]) == 1:
                    assert T.run_stream.direct_spatial[i][f] == int(a["frame_info2"][f][3]), what + " direct mode of coded frame %d" % f
        print("ok  ", what, "given up", gave, flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", what, "\n    ", str(ex)[:300].replace("\n", " "), flush=True)
print("done:", count, "configurations,", bad, "failures")
