"""Developer aid: writes the refslice_encode_stream job of a random stream configuration (tests/test_gpu_stream.py random_config(seed), chain k)
for oracle/_ref/msan/refslice_msan (oracle/msan_main.c).  python scratch/dump_ref_job.py <seed> <chain> <out>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import look_cases as K, test_gpu_stream as T
from oracle import refslice as rs
if sys.argv[1] == "cavlc":                      # python scratch/dump_ref_job.py cavlc <name> <out>: tests/test_gpu_cavlc.py's live configuration (clip t0 = 37)
    import test_gpu_cavlc as TC
    c = TC.CONFIGS[sys.argv[2]]
    y, u, v = rs.clip(c["w"], c["h"], c["n"], 37)
    p, e = rs.make_params(c["w"], c["h"], c["n"], **c["kw"]), rs.make_ext(write=1, **c.get("ext", {}))
    e.payload_cap = ((c["w"] + 15) // 16) * ((c["h"] + 15) // 16) * 800 + 4096
    with open(sys.argv[3], "wb") as f:
        f.write(np.int32(1).tobytes())
        f.write(np.int32(C.sizeof(p)).tobytes()); f.write(bytes(p)); f.write(np.int32(C.sizeof(e)).tobytes()); f.write(bytes(e))
        f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
    sys.exit(0)
if sys.argv[1] == "stream":                     # python scratch/dump_ref_job.py stream <name>:<chain> <live 0|1> <out>: tests/test_gpu_stream.py's fixture configurations
    name, k = sys.argv[2].split(":")
    c = T.chains(name, [s_ + 40 * int(sys.argv[3]) for s_ in T.SEEDS[name]])[int(k)]
    out = sys.argv[4]
elif sys.argv[1] == "postsc":                   # python scratch/dump_ref_job.py postsc <seed> <out>: a lookahead test clip without --pre-scenecut (post-encode scene cuts)
    c = dict(K.config(int(sys.argv[2])), pre_scenecut=0, subme=5, n_refs=2, inter=0x13)
    c["scenecut_threshold"] = 40 if c["scenecut_threshold"] < 0 else c["scenecut_threshold"]
    out = sys.argv[3]
else:
    seed, k, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    c = T.random_config(seed)
    c.update(t0=c["t0"] + 61 * k, slow=[c["slow"], 1 + (c["slow"] % 3)][k])
p = rs.make_params(c["w"], c["h"], c["frames"], qp=c["qp"], me_method=c["me"], subme=c["subme"], n_refs=c.get("n_refs", 2), inter=c.get("inter", 0x33),
                   intra=0x3, transform8x8=1, cabac=1, deblock=1, keyint=c["keyint"], mixed_refs=c.get("mixed_refs", 0), chroma_me=c.get("chroma_me", 1))
e = rs.make_ext(bframes=c["bframes"], b_adapt=c["b_adapt"], pre_scenecut=c["pre_scenecut"], scenecut_threshold=c["scenecut_threshold"],
                keyint_min=c["keyint_min"], crf=-1.0 if c["crf"] is None else c["crf"], bframe_bias=c["bframe_bias"], weightb=c["weightb"],
                aq_mode=c["aq"], aq_strength=1.0, trellis=c.get("trellis", 0), psy_rd=c.get("psy_rd", 0.0), direct_pred=c.get("direct_pred", 1))
e.payload_cap = ((c["w"] + 15) // 16) * ((c["h"] + 15) // 16) * 800 + 4096
y, u, v = K.clip(c["w"], c["h"], c["frames"], c["cut"], c["t0"], c["slow"])
with open(out, "wb") as f:
    f.write(np.int32(0).tobytes())
    f.write(np.int32(C.sizeof(p)).tobytes()); f.write(bytes(p)); f.write(np.int32(C.sizeof(e)).tobytes()); f.write(bytes(e))
    f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())
print(c)
