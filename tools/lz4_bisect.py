import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(15, exit=True)
import numpy as np
import compute_war_amd as cw
cw.init(0)
kind, bs, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if kind == "random":
    data = np.random.default_rng(1).integers(0, 256, bs * nb, dtype=np.uint8)
else:
    t = open(os.path.join(os.path.dirname(__file__), "..", "tests/golden/corpus/canterbury/alice29.txt"), "rb").read()
    data = np.frombuffer((t * (bs * nb // len(t) + 1))[: bs * nb], dtype=np.uint8)
print("calling", kind, bs, nb, os.environ.get("CW_LZ4_MODE"), flush=True)
try:
    sizes, payload = cw.compress_blocks("lz4", data, bs)
    print("ok sizes", sizes[:8], flush=True)
except Exception as e:
    print("error:", e, flush=True)
