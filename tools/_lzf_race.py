import sys, os, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests")
import numpy as np, torch, compute_war_amd as cw
from conftest import corpus_file, corpus_names
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
data = b"".join(corpus_file(n) for n in corpus_names())
bs = 4096
for nb in (49152, 65536, 81920, 131072, 262144):
    a = np.frombuffer((data * (nb * bs // len(data) + 1))[:nb * bs], dtype=np.uint8).copy()
    src = torch.from_numpy(a).cuda()
    stride = (cw.compress_bound("lzf", bs) + 15) // 16 * 16
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda"); sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
    outs = []
    for rep in range(6):
        sizes.fill_(-7)
        cw.dev_compress("lzf", src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
        torch.cuda.synchronize()
        z = sizes.cpu().numpy()
        outs.append(z.copy())
    ref = outs[0]
    diffs = [int((o != ref).sum()) for o in outs]
    print(nb, "sum", [int(o.astype(np.int64).sum()) for o in outs], "diff vs rep0", diffs, "untouched", [int((o == -7).sum()) for o in outs], cw.profile_kernels()["codec"][:40], flush=True)
    if max(diffs):
        o = outs[[i for i, d in enumerate(diffs) if d][0]]
        idx = np.nonzero(o != ref)[0]
        print("   first differing blocks", idx[:10], ref[idx[:10]], o[idx[:10]])
