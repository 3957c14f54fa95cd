"""Decoded bytes per second of cw_dev_decompress on lcet10-tiled text: `decode_probe.py [total MiB] [block sizes...]`
(CW_DECODE_LANES=0 keeps the wavefront decoders for every batch)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, compute_war_amd as cw
cw.init(0)
s = torch.cuda.current_stream().cuda_stream
t = open("tests/golden/corpus/canterbury/lcet10.txt", "rb").read()
for alg in ("lz4", "lzf"):
    for bs in ([int(x) for x in sys.argv[2:]] or [4096, 65536]):
        nb = ((int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20) // bs
        data = (t * (nb * bs // len(t) + 1))[: nb * bs]
        src = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        stride = (cw.compress_bound(alg, bs) + 15) // 16 * 16
        dst = torch.empty(nb * stride, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(nb, dtype=torch.int32, device="cuda")
        out = torch.empty(nb * bs, dtype=torch.uint8, device="cuda")
        st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        cw.dev_compress(alg, src.data_ptr(), bs, nb, dst.data_ptr(), stride, sizes.data_ptr(), s)
        cw.dev_decompress(alg, dst.data_ptr(), stride, sizes.data_ptr(), nb, out.data_ptr(), bs, st.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            cw.dev_decompress(alg, dst.data_ptr(), stride, sizes.data_ptr(), nb, out.data_ptr(), bs, st.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        ok = bool((st == 0).all()) and torch.equal(out, src)
        print(f"decode {alg} bs={bs}: {ms:.2f} ms  {nb*bs/ms/1e6:.1f} GB/s (decoded bytes)  ok={ok}")
