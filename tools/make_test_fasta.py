import sys, numpy as np, torch
sys.path.insert(0, ".")
from fmindex_collection_amd import datasets
lengths = [20_000_000, 15_000_000, 5_000_000]
text, st = datasets.genome_like_text(lengths, seed=3, device=torch.device("cuda", 0))
h = text.cpu().numpy()
lut = np.frombuffer(b"NACGT", dtype=np.uint8)
off = 0
with open("/tmp/test.fa", "wb") as f:
    for i, l in enumerate(lengths):
        f.write(b">chr%d test sequence\n" % (i + 1))
        s = lut[h[off: off + l]]
        s[1000:1100] = ord("N"); s[5000:5050] = ord("n")
        for p in range(0, l, 60_000_00):
            chunk = s[p: p + 60_000_00]
            lines = chunk.reshape(-1, 60) if chunk.size % 60 == 0 else None
            if lines is not None:
                f.write(b"\n".join(x.tobytes() for x in lines) + b"\n")
            else:
                f.write(chunk.tobytes() + b"\n")
        off += l
print("wrote", sum(lengths))
