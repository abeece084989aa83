import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gossamer_amd as g
text = b"ACGTNACGTACGTAAC" * 10
n = len(text)
buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
groups = (n + 15) // 16
dc = torch.zeros(groups + 1, dtype=torch.int32, device="cuda")
db = torch.zeros(groups + 1, dtype=torch.int16, device="cuda")
print("ptrs", hex(buf.data_ptr()), hex(dc.data_ptr()), hex(db.data_ptr()), flush=True)
with g.Context(25, 0, hbm_budget=64 << 20) as ctx:
    print("ctx", flush=True)
    ctx.pack_bases_device(buf.data_ptr(), n, dc.data_ptr(), db.data_ptr())
    print("packed", dc.cpu()[:3], db.cpu()[:3], flush=True)
