"""`goss build-kmer-set` on reads with poly-A / poly-T tails from a FASTQ file (the CLI's 24 GB arena and staged
chunks): does the build stay on the fused path?  usage: python tools/dbg/homo_cli.py [reads]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import gossamer_amd as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = 150
buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
with g.Context(25, 0, hbm_budget=1 << 30) as ctx:
    ctx.synth_reads(buf.data_ptr(), n, L, 20_000_000, seed=3)
torch.cuda.synchronize()
rows = buf.view(n, L + 1)
gen = torch.Generator(device="cuda"); gen.manual_seed(11)
part = rows[::6]
m = part.shape[0]
tlen = torch.randint(30, 81, (m, 1), device="cuda", generator=gen)
letter = torch.where((torch.arange(m, device="cuda") % 2 == 0).view(m, 1), torch.tensor(65, device="cuda"), torch.tensor(84, device="cuda")).to(torch.uint8)
mask = torch.arange(L, device="cuda").view(1, L) >= (L - tlen)
body = part[:, :L]
body[mask] = letter.expand(m, L)[mask]
seqs = rows[:, :L].cpu().numpy()
rec = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r"); rec[:, 2] = 10
rec[:, 3:3 + L] = seqs
rec[:, 3 + L] = 10; rec[:, 4 + L] = ord("+"); rec[:, 5 + L] = 10
rec[:, 6 + L:6 + 2 * L] = ord("I"); rec[:, 6 + 2 * L] = 10
d = "/dev/shm/goss_homo"
os.makedirs(d, exist_ok=True)
rec.tofile(os.path.join(d, "r.fq"))
del rec, seqs, buf
torch.cuda.empty_cache()
for env in ({}, {"GOSS_GPU_OVERFLOW_BY_SORT": "0"}):
    t = time.time()
    p = subprocess.run([os.path.join(ROOT, "gossamer_amd", "goss"), "build-kmer-set", "-k", "25", "-T", "32", "-i", os.path.join(d, "r.fq"), "-O", os.path.join(d, "ks"), "-v"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, GOSS_GPU_DEBUG="1", **env))
    dt = time.time() - t
    err = p.stderr.decode(errors="replace")
    print(env, "rc", p.returncode, "wall %.2f s" % dt, "by sort:", err.count("counted by sort"), "declined:", err.count("declined"), "overflowed:", err.count("overflowed"),
          [l.split("info")[-1].strip() for l in err.splitlines() if "total build time" in l or "k-mer windows" in l], flush=True)
import shutil; shutil.rmtree(d, ignore_errors=True)
