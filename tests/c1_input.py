"""BASELINE.json config C1 as data: 1 M synthetic 150 bp reads of a 5 Mbp genome (seed 1, the library's own
generator -- one N in every 97th read, strand flipped with p = 1/2), written as 4-line FASTQ.  Used by
tests/golden/make_golden.py (digests of the oracle's output, committed) and tests/test_gpu_c1.py."""
import numpy as np

NREADS, READ_LEN, GENOME, SEED, K = 1_000_000, 150, 5_000_000, 1, 25


def fastq_bytes(nreads=NREADS):
    import gossamer_amd as g
    raw = np.frombuffer(g.synth_reads_host(nreads, READ_LEN, GENOME, seed=SEED), dtype=np.uint8).reshape(nreads, READ_LEN + 1)
    rec = 2 + 7 + 1 + (READ_LEN + 1) + 2 + (READ_LEN + 1)          # @r%07d\n bases\n +\n quals\n
    a = np.empty((nreads, rec), dtype=np.uint8)
    a[:, 0] = ord("@"); a[:, 1] = ord("r")
    idx = np.arange(nreads, dtype=np.int64)
    for d in range(7):
        a[:, 2 + 6 - d] = (idx // (10 ** d)) % 10 + ord("0")
    a[:, 9] = ord("\n")
    a[:, 10:10 + READ_LEN + 1] = raw
    a[:, 10 + READ_LEN + 1] = ord("+"); a[:, 10 + READ_LEN + 2] = ord("\n")
    a[:, 10 + READ_LEN + 3:rec - 1] = ord("I"); a[:, rec - 1] = ord("\n")
    return a.tobytes()
