"""The end-to-end record of bench.py alone (FASTQ file -> goss build-kmer-set -> files closed, both runs with their phases)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
r = bench.e2e_record(n, 150, n, 1, 25, max(1, min(os.cpu_count() or 1, 64)))
print(json.dumps(r, indent=1))
