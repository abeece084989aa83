"""gossamer_amd -- MI355X-native k-mer counting / de Bruijn edge-set construction.

The product is the C ABI in include/goss_gpu.h (libgossgpu.so, HIP kernels for gfx950) and
the C++ host command `goss` (build-kmer-set / build-graph).  This Python package only binds
the C ABI for tests and bench.py.
"""
from .binding import (Context, GossGpuError, MODE_GRAPH, MODE_KMER_SET, SYMBOLS,  # noqa: F401
                      group_emit, group_exchange, group_route_exchange, load, synth_reads_host)
