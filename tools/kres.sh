#!/bin/bash
# Resource usage (LDS, VGPRs, SGPRs, spills) of the gfx950 kernels in libgossgpu.so whose name matches $1.
# usage: tools/kres.sh <pattern>
python3 - "$(dirname "$0")/../gossamer_amd/libgossgpu.so" <<'PY'
import struct, sys
data = open(sys.argv[1], 'rb').read()
i = data.find(b'__CLANG_OFFLOAD_BUNDLE__')
n = struct.unpack_from('<Q', data, i + 24)[0]
off = i + 32
for _ in range(n):
    o, sz, tl = struct.unpack_from('<QQQ', data, off); off += 24
    t = data[off:off + tl].decode(); off += tl
    if 'gfx950' in t:
        open('/tmp/gk.co', 'wb').write(data[i + o:i + o + sz])
PY
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/gk.co | awk -v pat="$1" '
/\.group_segment_fixed_size:/ {lds=$2} /\.name:/ {name=$2} /\.sgpr_count:/ {sg=$2} /\.vgpr_count:/ {vg=$2}
/\.vgpr_spill_count:/ {sp=$2; if (name ~ pat) printf "%-110s lds=%-7s vgpr=%-4s sgpr=%-4s spill=%s\n", substr(name,1,110), lds, vg, sg, sp}'
