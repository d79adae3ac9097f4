"""Print registers / LDS / spills of every kernel in a HIP shared library or object (reads the
clang offload bundle and the code object notes)."""
import re
import struct
import subprocess
import sys
import tempfile

data = open(sys.argv[1], 'rb').read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos = data.find(magic)
if pos < 0:
    raise SystemExit('no uncompressed offload bundle found')
n, = struct.unpack_from('<Q', data, pos + 24)
off = pos + 32
for _ in range(n):
    o, sz, tl = struct.unpack_from('<QQQ', data, off)
    triple = data[off + 24:off + 24 + tl].decode()
    off += 24 + tl
    if 'gfx' not in triple:
        continue
    with tempfile.NamedTemporaryFile(suffix='.co') as f:
        f.write(data[pos + o:pos + o + sz])
        f.flush()
        txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
    cur = {}
    for line in txt.splitlines():
        m = re.match(r'\s+\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)', line)
        if m:
            cur[m.group(1)] = m.group(2)
        if line.strip().startswith('.wavefront_size') or (m and m.group(1) == 'vgpr_spill_count'):
            pass
        if m and m.group(1) == 'vgpr_spill_count':
            pass
    # the notes list one mapping per kernel; re-parse per block
    for block in txt.split('- .agpr_count')[1:]:
        d = dict(re.findall(r'\.(name|vgpr_count|sgpr_count|vgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\S+)', block))
        ag = re.match(r':\s+(\S+)', block)
        name = subprocess.run(['c++filt', d.get('name', '?')], capture_output=True, text=True).stdout.strip()
        if pat in name:
            print(f"vgpr {d.get('vgpr_count'):>4} agpr {ag.group(1) if ag else '?':>4} sgpr {d.get('sgpr_count'):>4} spill {d.get('vgpr_spill_count'):>3} "
                  f"lds {d.get('group_segment_fixed_size'):>6} scratch {d.get('private_segment_fixed_size'):>5}  {name[:120]}")
