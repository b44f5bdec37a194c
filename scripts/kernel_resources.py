"""VGPR / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py gemm_b3_nt.hip [-DSM3_B3_SIGNED=0 ...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm3det_amd import build as B
src = sys.argv[1]
flags = sys.argv[2:]
cmd = [B._hipcc()] + B.BASE_FLAGS + B.FILE_FLAGS.get(os.path.basename(src), []) + flags + \
      ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(B.CSRC, src), '-o', '/dev/null']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for l in out.splitlines():
    m = re.search(r'remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPRs Spill): (\S+)', l)
    if not m:
        continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': v}
        rows.append(cur)
    else:
        cur[k.split(' ')[0] + ('Spill' if 'Spill' in k else '')] = v
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'sm3gemm::|\(sm3gemm::GemmParams\)|void ', '', n)
    print(f"{n[:100]:100s} vgpr {r.get('VGPRs','?'):>4s} spill {r.get('VGPRsSpill','?'):>3s} scratch {r.get('ScratchSize','?'):>4s} occ {r.get('Occupancy','?')} lds {r.get('LDS','?')}")
