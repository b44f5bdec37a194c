"""VGPR / scratch / occupancy of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py gemm_b3_nt.hip [-DSM3_B3_SIGNED=0 ...]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_resources(src, flags=()):
    """-> [{name (demangled), VGPRs, VGPRsSpill, ScratchSize, Occupancy, LDS}] for csrc/<src>"""
    from sm3det_amd import build as B
    cmd = [B._hipcc()] + B.BASE_FLAGS + B.FILE_FLAGS.get(os.path.basename(src), []) + list(flags) + \
          ['-Rpass-analysis=kernel-resource-usage', '--cuda-device-only', '-c', os.path.join(B.CSRC, src), '-o', '/dev/null']
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = {}, []
    for l in out.splitlines():
        m = re.search(r'remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|'
                      r'LDS Size \[bytes/block\]|VGPRs Spill): (\S+)', l)
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
        r['name'] = re.sub(r'sm3gemm::|\(sm3gemm::GemmParams\)|void ', '', n)
    return rows


if __name__ == '__main__':
    for r in kernel_resources(sys.argv[1], sys.argv[2:]):
        print(f"{r['name'][:100]:100s} vgpr {r.get('VGPRs','?'):>4s} spill {r.get('VGPRsSpill','?'):>3s} scratch "
              f"{r.get('ScratchSize','?'):>4s} occ {r.get('Occupancy','?')} lds {r.get('LDS','?')}")
