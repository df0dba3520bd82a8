"""Per-kernel VGPR / SGPR / scratch / occupancy / LDS of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-Rpass-analysis=kernel-resource-usage',
                    '-c', src, '-o', '/dev/null'] + sys.argv[3:], stderr=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in r.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = subprocess.run(['c++filt', t.split(':', 1)[1].strip()], stdout=subprocess.PIPE, text=True).stdout.strip()
        rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1)
        rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    if flt in name:
        short = re.sub(r'\(.*', '', name).replace('void otto::', '')
        print(f"{short[:60]:60s} VGPR {d.get('VGPRs','?'):>4} AGPR {d.get('AGPRs','?'):>3} SGPR {d.get('TotalSGPRs', d.get('SGPRs','?')):>4} "
              f"scratch {d.get('ScratchSize [bytes/lane]','?'):>4} occ {d.get('Occupancy [waves/SIMD]','?'):>2} LDS {d.get('LDS Size [bytes/block]','?'):>7}")
