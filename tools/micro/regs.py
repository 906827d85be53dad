"""Register / scratch / occupancy table of the kernels in a -Rpass-analysis=kernel-resource-usage log:
   python tools/micro/regs.py LOG [name-substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in txt.split('Function Name: ')[1:]:
    name = b.split('\n')[0].split(' [')[0].strip()
    dem = subprocess.run(['/usr/bin/c++filt', name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    short = re.sub(r'\(anonymous namespace\)::', '', dem.split('(')[0] if not dem.startswith('void (') else dem[5:].split('>(')[0] + '>')
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print(short[:90].ljust(90), 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'),
          'occ', g(r'Occupancy \[waves/SIMD\]'), 'lds', g(r'LDS Size \[bytes/block\]'))
