"""Per-kernel register / LDS / occupancy table from `make -C libtike-cufft_amd/csrc usage` output (file argument)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
rows = []; cur = {}
for l in txt:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        if cur: rows.append(cur)
        cur = {'name': m.group(1)}
    for key, lab in (('VGPRs:', 'v'), ('AGPRs', 'a'), ('VGPRs Spill', 'spill'), ('LDS Size', 'lds'), ('Occupancy', 'occ')):
        m = re.search(key + r'[^0-9]*([0-9]+)', l)
        if m: cur[lab] = m.group(1)
rows.append(cur)
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if re.search(pat, n):
        print("%-60s v %-4s a %-3s spill %-3s lds %-7s occ %s" % (n[:60], r.get('v'), r.get('a'), r.get('spill'), r.get('lds'), r.get('occ')))
